#!/bin/bash
# rocprofv3 evidence for bench.py, run ON THE GPU BOX:  bash tools/profile_bench.sh <tag>
# Pass 1: kernel trace + stats.  Passes 2..4: PMC counters, each in its own run (never combined with
# a trace domain other than the kernel trace).  Summaries land in gpurun_out/prof_<tag>/ ; copy the
# *.txt into profiles/<round>/ afterwards.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-run}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
db() { find "$1" -name '*.db' | head -1; }
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o bench -- $BENCH > "$OUT/kt.log" 2>&1
python "$ROOT/tools/rocpd_summary.py" "$(db "$OUT/kt")" > "$OUT/bench_em_kernel_stats.txt" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d "$OUT/$C" -o bench -- $BENCH > "$OUT/$C.log" 2>&1
  python "$ROOT/tools/rocpd_summary.py" "$(db "$OUT/$C")" --pmc > "$OUT/bench_em_pmc_$(echo $C | tr A-Z a-z).txt" 2>&1
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY \
  -d "$OUT/sq" -o bench -- $BENCH > "$OUT/sq.log" 2>&1
python "$ROOT/tools/rocpd_summary.py" "$(db "$OUT/sq")" --pmc > "$OUT/bench_em_pmc_sq.txt" 2>&1
tail -1 "$OUT/kt.log" > "$OUT/bench_under_rocprof.json"
rm -rf "$OUT"/kt "$OUT"/FETCH_SIZE "$OUT"/WRITE_SIZE "$OUT"/sq
ls -la "$OUT"
