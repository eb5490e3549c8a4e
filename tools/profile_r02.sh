#!/bin/bash
# rocprofv3 evidence of round 2, run ON THE GPU BOX:  bash tools/profile_r02.sh
# Kernel-trace summaries and PMC passes are separate runs (a --pmc run carries no trace domain except the kernel trace).
# Summaries land in gpurun_out/prof_r02/ ; the *.txt / *.json are copied into profiles/r02/ afterwards.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r02
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }
kt() { # kt <tag> <cmd...>: kernel trace + stats
  local tag=$1; shift
  ( cd $ROOT && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$tag" -o t -- "$@" > "$OUT/$tag.log" 2>&1 )
  python "$ROOT/tools/rocpd_summary.py" "$(db "$OUT/kt_$tag")" > "$OUT/${tag}_kernel_stats.txt" 2>&1
  rm -rf "$OUT/kt_$tag"
}
pmc() { # pmc <tag> <counters...> -- <cmd...>
  local tag=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  ( cd $ROOT && timeout 900 rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d "$OUT/pmc_$tag" -o t -- "$@" > "$OUT/pmc_$tag.log" 2>&1 )
  python "$ROOT/tools/rocpd_summary.py" "$(db "$OUT/pmc_$tag")" --pmc > "$OUT/${tag}.txt" 2>&1
  rm -rf "$OUT/pmc_$tag"
}
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
kt bench_em $BENCH
grep -h '^{"metric"' "$OUT/bench_em.log" > "$OUT/bench_em_under_rocprof.json"
pmc bench_em_pmc_fetch_size FETCH_SIZE -- $BENCH
pmc bench_em_pmc_write_size WRITE_SIZE -- $BENCH
pmc bench_em_pmc_sq SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- $BENCH
TV="python bench.py --workload tv --steps 2 --warmup 1 --no-cpu-baseline --tv-utterances 2048"
kt bench_tv $TV
grep -h '^{"metric"' "$OUT/bench_tv.log" > "$OUT/bench_tv_under_rocprof.json"
kt topc python tools/topc_bw.py
pmc topc_pmc_fetch_size FETCH_SIZE -- python tools/topc_bw.py
pmc topc_pmc_write_size WRITE_SIZE -- python tools/topc_bw.py
ls -la "$OUT"
