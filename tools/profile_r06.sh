#!/bin/bash
# rocprofv3 evidence of round 6, run ON THE GPU BOX:  bash tools/profile_r06.sh
# Kernel-trace summaries and PMC passes are separate runs (a --pmc run carries no trace domain except the kernel trace).
# Summaries land in gpurun_out/prof_r06/ ; tools/make_traffic.py (run in the build container afterwards) turns the PMC summaries
# into profiles/traffic.json, stamped with the sha256 of the libgmmiv.so that ran here; the *.txt / *.json are copied to profiles/r06/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r06
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
sha256sum $ROOT/lia_ral_amd/csrc/libgmmiv.so | cut -d' ' -f1 > "$OUT/libgmmiv_sha256.txt"
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
# 1. the driver's own command, plain: the line BENCH_r06 should reproduce
( cd $ROOT && timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2> "$OUT/bench_n1.err" | grep '^{"metric"' > "$OUT/bench_n1.json" )
# 2. the whole default line under the kernel trace (every block's kernels: EM, IvExtractor at 10 k x 3 k, ComputeTest, host layer, T-matrix EM, scoring)
kt bench_all python bench.py --steps 3 --warmup 1 --no-cpu-baseline
grep -h '^{"metric"' "$OUT/bench_all.log" > "$OUT/bench_all_under_rocprof.json"
# 3. the EM headline alone: kernel trace + the PMC passes (HBM bytes; instruction mix)
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
kt bench_em $BENCH
grep -h '^{"metric"' "$OUT/bench_em.log" > "$OUT/bench_em_under_rocprof.json"
pmc bench_em_pmc_fetch_size FETCH_SIZE -- $BENCH
pmc bench_em_pmc_write_size WRITE_SIZE -- $BENCH
pmc bench_em_pmc_sq SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- $BENCH
# 4. IvExtractor (configs[2]) as countable passes: 2 passes over 2560 utterances x 3000 frames (10 chunks of the 16 GiB scratch each)
IV="python tools/pmc_blocks.py iv 2560 2"
kt iv $IV
pmc iv_pmc_fetch_size FETCH_SIZE -- $IV
pmc iv_pmc_write_size WRITE_SIZE -- $IV
# 5. T-matrix EM (configs[3] per-GPU share is 6250 utterances; the counters on 2048): 3 iterations (2 timed + 1 warm-up)
TV="python bench.py --workload tv --steps 2 --warmup 1 --no-cpu-baseline --tv-utterances 2048"
kt bench_tv $TV
pmc bench_tv_pmc_fetch_size FETCH_SIZE -- $TV
pmc bench_tv_pmc_write_size WRITE_SIZE -- $TV
( cd $ROOT && timeout 900 python bench.py --workload tv --steps 3 --warmup 1 2> "$OUT/bench_tv_n1.err" | grep '^{"metric"' > "$OUT/bench_workload_tv_n1.json" )
# 6. scoring (configs[4]): 2 Mahalanobis calls at 100 k x 100 k
SC="python tools/pmc_blocks.py score 100000 2"
kt score $SC
pmc score_pmc_fetch_size FETCH_SIZE -- $SC
pmc score_pmc_write_size WRITE_SIZE -- $SC
# 7. the Cholesky family on its own (tools/chol_probe.hip)
( cd $ROOT && for nb in 32 256 1024; do tools/bin/chol_probe 400 $nb 5; done > "$OUT/chol_probe.txt" 2>&1 )
# 8. ComputeTest world pass
kt topc python tools/topc_bw.py
ls -la "$OUT"
