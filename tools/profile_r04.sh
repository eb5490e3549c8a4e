#!/bin/bash
# rocprofv3 evidence of round 3, run ON THE GPU BOX:  bash tools/profile_r04.sh
# Kernel-trace summaries and PMC passes are separate runs (a --pmc run carries no trace domain except the kernel trace).
# Summaries land in gpurun_out/prof_r04/ ; the *.txt / *.json are copied into profiles/r04/ afterwards.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r04
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
# 1. the driver's own command (default flags), plain: the line BENCH_r04 should reproduce
( cd $ROOT && timeout 900 python bench.py 2> "$OUT/bench_n1.err" | grep '^{"metric"' > "$OUT/bench_n1.json" )
# 2. the same EM workload under the kernel trace and the three PMC passes
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
kt bench_em $BENCH
grep -h '^{"metric"' "$OUT/bench_em.log" > "$OUT/bench_em_under_rocprof.json"
pmc bench_em_pmc_fetch_size FETCH_SIZE -- $BENCH
pmc bench_em_pmc_write_size WRITE_SIZE -- $BENCH
pmc bench_em_pmc_sq SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- $BENCH
# 3. configs[3] on one GPU at the per-GPU size (6250 utterances), plain and traced on 2048 utterances
( cd $ROOT && timeout 900 python bench.py --workload tv --steps 3 --warmup 1 2> "$OUT/bench_tv_n1.err" | grep '^{"metric"' > "$OUT/bench_workload_tv_n1.json" )
TV="python bench.py --workload tv --steps 2 --warmup 1 --no-cpu-baseline --tv-utterances 2048"
kt bench_tv $TV
pmc bench_tv_pmc_fetch_size FETCH_SIZE -- $TV
# 4. the multi-rank orchestration on this one GPU (ranks share it over the C ABI's shm transport): a correctness record, not a scaling number
( cd $ROOT && timeout 900 python bench.py --gpus 2 --share-gpu --workload tv --tv-utterances 1024 --steps 2 --warmup 1 2> "$OUT/bench_tv_2ranks_shared.err" | grep '^{"metric"' > "$OUT/bench_tv_2ranks_shared_gpu.json" )
( cd $ROOT && timeout 900 python bench.py --gpus 2 --share-gpu --workload tv --tv-utterances 1024 --steps 2 --warmup 1 --overlap 1 2>> "$OUT/bench_tv_2ranks_shared.err" | grep '^{"metric"' > "$OUT/bench_tv_2ranks_shared_gpu_overlap.json" )
( cd $ROOT && timeout 900 python bench.py --gpus 2 --share-gpu --frames 2000000 --steps 2 --warmup 1 --no-secondary 2> "$OUT/bench_em_2ranks_shared.err" | grep '^{"metric"' > "$OUT/bench_em_2ranks_shared_gpu.json" )
# 5. the Cholesky family on its own (tools/chol_probe.hip), with the in-kernel phase stamps of one uncontended workgroup
( cd $ROOT && for nb in 32 256 1024; do tools/bin/chol_probe 400 $nb 5; done > "$OUT/chol_probe.txt" 2>&1; tools/bin/chol_probe_prof 400 32 2 >> "$OUT/chol_probe.txt" 2>&1 )
# 6. ComputeTest world pass
kt topc python tools/topc_bw.py
# 7. the secondary metric (IvExtractor on 512 utterances) and the host layer under the kernel trace: the whole default line without the CPU legs
kt bench_all python bench.py --steps 2 --warmup 1 --no-cpu-baseline
grep -h '^{"metric"' "$OUT/bench_all.log" > "$OUT/bench_all_under_rocprof.json"
# 8. TrainWorld through the C++ host layer, stage by stage (LIAGPU_TRACE) and the first-call allocation cost by scratch budget
( cd $ROOT && LIAGPU_TRACE=1 timeout 600 python tools/host_world_time.py 10000000 1.0 0.4 > "$OUT/host_world_time.txt" 2>&1 )
( cd $ROOT && timeout 600 python tools/alloc_time.py > "$OUT/alloc_time.txt" 2>&1 )
ls -la "$OUT"
# 9. the secondary metric on its own under the kernel trace
kt iv_secondary python tools/iv_secondary.py
grep -h '^{"metric"' "$OUT/iv_secondary.log" > "$OUT/iv_secondary_under_rocprof.json"
