#!/bin/bash
# Re-measure everything the scoreboards of DESIGN.md / README.md quote, ON THE GPU BOX:  bash tools/refresh_evidence.sh [rNN]
# Output: gpurun_out/evidence_rNN/ (copy what is to be kept into profiles/rNN/).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
R=${1:-r02}
OUT=$ROOT/gpurun_out/evidence_$R
mkdir -p "$OUT"; cd "$ROOT"
j() { grep -h '^{' | tail -1; }
timeout 900 python bench.py --steps 10 --warmup 3 2> "$OUT/bench_n1.err" | j > "$OUT/bench_n1.json"
timeout 900 python bench.py --workload tv --steps 4 --warmup 1 2> "$OUT/bench_tv.err" | j > "$OUT/bench_workload_tv_n1.json"
timeout 900 python tools/bench_tv.py > "$OUT/bench_tv_tool.json" 2> "$OUT/bench_tv_tool.err"
timeout 900 python tools/host_tv_time.py > "$OUT/host_layer_tv_time.json" 2> "$OUT/host_tv.err"
timeout 900 python tools/chol_lds_ab.py > "$OUT/chol_lds_ab.txt" 2>&1
timeout 900 python tools/topc_bw.py > "$OUT/topc_bw.txt" 2>&1
timeout 1500 python tools/run_configs.py > "$OUT/configs_3_and_5_full_size.json" 2> "$OUT/run_configs.err"
[ -x tools/bin/gemm_probe ] && timeout 300 tools/bin/gemm_probe > "$OUT/gemm_probe.txt" 2>&1
ls -la "$OUT"; tail -3 "$OUT"/*.err
