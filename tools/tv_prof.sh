#!/bin/bash
# per-kernel breakdown of one phase of tools/bench_tv.py (run on the GPU box): bash tools/tv_prof.sh <phase> [U]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PH=${1:-estep}; export TV_U=${2:-1024}; export TV_PHASES=$PH
OUT=$ROOT/gpurun_out/tvprof_$PH; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o tv -- python $ROOT/tools/bench_tv.py > "$OUT/kt.log" 2>&1
python $ROOT/tools/rocpd_summary.py "$(find $OUT/kt -name '*.db' | head -1)" > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/kt"; tail -1 "$OUT/kt.log"; head -24 "$OUT/kernel_stats.txt"
