#!/bin/bash
# rocprofv3 kernel-trace summary of an arbitrary command (run on the GPU box): bash tools/prof_cmd.sh <tag> <cmd...>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd $ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o t -- "$@" > "$OUT/kt.log" 2>&1 )
python $ROOT/tools/rocpd_summary.py "$(find $OUT/kt -name '*.db' | head -1)" > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/kt"; tail -1 "$OUT/kt.log"; head -${LINES_OUT:-16} "$OUT/kernel_stats.txt"
