#!/bin/bash
# rocprofv3 SQ counters (MFMA / VALU instruction counts, busy cycles) of an arbitrary command, kernel trace only: bash tools/pmc_cmd.sh <tag> <cmd...>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd $ROOT && timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d "$OUT/sq" -o t -- "$@" > "$OUT/sq.log" 2>&1 )
python $ROOT/tools/rocpd_summary.py "$(find $OUT/sq -name '*.db' | head -1)" --pmc > "$OUT/pmc_sq.txt" 2>&1
rm -rf "$OUT/sq"; head -${LINES_OUT:-40} "$OUT/pmc_sq.txt"
