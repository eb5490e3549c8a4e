#!/bin/bash
# bash tools/chol_probe.sh build [extra -D flags]   -> tools/bin/chol_probe (+ tools/bin/chol_probe_prof with the in-kernel time stamps)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/lia_ral_amd/csrc
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function"
mkdir -p $ROOT/tools/bin
OTHERS=$(ls $CS/*.o | grep -v chol_fused.o)
shift || true
$HIPCC $FLAGS "$@" -c $CS/chol_fused.hip -o /tmp/chol_plain.o &
$HIPCC $FLAGS "$@" -DCHOL_PROF=1 -c $CS/chol_fused.hip -o /tmp/chol_prof.o &
wait
$HIPCC $FLAGS -c $ROOT/tools/chol_probe.hip -o /tmp/chol_probe_main.o
$HIPCC --offload-arch=gfx950 /tmp/chol_probe_main.o /tmp/chol_plain.o $OTHERS -o $ROOT/tools/bin/chol_probe -ldl &
$HIPCC --offload-arch=gfx950 /tmp/chol_probe_main.o /tmp/chol_prof.o $OTHERS -o $ROOT/tools/bin/chol_probe_prof -ldl &
wait
ls -la $ROOT/tools/bin/chol_probe*
