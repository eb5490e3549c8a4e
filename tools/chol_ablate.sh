#!/bin/bash
# Where does k_chol_left spend its time?  Builds libgmmiv with parts of the kernel compiled out (-DCHOL_ABL=bits, chol_fused.hip; results
# are WRONG, timing only) and reads the kernel's average duration from a rocprofv3 kernel trace of one T-matrix E-step batch.
#   bash tools/chol_ablate.sh build     (here)          bash tools/chol_ablate.sh run     (on the GPU box)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CS=$ROOT/lia_ral_amd/csrc
VARIANTS=${VARIANTS:-"0 1 2 4 8 15"}
if [ "$1" = build ]; then
    mkdir -p $CS/abl
    for v in $VARIANTS; do
        ( cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DCHOL_ABL=$v -c chol_fused.hip -o abl/chol_$v.o &&
          /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libgmmiv_c$v.so abl/chol_$v.o $(ls *.o | grep -v chol_fused.o) -ldl && rm abl/chol_$v.o ) &
    done
    wait; ls $CS/abl
elif [ "$1" = run ]; then
    for v in $VARIANTS; do
        GMMIV_LIB_PATH=$CS/abl/libgmmiv_c$v.so LINES_OUT=40 bash $ROOT/tools/prof_cmd.sh cabl_$v python $ROOT/tools/estep_prof.py 2>/dev/null | grep "k_chol_left" | awk -v v=$v '{print "CHOL_ABL=" v, "k_chol_left avg ns per 1024 systems:", $(NF-3), "calls", $(NF-5)}'
    done | tee $ROOT/gpurun_out/chol_ablate.txt
fi
