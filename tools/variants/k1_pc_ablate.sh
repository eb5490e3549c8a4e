#!/bin/bash
# Where does k_llk_pc (llk_pc.hip) lose against its own MFMA stream?  Builds libgmmiv with parts of the kernel compiled out
# (-DK1PC_ABL=bits / -DK1PC_PRIO=0; results WRONG when bits != 0) and times them with tools/k1_pc_ab.py.
#   bash tools/k1_pc_ablate.sh build   (cross-compiles here)      bash tools/k1_pc_ablate.sh run   (on the GPU box)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CS=$ROOT/lia_ral_amd/csrc
VARIANTS=${VARIANTS:-"0 1 2 3 4 8 p0"}
if [ "$1" = build ]; then
    mkdir -p $CS/abl
    for v in $VARIANTS; do
        if [ "$v" = p0 ]; then DEF="-DK1PC_PRIO=0"; else DEF="-DK1PC_ABL=$v"; fi
        ( cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $DEF ${EXTRA:-} -c llk_pc.hip -o abl/pc_$v.o &&
          /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libgmmiv_pc_$v.so abl/pc_$v.o $(ls *.o | grep -v llk_pc.o) -ldl && rm abl/pc_$v.o ) &
    done
    wait; ls -la $CS/abl
elif [ "$1" = run ]; then
    for v in $VARIANTS; do
        echo "== variant $v"
        GMMIV_LIB_PATH=$CS/abl/libgmmiv_pc_$v.so python $ROOT/tools/k1_pc_ab.py ${FRAMES:-3072000} 2>/dev/null | grep "k1_pc 1" | tail -1
    done | tee $ROOT/gpurun_out/k1_pc_ablate.txt
fi
