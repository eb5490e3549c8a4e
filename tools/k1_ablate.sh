#!/bin/bash
# Where does k_llk_mfma<WZ> spend the time next to its MFMAs?  Builds libgmmiv with parts of the kernel's epilogue compiled out
# (-DK1_ABL=bits, see gmm_kernels.hip; results are WRONG, timing only) and times the EM bench's kernels with each.
#   bash tools/k1_ablate.sh build      (here, cross-compiles)          bash tools/k1_ablate.sh run   (on the GPU box)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CS=$ROOT/lia_ral_amd/csrc
VARIANTS=${VARIANTS:-"0 1 2 4 8 10 14 16 17"}
if [ "$1" = build ]; then
    mkdir -p $CS/abl
    for v in $VARIANTS; do
        ( cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DK1_ABL=$v ${EXTRA:-} -c gmm_kernels.hip -o abl/gmm_$v.o &&
          /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libgmmiv_$v.so abl/gmm_$v.o $(ls *.o | grep -v gmm_kernels.o) -ldl && rm abl/gmm_$v.o ) &
    done
    wait; ls -la $CS/abl
elif [ "$1" = run ]; then
    mkdir -p $ROOT/gpurun_out
    for v in $VARIANTS; do
        GMMIV_LIB_PATH=$CS/abl/libgmmiv_$v.so python $ROOT/bench.py --frames ${FRAMES:-4000000} --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null |
            python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']; print('K1_ABL=%-3s k_llk_mfma %.3f ms/launch (%.1f Gpair/s, frac %.3f)   k_stats_z %.3f ms' % ('$v', k['k_llk_mfma']['ms_per_launch'], k['k_llk_mfma']['gpairs_per_s'], k['k_llk_mfma']['tflops'] / 78.6, k['k_stats_z']['ms_per_launch']))"
    done | tee $ROOT/gpurun_out/k1_ablate.txt
fi
# TC variant (k_llk_mfma<..., 2>, bits 64 / 128 / 256): the instrumented builds make the ranking kernel reject frames, so the kernel is timed
# by name from a rocprofv3 kernel trace of tools/topc_bw.py:   VARIANTS="0 64 128 256 448" bash tools/k1_ablate.sh build && ... bash tools/k1_ablate.sh topc
if [ "$1" = topc ]; then
    for v in $VARIANTS; do
        GMMIV_LIB_PATH=$CS/abl/libgmmiv_$v.so LINES_OUT=30 bash $ROOT/tools/prof_cmd.sh abl_$v python $ROOT/tools/topc_bw.py 2>/dev/null | grep "k_llk_mfma<15, float, 8, 2>" | awk -v v=$v '{print "K1_ABL=" v, "k_llk_mfma<TC> avg ns per launch:", $(NF-3)}'
    done | tee $ROOT/gpurun_out/k1_ablate_topc.txt
fi
