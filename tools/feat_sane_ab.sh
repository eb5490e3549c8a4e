#!/bin/bash
# A/B of the in-kernel handling of unusable feature values (csrc/devutil.h feat_sane): builds a second libgmmiv WITHOUT the compare + select
# (tools/bin/libgmmiv_nosane.so, -DGMMIV_FEAT_SANE_OFF) and runs the i-vector block and the EM headline on both, alternating.
# usage: tools/feat_sane_ab.sh build   (here, cross-compiles)   |   tools/feat_sane_ab.sh run   (on the GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p tools/bin/nosane
  for f in capi_gmm gmm_kernels stats_z capi_tv tv_kernels chol_fused topc_z capi_comm; do
    [ -f lia_ral_amd/csrc/$f.hip ] || continue
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DGMMIV_FEAT_SANE_OFF -c lia_ral_amd/csrc/$f.hip -o tools/bin/nosane/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libgmmiv_nosane.so tools/bin/nosane/*.o -ldl
  exit 0
fi
for i in 1 2 3; do
  for lib in "" "$PWD/tools/bin/libgmmiv_nosane.so"; do
    export GMMIV_LIB_PATH="$lib"
    python tools/iv_secondary.py 2>/dev/null | python -c "
import json,sys,os
s=json.load(sys.stdin); p=s['roofline']['parts']
print('%-8s i-vectors/s %.0f  runs %s  k1 %.1f  k3 %.1f' % ('nosane' if os.environ.get('GMMIV_LIB_PATH') else 'product', s['value'], ['%.1f'%r for r in s['timed_runs_ms']], p['k_llk_mfma']['ms'], p['k_stats_z(N,F)']['ms']))"
  done
done
