#!/bin/bash
# every (value, entry point, path) in its own process
for v in nan inf -inf 1e30 row; do
  for what in top em tv occ; do
    for o in "" "topc_fused=0" "topc_fused=0 topc_z=0" "stats_z=0" "wg_waves=4"; do
      case "$what:$o" in top:stats_z=0|em:topc*|tv:topc*|occ:topc*) continue;; esac
      timeout 150 python tools/degenerate_case.py $v $what $o 2>&1 | grep -v amdgpu.ids | tail -2
    done
  done
done
