#!/usr/bin/env python3
"""In-process A/B: likelihood scratch tile stride padded to an odd number of 4 KB granules (default) or not (dbg 64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
C, D, T = 2048, 60, 6_000_000
w, mean, iv = make_gmm(C, D, seed=0)
x = bench.synth_frames(w, mean, iv, T, torch.device("cuda", 0), seed=1)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1)
g = ctx.gmm(w, mean, iv)
acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device="cuda")
for rep in range(2):
    for dbg in (0, 64):
        ctx.set_option("dbg", dbg)
        acc.zero_(); g.em_accumulate(x, acc=acc); acc.zero_(); g.em_accumulate(x, acc=acc)
        print("dbg %d: k_llk %.2f ms  k_stats_z %.2f ms (%d launches)" % (dbg, ctx.kernel_ms("k_llk_mfma"), ctx.kernel_ms("k_stats_z"), ctx.kernel_launches("k_stats_z")))
ctx.set_option("dbg", 0)
g.close(); ctx.close()
