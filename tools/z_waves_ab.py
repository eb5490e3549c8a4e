#!/usr/bin/env python3
"""In-process A/B of k_stats_z workgroup shapes (z_waves 8 / 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
C, D, T = 2048, 60, 4_000_000
w, mean, iv = make_gmm(C, D, seed=0)
x = bench.synth_frames(w, mean, iv, T, torch.device("cuda", 0), seed=1)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1)
g = ctx.gmm(w, mean, iv)
acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device="cuda")
ref = None
for rep in range(2):
    for zw in (8, 4, 16):
        ctx.set_option("z_waves", zw)
        acc.zero_(); g.em_accumulate(x, acc=acc); acc.zero_(); g.em_accumulate(x, acc=acc)
        a = acc.cpu().numpy().copy()
        if ref is None: ref = a
        ms = ctx.kernel_ms("k_stats_z")
        print("z_waves %d: k_stats_z %.2f ms (%.1f TF MFMA-rate)  max rel diff %.1e" % (
            zw, ms, 256.0 * T * C / ms / 1e9, np.max(np.abs(a - ref)) / np.max(np.abs(ref))))
g.close(); ctx.close()
