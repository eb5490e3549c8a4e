#!/usr/bin/env python3
"""Ablation timing of k_stats_z (dbg >> 4: 1 no exp, 2 no logit loads, 3 no staging/barrier) and K1 with logit stores."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
C, D, T = 2048, 60, 2_000_000
w, mean, iv = make_gmm(C, D, seed=0)
x = bench.synth_frames(w, mean, iv, T, torch.device("cuda", 0), seed=1)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1)
g = ctx.gmm(w, mean, iv)
acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device="cuda")
for abl in (0, 1, 2, 3):
    ctx.set_option("dbg", abl << 4)
    g.em_accumulate(x, acc=acc); g.em_accumulate(x, acc=acc)
    k1, k2 = ctx.kernel_ms("k_llk_mfma"), ctx.kernel_ms("k_stats_z")
    print("abl %d: k_llk %.2f ms  k_stats_z %.2f ms (%d launches)  -> %.1f TF MFMA-rate" % (
        abl, k1, k2, ctx.kernel_launches("k_stats_z"), 256.0 * T * C / (k2 * 1e-3) / 1e12))
ctx.set_option("dbg", 0)
ctx.set_option("stats_z", 0)
g.em_accumulate(x, acc=acc); g.em_accumulate(x, acc=acc)
print("recompute path: k_llk %.2f ms  k_stats_mfma %.2f ms" % (ctx.kernel_ms("k_llk_mfma"), ctx.kernel_ms("k_stats_mfma")))
g.close(); ctx.close()
