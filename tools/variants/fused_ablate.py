#!/usr/bin/env python3
"""Ablation timing of the fused EM kernel (dbg bits: 1 no exchange, 2 no phase-A reductions, 4 no stats MFMAs)."""
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
ctx.set_option("timing", 1); ctx.set_option("em_fused", 1)
g = ctx.gmm(w, mean, iv)
acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device="cuda")
for dbg in (0, 1):
    ctx.set_option("dbg", dbg)
    g.em_accumulate(x, acc=acc); g.em_accumulate(x, acc=acc)
    ms = ctx.kernel_ms("k_em_fused")
    print("dbg %d: %.2f ms  (%.1f Gpair/s)" % (dbg, ms, T * C / ms / 1e6))
ctx.set_option("dbg", 0)
g.close(); ctx.close()
