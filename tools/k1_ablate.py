#!/usr/bin/env python3
"""Ablation timing of the LLK kernel (dbg option): where does the non-MFMA time go?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
C, D, T = 2048, 60, 4_000_000
for spread in (2.0, 0.3):
    w, mean, iv = make_gmm(C, D, seed=0, spread=spread)
    x = bench.synth_frames(w, mean, iv, T, torch.device("cuda", 0), seed=1)
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_option("timing", 1)
    g = ctx.gmm(w, mean, iv)
    out = torch.empty(T, dtype=torch.float64, device="cuda")
    for nw in (8, 4):
        ctx.set_option("wg_waves", nw)
        for dbg in (0, 1, 2):
            ctx.set_option("dbg", dbg)
            for glds in (1, 0):
                ctx.set_option("glds", glds)
                g.llk(x, -1e9, 1e9, out=out); g.llk(x, -1e9, 1e9, out=out)
                ms = ctx.kernel_ms("k_llk_mfma")
                print("spread %.1f nw %d dbg %d glds %d: %.2f ms  %.1f TF" % (spread, nw, dbg, glds, ms, 240.0 * T * C / ms / 1e9))
    ctx.set_option("dbg", 0)
    g.close(); ctx.close()
