#!/usr/bin/env python3
"""A/B of option "tv_overlap" (gmmiv_tv_stats: the log-likelihood kernel of chunk k + 1 beside the N / F statistics kernel of chunk k
on a side stream; 2 = the statistics kernel in its 4-wave / 51 KB shape so that both kernels fit one CU): wall time of the statistics
pass over U utterances x 3000 frames, results compared bitwise with the serial form.  usage: python tools/tv_overlap_ab.py [U]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
C, D, frames = 2048, 60, 3000
U = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev); torch.cuda.set_stream(side)
ctx = capi.Context(0, side.cuda_stream); ctx.set_option("assume_finite", 1); ctx.set_option("timing", 1)
w, mean, iv = make_gmm(C, D, seed=0)
g = ctx.gmm(w, mean, iv)
x = bench.synth_frames(w, mean, iv, U * frames, dev, seed=777)
ub = np.arange(U + 1, dtype=np.int64) * frames
ref = None
for mode in (0, 1, 2, 0, 1, 2):
    ctx.set_option("tv_overlap", mode)
    N = torch.empty((U, C), dtype=torch.float64, device=dev); F = torch.empty((U, C * D), dtype=torch.float64, device=dev)
    g.tv_stats(x, ub, N, F); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); g.tv_stats(x, ub, N, F); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = float(np.mean(ts))
    if ref is None:
        ref = (N.clone(), F.clone())
    print("tv_overlap %d: %.2f ms per pass over %d utterances (%.3f ms per 10^6 frames), N / F bitwise the serial form's: %s" % (
        mode, dt * 1e3, U, dt * 1e3 / (U * frames / 1e6), bool(torch.equal(N, ref[0]) and torch.equal(F, ref[1]))), flush=True)
    del N, F
