#!/usr/bin/env python3
"""A/B of the stream prefetch depth of k_stats_z (options z_depth_em / z_depth_tv: 2 or 4 register sets): EM statistics on 4 M frames and
N / F statistics of 1024 x 3000-frame utterances, 2048 Gaussians x 60 dims; results must be bit-identical."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
from bench import synth_frames
dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1)
C, D = 2048, 60
w, mean, iv = make_gmm(C, D, seed=0, spread=2.0)
g = ctx.gmm(w, mean, iv)
T = 4_000_000
x = synth_frames(w, mean, iv, T, dev, seed=1)
U, frames = 1024, 3000
ub = np.arange(U + 1, dtype=np.int64) * frames
N = torch.empty((U, C), dtype=torch.float64, device=dev); F = torch.empty((U, C * D), dtype=torch.float64, device=dev)
acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device=dev)
ref = None
for depth, tv4 in ((2, 1), (4, 0), (2, 1), (4, 0)):
    ctx.set_option("z_depth_em", depth); ctx.set_option("z_depth_tv", depth); ctx.set_option("z_tv4", tv4)
    for rep in range(2):
        acc.zero_(); g.em_accumulate(x, acc=acc); torch.cuda.synchronize()
    em = ctx.kernel_ms("k_stats_z")
    for rep in range(2):
        g.tv_stats(x[:U * frames], ub, N, F); torch.cuda.synchronize()
    tv = ctx.kernel_ms("k_stats_z")
    cur = (acc.clone(), N.clone(), F.clone())
    if ref is None:
        ref = cur
    diff = max(float((a - b).abs().max().item()) for a, b in zip(cur, ref))
    print("depth %d: EM k_stats_z %.2f ms (%.1f Gpair/s, %.3f of peak at 256 flop/pair) | N/F k_stats_z %.2f ms (%.1f Gpair/s, %.2f TB/s) | max abs diff vs first %.1e" % (
        depth, em, T * C / em / 1e6, T * C * 256 / em / 1e9 / 78.6, tv, U * frames * C / tv / 1e6, U * frames * C * 8 / tv / 1e9, diff))
