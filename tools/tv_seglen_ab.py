#!/usr/bin/env python3
"""k_stats_z in N / F mode against the utterance length (segments = utterances): the same 3 M frames cut into utterances of 3000 (BASELINE
config 3), 3008 (a multiple of the 64-frame tile), 1000, 30000 frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
C, D = 2048, 60
w, mean, iv = make_gmm(C, D, seed=0)
dev = torch.device("cuda", 0)
T = 3_080_192
x = bench.synth_frames(w, mean, iv, T, dev, seed=1)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1)
g = ctx.gmm(w, mean, iv)
for frames in (3000, 3008, 1000, 30000, 3000):
    U = T // frames
    ub = np.arange(U + 1, dtype=np.int64) * frames
    N = torch.empty((U, C), dtype=torch.float64, device=dev); F = torch.empty((U, C * D), dtype=torch.float64, device=dev)
    g.tv_stats(x, ub, N, F); g.tv_stats(x, ub, N, F); torch.cuda.synchronize()
    ms, ms1 = ctx.kernel_ms("k_stats_z"), ctx.kernel_ms("k_llk_mfma")
    n = U * frames
    print("%6d frames x %5d utterances: k_stats_z %.2f ms (%.1f Gpair/s, %.2f of the MFMA peak at 128 flop/pair), k_llk_mfma %.2f ms (%.1f Gpair/s)" % (
        frames, U, ms, n * C / ms / 1e6, n * C * 128 / ms / 1e9 / 78.6, ms1, n * C / ms1 / 1e6))
