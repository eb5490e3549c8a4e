#!/usr/bin/env python3
"""The IvExtractor and scoring blocks of bench.py as fixed, countable work for the rocprofv3 passes of tools/profile_r05.sh:
  python tools/pmc_blocks.py iv U PASSES      -- PASSES passes of (tv_stats + substractM + estimateW) over U utterances x 3000 frames
                                                 (TETt computed once before; nothing else launches after the setup marker)
  python tools/pmc_blocks.py score M PASSES    -- PASSES calls of gmmiv_score_mahalanobis on M x M vectors of dimension 400
tools/make_traffic.py divides the counter totals of the named kernels by PASSES (x U)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench

what, n, passes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev); torch.cuda.set_stream(side)
ctx = capi.Context(0, side.cuda_stream); ctx.set_option("assume_finite", 1)
C, D, R = bench.C, bench.D, 400
if what == "iv":
    w, mean, iv = make_gmm(C, D, seed=0)
    g = ctx.gmm(w, mean, iv)
    frames = 3000
    x = bench.synth_frames(w, mean, iv, n * frames, dev, seed=777)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    Tm = 0.01 * torch.randn((R, C * D), dtype=torch.float64, device=dev, generator=gen)
    invvar = torch.from_numpy(iv.ravel().copy()).to(dev); means = torch.from_numpy(mean.ravel().copy()).to(dev)
    tett = torch.empty((C, R * (R + 1) // 2), dtype=torch.float64, device=dev)
    ctx.tv_tett(Tm, invvar, C, D, out=tett)
    N = torch.empty((n, C), dtype=torch.float64, device=dev); F = torch.empty((n, C * D), dtype=torch.float64, device=dev)
    W = torch.empty((n, R), dtype=torch.float64, device=dev)
    ub = np.arange(n + 1, dtype=np.int64) * frames
    for _ in range(passes):
        g.tv_stats(x, ub, N, F)
        ctx.tv_subtract_m(N, F, means, C, D)
        ctx.tv_estimate_w(N, F, Tm, invvar, tett, C, D, out=W)
    torch.cuda.synchronize()
    print("iv: %d utterances x %d passes, finite %s" % (n, passes, bool(torch.isfinite(W).all().item())))
else:
    gen = torch.Generator(device=dev); gen.manual_seed(11)
    models = torch.randn((R, n), dtype=torch.float64, device=dev, generator=gen)
    segs = torch.randn((R, n), dtype=torch.float64, device=dev, generator=gen)
    Q = torch.randn((R, R), dtype=torch.float64, device=dev, generator=gen)
    Mah = (Q @ Q.T / R + torch.eye(R, dtype=torch.float64, device=dev)).contiguous()
    scores = torch.empty((n, n), dtype=torch.float64, device=dev)
    for _ in range(passes):
        ctx.score_mahalanobis(models, segs, Mah, out=scores)
    torch.cuda.synchronize()
    print("score: %d x %d x %d passes, checksum %.6e" % (n, n, passes, float(scores[:1000, :1000].sum().item())))
