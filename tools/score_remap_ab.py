#!/usr/bin/env python3
"""A/B of k_dgemm's tile order ("gemm_remap" 1 = an XCD walks its N-tile columns M-fastest, 2 = in 8 x 8 tile blocks) on the scoring
GEMM (M x M trials of dimension 400) and on the T-matrix E-step (tools/bench_tv.py shapes are timed by bench.py --workload tv).
usage: python tools/score_remap_ab.py [M]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lia_ral_amd import capi
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
R = 400
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev); torch.cuda.set_stream(side)
ctx = capi.Context(0, side.cuda_stream); ctx.set_option("timing", 1)
gen = torch.Generator(device=dev); gen.manual_seed(11)
models = torch.randn((R, M), dtype=torch.float64, device=dev, generator=gen)
segs = torch.randn((R, M), dtype=torch.float64, device=dev, generator=gen)
Q = torch.randn((R, R), dtype=torch.float64, device=dev, generator=gen)
Mah = (Q @ Q.T / R + torch.eye(R, dtype=torch.float64, device=dev)).contiguous()
scores = torch.empty((M, M), dtype=torch.float64, device=dev)
ref = None
for rm in (1, 2, 1, 2):
    ctx.set_option("gemm_remap", rm)
    ctx.score_mahalanobis(models, segs, Mah, out=scores); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ctx.score_mahalanobis(models, segs, Mah, out=scores); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = min(ts)
    cs = float(scores[::997, ::991].sum().item())
    print("gemm_remap %d: %.2f ms = %.1f G trials/s (%.1f TF), scoring GEMM %.2f ms, checksum %.12e" % (rm, dt * 1e3, M * M / dt / 1e9, 800.0 * M * M / dt / 1e12, ctx.kernel_ms("k_dgemm(score)"), cs), flush=True)
