#!/usr/bin/env python3
"""A/B of the N / F statistics (gmmiv_tv_stats) shapes of k_stats_z: option z_tv4 (4 Gaussian tiles per wave) on / off."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
sys.path.insert(0, ROOT)
from bench import synth_frames
dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1)
C, D, U, frames = 2048, 60, 512, 3000
w, mean, iv = make_gmm(C, D, seed=0, spread=2.0)
x = synth_frames(w, mean, iv, U * frames, dev, seed=777)
g = ctx.gmm(w, mean, iv)
ub = np.arange(U + 1, dtype=np.int64) * frames
N = torch.empty((U, C), dtype=torch.float64, device=dev); F = torch.empty((U, C * D), dtype=torch.float64, device=dev)
out = {}
ref = None
for mode in (0, 1, 0, 1):
    ctx.set_option("z_tv4", mode)
    g.tv_stats(x, ub, N, F); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g.tv_stats(x, ub, N, F)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    out.setdefault("tv4" if mode else "tv2", []).append({"ms": round(ms, 3), "k_stats_z_ms": round(ctx.kernel_ms("k_stats_z"), 3), "k_llk_ms": round(ctx.kernel_ms("k_llk_mfma"), 3)})
    if ref is None:
        ref = (N.clone(), F.clone())
    else:
        out["max_rel_diff"] = max(float(((N - ref[0]).abs().max() / ref[0].abs().max()).item()), float(((F - ref[1]).abs().max() / ref[1].abs().max()).item()))
print(json.dumps(out))
