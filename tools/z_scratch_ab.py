#!/usr/bin/env python3
"""EM pass (10 M frames, 2048 x 60) and IvExtractor statistics against the likelihood-scratch budget (option z_scratch_mb): fewer, larger
chunks = fewer launch boundaries.  usage: python tools/z_scratch_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev); torch.cuda.set_stream(side)
ctx = capi.Context(0, side.cuda_stream); ctx.set_option("timing", 1)
w, mean, iv = make_gmm(bench.C, bench.D, seed=0)
g = ctx.gmm(w, mean, iv)
T = 10_000_000
x = bench.synth_frames(w, mean, iv, T, dev, seed=1234)
acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device=dev)
ref = None
for mb in (16384, 32768, 65536, 16384):
    ctx.set_option("z_scratch_mb", mb)
    g.em_accumulate(x, acc=acc); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        acc.zero_(); t0 = time.perf_counter(); g.em_accumulate(x, acc=acc); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    a = acc.cpu().numpy().copy()
    if ref is None: ref = a
    print("z_scratch_mb %6d: %.2f ms  (%s)  launches %d  K1 %.2f K2 %.2f  max rel diff vs 16384: %.2e" % (mb, 1e3 * np.median(ts), " ".join("%.1f" % (1e3 * t) for t in ts),
          ctx.kernel_launches("k_llk_mfma"), ctx.kernel_ms("k_llk_mfma"), ctx.kernel_ms("k_stats_z"), np.max(np.abs(a - ref)) / np.max(np.abs(ref))), flush=True)
