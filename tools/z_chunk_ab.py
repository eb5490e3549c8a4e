#!/usr/bin/env python3
"""A/B inside one process: EM statistics pass (10 M frames) against the logit scratch budget."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
C, D, T = 2048, 60, 10_000_000
w, mean, iv = make_gmm(C, D, seed=0)
x = bench.synth_frames(w, mean, iv, T, torch.device("cuda", 0), seed=1)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1)
g = ctx.gmm(w, mean, iv)
acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device="cuda")
for rep in range(2):
    for mb in (4096, 16384, 65536, 131072):
        ctx.set_option("z_scratch_mb", mb)
        g.em_accumulate(x, acc=acc); torch.cuda.synchronize()
        t0 = time.perf_counter(); g.em_accumulate(x, acc=acc); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("budget %6d MiB: %d launches, total %.1f ms, k_llk %.1f ms, k_stats_z %.1f ms" % (
            mb, ctx.kernel_launches("k_stats_z"), dt * 1e3, ctx.kernel_ms("k_llk_mfma"), ctx.kernel_ms("k_stats_z")))
g.close(); ctx.close()
