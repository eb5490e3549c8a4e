#!/usr/bin/env python3
"""First-call cost of the likelihood scratch: hipMalloc time by size (GMMIV_TRACE_ALLOC=1 prints every workspace allocation) and the
EM pass rate against the scratch budget."""
import os, sys, time
os.environ["GMMIV_TRACE_ALLOC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
C, D, T = 2048, 60, 10_000_000
w, mean, iv = make_gmm(C, D, seed=0)
x = bench.synth_frames(w, mean, iv, T, torch.device("cuda", 0), seed=1)
for mb in (8192, 16384, 32768, 65536):
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_option("z_scratch_mb", mb); ctx.set_option("timing", 1)
    g = ctx.gmm(w, mean, iv)
    acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.em_accumulate(x, acc=acc); torch.cuda.synchronize(); first = time.perf_counter() - t0
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); g.em_accumulate(x, acc=acc); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("budget %6d MiB: first call %.1f ms, then %.1f ms per pass (%d launches): %.1f G pairs/s" % (
        mb, first * 1e3, np.mean(ts) * 1e3, ctx.kernel_launches("k_stats_z"), T * C / np.mean(ts) / 1e9), flush=True)
    g.close(); ctx.close()
