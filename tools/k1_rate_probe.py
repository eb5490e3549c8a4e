#!/usr/bin/env python3
"""k_llk_mfma<WZ> per frame in the EM call and in the N / F call, seed model, by call length (why is IvExtractor's K1 slower per frame?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
C, D = 2048, 60
w, mean, iv = make_gmm(C, D, seed=0)
dev = torch.device("cuda", 0)
x = bench.synth_frames(w, mean, iv, 4_000_000, dev, seed=777)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1); ctx.set_option("assume_finite", 1)
g = ctx.gmm(w, mean, iv)
acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device=dev)
for T in (786432, 768000, 1536000, 3145728):
    for rep in range(3):
        acc.zero_(); g.em_accumulate(x[:T], acc=acc); torch.cuda.synchronize()
    k1, k2, n1 = ctx.kernel_ms("k_llk_mfma"), ctx.kernel_ms("k_stats_z"), ctx.kernel_launches("k_llk_mfma")
    U = T // 3000
    N = torch.empty((U, C), dtype=torch.float64, device=dev); F = torch.empty((U, C * D), dtype=torch.float64, device=dev)
    ub = np.arange(U + 1, dtype=np.int64) * 3000
    for rep in range(3):
        g.tv_stats(x[:U * 3000], ub, N, F); torch.cuda.synchronize()
    t1, t3, m1 = ctx.kernel_ms("k_llk_mfma"), ctx.kernel_ms("k_stats_z"), ctx.kernel_launches("k_llk_mfma")
    print("T %8d: EM  K1 %.3f ms/Mframe (%d launches)  K2 %.3f | TV (%d utt) K1 %.3f ms/Mframe (%d launches)  K3 %.3f" % (
        T, k1 / T * 1e6, n1, k2 / T * 1e6, U, t1 / (U * 3000) * 1e6, m1, t3 / (U * 3000) * 1e6), flush=True)
    del N, F

# what precedes K1 matters?  EM's K1 right after a tv_stats call (K3 last) against EM's K1 right after an EM call (K2 last)
T = 768000; U = 256
N = torch.empty((U, C), dtype=torch.float64, device=dev); F = torch.empty((U, C * D), dtype=torch.float64, device=dev)
ub = np.arange(U + 1, dtype=np.int64) * 3000
for label, pre in (("after EM (K2)", lambda: g.em_accumulate(x[:T], acc=acc)), ("after tv_stats (K3)", lambda: g.tv_stats(x[:T], ub, N, F)),
                   ("after a 1 GB memset", lambda: F.zero_())):
    ts = []
    for rep in range(4):
        pre(); torch.cuda.synchronize()
        acc.zero_(); g.em_accumulate(x[:T], acc=acc); torch.cuda.synchronize()
        ts.append(ctx.kernel_ms("k_llk_mfma"))
    print("EM K1 on %d frames %s: %s ms" % (T, label, ["%.3f" % t for t in ts]), flush=True)
ts = []
for rep in range(4):
    g.em_accumulate(x[:T], acc=acc); torch.cuda.synchronize()
    g.tv_stats(x[:T], ub, N, F); torch.cuda.synchronize(); ts.append(ctx.kernel_ms("k_llk_mfma"))
print("TV K1 on %d frames after EM (K2): %s ms" % (T, ["%.3f" % t for t in ts]), flush=True)
