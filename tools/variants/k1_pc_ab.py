#!/usr/bin/env python3
"""A/B of the producer / consumer log-likelihood kernel (llk_pc.hip, option "k1_pc") against k_llk_mfma on the EM pass and on the
plain log-likelihood: kernel ms from the library's HIP events, results compared bitwise.  usage: python tools/k1_pc_ab.py [frames]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import make_gmm
from lia_ral_amd import capi
from bench import synth_frames

C, D = 2048, 60
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3_072_000
dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1); ctx.set_option("assume_finite", 1)
w, mean, iv = make_gmm(C, D, seed=0)
g = ctx.gmm(w, mean, iv)
x = synth_frames(w, mean, iv, T, dev, seed=1)
res = {}
for pc in (0, 1, 0, 1):
    ctx.set_option("k1_pc", pc)
    acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device=dev)
    g.em_accumulate(x, acc=acc); torch.cuda.synchronize()
    acc.zero_(); g.em_accumulate(x, acc=acc); torch.cuda.synchronize()
    k1, k2, nl = ctx.kernel_ms("k_llk_mfma"), ctx.kernel_ms("k_stats_z"), ctx.kernel_launches("k_llk_mfma")
    l = torch.empty(T, dtype=torch.float64, device=dev)
    import ctypes as ct
    capi._chk(capi.lib.gmmiv_llk(ctx._h, g._h, capi._ptr(x), capi.F32, ct.c_int64(T), ct.c_int64(D), ct.c_double(-200.0), ct.c_double(200.0), capi._ptr(l), None))
    torch.cuda.synchronize()
    capi._chk(capi.lib.gmmiv_llk(ctx._h, g._h, capi._ptr(x), capi.F32, ct.c_int64(T), ct.c_int64(D), ct.c_double(-200.0), ct.c_double(200.0), capi._ptr(l), None))
    torch.cuda.synchronize()
    kp = ctx.kernel_ms("k_llk_mfma")
    tf = 240.0 * T * C / (k1 * 1e-3) / 1e12
    print("k1_pc %d: K1<WZ> %.3f ms (%d launches, %.1f TF = %.3f of 78.6), K2 %.3f ms, plain llk %.3f ms (%.1f TF)" % (pc, k1, nl, tf, tf / 78.6, k2, kp, 240.0 * T * C / (kp * 1e-3) / 1e12), flush=True)
    res.setdefault(pc, (acc.clone(), l.clone()))
print("EM accumulator bitwise equal:", bool(torch.equal(res[0][0], res[1][0])), " plain llk bitwise equal:", bool(torch.equal(res[0][1], res[1][1])))
print("max |acc diff| rel:", float(((res[0][0] - res[1][0]).abs().max() / res[0][0].abs().max()).item()))
