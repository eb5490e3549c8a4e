#!/usr/bin/env python3
"""A/B of the batched Cholesky behind the i-vector solve (gmmiv_tv_estimate_w) and the T-matrix E-step
(gmmiv_tv_estimate_a_and_c): k_chol_left (one workgroup per matrix) vs the GEMM-built factorisation."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi

dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
C, D = 2048, 60
R = int(os.environ.get("R", "400")); U = int(os.environ.get("U", "512"))
P = R * (R + 1) // 2
g = torch.Generator(device=dev); g.manual_seed(0)
w, mean, iv = make_gmm(C, D, seed=0)
N = torch.rand((U, C), dtype=torch.float64, device=dev, generator=g) * 3.0
F = torch.randn((U, C * D), dtype=torch.float64, device=dev, generator=g)
Tm = 0.01 * torch.randn((R, C * D), dtype=torch.float64, device=dev, generator=g)
invvar = torch.from_numpy(iv.ravel().copy()).to(dev)
tett = torch.empty((C, P), dtype=torch.float64, device=dev)
ctx.tv_tett(Tm, invvar, C, D, out=tett)
W = torch.empty((U, R), dtype=torch.float64, device=dev)
acc = dict(A=torch.zeros((C, P), dtype=torch.float64, device=dev), Cmx=torch.zeros((R, C * D), dtype=torch.float64, device=dev),
           Rm=torch.zeros((R, R), dtype=torch.float64, device=dev), r=torch.zeros(R, dtype=torch.float64, device=dev),
           meanW=torch.zeros(R, dtype=torch.float64, device=dev), W=torch.empty((U, R), dtype=torch.float64, device=dev))

def t(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

ctx.set_option("tv_batch", int(os.environ.get("TV_BATCH", "256")))
out = {"R": R, "U": U, "tv_batch": int(os.environ.get("TV_BATCH", "256"))}
res = {}
for mode in (1, 0, 1, 0):
    ctx.set_option("chol_gemm", mode)
    key = "gemm" if mode else "left"
    ms_w = t(lambda: ctx.tv_estimate_w(N, F, Tm, invvar, tett, C, D, out=W))
    res[key + "_W"] = W.clone()
    def estep():
        for k in ("A", "Cmx", "Rm", "r", "meanW"):
            acc[k].zero_()
        ctx.tv_estimate_a_and_c(N, F, Tm, invvar, tett, C, D, acc=acc)
    ms_e = t(estep, 2)
    res[key + "_A"] = acc["A"].clone()
    out.setdefault(key, []).append({"estimate_w_ms": ms_w, "estep_ms": ms_e})
out["max_rel_diff_W"] = float(((res["left_W"] - res["gemm_W"]).abs().max() / res["gemm_W"].abs().max()).item())
out["max_rel_diff_A"] = float(((res["left_A"] - res["gemm_A"]).abs().max() / res["gemm_A"].abs().max()).item())
print(json.dumps(out))
