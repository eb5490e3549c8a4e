#!/usr/bin/env python3
"""A/B of chol_fused.hip's panel-row source (option "chol_lds": 1 = staged once per workgroup in LDS, 0 = fetched by every wave):
T-matrix E-step, M-step (by substitution, and with the explicit inverse: option "tv_mstep_solve") and i-vector solve at C = 2048,
R = 400; results must agree to rounding."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1)
C, D, R, U = 2048, 60, int(os.environ.get("R", "400")), int(os.environ.get("U", "1024"))
P = R * (R + 1) // 2
g = torch.Generator(device=dev); g.manual_seed(0)
w, mean, iv = make_gmm(C, D, seed=0)
N = torch.rand((U, C), dtype=torch.float64, device=dev, generator=g) * 3.0
F = torch.randn((U, C * D), dtype=torch.float64, device=dev, generator=g)
Tm = 0.01 * torch.randn((R, C * D), dtype=torch.float64, device=dev, generator=g)
invvar = torch.from_numpy(iv.ravel().copy()).to(dev)
tett = torch.empty((C, P), dtype=torch.float64, device=dev); ctx.tv_tett(Tm, invvar, C, D, out=tett)
W = torch.empty((U, R), dtype=torch.float64, device=dev); Tn = torch.empty_like(Tm)
z = lambda *s: torch.zeros(s, dtype=torch.float64, device=dev)
acc = dict(A=z(C, P), Cmx=z(R, C * D), Rm=z(R, R), r=z(R), meanW=z(R), W=torch.empty((U, R), dtype=torch.float64, device=dev))
def t(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def estep():
    for k in ("A", "Cmx", "Rm", "r", "meanW"): acc[k].zero_()
    ctx.tv_estimate_a_and_c(N, F, Tm, invvar, tett, C, D, acc=acc)
out = {"R": R, "U": U}; keep = {}
for mode in (0, 1, 0, 1):
    ctx.set_option("chol_lds", mode)
    key = "lds" if mode else "mem"
    a = t(lambda: ctx.tv_estimate_w(N, F, Tm, invvar, tett, C, D, out=W)); keep[key + "W"] = W.clone()
    b = t(estep, 2); keep[key + "A"] = acc["A"].clone()
    c = t(lambda: ctx.tv_update_t(acc["A"], acc["Cmx"], C, D, out=Tn), 2); keep[key + "T"] = Tn.clone()
    ctx.set_option("tv_mstep_solve", 0)
    c0 = t(lambda: ctx.tv_update_t(acc["A"], acc["Cmx"], C, D, out=Tn), 2); keep[key + "Tinv"] = Tn.clone()
    ctx.set_option("tv_mstep_solve", 1)
    out.setdefault(key, []).append({"estimate_w_ms": a, "estep_ms": b, "mstep_ms": c, "mstep_explicit_inverse_ms": c0})
for k in "WAT":
    out["max_rel_diff_" + k] = float(((keep["lds" + k] - keep["mem" + k]).abs().max() / keep["mem" + k].abs().max()).item())
out["max_rel_diff_T_solve_vs_inverse"] = float(((keep["ldsT"] - keep["ldsTinv"]).abs().max() / keep["ldsTinv"].abs().max()).item())
print(json.dumps(out))
