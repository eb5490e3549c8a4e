#!/usr/bin/env python3
"""One T-matrix E-step (gmmiv_tv_estimate_a_and_c) and one extraction (gmmiv_tv_estimate_w) on U utterances, for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_gmm
from lia_ral_amd import capi
dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
C, D = 2048, 60
R = int(os.environ.get("R", "400")); U = int(os.environ.get("U", "1024"))
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
what = os.environ.get("WHAT", "estep")
for _ in range(3):
    try:
        if what == "estep":
            ctx.tv_estimate_a_and_c(N, F, Tm, invvar, tett, C, D, acc=acc)
        else:
            ctx.tv_estimate_w(N, F, Tm, invvar, tett, C, D, out=W)
    except capi.GmmivError as e:       # instrumented builds (tools/chol_ablate.sh) produce non-SPD garbage on purpose
        print("call failed:", e)
torch.cuda.synchronize()
