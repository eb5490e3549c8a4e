#!/usr/bin/env python3
"""Side measurements for DESIGN.md (not the headline bench): one T-matrix EM iteration on a slice of
BASELINE.json configs[3] (2048-g, rank 400) and the scoring rules on a slice of configs[4]
(rank-400 vectors), single GPU, everything resident in HBM."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi

dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("gemm_clamp", int(os.environ.get("GEMM_CLAMP", "1")))
ctx.set_option("gemm_remap", int(os.environ.get("GEMM_REMAP", "1")))
if os.environ.get("TV_BATCH"):
    ctx.set_option("tv_batch", int(os.environ["TV_BATCH"]))   # utterances per batch of the E-step (default 1024)
ctx.set_option("gemm_nt80", int(os.environ.get("GEMM_NT80", "1")))   # aux on 128 x 80 tiles (R = 400 = 5 x 80) instead of 128 x 128 + strip
C, D, R = 2048, 60, 400
P = R * (R + 1) // 2
out = {}

def sync_time(f, reps=2):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / max(reps, 1)

# ---- T-matrix EM iteration on U utterances (statistics given)
U = int(os.environ.get("TV_U", "1024"))
g = torch.Generator(device=dev); g.manual_seed(0)
w, mean, iv = make_gmm(C, D, seed=0)
N = torch.rand((U, C), dtype=torch.float64, device=dev, generator=g) * 3.0
F = torch.randn((U, C * D), dtype=torch.float64, device=dev, generator=g)
Tm = 0.01 * torch.randn((R, C * D), dtype=torch.float64, device=dev, generator=g)
invvar = torch.from_numpy(iv.ravel().copy()).to(dev)
tett = torch.empty((C, P), dtype=torch.float64, device=dev)
acc = dict(A=torch.zeros((C, P), dtype=torch.float64, device=dev), Cmx=torch.zeros((R, C * D), dtype=torch.float64, device=dev),
           Rm=torch.zeros((R, R), dtype=torch.float64, device=dev), r=torch.zeros(R, dtype=torch.float64, device=dev),
           meanW=torch.zeros(R, dtype=torch.float64, device=dev), W=torch.empty((U, R), dtype=torch.float64, device=dev))
Tn = torch.empty_like(Tm)
PH = os.environ.get("TV_PHASES", "tett,estep,mstep,scoring").split(",")
t_tett = sync_time(lambda: ctx.tv_tett(Tm, invvar, C, D, out=tett), 2 if "tett" in PH else 0)
def estep():
    for k in ("A", "Cmx", "Rm", "r", "meanW"):
        acc[k].zero_()
    ctx.tv_estimate_a_and_c(N, F, Tm, invvar, tett, C, D, acc=acc)
t_e = sync_time(estep, 1) if "estep" in PH else float("nan")
if "estep" not in PH:
    estep()          # the M-step needs a valid A
t_m = sync_time(lambda: ctx.tv_update_t(acc["A"], acc["Cmx"], C, D, out=Tn), 1) if "mstep" in PH else float("nan")
flop_e = U * (2.0 * C * P + 2.0 * C * D * R + 2.0 * C * P + 2.0 * R * C * D)      # L, aux, A (packed), Cmx
out["tv_em"] = {"utterances": U, "tett_ms": t_tett * 1e3, "estep_ms": t_e * 1e3, "mstep_ms": t_m * 1e3,
                "estep_ms_per_utterance": t_e * 1e3 / U, "estep_gemm_tflops": flop_e / t_e / 1e12,
                "finite": bool(torch.isfinite(Tn).all().item())}
del N, F, acc, tett, Tn
torch.cuda.empty_cache()

if "scoring" not in PH:
    print(json.dumps(out)); sys.exit(0)
# ---- scoring M x S
M = S = int(os.environ.get("SC_N", "20000"))
models = torch.randn((R, M), dtype=torch.float64, device=dev, generator=g)
segs = torch.randn((R, S), dtype=torch.float64, device=dev, generator=g)
models /= models.norm(dim=0, keepdim=True); segs /= segs.norm(dim=0, keepdim=True)
scores = torch.empty((M, S), dtype=torch.float64, device=dev)
Q = torch.randn((R, R), dtype=torch.float64, device=dev, generator=g)
Mah = (Q @ Q.T / R + torch.eye(R, dtype=torch.float64, device=dev)).contiguous()
res = {}
res["cosine"] = sync_time(lambda: ctx.score_cosine(models, segs, out=scores))
res["mahalanobis"] = sync_time(lambda: ctx.score_mahalanobis(models, segs, Mah, out=scores))
Gm = (Q / R).contiguous(); Hm = (Q.T / R).contiguous()
res["twocov"] = sync_time(lambda: ctx.score_twocov(models, segs, Gm, Hm, out=scores))
rf = 200
Fp = torch.randn((R, rf), dtype=torch.float64, device=dev, generator=g)
FTJF = (Fp.T @ Fp / R).contiguous()
mp = torch.randn((rf, M), dtype=torch.float64, device=dev, generator=g); sp = torch.randn((rf, S), dtype=torch.float64, device=dev, generator=g)
nsess = np.ones(M, np.int64)
res["plda(rankF=200)"] = sync_time(lambda: ctx.score_plda(mp, nsess, sp, FTJF, out=scores))
out["scoring"] = {"M": M, "S": S, "dim": R,
                  **{k: {"ms": v * 1e3, "Gtrials_per_s": M * S / v / 1e9, "tflops": 2.0 * (rf if "plda" in k else R) * M * S / v / 1e12}
                     for k, v in res.items()}}
print(json.dumps(out))
