#!/usr/bin/env python3
"""IvTest scoring kernels at dimension DIM (400: cosine / Mahalanobis / two-covariance; rank RF: PLDA) on M x S device-resident vectors:
wall time of the call and HIP-event time of the scoring GEMM inside it (k_dgemm with the rule's epilogue)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lia_ral_amd import capi
M = int(os.environ.get("M", "40000")); S = int(os.environ.get("S", "40000")); DIM = int(os.environ.get("DIM", "400")); RF = int(os.environ.get("RF", "200"))
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(); torch.cuda.set_stream(side)
ctx = capi.Context(0, side.cuda_stream)
ctx.set_option("timing", 1)
g = torch.Generator(device=dev); g.manual_seed(0)
rnd = lambda *s: torch.randn(s, dtype=torch.float64, device=dev, generator=g)
models, segs = rnd(DIM, M), rnd(DIM, S)
A = rnd(DIM, DIM); Mah = A @ A.T / DIM + torch.eye(DIM, dtype=torch.float64, device=dev)
scores = torch.empty((M, S), dtype=torch.float64, device=dev)
pm, ps = rnd(RF, M), rnd(RF, S)
B = rnd(RF, RF); FTJF = (B @ B.T / RF).cpu().numpy()
nsess = np.full(M, 3, np.int64) if os.environ.get("RUNS", "1") == "1" else np.sort(np.random.default_rng(3).integers(1, 4, M)).astype(np.int64)   # RUNS=3: three runs of equal session count
def tm(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for name, f, k in (("cosine", lambda: ctx.score_cosine(models, segs, out=scores), DIM),
                   ("mahalanobis", lambda: ctx.score_mahalanobis(models, segs, Mah, out=scores), DIM),
                   ("plda rank %d" % RF, lambda: ctx.score_plda(pm, nsess, ps, FTJF, out=scores), RF)):
    ms = tm(f)
    kms = ctx.kernel_ms("k_dgemm(score)")
    print("%-14s %5d x %5d: call %8.2f ms = %6.1f G trials/s (%5.1f TF) | scoring GEMM alone %8.2f ms = %5.1f TF" %
          (name, M, S, ms, M * S / ms / 1e6, 2.0 * k * M * S / ms / 1e9, kms, 2.0 * k * M * S / kms / 1e9))
