#!/usr/bin/env python3
"""Wall time of one PldaModel::em_iteration (gmmiv_plda_em_iteration) at a realistic back-end size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lia_ral_amd import capi
ctx = capi.Context(0)
dim, rf, rg = 400, int(os.environ.get("RF", "200")), int(os.environ.get("RG", "0"))
nspk, per = int(os.environ.get("NSPK", "2000")), 10
rng = np.random.default_rng(0)
sps = np.full(nspk, per, np.int64); n = int(sps.sum())
spk = rng.normal(size=(dim, nspk)); X = np.repeat(spk, per, axis=1) + 0.5 * rng.normal(size=(dim, n))
F = rng.normal(size=(dim, rf)) * 0.1; G = rng.normal(size=(dim, rg)) * 0.1; Sigma = np.eye(dim); Delta = X.mean(1)
for it in range(2):
    t0 = time.perf_counter()
    ctx.plda_em_iteration(X, sps, F, G, Sigma, Delta)
    print("iteration %d: %.1f ms (dim %d, rankF %d, rankG %d, %d sessions of %d speakers), finite %s" % (it, (time.perf_counter() - t0) * 1e3, dim, rf, rg, n, nspk, bool(np.isfinite(F).all())))
