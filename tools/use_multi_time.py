#!/usr/bin/env python3
"""ComputeTest's client loop on one test segment: N client models scored on the world's top-10 indices, client by client
(gmmiv_llk_use_top, host result per call, like the reference's loop) against one gmmiv_llk_use_top_multi call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_gmm, make_frames
from lia_ral_amd import capi
ctx = capi.Context(0)
C, D, ctop = 2048, 60, 10
w, mean, iv = make_gmm(C, D, seed=0)
world = ctx.gmm(w, mean, iv)
rng = np.random.default_rng(1)
NC = int(os.environ.get("CLIENTS", "100"))
clients = [ctx.gmm(w, mean + rng.normal(0, 0.1, mean.shape), iv) for _ in range(NC)]
for T in (3000, 30000, 300000):
    x = make_frames(w, mean, iv, T, seed=T).astype(np.float32)
    d = world.llk_determine_top(x, ctop, True)
    def loop():
        return np.stack([g.llk_use_top(x, d["idx"], d["nontop_llk"], True) for g in clients])
    def multi():
        return capi.Gmm.llk_use_top_multi(clients, x, d["idx"], d["nontop_llk"], True)
    a, b = loop(), multi()
    assert np.array_equal(a, b)
    res = []
    for f in (loop, multi):
        t0 = time.perf_counter(); f(); f(); dt = (time.perf_counter() - t0) / 2
        res.append(dt)
    print("T=%d, %d clients: client by client %.2f ms (%.1f M frame-clients/s) | one call %.2f ms (%.1f M frame-clients/s)"
          % (T, NC, res[0] * 1e3, T * NC / res[0] / 1e6, res[1] * 1e3, T * NC / res[1] / 1e6))
