#!/usr/bin/env python3
"""What the kernels do with degenerate inputs (exploration; the rules are stated in include/gmmiv.h and tested in
tests/test_gpu_degenerate.py): non-finite frames, a Gaussian of weight 0, identical Gaussians (ties over a whole row), T = 0."""
import os, sys, signal
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import functools
import numpy as np
print = functools.partial(print, flush=True)
from conftest import make_frames, make_gmm
from lia_ral_amd import capi

signal.alarm(240)
C, D, T = int(os.environ.get("PC", "256")), 60, 600
w, mean, iv = make_gmm(C, D, seed=1)
x = make_frames(w, mean, iv, T, seed=2)
bad = {5: np.nan, 9: np.inf, 11: -np.inf, 13: 1e30, 300: np.nan}
for t, v in bad.items():
    x[t, 7] = v
x[17, :] = 3e38            # finite float32, the distance overflows fp64? no: 9e76 * iv fits; stays finite
for waves in (8, 4):
    for opts in ({}, {"stats_z": 0}, {"topc_fused": 0}, {"topc_fused": 0, "topc_z": 0}):
        ctx = capi.Context(0)
        ctx.set_option("wg_waves", waves)
        for k, v in opts.items():
            ctx.set_option(k, v)
        g = ctx.gmm(w, mean, iv)
        tag = "waves %d %s" % (waves, opts)
        try:
            l = g.llk(x)
            print(tag, "llk bad frames:", {t: l[t] for t in list(bad) + [17]}, "finite elsewhere:", np.isfinite(np.delete(l, list(bad) + [17])).all())
            d = g.llk_determine_top(x, 10)
            okidx = (d["idx"] >= 0).all() and (d["idx"] < C).all()
            print(tag, "top idx in range:", okidx, "bad rows:", {t: (d["idx"][t][:4].tolist(), d["llk"][t], d["nontop_llk"][t]) for t in (5, 9, 13)})
            a = g.split_acc(g.em_accumulate(x))
            print(tag, "em: occ finite", np.isfinite(a["occ"]).all(), "sx finite", np.isfinite(a["sx"]).all(), "sum occ", a["occ"].sum(), "llk", a["llk"], "count", a["count"])
            o = g.occ(x[:20])
            print(tag, "occ rows sums", {t: o[t].sum() for t in (4, 5, 9, 11, 13, 17)})
            ub = np.array([0, 100, 100, 600])
            import torch
            N = np.zeros((3, C)); F = np.zeros((3, C * D))
            g.tv_stats(x, ub, N, F)
            print(tag, "tv_stats N finite", np.isfinite(N).all(), "F finite", np.isfinite(F).all(), "N sums", N.sum(1))
        except capi.GmmivError as e:
            print(tag, "ERROR", e)
        g.close(); ctx.close()
# weight 0 and identical Gaussians
ctx = capi.Context(0)
w0 = w.copy(); w0[3] = 0.0; w0 /= w0.sum()
g = ctx.gmm(w0, mean, iv)
xs = make_frames(w, mean, iv, 400, seed=3)
l = g.llk(xs); d = g.llk_determine_top(xs, 10); a = g.split_acc(g.em_accumulate(xs))
print("weight 0: llk finite", np.isfinite(l).all(), "idx never 3:", not (d["idx"] == 3).any(), "occ[3]", a["occ"][3], "sx[3] finite", np.isfinite(a["sx"][3]).all())
wm, mm, cc = g.em_get(g.em_accumulate(xs), mean, 1.0 / iv)
print("em_get with occ 0: w[3]", wm[3], "mean[3]==prev", np.array_equal(mm[3], mean[3]), "cov[3]==prev", np.array_equal(cc[3], (1.0 / iv)[3]))
g.close()
we = np.full(C, 1.0 / C); me = np.tile(mean[0], (C, 1)); ive = np.tile(iv[0], (C, 1))
g = ctx.gmm(we, me, ive)
for fused, z in ((1, 1), (0, 1), (0, 0)):
    ctx.set_option("topc_fused", fused); ctx.set_option("topc_z", z)
    d = g.llk_determine_top(xs, 10)
    print("ties fused %d z %d: idx rows are 0..9:" % (fused, z), bool((d["idx"] == np.arange(10)).all()), d["idx"][0].tolist(), "llk finite", np.isfinite(d["llk"]).all())
a = g.split_acc(g.em_accumulate(xs)); print("ties em occ all equal:", np.allclose(a["occ"], a["occ"][0]), a["occ"][:3])
# T = 0
e = np.zeros((0, D), np.float32)
print("T=0:", g.llk(e).shape, g.llk_determine_top(e, 10)["idx"].shape, g.em_accumulate(e).sum(), g.occ(e).shape)
print("done")
