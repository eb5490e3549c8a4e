#!/usr/bin/env python3
"""One degenerate-input case per process (a fault kills only that process): python tools/degenerate_case.py <value> <what> [opt=val ...]"""
import os, sys, signal
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_frames, make_gmm
from lia_ral_amd import capi
signal.alarm(120)
val = float(sys.argv[1]); what = sys.argv[2]
C, D, T = 256, 60, 600
w, mean, iv = make_gmm(C, D, seed=1)
x = make_frames(w, mean, iv, T, seed=2)
if sys.argv[1] == "row":
    x[17, :] = 3e38
else:
    x[5, 7] = val
ctx = capi.Context(0)
for a in sys.argv[3:]:
    k, v = a.split("="); ctx.set_option(k, int(v))
g = ctx.gmm(w, mean, iv)
if what == "top":
    d = g.llk_determine_top(x, 10)
    print(sys.argv[1:], "ok idx in range", bool(((d["idx"] >= 0) & (d["idx"] < C)).all()), d["idx"][5].tolist(), d["llk"][5], d["nontop_llk"][5], d["idx"][17][:3].tolist(), d["llk"][17])
elif what == "em":
    a = g.split_acc(g.em_accumulate(x)); print(sys.argv[1:], "occ finite", np.isfinite(a["occ"]).all(), a["occ"].sum(), a["llk"], a["count"])
elif what == "tv":
    N = np.zeros((3, C)); F = np.zeros((3, C * D)); g.tv_stats(x, np.array([0, 100, 100, 600]), N, F); print(sys.argv[1:], np.isfinite(N).all(), np.isfinite(F).all(), N.sum(1))
elif what == "occ":
    o = g.occ(x[:20]); print(sys.argv[1:], o[5].sum(), o[17].sum(), o[4].sum())
