#!/usr/bin/env python3
"""What do the kernels do with unusable frames when NOTHING screens them (assume_finite 1)?  Every section of
tests/test_gpu_degenerate.py::test_zero_likelihood_frames_follow_the_rule_on_every_path as a soft check, per kernel path."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_degenerate import _case, PATHS, C, D, T, BAD
from oracle import oracle as orc
from lia_ral_amd import capi

af = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w, mean, iv, x, zero, good = _case()
og = orc.Gmm(w, mean, iv); xg = x[good].astype(np.float64)
rel = lambda p, q: np.max(np.abs(p - q)) / np.max(np.abs(q))
for opts in PATHS:
    ctx = capi.Context(0)
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.set_option("assume_finite", af)
    g = ctx.gmm(w, mean, iv)
    res = {}
    def chk(name, fn):
        try:
            r = fn(); res[name] = "ok" if (isinstance(r, (bool, np.bool_)) and bool(r)) else "FAIL %s" % (r,)
        except Exception as e:
            res[name] = "EXC %s" % str(e)[:80]
    def t_llk():
        sums = np.zeros(2); l = g.llk(x, sums=sums)
        return (np.all(l[zero] == -200.0) and np.max(np.abs(l[good] - orc.llk(og, xg))) < 1e-9 and sums[1] == T and abs(sums[0] - l.sum()) < 1e-6) or [l[zero].tolist(), sums.tolist()]
    chk("llk", t_llk)
    dd = {}
    def t_top():
        d = g.llk_determine_top(x, 10); dd["d"] = d
        do = orc.llk_determine_top(og, xg, 10, True)
        a = np.array_equal(d["idx"][good], do["idx"]) and np.max(np.abs(d["llk"][good] - do["llk"])) < 1e-9
        b = all(d["idx"][t].tolist() == list(range(10)) and np.all(d["lk"][t] == 0.0) and d["nontop_lk"][t] == 0.0 and d["nontop_llk"][t] == -np.inf and d["llk"][t] == -200.0 for t in zero)
        return (a and b) or [a, b, [(t, d["idx"][t][:3].tolist(), d["llk"][t], d["nontop_llk"][t]) for t in zero]]
    chk("determine_top", t_top)
    def t_use():
        rng = np.random.default_rng(0)
        cl = [ctx.gmm(w, mean + rng.normal(0, 0.1, mean.shape), iv) for _ in range(3)]
        d = dd["d"]
        u = cl[0].llk_use_top(x, d["idx"], d["nontop_llk"]); um = type(cl[0]).llk_use_top_multi(cl, x, d["idx"], d["nontop_llk"])
        return (np.all(u[zero] == -200.0) and np.all(um[:, zero] == -200.0) and np.isfinite(u).all() and np.isfinite(um).all() and np.array_equal(um[0], u)) or [u[zero].tolist()]
    chk("use_top", t_use)
    def t_occ():
        o = g.occ(x[:40])
        return all((np.all(o[t] == 0.0) if t in zero else abs(o[t].sum() - 1.0) < 1e-9) for t in range(40)) or [o[5][:4].tolist(), o[9][:4].tolist(), o[17][:4].tolist()]
    chk("occ", t_occ)
    def t_em():
        a = g.split_acc(g.em_accumulate(x)); ref = orc.em_accumulate(og, xg)
        ok = a["count"] == len(good) and rel(a["occ"], ref["occ"]) < 1e-9 and rel(a["sx"], ref["sx"]) < 1e-9 and rel(a["sxx"], ref["sxx"]) < 1e-9 and abs(a["llk"] - ref["llk"]) < 1e-6 * abs(ref["llk"])
        return ok or [a["count"], len(good), np.isfinite(a["occ"]).all(), np.isfinite(a["sx"]).all(), np.isfinite(a["sxx"]).all(), a["llk"]]
    chk("em", t_em)
    def t_tv():
        ub = np.array([0, 6, 6, 301, T]); N = np.zeros((4, C)); F = np.zeros((4, C * D))
        g.tv_stats(x, ub, N, F)
        utt = np.searchsorted(ub, good, side="right") - 1
        No, Fo = orc.tv_stats(og, xg, utt, 4)
        return (np.isfinite(N).all() and np.isfinite(F).all() and rel(N, No) < 1e-9 and rel(F, Fo) < 1e-9) or [np.isfinite(N).all(), np.isfinite(F).all()]
    chk("tv_stats", t_tv)
    print(opts, "screened", ctx.set_option("screened_frames", 0), "zero_llk", ctx.set_option("zero_llk_frames", 0))
    for k, v in res.items(): print("   %-14s %s" % (k, v))
    g.close(); ctx.close()
