#!/usr/bin/env python3
"""Boundary-shape sweep of the C ABI against the oracle (round 6: the vectSize-1 bug was found by ONE fixture that happened to have
that shape -- this walks the boundaries on purpose).  Soft checks: prints every (entry point, shape) whose error exceeds the tolerance
or that raises; exit status 1 if any.   usage: python tools/shape_sweep.py [gmm|tv|score|backend ...]"""
import os, sys, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_frames, make_gmm
from oracle import oracle as orc
from lia_ral_amd import capi

ctx = None
bad = []
nchk = [0]


def run(which):
    """-> (checks, failures) of the named sweeps (tests/test_gpu_shape_sweep.py)"""
    global ctx
    ctx = capi.Context(0)
    del bad[:]; nchk[0] = 0
    try:
        for name in which:
            {"gmm": gmm_sweep, "gmm_paths": gmm_paths_sweep, "tv": tv_sweep, "score": score_sweep, "backend": backend_sweep}[name]()
    finally:
        ctx.close(); ctx = None
    return nchk[0], list(bad)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape:
        return np.inf
    if a.size == 0:
        return 0.0
    if not np.isfinite(a).all():
        return np.inf
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def chk(name, shape, fn, tol=1e-9):
    nchk[0] += 1
    try:
        e = fn()
        if not (e <= tol):
            bad.append((name, shape, e)); print("FAIL %-28s %-28s err %.3g" % (name, shape, e), flush=True)
    except Exception as ex:     # noqa: BLE001
        bad.append((name, shape, repr(ex)[:120])); print("EXC  %-28s %-28s %s" % (name, shape, repr(ex)[:160]), flush=True)


GMM_PATHS = [{"stats_z": 0}, {"wg_waves": 4}, {"short_calls": 0}, {"topc_fused": 0}, {"topc_fused": 0, "topc_z": 0}, {"glds": 0}, {"topc_rank2": 0},
             {"z_waves": 4}, {"z_waves": 16}, {"tv_stats_split": 0}]


def gmm_paths_sweep():
    """the non-default kernel paths of the GMM side (options of the context) on a shorter list of boundary shapes"""
    shapes = [(1, 1, 1), (2, 1, 26), (3, 2, 17), (16, 3, 64), (17, 4, 65), (33, 5, 129), (2, 15, 33), (5, 16, 257), (64, 17, 31), (65, 31, 16), (15, 32, 15),
              (31, 33, 63), (7, 59, 2), (128, 60, 300), (3, 61, 127), (16, 63, 255), (32, 64, 64), (9, 65, 65), (2, 79, 17), (40, 80, 40), (2048, 1, 50), (2048, 60, 70)]
    defaults = {}
    for opts in GMM_PATHS:
        for k, v in opts.items():
            defaults[k] = ctx.set_option(k, v)
        tag = ",".join("%s=%s" % kv for kv in opts.items())
        try:
            for C, D, T in shapes:
                w, mean, iv = make_gmm(C, D, seed=C * 7 + D, spread=0.5 if D > 60 else 2.0)
                x = make_frames(w, mean, iv, T, seed=T * 3 + D)
                xo = x.astype(np.float64)
                g = ctx.gmm(w, mean, iv); og = orc.Gmm(w, mean, iv)
                sh = "C%d D%d T%d [%s]" % (C, D, T, tag)
                chk("llk", sh, lambda: float(np.max(np.abs(g.llk(x, -1e9, 1e9) - orc.llk(og, xo, -1e9, 1e9)))), 1e-9)
                ref = orc.em_accumulate(og, xo)
                def em():
                    a = g.split_acc(g.em_accumulate(x))
                    return max(rel(a["occ"], ref["occ"]), rel(a["sx"], ref["sx"]), rel(a["sxx"], ref["sxx"]), abs(a["count"] - T))
                chk("em_accumulate", sh, em)
                chk("occ", sh, lambda: rel(g.occ(x), orc.occ(og, xo)))
                for ctop in sorted({1, min(C, 3), min(C, 16), min(C, 20)}):
                    do = orc.llk_determine_top(og, xo, ctop, True)
                    def top():
                        d = g.llk_determine_top(x, ctop, True)
                        return max(0.0 if np.array_equal(d["idx"], do["idx"]) else np.inf, float(np.max(np.abs(d["llk"] - do["llk"]))), rel(d["lk"], do["lk"]))
                    chk("determine_top c%d" % ctop, sh, top)
                ub = np.array([0, T // 3, T // 3, T])
                utt = np.minimum(np.searchsorted(ub, np.arange(T), side="right") - 1, 2)
                def tvs():
                    N, F = g.tv_stats(x, ub)
                    No, Fo = orc.tv_stats(og, xo, utt, 3)
                    return max(rel(N, No), rel(F, Fo))
                chk("tv_stats", sh, tvs)
                g.close()
        finally:
            for k in opts:
                ctx.set_option(k, defaults[k])


def gmm_sweep():
    Ds = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 59, 60, 61, 63, 64, 65, 79, 80, 81]
    Cs = [1, 2, 3, 15, 16, 17, 31, 32, 33, 64, 65]
    Ts = [1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 255, 257]
    rng = np.random.default_rng(0)
    combos = [(C, D, T) for D in Ds for C in (Cs[rng.integers(len(Cs))], Cs[rng.integers(len(Cs))]) for T in (Ts[rng.integers(len(Ts))],)]
    combos += [(C, 1, T) for C in (1, 2, 16, 33) for T in (1, 17, 64, 257)] + [(1, D, 5) for D in (1, 2, 60, 81)] + [(2048, 1, 40), (2048, 2, 40), (4097, 3, 9)]
    for C, D, T in combos:
        w, mean, iv = make_gmm(C, D, seed=C * 131 + D, spread=0.5 if D > 60 else 2.0)
        for dtype in (np.float32, np.float64):
            x = make_frames(w, mean, iv, T, seed=T + D, dtype=dtype)
            xo = x.astype(np.float64)
            g = ctx.gmm(w, mean, iv); og = orc.Gmm(w, mean, iv)
            sh = "C%d D%d T%d %s" % (C, D, T, np.dtype(dtype).name)
            chk("llk", sh, lambda: float(np.max(np.abs(g.llk(x, -1e9, 1e9) - orc.llk(og, xo, -1e9, 1e9)))), 1e-9)
            ref = orc.em_accumulate(og, xo)
            def em():
                a = g.split_acc(g.em_accumulate(x))
                return max(rel(a["occ"], ref["occ"]), rel(a["sx"], ref["sx"]), rel(a["sxx"], ref["sxx"]), abs(a["llk"] - ref["llk"]) / max(1.0, abs(ref["llk"])), abs(a["count"] - T))
            chk("em_accumulate", sh, em)
            chk("occ", sh, lambda: rel(g.occ(x), orc.occ(og, xo)))
            for ctop in sorted({1, min(C, 2), min(C, 10), min(C, 17), min(C, 65)}):
                do = orc.llk_determine_top(og, xo, ctop, True)
                def top():
                    d = g.llk_determine_top(x, ctop, True)
                    e = 0.0 if np.array_equal(d["idx"], do["idx"]) else np.inf
                    return max(e, float(np.max(np.abs(d["llk"] - do["llk"]))), rel(d["lk"], do["lk"]))
                chk("determine_top c%d" % ctop, sh, top)
                def use():
                    u = g.llk_use_top(x, do["idx"].astype(np.int32), np.log(np.maximum(do["nontop_lk"], 0)) if True else None, True)
                    return float(np.max(np.abs(u - orc.llk_use_top(og, xo, do["idx"], do["nontop_lk"], True))))
                with np.errstate(divide="ignore"):
                    chk("use_top c%d" % ctop, sh, use)
            cuts = sorted(set([0, T] + list(np.random.default_rng(T).integers(0, T + 1, 3))))
            ub = np.array([cuts[0]] + cuts[1:] + [cuts[-1]])          # a trailing empty utterance
            utt = np.searchsorted(ub, np.arange(T), side="right") - 1
            utt = np.minimum(utt, len(ub) - 2)
            def tvs():
                N, F = g.tv_stats(x, ub)
                No, Fo = orc.tv_stats(og, xo, utt, len(ub) - 1)
                return max(rel(N, No), rel(F, Fo))
            chk("tv_stats", sh, tvs)
            chk("frame_moments", sh, lambda: rel(ctx.frame_moments(x), np.concatenate([xo.sum(0), (xo * xo).sum(0), [T]])), 1e-12)
            g.close()


def tv_sweep():
    Rs = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 47, 63, 64, 65, 96, 97, 129]
    rng = np.random.default_rng(1)
    for R in Rs:
        for C, D, U in ((1, 1, 1), (2, 3, 2), (3, 5, 7), (5, 2, 33), (1, 60, 3)):
            sh = "R%d C%d D%d U%d" % (R, C, D, U)
            N = rng.uniform(0.5, 30.0, (U, C)); means = rng.normal(size=C * D); invvar = rng.uniform(0.5, 2.0, C * D)
            F = rng.normal(size=(U, C * D)) * np.repeat(N, D, axis=1)
            Tm = 0.1 * rng.normal(size=(R, C * D))
            te_o = orc.tv_tett(Tm, invvar, C, D)
            il = np.tril_indices(R)
            chk("tv_tett", sh, lambda: rel(ctx.tv_tett(Tm, invvar, C, D), te_o[:, il[0], il[1]]))
            te = ctx.tv_tett(Tm, invvar, C, D)
            chk("tv_subtract_m", sh, lambda: rel(ctx.tv_subtract_m(N, F.copy(), means, C, D), orc.tv_subtract_m(N, F, means)), 1e-12)
            chk("tv_estimate_w", sh, lambda: rel(ctx.tv_estimate_w(N, F, Tm, invvar, te, C, D), orc.tv_estimate_w(N, F, Tm, invvar, te_o)))
            eo = orc.tv_estimate_a_and_c(N, F, Tm, invvar, te_o)
            def ac():
                a = ctx.tv_estimate_a_and_c(N, F, Tm, invvar, te, C, D)
                return max(rel(a["A"], eo["A"].reshape(C, R, R)[:, il[0], il[1]]), rel(a["Cmx"], eo["Cmx"]), rel(a["W"], eo["W"]), rel(a["Rm"], eo["Rm"]), rel(a["r"], eo["r"]),
                           rel(a["meanW"] / U, eo["meanW"]))
            chk("tv_estimate_a_and_c", sh, ac)
            Ap = np.ascontiguousarray(eo["A"].reshape(C, R, R)[:, il[0], il[1]])
            To = orc.tv_update_t(eo["A"], eo["Cmx"], C, D)
            chk("tv_update_t", sh, lambda: rel(ctx.tv_update_t(Ap, eo["Cmx"], C, D), To), 1e-8)
            def md():
                m2, T2 = ctx.tv_min_divergence(eo["Rm"].copy(), eo["r"].copy(), eo["meanW"].copy(), means.copy(), To.copy(), U, C, D)
                mo, Tq = orc.tv_min_divergence(eo["Rm"], eo["r"], eo["meanW"], means, To, U, C, D)
                return max(rel(m2, mo), rel(T2, Tq))
            chk("tv_min_divergence", sh, md, 1e-8)
            chk("tv_norm_statistics", sh, lambda: rel(ctx.tv_norm_statistics(N, F.copy(), means, invvar, C, D), orc.tv_norm_statistics(N, F, means, invvar)), 1e-12)
            W = rng.normal(size=(U, R))
            chk("tv_subtract_m_plus_tw", sh, lambda: rel(ctx.tv_subtract_m_plus_tw(N, F.copy(), means, Tm, W, C, D), orc.tv_subtract_m_plus_tw(N, F, means, Tm, W)), 1e-11)
            wgt = rng.uniform(0.1, 1.0, C)
            chk("tv_weighted_cov", sh, lambda: rel(ctx.tv_weighted_cov(Tm, wgt, C, D), orc.tv_weighted_cov(Tm, wgt)), 1e-11)
            # approximate extractors + JFA pieces on the same boundary shapes
            Q = np.linalg.qr(rng.normal(size=(R, R)))[0]
            Do = orc.tv_approximate_tctc(Tm, Q, C)
            chk("tv_approximate_tctc", sh, lambda: rel(ctx.tv_approximate_tctc(Tm, Q, C, D), Do), 1e-11)
            Wo = orc.tv_weighted_cov(Tm, wgt)
            chk("tv_estimate_w_ubm_weight", sh, lambda: rel(ctx.tv_estimate_w_ubm_weight(N, F, Tm, Wo, C, D), orc.tv_estimate_w_ubm_weight(N, F, Tm, Wo)), 1e-8)
            chk("tv_estimate_w_eigen", sh, lambda: rel(ctx.tv_estimate_w_eigen(N, F, Tm, Do, Q, C, D), orc.tv_estimate_w_eigen(N, F, Tm, Do, Q)), 1e-8)
            Dm = rng.uniform(0.1, 1.0, C * D); Z = rng.normal(size=(U, C * D))
            chk("jfa_subtract", sh, lambda: rel(ctx.jfa_subtract(N, F.copy(), C, D, means=means, T=Tm, W=W, Dm=Dm, Z=Z), orc.jfa_subtract(N, F, None, means, Tm, W, Dm, Z)), 1e-11)
            chk("jfa_estimate_z", sh, lambda: rel(ctx.jfa_estimate_z(N, F, invvar, Dm, C, D), orc.jfa_estimate_z(N, F, invvar, Dm)), 1e-11)
            chk("jfa_estimate_z_map", sh, lambda: rel(ctx.jfa_estimate_z(N, F, invvar, Dm, C, D, tau=7.0), orc.jfa_estimate_z(N, F, invvar, Dm, 7.0)), 1e-11)
            if R > 1:
                Tq = Tm.copy()
                chk("tv_orthonormalize_t", sh, lambda: rel(ctx.tv_orthonormalize_t(Tq.copy()) if R <= C * D else orc.tv_orthonormalize_t(Tm), orc.tv_orthonormalize_t(Tm)) if R <= C * D else 0.0, 1e-8)


def score_sweep():
    rng = np.random.default_rng(2)
    for dim in (1, 2, 3, 4, 15, 16, 17, 33, 64, 65):
        for M, S in ((1, 1), (1, 17), (17, 1), (2, 3), (16, 16), (65, 33), (129, 130)):
            sh = "dim%d M%d S%d" % (dim, M, S)
            models = rng.normal(size=(dim, M)); segs = rng.normal(size=(dim, S))
            Q = rng.normal(size=(dim, dim)); Q = Q @ Q.T / dim + np.eye(dim)
            chk("score_cosine", sh, lambda: rel(ctx.score_cosine(models, segs), orc.score_cosine(models, segs)), 1e-11)
            chk("score_mahalanobis", sh, lambda: rel(ctx.score_mahalanobis(models, segs, Q), orc.score_mahalanobis(models, segs, Q)), 1e-10)
            G = rng.normal(size=(dim, dim)); G = G + G.T; H = rng.normal(size=(dim, dim)); H = H + H.T
            chk("score_twocov", sh, lambda: rel(ctx.score_twocov(models, segs, G, H), orc.score_twocov(models, segs, G, H)), 1e-10)
            for rf in sorted({1, min(dim, 2), dim}):
                ms = rng.normal(size=(rf, M)); sg = rng.normal(size=(rf, S)); ns = rng.integers(1, 4, M).astype(np.int64)
                Fm = rng.normal(size=(rf, rf)); FTJF = Fm @ Fm.T / rf + 0.1 * np.eye(rf)
                chk("score_plda rf%d" % rf, sh, lambda: rel(ctx.score_plda(ms, ns, sg, FTJF), orc.score_plda(ms, ns, sg, FTJF)), 1e-9)
            for ln in (True, False):
                Mx = rng.normal(size=(dim, dim)); mu = rng.normal(size=dim)
                chk("iv_normalize ln%d" % ln, sh, lambda: rel(ctx.iv_normalize(segs, mu, Mx, ln), orc.iv_normalize(segs, mu, Mx, ln)), 1e-11)


def backend_sweep():
    rng = np.random.default_rng(3)
    for dim in (1, 2, 3, 16, 17, 33):
        for sps in ([1], [2], [1, 1], [3, 1, 2], [2] * 9, [5, 1, 1, 7, 2, 2, 3]):
            sps = np.array(sps, np.int64); n = int(sps.sum())
            sh = "dim%d sps%s" % (dim, list(sps)[:4])
            X = rng.normal(size=(dim, n)) + np.repeat(rng.normal(size=(dim, len(sps))), sps, axis=1)
            chk("dev_means", sh, lambda: max(rel(a, b) for a, b in zip(ctx.dev_means(X, sps), orc.dev_means(X, sps))), 1e-12)
            chk("dev_cov_mat", sh, lambda: max(rel(a, b) for a, b in zip(ctx.dev_cov_mat(X, sps), orc.dev_cov_mat(X, sps))), 1e-11)
            chk("dev_scatter_mat", sh, lambda: max(rel(a, b) for a, b in zip(ctx.dev_scatter_mat(X, sps), orc.dev_scatter_mat(X, sps))), 1e-11)
            if n > dim + len(sps):      # W is SPD only with enough sessions
                chk("dev_wccn_chol", sh, lambda: rel(ctx.dev_wccn_chol(X, sps), orc.dev_wccn_chol(X, sps)), 1e-8)
            A = rng.normal(size=(dim, dim)); A = A @ A.T + np.eye(dim)
            def eig():
                v, l = ctx.sym_eigen(A); vo, lo = orc.sym_eigen(A)
                return max(rel(np.sort(np.diag(l) if np.ndim(l) == 2 else l), np.sort(np.diag(lo) if np.ndim(lo) == 2 else lo)), rel(np.abs(v), np.abs(vo)) if dim < 4 else 0.0)
            chk("sym_eigen", sh, eig, 1e-8)
            chk("dev_efr_matrix", sh, lambda: rel(ctx.dev_efr_matrix(A) @ A @ ctx.dev_efr_matrix(A).T, np.eye(dim)), 1e-8)


if __name__ == "__main__":
    n, failed = run(sys.argv[1:] or ["gmm", "gmm_paths", "tv", "score", "backend"])
    print("%d checks, %d failed" % (n, len(failed)))
    sys.exit(1 if failed else 0)
