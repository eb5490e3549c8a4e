#!/usr/bin/env python3
"""Default-path results of a small TrainWorld iteration, a top-C pass and an IvExtractor / T-matrix E-step run, as BITS.
  tools/bitwise_fixture.py write out.json     compute with the library capi loads (GMMIV_LIB_PATH selects another build), store SHA-256 of every array's bytes
  tools/bitwise_fixture.py check ref.json     compute again and compare the digests, key by key
tests/golden/r05_bitwise.json was written by the round-5 library (commit 4b2296d, built into tools/bin/libgmmiv_r05.so) on an MI355X: the
round-6 removal of the measured-slower kernel variants from libgmmiv.so must not move one bit of any default path
(tests/test_gpu_gmm.py::test_default_paths_are_bitwise_the_round_5_results)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def compute():
    from conftest import make_frames, make_gmm
    from lia_ral_amd import capi
    ctx = capi.Context(0)
    out = {}
    for C, D, T in ((128, 60, 5000), (2048, 60, 3000), (37, 13, 700)):
        w, mean, iv = make_gmm(C, D, seed=C + D)
        x = make_frames(w, mean, iv, T, seed=T)
        g = ctx.gmm(w, mean, iv)
        k = "%dx%dx%d" % (C, D, T)
        out["em_" + k] = g.em_accumulate(x)
        out["llk_" + k] = g.llk(x)
        wn, mn, cn = g.em_get(out["em_" + k], mean, 1.0 / iv)
        out["emget_w_" + k], out["emget_mean_" + k], out["emget_cov_" + k] = wn, mn, cn
        d = g.llk_determine_top(x, 10)
        for f in ("idx", "lk", "nontop_llk", "nontop_w", "llk"):
            out["top10_%s_%s" % (f, k)] = d[f]
        out["usetop_" + k] = g.llk_use_top(x, d["idx"], d["nontop_llk"])
        d20 = g.llk_determine_top(x, min(20, C))
        out["top20_idx_" + k] = d20["idx"]; out["top20_lk_" + k] = d20["lk"]
        out["occ_" + k] = g.occ(x[:64])
        lens = [300, 0, 1, 257, T - 558]
        ub = np.concatenate([[0], np.cumsum(lens)])
        N, F = g.tv_stats(x, ub)
        out["N_" + k], out["F_" + k] = N, F
        # IvExtractor + one T-matrix E-step / M-step on those statistics
        R = 24
        rng = np.random.default_rng(7)
        Tm = 0.05 * rng.normal(size=(R, C * D))
        invvar = iv.ravel().copy(); means = mean.ravel().copy()
        Fc = ctx.tv_subtract_m(N, F.copy(), means, C, D)
        te = ctx.tv_tett(Tm, invvar, C, D)
        out["tett_" + k] = te
        out["W_" + k] = ctx.tv_estimate_w(N, Fc, Tm, invvar, te, C, D)
        acc = ctx.tv_estimate_a_and_c(N, Fc, Tm, invvar, te, C, D)
        for f in ("A", "Cmx", "Rm", "r", "meanW", "W"):
            out["estep_%s_%s" % (f, k)] = np.asarray(acc[f])
        Tn = ctx.tv_update_t(acc["A"], acc["Cmx"], C, D)
        out["Tnew_" + k] = np.asarray(Tn).copy()
        m2 = means.copy()
        ctx.tv_min_divergence(acc["Rm"].copy(), acc["r"].copy(), acc["meanW"] / len(lens), m2, Tn, len(lens), C, D)
        out["Tmd_" + k] = np.asarray(Tn); out["means_md_" + k] = m2
        g.close()
    # the Cholesky family at the order of the BASELINE configs (rank 400: 13 panels per system, several row-tile rounds per wave) and at
    # orders with a partial last block -- the i-vector solve and the E-step accumulators of a few utterances
    for R, C, D, U in ((400, 8, 12, 7), (200, 4, 12, 5), (130, 3, 5, 4)):
        rng = np.random.default_rng(R)
        N = rng.uniform(0.5, 40.0, (U, C)); invvar = rng.uniform(0.5, 2.0, C * D)
        F = rng.normal(size=(U, C * D)) * np.repeat(N, D, axis=1)
        Tm = 0.05 * rng.normal(size=(R, C * D))
        te = ctx.tv_tett(Tm, invvar, C, D)
        k = "R%d" % R
        out["W_" + k] = ctx.tv_estimate_w(N, F, Tm, invvar, te, C, D)
        acc = ctx.tv_estimate_a_and_c(N, F, Tm, invvar, te, C, D)
        for f in ("A", "Cmx", "Rm", "r", "meanW", "W"):
            out["estep_%s_%s" % (f, k)] = np.asarray(acc[f])
        out["Tnew_" + k] = np.asarray(ctx.tv_update_t(acc["A"], acc["Cmx"], C, D)).copy()
    # scoring
    rng = np.random.default_rng(3)
    M, S, dim = 70, 130, 40
    models = rng.normal(size=(dim, M)); segs = rng.normal(size=(dim, S))
    Q = rng.normal(size=(dim, dim)); Q = Q @ Q.T / dim + np.eye(dim)
    out["cos"] = ctx.score_cosine(models, segs)
    out["mah"] = ctx.score_mahalanobis(models, segs, Q)
    ctx.close()
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


def digests(arrs):
    import hashlib
    return {k: {"sha256": hashlib.sha256(v.tobytes()).hexdigest(), "shape": list(v.shape), "dtype": str(v.dtype)} for k, v in arrs.items()}


if __name__ == "__main__":
    import json
    mode, path = sys.argv[1], sys.argv[2]
    got = digests(compute())
    if mode == "write":
        json.dump({"library": os.environ.get("GMMIV_LIB_PATH") or "in-tree", "arrays": got}, open(path, "w"), indent=0, sort_keys=True)
        print("wrote the digests of %d arrays to %s" % (len(got), path))
    else:
        ref = json.load(open(path))["arrays"]
        bad = [k for k in ref if got.get(k) != ref[k]]
        print("%d arrays, %d differ: %s" % (len(ref), len(bad), bad[:20]))
        sys.exit(1 if bad else 0)
