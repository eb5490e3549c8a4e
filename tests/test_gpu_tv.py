"""GPU parity for the i-vector path (TVAcc maths) and i-vector scoring vs the CPU oracle.
Tolerance: 1e-6 relative on i-vectors (north_star), tighter where conditioning allows."""
import numpy as np
import pytest

from conftest import make_frames, make_gmm
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from lia_ral_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def tv_problem(C, D, R, U, seed=0, frames=200):
    rng = np.random.default_rng(seed)
    w, mean, iv = make_gmm(C, D, seed=seed)
    lens = rng.integers(frames // 2, frames, U)
    ub = np.concatenate([[0], np.cumsum(lens)])
    x = make_frames(w, mean, iv, int(ub[-1]), seed=seed + 1).astype(np.float64)
    utt = np.repeat(np.arange(U), lens)
    N, F = orc.tv_stats(orc.Gmm(w, mean, iv), x, utt, U)
    Tm = rng.normal(0, 0.05, (R, C * D))
    return dict(w=w, mean=mean, iv=iv, N=N, F=F, Tm=Tm, C=C, D=D, R=R, U=U)


@pytest.mark.parametrize("C,D,R", [(5, 60, 400), (3, 13, 35), (2, 64, 450), (7, 1, 16), (4, 60, 1), (3, 38, 161), (2, 66, 40)])
def test_tett_packed_kernel_matches_the_gemm_form_and_the_oracle(ctx, C, D, R):
    """estimateTETt: k_tett_packed (lower triangle only, written packed; the default) against the batched GEMM + pack form and the
    oracle -- orders with a partial last tile, more rows than one LDS pass holds, dimension counts that are not a multiple of 4,
    and one (D = 66) that the kernel does not serve (GEMM form either way)."""
    rng = np.random.default_rng(C * 1000 + D * 10 + R)
    Tm = rng.normal(0, 0.3, (R, C * D))
    invvar = rng.uniform(0.5, 2.0, C * D)
    out = {}
    for direct in (1, 0):
        ctx.set_option("tv_tett_direct", direct)
        out[direct] = ctx.tv_tett(Tm, invvar, C, D)
    ctx.set_option("tv_tett_direct", 1)
    il = np.tril_indices(R)
    ref = orc.tv_tett(Tm, invvar, C, D)[:, il[0], il[1]]
    assert out[1].shape == ref.shape
    assert relerr(out[1], ref) < 1e-13 and relerr(out[0], ref) < 1e-13 and relerr(out[1], out[0]) < 1e-13


@pytest.mark.parametrize("R,where", [(96, 0), (400, 0), (400, 37), (400, 399), (512, 130), (35, 20)])
def test_ivector_extraction_reports_a_system_that_is_not_positive_definite(ctx, R, where):
    """The batched factorisation flags a non-positive pivot (first column, inside a diagonal block, last column; every kernel variant:
    LDS panels, per-wave panel rows at order 512, the GEMM-built path at an odd order) and the call fails with a numeric error instead
    of returning i-vectors of a broken system; the healthy utterances of the same batch are not what decides the status."""
    from lia_ral_amd import capi
    C, D, U = 4, 12, 5
    p = tv_problem(C, D, R, U, seed=R + where)
    invvar = p["iv"].ravel()
    te = ctx.tv_tett(p["Tm"], invvar, C, D)                    # [C, P] packed lower rows
    F0 = orc.tv_subtract_m(p["N"], p["F"], p["mean"].ravel())
    W = ctx.tv_estimate_w(p["N"], F0, p["Tm"], invvar, te, C, D)
    assert np.isfinite(W).all()
    bad = te.copy()
    d = where * (where + 1) // 2 + where                        # packed position of diagonal element `where`
    bad[:, d] = -1e6                                            # L_u = I + sum_c N_uc TETt_c gets a hugely negative pivot there
    with pytest.raises(capi.GmmivError):
        ctx.tv_estimate_w(p["N"], F0, p["Tm"], invvar, bad, C, D)
    W2 = ctx.tv_estimate_w(p["N"], F0, p["Tm"], invvar, te, C, D)   # the context is usable afterwards
    assert np.array_equal(W2, W)


@pytest.mark.parametrize("C,D,R,U", [(16, 12, 4, 5), (64, 60, 50, 9), (32, 20, 100, 40), (128, 60, 400, 3),
                                     (16, 12, 35, 6), (16, 12, 34, 300), (8, 12, 450, 4),    # odd order (GEMM-built path), > 1 tile pass
                                     (4, 12, 512, 3)])   # order 512: the panel rows no longer fit LDS -- the kernels that fetch them per wave
def test_ivector_extraction(ctx, C, D, R, U):
    p = tv_problem(C, D, R, U, seed=R)
    F0 = orc.tv_subtract_m(p["N"], p["F"], p["mean"].ravel())
    Fg = ctx.tv_subtract_m(p["N"], p["F"].copy(), p["mean"].ravel(), C, D)
    # restore + substractM in one pass (gmmiv_tv_subtract_m_to): out of place and in place, host and device arrays, bitwise the two-step result
    assert np.array_equal(ctx.tv_subtract_m_to(p["N"], p["F"], np.empty_like(p["F"]), p["mean"].ravel(), C, D), Fg)
    import torch
    Fd = torch.from_numpy(p["F"].copy()).cuda()
    ctx.tv_subtract_m_to(torch.from_numpy(p["N"]).cuda(), Fd, Fd, torch.from_numpy(p["mean"].ravel().copy()).cuda(), C, D)
    torch.cuda.synchronize(); ctx.sync()
    assert np.array_equal(Fd.cpu().numpy(), Fg)
    assert relerr(Fg, F0) < 1e-13
    invvar = p["iv"].ravel()
    te_o = orc.tv_tett(p["Tm"], invvar, C, D)
    te_g = ctx.tv_tett(p["Tm"], invvar, C, D)
    il = np.tril_indices(R)
    assert relerr(te_g, te_o[:, il[0], il[1]]) < 1e-12
    W_o = orc.tv_estimate_w(p["N"], F0, p["Tm"], invvar, te_o)
    W_g = ctx.tv_estimate_w(p["N"], F0, p["Tm"], invvar, te_g, C, D)
    assert relerr(W_g, W_o) < 1e-9          # north_star bar is 1e-6


@pytest.mark.parametrize("C,D,R,U", [(16, 12, 4, 30), (32, 20, 40, 300), (8, 12, 35, 20), (8, 12, 450, 5), (4, 12, 512, 4),   # 512: inverse through the non-LDS kernels
                                     (32, 12, 80, 33), (32, 60, 160, 70), (64, 60, 400, 9)])   # R = 80 k, C D = 128 m (no strips in N), odd K = 17 / 35 / 5 in the A^T B products
def test_tv_em_iteration(ctx, C, D, R, U):
    p = tv_problem(C, D, R, U, seed=7, frames=120)
    invvar = p["iv"].ravel()
    F0 = orc.tv_subtract_m(p["N"], p["F"], p["mean"].ravel())
    te_o = orc.tv_tett(p["Tm"], invvar, C, D)
    te_g = ctx.tv_tett(p["Tm"], invvar, C, D)
    o = orc.tv_estimate_a_and_c(p["N"], F0, p["Tm"], invvar, te_o)
    # two calls on disjoint utterance halves accumulate like one (what ranks do before the all-reduce)
    h = U // 2
    g = ctx.tv_estimate_a_and_c(p["N"][:h], F0[:h], p["Tm"], invvar, te_g, C, D)
    W1 = g["W"].copy()
    g = ctx.tv_estimate_a_and_c(p["N"][h:], F0[h:], p["Tm"], invvar, te_g, C, D, acc=g)
    W = np.vstack([W1, g["W"]])
    il = np.tril_indices(R)
    A_o = o["A"].reshape(C, R, R)[:, il[0], il[1]]
    assert relerr(W, o["W"]) < 1e-9
    assert relerr(g["A"], A_o) < 1e-9
    assert relerr(g["Cmx"], o["Cmx"]) < 1e-9
    assert relerr(g["Rm"], o["Rm"]) < 1e-9
    assert relerr(g["r"], o["r"]) < 1e-9
    assert relerr(g["meanW"] / U, o["meanW"]) < 1e-9
    T_o = orc.tv_update_t(o["A"], o["Cmx"], C, D)
    T_g = ctx.tv_update_t(g["A"], g["Cmx"], C, D)
    assert relerr(T_g, T_o) < 1e-7
    m_o, Tm_o = orc.tv_min_divergence(o["Rm"], o["r"], o["meanW"], p["mean"].ravel(), T_o, U, C, D)
    means = p["mean"].ravel().copy()
    m_g, Tm_g = ctx.tv_min_divergence(g["Rm"].copy(), g["r"].copy(), g["meanW"] / U, means, T_g.copy(), U, C, D)
    assert relerr(m_g, m_o) < 1e-7 and relerr(Tm_g, Tm_o) < 1e-7


def test_dgemm_shapes_through_scoring(ctx):
    """The MFMA GEMM behind every TV step: odd sizes, all transposes exercised via the score rules."""
    rng = np.random.default_rng(1)
    # (64, 400, 432), (400, 190, 150), (30, 258, 322): even sizes a little over a multiple of 128 -> the 32-wide strip tiles, both sides
    for dim, M, S in [(5, 3, 7), (50, 130, 129), (400, 260, 17), (33, 1, 300), (64, 400, 432), (400, 190, 150), (30, 258, 322)]:
        m = rng.normal(size=(dim, M)); s = rng.normal(size=(dim, S))
        assert relerr(ctx.score_cosine(m, s), orc.score_cosine(m, s)) < 1e-12
        Q = rng.normal(size=(dim, dim)); Mah = Q @ Q.T / dim + np.eye(dim)
        assert relerr(ctx.score_mahalanobis(m, s, Mah), orc.score_mahalanobis(m, s, Mah)) < 1e-11
        G = rng.normal(size=(dim, dim)) / dim; H = rng.normal(size=(dim, dim)) / dim
        assert relerr(ctx.score_twocov(m, s, G, H), orc.score_twocov(m, s, G, H)) < 1e-11


def test_plda_scoring(ctx):
    rng = np.random.default_rng(2)
    rf, M, S = 40, 37, 53
    Fm = rng.normal(size=(60, rf))
    FTJF = Fm.T @ Fm / 60
    models = rng.normal(size=(rf, M)); segs = rng.normal(size=(rf, S))
    nsess = np.array([1] * 10 + [3] * 20 + [1] * 7)
    got = ctx.score_plda(models, nsess, segs, FTJF)
    ref = orc.score_plda(models, nsess, segs, FTJF)
    assert relerr(got, ref) < 1e-10


def test_iv_normalisation_and_orthonormalize(ctx):
    rng = np.random.default_rng(5)
    din, dout, n = 60, 40, 333
    X = rng.normal(size=(din, n)) + 0.5; mu = X.mean(1); M = rng.normal(size=(dout, din))
    for mean, rot, ln in [(mu, M, True), (mu, None, False), (None, M, False), (None, None, True), (mu, None, True)]:
        got = ctx.iv_normalize(X, mean, rot, ln)
        ref = orc.iv_normalize(X, mean, rot, ln)
        assert relerr(got, ref) < 1e-12
    # sphericalNuisanceNormalization, 2 iterations == numpy
    M2 = rng.normal(size=(dout, dout)); mu2 = rng.normal(size=dout) * 0.01
    Y = ctx.iv_normalize(ctx.iv_normalize(X, mu, M, True), mu2, M2, True)
    Z = M @ (X - mu[:, None]); Z /= np.linalg.norm(Z, axis=0)
    Z = M2 @ (Z - mu2[:, None]); Z /= np.linalg.norm(Z, axis=0)
    assert relerr(Y, Z) < 1e-12 and np.allclose(np.linalg.norm(Y, axis=0), 1.0, atol=1e-13)
    # classical Gram-Schmidt of T rows
    T = rng.normal(size=(24, 1000))
    T[5] = 0.0                                     # a zero row stays zero (AccumulateTVStat.cpp:1577-1581)
    Q = ctx.tv_orthonormalize_t(T.copy())
    Qo = orc.tv_orthonormalize_t(T)
    assert relerr(Q, Qo) < 1e-10
    keep = [i for i in range(24) if i != 5]
    assert np.allclose(Q[keep] @ Q[keep].T, np.eye(23), atol=1e-10) and not Q[5].any()
    # full-rank T: the Gram-matrix Cholesky route (Q = L^-1 T), same Q as the step-by-step scheme
    for R, SV in [(24, 1000), (100, 7680)]:
        T = rng.normal(size=(R, SV)) * 0.05 + 0.01
        Q = ctx.tv_orthonormalize_t(T.copy())
        assert relerr(Q, orc.tv_orthonormalize_t(T)) < 1e-10
        assert np.allclose(Q @ Q.T, np.eye(R), atol=1e-10)


@pytest.mark.parametrize("U,C,D,R", [(7, 16, 12, 20), (300, 64, 60, 40), (3, 128, 60, 100)])
def test_approximate_extractors_match_oracle(ctx, U, C, D, R):
    """IvExtractor modes ubmWeight and eigenDecomposition (AccumulateTVStat.cpp:1225-1242, 1600-1609, 2837-2855,
    3116-3136, 2348-2396, 2566-2609) and substractMplusTW (:1379-1399) against the oracle."""
    rng = np.random.default_rng(U + R)
    SV = C * D
    N = rng.uniform(0.0, 40.0, (U, C)); F = rng.normal(size=(U, SV)) * 5
    means = rng.normal(size=SV); iv = rng.uniform(0.5, 2.0, SV); T = 0.05 * rng.normal(size=(R, SV))
    wgt = rng.dirichlet(np.ones(C)); Wv = rng.normal(size=(U, R))
    Fn = ctx.tv_norm_statistics(N, F.copy(), means, iv, C, D)
    assert relerr(Fn, orc.tv_norm_statistics(N, F, means, iv)) < 1e-13
    Fs = ctx.tv_subtract_m_plus_tw(N, F.copy(), means, T, Wv, C, D)
    assert relerr(Fs, orc.tv_subtract_m_plus_tw(N, F, means, T, Wv)) < 1e-12
    Tn = ctx.tv_norm_t(T.copy(), iv, C, D)
    assert relerr(Tn, orc.tv_norm_t(T, iv, C)) < 1e-14
    Wm = ctx.tv_weighted_cov(Tn, wgt, C, D)
    Wo = orc.tv_weighted_cov(Tn, wgt)
    assert relerr(Wm, Wo) < 1e-12
    Q = np.linalg.qr(rng.normal(size=(R, R)))[0]
    Dm = ctx.tv_approximate_tctc(Tn, Q, C, D)
    Do = orc.tv_approximate_tctc(Tn, Q, C)
    assert relerr(Dm, Do) < 1e-12
    assert relerr(ctx.tv_approximate_tctc(Tn, Q, C, D, out=Dm.copy()), 2 * Do) < 1e-12   # accumulates like the reference
    w1 = ctx.tv_estimate_w_ubm_weight(N, Fn, Tn, Wo, C, D)
    assert relerr(w1, orc.tv_estimate_w_ubm_weight(N, Fn, Tn, Wo)) < 1e-9
    w2 = ctx.tv_estimate_w_eigen(N, Fn, Tn, Do, Q, C, D)
    assert relerr(w2, orc.tv_estimate_w_eigen(N, Fn, Tn, Do, Q)) < 1e-9
    acc = np.ones((U, R))
    assert relerr(ctx.tv_estimate_w_eigen(N, Fn, Tn, Do, Q, C, D, out=acc) - 1.0, w2) < 1e-9


@pytest.mark.parametrize("dim,rf,rg", [(40, 10, 6), (200, 50, 0), (400, 100, 80)])
def test_plda_precompute_and_native_scoring(ctx, dim, rf, rg):
    """PldaModel::preComputation + FTJ / FTJF (PldaTools.cpp:2950-2972, 4494-4496), then the native scoring chain
    rotateLeft(FTJ) -> pldaScoring against the oracle."""
    rng = np.random.default_rng(dim)
    F = rng.normal(size=(dim, rf)) / np.sqrt(dim); G = rng.normal(size=(dim, rg)) / np.sqrt(dim) if rg else None
    A = rng.normal(size=(dim, dim)); S = A @ A.T / dim + np.eye(dim)
    FTJ, FTJF = ctx.plda_precompute(F, G, S)
    oJ, oJF = orc.plda_precompute(F, G, S)
    assert relerr(FTJ, oJ) < 1e-10 and relerr(FTJF, oJF) < 1e-10
    M, Sg = 9, 14
    models = rng.normal(size=(dim, M)); segs = rng.normal(size=(dim, Sg))
    pm = ctx.iv_normalize(np.ascontiguousarray(models), None, FTJ, length_norm=False)
    ps = ctx.iv_normalize(np.ascontiguousarray(segs), None, FTJ, length_norm=False)
    ns = np.array([1, 1, 2, 2, 2, 3, 1, 1, 4])
    got = ctx.score_plda(np.ascontiguousarray(pm * ns), ns, ps, FTJF)
    ref = orc.score_plda((oJ @ models) * ns, ns, oJ @ segs, oJF)
    assert relerr(got, ref) < 1e-9


def _dev_set(dim, sps, seed):
    rng = np.random.default_rng(seed)
    k, n = len(sps), int(np.sum(sps))
    cls = np.repeat(np.arange(k), sps)
    X = np.ascontiguousarray((rng.normal(size=(dim, k)) * 1.5)[:, cls] + rng.normal(size=(dim, n)))
    return X


@pytest.mark.parametrize("dim,nspk,seed", [(10, 6, 1), (64, 40, 2), (400, 300, 3)])
def test_backend_estimation_matches_oracle(ctx, dim, nspk, seed):
    """PldaDev::computeAll / computeCovMat / computeWccnChol / computeMahalanobis / computeScatterMat
    (PldaTools.cpp:353-387, 527-566, 1124-1176, 1366-1378, 1610-1644) on the device against the oracle."""
    rng = np.random.default_rng(seed)
    sps = rng.integers(2, 9, nspk)
    sps[0] = 1
    X = _dev_set(dim, sps, seed)
    mean, sm = ctx.dev_means(X, sps)
    om, osm = orc.dev_means(X, sps)
    assert relerr(mean, om) < 1e-13 and relerr(sm, osm) < 1e-13
    S, W, B = ctx.dev_cov_mat(X, sps)
    oS, oW, oB = orc.dev_cov_mat(X, sps)
    assert relerr(S, oS) < 1e-12 and relerr(W, oW) < 1e-12 and relerr(B, oB) < 1e-12
    SB, SW = ctx.dev_scatter_mat(X, sps)
    oSB, oSW = orc.dev_scatter_mat(X, sps)
    assert relerr(SB, oSB) < 1e-12 and relerr(SW, oSW) < 1e-12
    if X.shape[1] > 2 * dim:   # W must be invertible
        assert relerr(ctx.dev_mahalanobis(X, sps), np.linalg.inv(oW)) < 1e-8
        U = ctx.dev_wccn_chol(X, sps)
        assert relerr(U, orc.dev_wccn_chol(X, sps)) < 1e-8 and np.allclose(U, np.triu(U))


@pytest.mark.parametrize("dim", [12, 100])
def test_efr_lda_and_eigen(ctx, dim):
    """computeEigenProblem on symmetric input, the EFR / sphNorm matrix and LDA (PldaTools.cpp:1490-1535, 1852-1902,
    1381-1413): same results as the oracle, and the defining properties."""
    sps = np.full(6 * dim // 4, 4)
    X = _dev_set(dim, sps, dim)
    S, W, B = orc.dev_cov_mat(X, sps)
    vect, val = ctx.sym_eigen(S)
    assert np.all(np.diff(val) <= 0) and np.allclose(vect @ np.diag(val) @ vect.T, S, atol=1e-10 * val[0])
    ov, ol = orc.sym_eigen(S)
    assert relerr(val, ol) < 1e-12 and relerr(np.abs(vect), np.abs(ov)) < 1e-8
    M = ctx.dev_efr_matrix(S)
    assert np.allclose(M @ S @ M.T, np.eye(dim), atol=1e-9)
    assert relerr(np.abs(M), np.abs(orc.dev_efr_matrix(S))) < 1e-8
    rank = min(5, dim - 1)
    L, lam = ctx.dev_lda(W, B, rank)
    oL, olam = orc.dev_lda(W, B, rank)
    assert relerr(lam, olam) < 1e-10 and relerr(np.abs(L), np.abs(oL)) < 1e-7
    EP = np.linalg.inv(W) @ B
    for j in range(rank):
        assert np.allclose(EP @ L[j], lam[j] * L[j], atol=1e-8 * lam[0])
    # the EFR training loop of PldaDev::sphericalNuisanceNormalization: covariance -> matrix -> center / rotate / lengthNorm
    Y = ctx.iv_normalize(X, X.mean(1), M, length_norm=True)
    assert np.allclose(np.linalg.norm(Y, axis=0), 1.0)
    Yo = M @ (X - X.mean(1)[:, None]); Yo /= np.linalg.norm(Yo, axis=0)
    assert relerr(Y, Yo) < 1e-10


@pytest.mark.parametrize("dim,rf,rg,nspk", [(8, 3, 2, 8), (60, 20, 10, 120), (200, 60, 40, 400), (40, 12, 0, 60)])   # rankG 0: simplified PLDA
def test_plda_em_iterations_match_oracle(ctx, dim, rf, rg, nspk):
    """PldaModel::em_iteration (PldaTools.cpp:2329-2343, 2359-2484, 2790-2815): three iterations, every quantity
    (centred data, F, G, Sigma, Delta) against the oracle; numpy arrays in place and a device-resident X."""
    import torch
    rng = np.random.default_rng(dim)
    sps = rng.integers(1, 6, nspk)
    k, n = nspk, int(sps.sum())
    cls = np.repeat(np.arange(k), sps)
    Ft = rng.normal(size=(dim, rf)); Gt = 0.5 * rng.normal(size=(dim, rg))
    X = Ft @ rng.normal(size=(rf, k))[:, cls] + Gt @ rng.normal(size=(rg, n)) + 0.3 * rng.normal(size=(dim, n)) + 0.2
    F = rng.normal(size=(dim, rf)); G = 0.5 * rng.normal(size=(dim, rg)); Sigma = np.cov(X) + 0.1 * np.eye(dim); Delta = np.zeros(dim)
    ref = (X.copy(), F.copy(), G.copy(), Sigma.copy(), Delta.copy())
    Xg = np.ascontiguousarray(X.copy()); Xd = torch.from_numpy(X.copy()).cuda()
    Fd, Gd, Sd, Dd = F.copy(), G.copy(), Sigma.copy(), Delta.copy()
    for it in range(3):
        ref = orc.plda_em_iteration(ref[0], sps, *ref[1:])
        ctx.plda_em_iteration(Xg, sps, F, G, Sigma, Delta)
        ctx.plda_em_iteration(Xd, sps, Fd, Gd, Sd, Dd)
        for got, want in zip((Xg, F, G, Sigma, Delta), ref):
            if want.size:
                assert relerr(got, want) < 1e-8, it
        assert relerr(Xd.cpu().numpy(), ref[0]) < 1e-8 and relerr(Fd, ref[1]) < 1e-8 and relerr(Sd, ref[3]) < 1e-8


@pytest.mark.parametrize("C,D,R,nspk", [(6, 5, 3, 7), (64, 60, 40, 30), (16, 12, 10, 400)])
def test_jfa_steps_match_oracle(ctx, C, D, R, nspk):
    """JFAAcc steps (AccumulateJFAStat.cpp): the subtract family, substractUX over the sessions of each speaker, estimateZ /
    estimateZMAP / estimateZandD, and estimateAndInverseL_EV + estimateYandV through the TV E-step entry point."""
    rng = np.random.default_rng(C + R)
    SV = C * D
    nses = rng.integers(1, 4, nspk); sb = np.concatenate([[0], np.cumsum(nses)]); nsess = int(sb[-1])
    owner = np.repeat(np.arange(nspk), nses)
    N = rng.uniform(0.5, 20, (nspk, C)); F = rng.normal(size=(nspk, SV))
    Nh = rng.uniform(0.2, 8, (nsess, C)); Fh = rng.normal(size=(nsess, SV))
    m = rng.normal(size=SV); V = rng.normal(size=(R, SV)) * 0.3; U = rng.normal(size=(R, SV)) * 0.2
    Y = rng.normal(size=(nspk, R)); X = rng.normal(size=(nsess, R)); Dm = rng.uniform(0.1, 1, SV); Z = rng.normal(size=(nspk, SV))
    iv = rng.uniform(0.5, 2, SV)
    assert relerr(ctx.jfa_subtract(N, F.copy(), C, D, means=m, Dm=Dm, Z=Z), orc.jfa_subtract(N, F, None, m, None, None, Dm, Z)) < 1e-13
    assert relerr(ctx.jfa_subtract(N, F.copy(), C, D, means=m, T=V, W=Y), orc.jfa_subtract(N, F, None, m, V, Y)) < 1e-12
    assert relerr(ctx.jfa_subtract(Nh, Fh.copy(), C, D, owner=owner, means=m, T=V, W=Y, Dm=Dm, Z=Z),
                  orc.jfa_subtract(Nh, Fh, owner, m, V, Y, Dm, Z)) < 1e-12
    assert relerr(ctx.jfa_subtract(Nh, Fh.copy(), C, D, T=U, W=X), orc.jfa_subtract(Nh, Fh, None, None, U, X)) < 1e-12   # UX alone
    assert relerr(ctx.jfa_subtract_sessions(sb, Nh, F.copy(), U, X, C, D), orc.jfa_subtract_sessions(sb, Nh, F, U, X)) < 1e-12
    assert relerr(ctx.jfa_estimate_z(N, F, iv, Dm, C, D), orc.jfa_estimate_z(N, F, iv, Dm)) < 1e-13
    assert relerr(ctx.jfa_estimate_z(N, F, iv, Dm, C, D, tau=14.0), orc.jfa_estimate_z(N, F, iv, Dm, 14.0)) < 1e-13
    Dg = Dm.copy()
    Zg = ctx.jfa_estimate_z_and_d(N, F, iv, Dg, C, D)
    Zo, Do = orc.jfa_estimate_z_and_d(N, F, iv, Dm)
    assert relerr(Zg, Zo) < 1e-13 and relerr(Dg, Do) < 1e-12
    from lia_ral_amd import capi
    with pytest.raises(capi.GmmivError):
        ctx.jfa_subtract(Nh, Fh.copy(), C, D, owner=owner + 1, means=m, T=V, W=Y)          # owner out of range
    # eigenvoice E-step: packed A of the TV entry point against the full matrices of the JFA loops
    te_o = orc.tv_tett(V, iv, C, D)
    Yo, Ao, Co = orc.jfa_estimate_y_and_v(N, F, V, iv, te_o)
    te_g = ctx.tv_tett(V, iv, C, D)
    g = ctx.tv_estimate_a_and_c(N, F, V, iv, te_g, C, D)
    il = np.tril_indices(R)
    assert relerr(g["W"], Yo) < 1e-9 and relerr(g["A"], Ao[:, il[0], il[1]]) < 1e-9 and relerr(g["Cmx"], Co) < 1e-9


def test_twocov_mix_part_trials_mask_and_model_blocks(ctx):
    """PldaTest::twoCovScoringMixPart (PldaTools.cpp:3923-3949, accumulating), the _trials mask of cosineDistance /
    mahalanobisDistance (:3871, :3889) and the model-block tiling of the score matrix over ranks."""
    from lia_ral_amd import capi
    from lia_ral_amd.dist import score_model_block
    rng = np.random.default_rng(9)
    dim, M, S = 60, 137, 211
    m = rng.normal(size=(dim, M)); s = rng.normal(size=(dim, S))
    G = rng.normal(size=(dim, dim)) / dim
    base = rng.normal(size=(M, S))
    got = ctx.score_twocov_mix_part(m, s, G, base.copy())
    ref = base + orc.score_twocov(m, s, G, np.zeros((dim, dim)))          # H = 0 leaves (m+s)'G(m+s)
    assert relerr(got, ref) < 1e-11
    # the same accumulating epilogue on the 32-wide strip tiles of k_dgemm (M, S a little over multiples of 128, both even)
    m2 = rng.normal(size=(dim, 400)); s2 = rng.normal(size=(dim, 278)); base2 = rng.normal(size=(400, 278))
    got2 = ctx.score_twocov_mix_part(m2, s2, G, base2.copy())
    assert relerr(got2, base2 + orc.score_twocov(m2, s2, G, np.zeros((dim, dim)))) < 1e-11
    # twoCovScoring == mix part on zeros minus the model / segment terms (:4127-4171)
    H = rng.normal(size=(dim, dim)) / dim
    full = ctx.score_twocov(m, s, G, H)
    mp = ctx.score_twocov_mix_part(m, s, G, np.zeros((M, S)))
    md = np.einsum("im,ik,km->m", m, H, m); sd = np.einsum("is,ik,ks->s", s, H, s)
    assert relerr(mp - md[:, None] - sd[None, :], full) < 1e-11
    # trials mask
    trials = rng.random((M, S)) < 0.3
    Q = rng.normal(size=(dim, dim)); Mah = Q @ Q.T / dim + np.eye(dim)
    gc = ctx.score_apply_trials(trials, ctx.score_cosine(m, s))
    gm = ctx.score_apply_trials(trials, ctx.score_mahalanobis(m, s, Mah))
    assert relerr(gc, orc.score_cosine(m, s, trials)) < 1e-12 and relerr(gm, orc.score_mahalanobis(m, s, Mah, trials)) < 1e-11
    assert np.all(gc[~trials] == 0.0) and np.all(gm[~trials] == 0.0)
    # model blocks: three "ranks" reproduce the rows of the one-call matrix bit for bit, PLDA with its per-model session counts too
    whole = ctx.score_mahalanobis(m, s, Mah)
    nsess = np.sort(rng.integers(1, 4, M)); FTJF = G @ G.T + np.eye(dim)
    whole_p = ctx.score_plda(m * nsess, nsess, s, FTJF)
    rows = 0
    for r in range(3):
        m0, m1, blk = score_model_block(lambda a, b: ctx.score_mahalanobis(a, b, Mah), m, s, r, 3)
        assert (m0, m1) == capi.shard_range(M, r, 3) and np.array_equal(blk, whole[m0:m1])
        _, _, bp = score_model_block(lambda a, b, n: ctx.score_plda(a, n, b, FTJF), m * nsess, s, r, 3, nsess=nsess)
        assert relerr(bp, whole_p[m0:m1]) < 1e-13
        rows += m1 - m0
    assert rows == M


def test_min_divergence_device_factor_matches_host_and_oracle(ctx):
    """gmmiv_tv_min_divergence: R normalised and factored by one workgroup of k_chol_left (default, even R) == the host route
    (option tv_md_device 0, and odd R) == the oracle (TVAcc::minDivergence, AccumulateTVStat.cpp:1003-1047); a non-positive
    definite R is an error on both routes, never a NaN matrix."""
    from lia_ral_amd import capi
    rng = np.random.default_rng(11)
    C, D, U = 6, 10, 50
    for R in (24, 66, 25):
        W = rng.normal(size=(U, R)) + 0.3
        Rm = W.T @ W + 0.1 * U * np.eye(R); r = W.sum(0); meanW = r / U
        means = rng.normal(size=C * D); T = rng.normal(size=(R, C * D))
        m_o, T_o = orc.tv_min_divergence(Rm.copy(), r.copy(), meanW, means.copy(), T.copy(), U, C, D)
        outs = []
        for dev_route in (1, 0):
            prev = ctx.set_option("tv_md_device", dev_route)
            Rg, rg, mg, Tg = Rm.copy(), r.copy(), means.copy(), T.copy()
            ctx.tv_min_divergence(Rg, rg, meanW, mg, Tg, U, C, D)
            ctx.set_option("tv_md_device", prev)
            assert relerr(mg, m_o) < 1e-12 and relerr(Tg, T_o) < 1e-11
            assert relerr(rg, r / U) < 1e-14 and relerr(Rg, Rm / U - np.outer(r / U, r / U)) < 1e-13
            outs.append(Tg)
        assert relerr(outs[0], outs[1]) < 1e-12
    bad = -np.eye(24)
    for dev_route in (1, 0):
        prev = ctx.set_option("tv_md_device", dev_route)
        with pytest.raises(capi.GmmivError):
            ctx.tv_min_divergence(bad.copy(), np.zeros(24), np.zeros(24), np.zeros(C * D), np.ones((24, C * D)), U, C, D)
        ctx.set_option("tv_md_device", prev)


def test_tv_estep_batching_does_not_change_the_statistics(ctx):
    """gmmiv_tv_estimate_a_and_c: utterances go through the solve in batches of tv_batch, their E_u are kept for a super-batch
    (tv_acc_mb) and A / Cmx / R / r are updated once per super-batch.  Ragged batch and super-batch boundaries (U = 75 with
    tv_batch 16: super-batches of 64 + 11 utterances when only one batch's worth of E fits, one of 80 by default) == the oracle."""
    C, D, R, U = 16, 12, 40, 75
    p = tv_problem(C, D, R, U, seed=21, frames=100)
    invvar = p["iv"].ravel()
    F0 = orc.tv_subtract_m(p["N"], p["F"], p["mean"].ravel())
    o = orc.tv_estimate_a_and_c(p["N"], F0, p["Tm"], invvar, orc.tv_tett(p["Tm"], invvar, C, D))
    te = ctx.tv_tett(p["Tm"], invvar, C, D)
    il = np.tril_indices(R)
    A_o = o["A"].reshape(C, R, R)[:, il[0], il[1]]
    outs = []
    prev_b = ctx.set_option("tv_batch", 16)
    for acc_mb in (0, 8192):
        prev_a = ctx.set_option("tv_acc_mb", acc_mb)
        g = ctx.tv_estimate_a_and_c(p["N"], F0, p["Tm"], invvar, te, C, D)
        ctx.set_option("tv_acc_mb", prev_a)
        for k, ref in (("W", o["W"]), ("A", A_o), ("Cmx", o["Cmx"]), ("Rm", o["Rm"]), ("r", o["r"])):
            assert relerr(g[k], ref) < 1e-9, (acc_mb, k)
        outs.append(g)
    ctx.set_option("tv_batch", prev_b)
    assert relerr(outs[0]["A"], outs[1]["A"]) < 1e-12 and np.array_equal(outs[0]["W"], outs[1]["W"])


def test_options_are_state_of_the_context_not_of_the_thread():
    """gmmiv_ctx_set_option: the kernel-launcher options ("z_waves", "z_depth_*", "z_tv4", "chol_lds", "chol_gemm", "gemm_*")
    live in the context.  Two contexts on ONE thread keep different settings and each call runs with its own context's set; one
    context driven from a SECOND thread shows that thread the same settings."""
    import threading
    import torch
    from lia_ral_amd import capi
    a = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    b = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    assert a.set_option("z_waves", 4) == 8 and b.set_option("z_waves", 16) == 8         # b did not see a's 4
    assert a.set_option("chol_gemm", 1) == 0 and b.set_option("chol_gemm", 0) == 0
    assert a.set_option("gemm_narrow", 0) == 1 and b.set_option("gemm_narrow", 1) == 1
    assert a.set_option("z_depth_tv", 2) == 4 and b.set_option("z_depth_tv", 4) == 4
    C, D, R, U = 8, 12, 12, 9
    rng = np.random.default_rng(5)
    N = rng.gamma(0.8, 3.0, (U, C)); F = rng.normal(size=(U, C * D)); Tm = rng.normal(0, 0.1, (R, C * D)); iv = rng.uniform(0.5, 2, C * D)
    Wo = orc.tv_estimate_w(N, F, Tm, iv, orc.tv_tett(Tm, iv, C, D))

    def run(ctx):
        return ctx.tv_estimate_w(N, F, Tm, iv, ctx.tv_tett(Tm, iv, C, D), C, D)
    Wa = run(a)
    assert a.set_option("kopts_bound", 0) == 1 and b.set_option("kopts_bound", 0) == 0  # a's call bound a's set to this thread
    Wb = run(b)
    assert a.set_option("kopts_bound", 0) == 0 and b.set_option("kopts_bound", 0) == 1
    assert relerr(Wa, Wo) < 1e-9 and relerr(Wb, Wo) < 1e-9                               # GEMM-built and fused factorisations
    w, mean, ivm = make_gmm(32, 20, seed=2)
    x = make_frames(w, mean, ivm, 700, seed=3)
    ga, gb = a.gmm(w, mean, ivm), b.gmm(w, mean, ivm)
    Na, Fa = ga.tv_stats(x, [0, 300, 700]); Nb, Fb = gb.tv_stats(x, [0, 300, 700])      # k_stats_z <4 waves> and <16 waves>
    assert np.array_equal(Na, Nb) and np.array_equal(Fa, Fb)                              # the shapes are bit-identical by construction
    seen = {}

    def other_thread():
        seen["bound_before"] = (a.set_option("kopts_bound", 0), b.set_option("kopts_bound", 0))
        seen["a"] = (a.set_option("z_waves", 4), a.set_option("chol_gemm", 1), a.set_option("gemm_narrow", 0), a.set_option("z_depth_tv", 2))
        seen["b"] = (b.set_option("z_waves", 16), b.set_option("chol_gemm", 0), b.set_option("gemm_narrow", 1), b.set_option("z_depth_tv", 4))
        seen["W"] = run(a)
        seen["bound_after"] = (a.set_option("kopts_bound", 0), b.set_option("kopts_bound", 0))
    th = threading.Thread(target=other_thread)
    th.start(); th.join()
    assert seen["bound_before"] == (0, 0) and seen["bound_after"] == (1, 0)
    assert seen["a"] == (4, 1, 0, 2) and seen["b"] == (16, 0, 1, 4)                       # the settings followed the contexts
    assert np.array_equal(seen["W"], Wa)                                                  # same context, other thread: same launches
    assert a.set_option("no_such_option", 1) == -1
    ga.close(); gb.close(); a.close(); b.close()


@pytest.mark.parametrize("R", [160, 400])
def test_aux_on_80_wide_tiles(R):
    """aux = F (T Sigma^-1)^T of estimateW (AccumulateTVStat.cpp:2147-2152) is a split-K NT product with N = R columns: for R a multiple of
    80 (and not of 128) on full row tiles it runs on 128 x 80 tiles with a 4 x 1 wave grid instead of 128 x 128 tiles plus a strip that
    re-reads all of F (option "gemm_nt80").  256 utterances (two row tiles), K = 3840 (split-K): same i-vectors with the option on and off,
    and the oracle's on a few rows."""
    import torch
    from lia_ral_amd import capi
    C, D, U = 64, 60, 256
    rng = np.random.default_rng(R)
    N = rng.gamma(0.8, 3.0, (U, C)); F = rng.normal(size=(U, C * D)) * np.sqrt(np.repeat(N, D, 1) + 0.1)
    Tm = rng.normal(0, 0.03, (R, C * D)); iv = rng.uniform(0.5, 2, C * D)
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    te = ctx.tv_tett(Tm, iv, C, D)
    W = {}
    for on in (1, 0):
        ctx.set_option("gemm_nt80", on)
        W[on] = ctx.tv_estimate_w(N, F, Tm, iv, te, C, D)
    ctx.set_option("gemm_nt80", 1)
    assert relerr(W[1], W[0]) < 1e-13
    rows = [0, 1, 127, 128, 255]
    Wo = orc.tv_estimate_w(N[rows], F[rows], Tm, iv, orc.tv_tett(Tm, iv, C, D))
    assert relerr(W[1][rows], Wo) < 1e-10
    ctx.close()
