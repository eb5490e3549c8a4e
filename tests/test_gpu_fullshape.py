"""GPU parity at the REAL shapes of BASELINE.json configs 3, 4 and 5 (C = 2048 Gaussians, D = 60, rank R = 400,
dim-400 scoring with M, S >= 4096 that are not multiples of the 128-wide GEMM tile).

The small-shape tests (test_gpu_tv.py) cannot reach the code these configurations actually run: the interior 128x128 tiles of
k_dgemm next to its clamped edge-tile side stream, split-K over SV = 122 880, the packed L / A GEMMs with P = 80 200 columns,
the 400-order chol_fused kernels with 1024 systems per batch.  The oracle is a scalar restatement, so every check is sized
to keep it under about a minute:
  * per-Gaussian quantities (TETt, T_c = A_c^-1 Cmx_c) are compared on a handful of Gaussians -- they are independent per
    Gaussian (AccumulateTVStat.cpp:777-805, 981-1000) -- out of a full-shape GPU call;
  * per-utterance quantities (i-vectors) on a handful of utterances out of a batch that spans interior AND edge tiles;
  * the accumulators A / Cmx / R / r (sums over utterances) against the oracle at U = 4, plus the size-independent property
    that a 300-utterance batch equals the sum of 75 four-utterance batches (the oracle-checked shape);
  * scores on sampled rows x columns (every trial is independent, PldaTools.cpp:3882-3909, 4252-4268).
Tolerances: 1e-9 relative on i-vectors and accumulators (north_star bar: 1e-6), 1e-6 on T after 2048 inverses."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

C, D, R = 2048, 60, 400
SV = C * D
SEL = [0, 1, 1023, 1024, 2046, 2047]      # Gaussians compared with the oracle (first / middle / last GEMM batch entries)


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def ctx():
    from lia_ral_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def prob():
    """Statistics as an IvExtractor run sees them after substractM: N ~ occupancies of ~3000-frame utterances, centred F."""
    rng = np.random.default_rng(2024)
    U = 300
    N = rng.gamma(0.6, 2.5, (U, C))                       # mean 1.5 per Gaussian = 3000 frames / 2048
    N[1, :7] = 0.0                                        # Gaussians an utterance never visits
    F = rng.normal(size=(U, SV)) * np.sqrt(np.repeat(N, D, axis=1) + 0.05)
    Tm = rng.normal(0.0, 0.02, (R, SV))
    invvar = rng.uniform(0.5, 2.0, SV)
    return dict(U=U, N=N, F=F, Tm=Tm, invvar=invvar)


@pytest.fixture(scope="module")
def tett(ctx, prob):
    """TETt of the full model on the GPU (packed) and from the oracle (full R x R blocks, ~20 s scalar)."""
    te_g = ctx.tv_tett(prob["Tm"], prob["invvar"], C, D)
    te_o = orc.tv_tett(prob["Tm"], prob["invvar"], C, D)
    return te_g, te_o


def test_tett_full_shape(tett):
    """estimateTETt (AccumulateTVStat.cpp:777-805) at C = 2048, R = 400: every Gaussian, packed lower triangle."""
    te_g, te_o = tett
    il = np.tril_indices(R)
    assert te_g.shape == (C, R * (R + 1) // 2)
    for c0 in range(0, C, 256):                           # blockwise to keep the temporary small
        assert relerr(te_g[c0:c0 + 256], te_o[c0:c0 + 256][:, il[0], il[1]]) < 1e-12


def test_ivectors_full_shape(ctx, prob, tett):
    """estimateW (AccumulateTVStat.cpp:2114-2169) on a 300-utterance batch: the L GEMM is 300 x 80200 x 2048 (two interior
    row tiles + one cut to 44 rows), aux is split-K over 122 880; six utterances against the oracle."""
    te_g, te_o = tett
    W_g = ctx.tv_estimate_w(prob["N"], prob["F"], prob["Tm"], prob["invvar"], te_g, C, D)
    rows = [0, 1, 127, 128, 255, 299]
    W_o = orc.tv_estimate_w(prob["N"][rows], prob["F"][rows], prob["Tm"], prob["invvar"], te_o)
    assert np.all(np.isfinite(W_g))
    assert relerr(W_g[rows], W_o) < 1e-9                  # north_star bar is 1e-6
    # a different batching (tv_batch = 128: three GEMMs with other tile cuts) gives the same vectors
    prev = ctx.set_option("tv_batch", 128)
    try:
        W_b = ctx.tv_estimate_w(prob["N"], prob["F"], prob["Tm"], prob["invvar"], te_g, C, D)
    finally:
        ctx.set_option("tv_batch", prev)
    assert relerr(W_b, W_g) < 1e-11


def test_tv_em_full_shape(ctx, prob, tett):
    """estimateAandC + updateTestimate + minDivergence (AccumulateTVStat.cpp:1702-1795, 974-1005, 2056-2099) at full shape."""
    te_g, te_o = tett
    N, F, Tm, invvar = prob["N"], prob["F"], prob["Tm"], prob["invvar"]
    il = np.tril_indices(R)
    # (1) U = 4 against the oracle: every accumulator, every Gaussian
    o = orc.tv_estimate_a_and_c(N[:4], F[:4], Tm, invvar, te_o)
    g4 = ctx.tv_estimate_a_and_c(N[:4], F[:4], Tm, invvar, te_g, C, D)
    assert relerr(g4["W"], o["W"]) < 1e-9
    Ao = o["A"].reshape(C, R, R)
    for c0 in range(0, C, 256):
        assert relerr(g4["A"][c0:c0 + 256], Ao[c0:c0 + 256][:, il[0], il[1]]) < 1e-9
    assert relerr(g4["Cmx"], o["Cmx"]) < 1e-9 and relerr(g4["Rm"], o["Rm"]) < 1e-9
    assert relerr(g4["r"], o["r"]) < 1e-9 and relerr(g4["meanW"] / 4, o["meanW"]) < 1e-9
    del o, Ao
    # (2) linearity: one 300-utterance batch (A += N^T E is 2048 x 80200 x 300, Cmx += W^T F is 400 x 122880 x 300:
    #     interior tiles, clamped edge tiles, a K that is not a multiple of 16) == 75 batches of the oracle-checked shape
    gall = ctx.tv_estimate_a_and_c(N, F, Tm, invvar, te_g, C, D)
    prev = ctx.set_option("tv_batch", 4)
    try:
        gsm = ctx.tv_estimate_a_and_c(N, F, Tm, invvar, te_g, C, D)
    finally:
        ctx.set_option("tv_batch", prev)
    for k in ("W", "A", "Cmx", "Rm", "r", "meanW"):
        assert relerr(gall[k], gsm[k]) < 1e-11, k
    # (3) M-step on all 2048 Gaussians, six of them against the oracle (fed with the GPU's accumulators)
    T_g = ctx.tv_update_t(gall["A"], gall["Cmx"], C, D)
    assert np.all(np.isfinite(T_g))
    A_sel = np.zeros((len(SEL), R, R))
    for k, c in enumerate(SEL):
        A_sel[k][il] = gall["A"][c]
        A_sel[k] = A_sel[k] + np.tril(A_sel[k], -1).T
    C_sel = np.concatenate([gall["Cmx"][:, c * D:(c + 1) * D] for c in SEL], axis=1)
    T_o = orc.tv_update_t(A_sel.reshape(len(SEL), R * R), C_sel, len(SEL), D)
    T_gs = np.concatenate([T_g[:, c * D:(c + 1) * D] for c in SEL], axis=1)
    assert relerr(T_gs, T_o) < 1e-6
    # (4) minimum divergence: T <- chol(R/n - r r^T) T is a 400 x 122880 x 400 GEMM; mean += T^T meanW
    U = prob["U"]
    means = np.random.default_rng(5).normal(size=SV)
    m_o, Tm_o = orc.tv_min_divergence(gall["Rm"], gall["r"], gall["meanW"] / U, means, T_g, U, C, D)
    m_g, Tm_g = ctx.tv_min_divergence(gall["Rm"].copy(), gall["r"].copy(), gall["meanW"] / U, means.copy(), T_g.copy(), U, C, D)
    assert relerr(m_g, m_o) < 1e-9 and relerr(Tm_g, Tm_o) < 1e-9


@pytest.mark.parametrize("M,S", [(4100, 4233), (4097, 8191)])
def test_scoring_full_dim(ctx, M, S):
    """cosine / Mahalanobis / two-covariance / PLDA at dim 400 (rankF 200) on an M x S grid that is not a multiple of the GEMM
    tile; sampled rows x columns (first, interior-tile, last-tile and last entries) against the oracle."""
    rng = np.random.default_rng(M)
    dim, rf = 400, 200
    models = rng.normal(size=(dim, M)); segs = rng.normal(size=(dim, S))
    models /= np.linalg.norm(models, axis=0); segs /= np.linalg.norm(segs, axis=0)
    rows = np.unique(np.concatenate([[0, 1, 127, 128, 129, M - 129, M - 2, M - 1], rng.integers(0, M, 12)]))
    cols = np.unique(np.concatenate([[0, 63, 127, 128, S - 130, S - 2, S - 1], rng.integers(0, S, 12)]))
    sub = np.ix_(rows, cols)
    ms, ss = np.ascontiguousarray(models[:, rows]), np.ascontiguousarray(segs[:, cols])
    got = ctx.score_cosine(models, segs)
    assert got.shape == (M, S) and relerr(got[sub], orc.score_cosine(ms, ss)) < 1e-12
    Q = rng.normal(size=(dim, dim)); Mah = Q @ Q.T / dim + np.eye(dim)
    got = ctx.score_mahalanobis(models, segs, Mah)
    assert relerr(got[sub], orc.score_mahalanobis(ms, ss, Mah)) < 1e-10
    G = rng.normal(size=(dim, dim)) / dim; G = G + G.T; H = rng.normal(size=(dim, dim)) / dim; H = H + H.T
    got = ctx.score_twocov(models, segs, G, H)
    assert relerr(got[sub], orc.score_twocov(ms, ss, G, H)) < 1e-10
    # PLDA on FTJ-projected vectors: rankF 200, models = sums over 1..3 enrolment sessions in runs like PldaTest keeps them
    Fm = rng.normal(size=(dim, rf)) / np.sqrt(dim)
    FTJF = Fm.T @ Fm + 0.1 * np.eye(rf)
    nsess = np.sort(rng.integers(1, 4, M))
    pm = rng.normal(size=(rf, M)) * nsess; ps = rng.normal(size=(rf, S))
    got = ctx.score_plda(pm, nsess, ps, FTJF)
    ref = orc.score_plda(np.ascontiguousarray(pm[:, rows]), nsess[rows], np.ascontiguousarray(ps[:, cols]), FTJF)
    assert relerr(got[sub], ref) < 1e-10
