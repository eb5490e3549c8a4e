"""The i-vector / scoring rows have no reference fixture ("parity unpinned"); the
C oracle for them is cross-checked here against an independent numpy restatement of the same
formulas, so that a slip in the oracle cannot silently become the GPU's target."""
import numpy as np
import pytest

from conftest import make_frames, make_gmm
from oracle import oracle as orc


def test_posteriors_and_stats_against_numpy():
    w, mean, iv = make_gmm(24, 10, seed=3)
    x = make_frames(w, mean, iv, 200, seed=4).astype(np.float64)
    D = 10
    logn = (np.log(w) - 0.5 * D * np.log(2 * np.pi) + 0.5 * np.log(iv).sum(1)
            - 0.5 * (((x[:, None, :] - mean[None]) ** 2) * iv[None]).sum(-1))
    lse = np.logaddexp.reduce(logn, axis=1)
    gam = np.exp(logn - lse[:, None])
    g = orc.Gmm(w, mean, iv)
    assert np.allclose(orc.llk(g, x, -1e9, 1e9), lse, rtol=0, atol=1e-10)
    assert np.allclose(orc.occ(g, x), gam, atol=1e-12)
    a = orc.em_accumulate(g, x)
    assert np.allclose(a["occ"], gam.sum(0)) and np.allclose(a["sx"], gam.T @ x) and np.allclose(a["sxx"], gam.T @ x ** 2)
    utt = np.repeat([0, 1, 2], [50, 70, 80])
    N, F = orc.tv_stats(g, x, utt, 3)
    for u in range(3):
        assert np.allclose(N[u], gam[utt == u].sum(0))
        assert np.allclose(F[u].reshape(24, 10), gam[utt == u].T @ x[utt == u])


def test_tv_maths_against_numpy():
    rng = np.random.default_rng(0)
    C, D, R, U = 6, 5, 7, 11
    N = rng.uniform(0.5, 20, (U, C)); F = rng.normal(size=(U, C * D)); T = rng.normal(size=(R, C * D)) * 0.3
    iv = rng.uniform(0.5, 2, C * D); means = rng.normal(size=C * D)
    F0 = orc.tv_subtract_m(N, F, means)
    assert np.allclose(F0, F - np.repeat(N, D, axis=1) * means)
    te = orc.tv_tett(T, iv, C, D)
    for c in range(C):
        Tc = T[:, c * D:(c + 1) * D]
        assert np.allclose(te[c], (Tc * iv[c * D:(c + 1) * D]) @ Tc.T)
    W = orc.tv_estimate_w(N, F0, T, iv, te)
    o = orc.tv_estimate_a_and_c(N, F0, T, iv, te)
    A = np.zeros((C, R, R)); Cm = np.zeros((R, C * D)); Rm = np.zeros((R, R))
    for u in range(U):
        L = np.eye(R) + np.einsum("c,cij->ij", N[u], te)
        wv = np.linalg.solve(L, T @ (iv * F0[u]))
        assert np.allclose(W[u], wv, rtol=1e-10) and np.allclose(o["W"][u], wv, rtol=1e-10)
        E = np.linalg.inv(L) + np.outer(wv, wv)
        A += N[u][:, None, None] * E; Cm += np.outer(wv, F0[u]); Rm += E
    assert np.allclose(o["A"].reshape(C, R, R), A) and np.allclose(o["Cmx"], Cm) and np.allclose(o["Rm"], Rm)
    assert np.allclose(o["r"], W.sum(0)) and np.allclose(o["meanW"], W.mean(0))
    Tn = orc.tv_update_t(o["A"], o["Cmx"], C, D)
    for c in range(C):
        assert np.allclose(Tn[:, c * D:(c + 1) * D], np.linalg.solve(A[c], Cm[:, c * D:(c + 1) * D]))
    m2, T2 = orc.tv_min_divergence(o["Rm"], o["r"], o["meanW"], means, Tn, U, C, D)
    rbar = o["r"] / U
    Ch = np.linalg.cholesky(o["Rm"] / U - np.outer(rbar, rbar)).T          # upper, R = Ch^T Ch
    assert np.allclose(T2, Ch @ Tn) and np.allclose(m2, means + Tn.T @ o["meanW"])
    Q = orc.tv_orthonormalize_t(T)
    assert np.allclose(Q @ Q.T, np.eye(R), atol=1e-10)


def test_scoring_against_numpy():
    rng = np.random.default_rng(1)
    d, M, S = 9, 6, 8
    m = rng.normal(size=(d, M)); s = rng.normal(size=(d, S))
    cos = (m.T @ s) / np.outer(np.linalg.norm(m, axis=0), np.linalg.norm(s, axis=0))
    assert np.allclose(orc.score_cosine(m, s), cos)
    tr = rng.random((M, S)) > 0.5
    assert np.allclose(orc.score_cosine(m, s, tr), np.where(tr, cos, 0))
    mu = m.mean(1); Mr = rng.normal(size=(5, d))
    y = Mr @ (m - mu[:, None]); y /= np.linalg.norm(y, axis=0)
    assert np.allclose(orc.iv_normalize(m, mu, Mr, True), y)
    Q = rng.normal(size=(d, d)); Mah = Q @ Q.T + np.eye(d)
    diff = m[:, :, None] - s[:, None, :]
    assert np.allclose(orc.score_mahalanobis(m, s, Mah), -0.5 * np.einsum("ims,ij,jms->ms", diff, Mah, diff))
    Wm = Q @ Q.T / d + np.eye(d); Bm = 2 * np.eye(d) + 0.1 * (Q + Q.T) @ (Q + Q.T).T / d
    G, H = orc.twocov_model(Wm, Bm)
    iW, iB = np.linalg.inv(Wm), np.linalg.inv(Bm)
    assert np.allclose(G, iW @ np.linalg.inv(iB + 2 * iW) @ iW) and np.allclose(H, iW @ np.linalg.inv(iB + iW) @ iW)
    sm = m[:, :, None] + s[:, None, :]
    ref = np.einsum("ims,ij,jms->ms", sm, G, sm) - np.einsum("im,ij,jm->m", m, H, m)[:, None] - np.einsum("is,ij,js->s", s, H, s)[None]
    assert np.allclose(orc.score_twocov(m, s, G, H), ref)
    Fm = rng.normal(size=(20, d)); FTJF = Fm.T @ Fm / 20
    nsess = np.array([1, 1, 2, 2, 2, 1])
    K = lambda n: np.linalg.inv(n * FTJF + np.eye(d))
    al = lambda n: np.linalg.slogdet(K(n))[1]
    ref = np.empty((M, S))
    for i in range(M):
        L = nsess[i]
        for j in range(S):
            v = s[:, j] + m[:, i]
            ref[i, j] = 0.5 * (v @ K(L + 1) @ v - m[:, i] @ K(L) @ m[:, i] - s[:, j] @ K(1) @ s[:, j]) + 0.5 * (al(L + 1) - al(L) - al(1))
    assert np.allclose(orc.score_plda(m, nsess, s, FTJF), ref)


def test_approximate_extractors_and_plda_precompute_against_numpy():
    """Independent numpy restatement of the approximate i-vector extractors (AccumulateTVStat.cpp:1225-1242,
    1379-1399, 1600-1609, 2348-2396, 2566-2609, 2837-2855, 3116-3136) and of PldaModel::preComputation
    (PldaTools.cpp:2950-2972, 4494-4496); no reference fixture exists for them."""
    rng = np.random.default_rng(5)
    U, C, D, R = 5, 6, 4, 7
    SV = C * D
    N = rng.uniform(0.1, 30.0, (U, C)); F = rng.normal(size=(U, SV)); means = rng.normal(size=SV)
    iv = rng.uniform(0.5, 2.0, SV); T = 0.3 * rng.normal(size=(R, SV)); W = rng.normal(size=(U, R)); wgt = rng.dirichlet(np.ones(C))
    Nrep = np.repeat(N, D, axis=1)
    assert np.allclose(orc.tv_norm_statistics(N, F, means, iv), (F - means * Nrep) * np.sqrt(iv), rtol=1e-13, atol=1e-13)
    assert np.allclose(orc.tv_subtract_m_plus_tw(N, F, means, T, W), F - (means + W @ T) * Nrep, rtol=1e-12, atol=1e-12)
    Tn = orc.tv_norm_t(T, iv, C)
    assert np.allclose(Tn, T * np.sqrt(iv), rtol=1e-14)
    Wm = orc.tv_weighted_cov(Tn, wgt)
    assert np.allclose(Wm, (Tn * np.repeat(wgt, D)) @ Tn.T, rtol=1e-12, atol=1e-14)
    Q = np.linalg.qr(rng.normal(size=(R, R)))[0]
    Dm = orc.tv_approximate_tctc(Tn, Q, C)
    ref = np.stack([((Tn[:, c * D:(c + 1) * D].T @ Q) ** 2).sum(0) for c in range(C)])
    assert np.allclose(Dm, ref, rtol=1e-12)
    Fn = orc.tv_norm_statistics(N, F, means, iv)
    w1 = orc.tv_estimate_w_ubm_weight(N, Fn, Tn, Wm)
    ref1 = np.stack([np.linalg.solve(np.eye(R) + N[u].sum() * Wm, Tn @ Fn[u]) for u in range(U)])
    assert np.allclose(w1, ref1, rtol=1e-10, atol=1e-12)
    w2 = orc.tv_estimate_w_eigen(N, Fn, Tn, Dm, Q)
    ref2 = np.stack([Q @ ((Q.T @ (Tn @ Fn[u])) / (1.0 + N[u] @ Dm)) for u in range(U)])
    assert np.allclose(w2, ref2, rtol=1e-10, atol=1e-12)
    dim, rf, rg = 9, 4, 3
    Fm = rng.normal(size=(dim, rf)); G = rng.normal(size=(dim, rg)); A = rng.normal(size=(dim, dim)); S = A @ A.T + dim * np.eye(dim)
    FTJ, FTJF = orc.plda_precompute(Fm, G, S)
    Si = np.linalg.inv(S)
    ref = Fm.T @ Si - Fm.T @ Si @ G @ np.linalg.inv(G.T @ Si @ G + np.eye(rg)) @ G.T @ Si
    assert np.allclose(FTJ, ref, rtol=1e-10, atol=1e-12) and np.allclose(FTJF, ref @ Fm, rtol=1e-10, atol=1e-12)
    FTJ0, _ = orc.plda_precompute(Fm, None, S)
    assert np.allclose(FTJ0, Fm.T @ Si, rtol=1e-10, atol=1e-12)


def test_backend_estimation_against_numpy():
    """PldaDev (PldaTools.cpp:353-387, 527-566, 1124-1176, 1381-1413, 1610-1644, 1852-1902) restated with numpy."""
    rng = np.random.default_rng(11)
    dim, sps = 7, np.array([3, 1, 5, 2, 4])
    n, k = int(sps.sum()), len(sps)
    spk = rng.normal(size=(dim, k)) * 2
    cls = np.repeat(np.arange(k), sps)
    X = spk[:, cls] + rng.normal(size=(dim, n))
    mean, sm = orc.dev_means(X, sps)
    assert np.allclose(mean, X.mean(1)) and np.allclose(sm, np.stack([X[:, cls == c].mean(1) for c in range(k)], 1))
    S, W, B = orc.dev_cov_mat(X, sps)
    Xc = X - mean[:, None]; Xw = X - sm[:, cls]; Mb = (sm - mean[:, None]) * np.sqrt(sps)
    assert np.allclose(S, Xc @ Xc.T / n) and np.allclose(W, Xw @ Xw.T / n) and np.allclose(B, Mb @ Mb.T / n)
    assert np.allclose(S, W + B)
    Wc = sum((Xw[:, cls == c] @ Xw[:, cls == c].T) / sps[c] for c in range(k)) / k
    U = orc.dev_wccn_chol(X, sps)
    assert np.allclose(U, np.triu(U)) and np.allclose(U.T @ U, np.linalg.inv(Wc), rtol=1e-9)
    SB, SW = orc.dev_scatter_mat(X, sps)
    assert np.allclose(SB, (sm - mean[:, None]) @ (sm - mean[:, None]).T)
    assert np.allclose(SW, Xw[:, :sps[-1]] @ Xw[:, :sps[-1]].T / sps[-1])          # the reference's quirk, kept
    vect, val = orc.sym_eigen(S)
    assert np.all(np.diff(val) <= 0) and np.allclose(vect @ np.diag(val) @ vect.T, S) and np.allclose(vect.T @ vect, np.eye(dim))
    M = orc.dev_efr_matrix(S)
    assert np.allclose(M @ S @ M.T, np.eye(dim), atol=1e-10)                         # whitening
    L, lam = orc.dev_lda(W, B, 3)
    EP = np.linalg.inv(W) @ B
    for j in range(3):
        assert np.allclose(EP @ L[j], lam[j] * L[j], atol=1e-9 * abs(lam[0])) and abs(np.linalg.norm(L[j]) - 1) < 1e-12
    ev = np.sort(np.linalg.eigvals(EP).real)[::-1]
    assert np.allclose(lam, ev[:3], rtol=1e-9, atol=1e-12)


def _plda_em_numpy(X, sps, F, G, Sigma, Delta):
    """PldaModel::em_iteration (PldaTools.cpp:2329-2343, 2359-2484, 2790-2815) with numpy, speaker by speaker."""
    dim, n = X.shape; rf, rg = F.shape[1], G.shape[1]
    X = X - Delta[:, None]
    sigObs = X @ X.T
    Si = np.linalg.inv(Sigma); Ftw = F.T @ Si; Gtw = G.T @ Si
    iGG = np.linalg.inv(Gtw @ G + np.eye(rg)); FtwG = Ftw @ G
    S = iGG @ FtwG.T; A = Ftw @ F - FtwG @ iGG @ FtwG.T
    Ehh = np.zeros((rf + rg, rf + rg)); xh = np.zeros((dim, rf + rg)); U = np.zeros(rf + rg)
    s0 = 0
    for ns in sps:
        Xs = X[:, s0:s0 + ns]; s0 += ns
        M = np.linalg.inv(ns * A + np.eye(rf)); MsT = M @ S.T
        fi = Ftw @ Xs; gi = Gtw @ Xs
        h = M @ (fi.sum(1) - S.T @ gi.sum(1))
        w = iGG @ gi - (S @ h)[:, None]
        Eh = np.vstack([np.repeat(h[:, None], ns, 1), w])
        tmpM = np.block([[M, -MsT], [-MsT.T, iGG + S @ MsT]])
        Ehh += ns * tmpM + Eh @ Eh.T; xh += Xs @ Eh.T; U += Eh.sum(1)
    FG = xh @ np.linalg.inv(Ehh)
    Sig = (sigObs - FG @ xh.T) / n
    U /= n
    c = Ehh / n - np.outer(U, U)
    Rh = np.linalg.cholesky(c[:rf, :rf]).T; Rw = np.linalg.cholesky(c[rf:, rf:]).T
    return X, FG[:, :rf] @ Rh.T, FG[:, rf:] @ Rw.T, Sig, Delta + FG @ U


def test_plda_em_iteration_against_numpy():
    rng = np.random.default_rng(3)
    dim, rf, rg = 8, 3, 2
    sps = np.array([2, 2, 3, 3, 3, 1, 4, 2])
    k, n = len(sps), int(sps.sum())
    cls = np.repeat(np.arange(k), sps)
    X = (rng.normal(size=(dim, k)))[:, cls] + 0.5 * rng.normal(size=(dim, n))
    F = rng.normal(size=(dim, rf)); G = 0.5 * rng.normal(size=(dim, rg)); Sigma = np.cov(X) + 0.1 * np.eye(dim); Delta = np.zeros(dim)
    state = (X, F, G, Sigma, Delta)
    ref = state
    for it in range(3):
        state = orc.plda_em_iteration(state[0], sps, *state[1:])
        ref = _plda_em_numpy(ref[0], sps, *ref[1:])
        for a, b in zip(state, ref):
            assert np.allclose(a, b, rtol=1e-8, atol=1e-10)


def test_jfa_pieces_against_numpy():
    """JFA restatements (AccumulateJFAStat.cpp) against plain numpy; estimateYandV is the TV E-step under another name."""
    rng = np.random.default_rng(11)
    C, D, R, nspk = 5, 4, 3, 6
    SV = C * D
    sb = np.array([0, 2, 3, 6, 7, 9, 12]); nsess = sb[-1]
    owner = np.repeat(np.arange(nspk), np.diff(sb))
    N = rng.uniform(0.5, 20, (nspk, C)); F = rng.normal(size=(nspk, SV))
    Nh = rng.uniform(0.2, 8, (nsess, C)); Fh = rng.normal(size=(nsess, SV))
    m = rng.normal(size=SV); V = rng.normal(size=(R, SV)) * 0.3; U = rng.normal(size=(R, SV)) * 0.2
    Y = rng.normal(size=(nspk, R)); X = rng.normal(size=(nsess, R)); Dm = rng.uniform(0.1, 1, SV); Z = rng.normal(size=(nspk, SV))
    iv = rng.uniform(0.5, 2, SV)
    rep = lambda n: np.repeat(n, D, axis=1)
    assert np.abs(orc.jfa_subtract(N, F, None, m, None, None, Dm, Z) - (F - rep(N) * (m + Dm * Z))).max() < 1e-12        # M + DZ
    assert np.abs(orc.jfa_subtract(N, F, None, m, V, Y) - (F - rep(N) * (m + Y @ V))).max() < 1e-12                    # M + VY
    ref = Fh - rep(Nh) * (m + Y[owner] @ V + Dm * Z[owner])
    assert np.abs(orc.jfa_subtract(Nh, Fh, owner, m, V, Y, Dm, Z) - ref).max() < 1e-12                                 # M + VY + DZ per session
    ref = F.copy()
    for s in range(nspk):
        for h in range(sb[s], sb[s + 1]):
            ref[s] -= np.repeat(Nh[h], D) * (X[h] @ U)
    assert np.abs(orc.jfa_subtract_sessions(sb, Nh, F, U, X) - ref).max() < 1e-12                                      # UX
    ref2 = F.copy()
    for s in range(nspk):
        for h in range(sb[s], sb[s + 1]):
            ref2[s] -= np.repeat(Nh[h], D) * (m + X[h] @ U)
    assert np.abs(orc.jfa_subtract_m_plus_ux(sb, Nh, F, m, U, X) - ref2).max() < 1e-12                                  # M + UX, speaker rows
    L = 1 + rep(N) * iv * Dm * Dm; z = F * iv * Dm / L
    Zo, Dn = orc.jfa_estimate_z_and_d(N, F, iv, Dm)
    assert np.abs(Zo - z).max() < 1e-13 and np.abs(Dn - (z * F).sum(0) / ((1 / L + z * z) * rep(N)).sum(0)).max() < 1e-12
    assert np.abs(orc.jfa_estimate_z(N, F, iv, Dm) - z).max() < 1e-13
    assert np.abs(orc.jfa_estimate_z(N, F, iv, Dm, 3.0) - (3.0 / (3.0 + rep(N))) * Dm * iv * F).max() < 1e-13
    # estimateAndInverseL_EV + estimateYandV == TVAcc::estimateAandC
    te = orc.tv_tett(V, iv, C, D)
    Yj, Aj, Cj = orc.jfa_estimate_y_and_v(N, F, V, iv, te)
    o = orc.tv_estimate_a_and_c(N, F, V, iv, te)
    assert np.abs(Yj - o["W"]).max() < 1e-10 * np.abs(Yj).max()
    assert np.abs(Aj.reshape(C, R * R) - o["A"]).max() < 1e-10 * np.abs(Aj).max()
    assert np.abs(Cj - o["Cmx"]).max() < 1e-10 * np.abs(Cj).max()


def test_init_t_box_muller_chain_matches_numpy_restatement():
    """orc_tv_init_t against an independent restatement of ScoreWarp.cpp:68-81 driven by the same glibc rand() stream."""
    import ctypes as ct
    libc = ct.CDLL("libc.so.6")
    R, SV, seed = 3, 40, 5
    iv = np.linspace(0.5, 2.0, SV)
    got = orc.tv_init_t(R, iv, seed)
    libc.srand(seed)
    RAND_MAX = 2147483647
    u = lambda: float(np.float32(libc.rand()) / np.float32(RAND_MAX))
    x1 = u()
    ref = np.empty(R * SV)
    for e in range(R * SV):
        while True:
            x2, x1 = x1, u()
            with np.errstate(divide="ignore", invalid="ignore"):
                y = np.sqrt(-2.0 * np.log(x1)) * np.cos(2 * np.pi * x2)
            if np.isfinite(y):
                break
        ref[e] = y * iv.sum() * 0.001
    assert np.allclose(got.ravel(), ref, rtol=1e-13, atol=0)


def test_mixture_init_against_a_python_walk_with_glibc_rand():
    """orc.mixture_init (TrainTools.cpp:674-766 + GeneralTools.cpp:330-390) against a straight Python walk that calls glibc's
    srand / rand through ctypes: chunk-major, component-minor draws; seed (stream + 1) * 100 + it + 1; p folded into several
    passes when it exceeds one."""
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.rand.restype = ctypes.c_int
    RAND_MAX = 2147483647
    rng = np.random.default_rng(1)
    for T, segs, select in ((900, [(0, 400), (450, 450)], 30.0), (120, [(5, 100)], 700.0)):   # p = 7: three passes of 6 / 7
        C, D = 5, 3
        x = rng.normal(size=(T, D))
        sb = [b for b, _ in segs]; sl = [n for _, n in segs]
        mean, cnt = orc.mixture_init(C, x, sb, sl, nb_frame_to_select=select)
        total = sum(sl)
        proba = select / total
        nb_it, tmp = 1, proba
        while tmp > 1:
            nb_it += 1
            tmp /= proba / nb_it
        proba = tmp
        s = np.zeros((C, D)); n = np.zeros(C)
        for it in range(nb_it):
            libc.srand(100 + it + 1)
            for b, ln in segs:
                while ln > 0:
                    length = min(ln, min(max(ln, 3), 7))
                    for c in range(C):
                        if libc.rand() / RAND_MAX < proba:
                            s[c] += x[b:b + length].sum(0); n[c] += length
                    b += length; ln -= length
        assert np.array_equal(n, cnt) and n.min() > 0
        assert np.allclose(mean, s / n[:, None], rtol=1e-13, atol=1e-13)
    with pytest.raises(ValueError, match="does not terminate"):     # 1 < p < 4.9: the reference's fold loop never ends
        orc.mixture_init(5, rng.normal(size=(120, 3)), [5], [100], nb_frame_to_select=160.0)


def test_threaded_tv_iteration_equals_the_unthreaded_steps():
    """oracle_mt.c's T-matrix EM iteration (the reference's thread partition: AccumulateTVStat.cpp:826-950, :1831-2052; bench.py's
    cpu_baseline of the T-matrix workload, -O3 -ffast-math) against the strict single-thread oracle steps on the same statistics."""
    rng = np.random.default_rng(3)
    U, C, D, R = 23, 6, 5, 7
    N = rng.uniform(0.5, 40.0, (U, C))
    means = rng.normal(size=C * D)
    invvar = rng.uniform(0.5, 2.0, C * D)
    F = rng.normal(size=(U, C * D)) * np.repeat(N, D, axis=1)
    Tm = 0.1 * rng.normal(size=(R, C * D))
    te = orc.tv_tett(Tm, invvar, C, D)
    e = orc.tv_estimate_a_and_c(N, F, Tm, invvar, te)
    Tn = orc.tv_update_t(e["A"], e["Cmx"], C, D)
    m_ref, T_ref = orc.tv_min_divergence(e["Rm"], e["r"], e["meanW"], means, Tn, U, C, D)
    for threads, upd in ((1, 1), (4, 1), (5, 3), (64, 8)):          # 64 > U: clamped to U like :1949
        got = orc.tv_em_iteration_mt(N, F, Tm, invvar, means, threads=threads, upd_threads=upd)
        assert np.max(np.abs(got["W"] - e["W"])) < 1e-10 * np.max(np.abs(e["W"]))
        assert np.max(np.abs(got["T"] - T_ref)) < 1e-9 * np.max(np.abs(T_ref))
        assert np.max(np.abs(got["means"] - m_ref)) < 1e-10 * np.max(np.abs(m_ref))
        assert got["phase_s"].shape == (4,) and np.all(got["phase_s"] >= 0)
