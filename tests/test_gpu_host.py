"""GPU parity of the C++ host layer (lia_ral_amd/host: trainModelStream, ComputeTest LLR loop,
IvExtractor, TotalVariability mirrors) against the same pipelines composed from oracle primitives."""
import os

import numpy as np
import pytest

from conftest import make_frames, make_gmm
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def oracle_train_world(x, seg_begin, seg_len, w, mean, cov, nb_it, p, fl0, fl1, ce0, ce1, init_rand=0):
    """TrainWorld.cpp:101-191 / TrainTools.cpp:1030-1110 from oracle pieces (single stream)."""
    x = x.astype(np.float64)
    sel = np.concatenate([np.arange(b, b + n) for b, n in zip(seg_begin, seg_len)])
    s, ss, n = orc.frame_acc(x[sel])
    gmean, gcov = orc.frame_mean_cov(s, ss, n)
    llks = []
    for it in range(nb_it):
        floor = orc.set_it_parameter(fl0, fl1, nb_it, it)
        ceil = orc.set_it_parameter(ce0, ce1, nb_it, it)
        seed = ((it + 1 + init_rand) * 200) + ((0 + 1) * 20) + (0 + 1)
        bb, bl, _ = orc.bagged_segments(seed, seg_begin, seg_len, p, 3, 7)
        fr = np.concatenate([np.arange(b, b + n) for b, n in zip(bb, bl)]) if len(bb) else np.zeros(0, int)
        acc = orc.em_accumulate(orc.Gmm(w, mean, 1.0 / cov), x[fr])
        llks.append(acc["llk"] / acc["count"])
        w, mean, cov = orc.em_get(acc, mean, cov)
        cov, _, _ = orc.variance_control(cov, floor, ceil, gcov)
    return dict(w=w, mean=mean, cov=cov, llk=np.array(llks), global_mean=gmean, global_cov=gcov)


@pytest.mark.parametrize("p", [1.0, 0.4])
def test_train_model_stream(p):
    from lia_ral_amd import host_capi as h
    C, D, T = 32, 20, 6000
    w, mean, iv = make_gmm(C, D, seed=5)
    x = make_frames(w, mean, iv, T, seed=6)
    rng = np.random.default_rng(0)
    w0 = np.full(C, 1.0 / C); mean0 = mean + rng.normal(0, 0.5, mean.shape); cov0 = np.ones((C, D)) * 2.0
    seg_begin = [0, 2500, 4000]; seg_len = [2000, 1400, 2000]          # label segments (gaps are not selected)
    args = dict(nb_it=3, bagged_p=p, init_floor=0.5, final_floor=0.05, init_ceil=5.0, final_ceil=10.0)
    got = h.train_world(x, seg_begin, seg_len, w0, mean0, cov0, **args)
    ref = oracle_train_world(x, np.array(seg_begin), np.array(seg_len), w0, mean0, cov0, 3, p, 0.5, 0.05, 5.0, 10.0)
    assert relerr(got["global_cov"], ref["global_cov"]) < 1e-12
    assert np.max(np.abs(got["llk"] - ref["llk"])) < 1e-9
    assert relerr(got["w"], ref["w"]) < 1e-8 and relerr(got["mean"], ref["mean"]) < 1e-8 and relerr(got["cov"], ref["cov"]) < 1e-8


def test_compute_test_kat1_through_host_layer(golden_dir):
    """LIA_SpkDet/ComputeTest/test/test1.validate.res via computeTestLLR (segmental mode)."""
    from lia_ral_amd import host_capi as h
    k = np.load(os.path.join(golden_dir, "kat1_computetest.npz"))
    world = (k["w"], k["mean_world"], 1.0 / k["covinv"])
    cl1 = (k["w_client"], k["mean_client"], 1.0 / k["covinv_client"])
    llr = h.compute_test(k["x"], k["seg_begin"], k["seg_len"], world, [cl1, world], top_c=int(k["top_c"]), complete=True,
                         segmental=True)
    assert np.allclose(llr[:, 0], k["expected_llr"], atol=float(k["abs_tol"]), rtol=0), llr
    assert np.all(np.abs(llr[:, 1]) < 1e-12)            # client == world -> 0 (validate.res lines 2,4)
    # file mode: one LLR over all selected frames == frame-weighted mean of the segment LLRs
    one = h.compute_test(k["x"], k["seg_begin"], k["seg_len"], world, [cl1], top_c=int(k["top_c"]), complete=True)
    wts = k["seg_len"] / k["seg_len"].sum()
    assert abs(one[0, 0] - (llr[:, 0] * wts).sum()) < 1e-9


def test_train_target_kat2_through_host_layer(golden_dir):
    """LIA_SpkDet/TrainTarget/test/test1.validate.gmm: adaptModel + computeMAPOccDep on the HIP path."""
    from lia_ral_amd import host_capi as h
    k = np.load(os.path.join(golden_dir, "kat2_traintarget.npz"))
    world = (k["w"], k["mean_world"], 1.0 / k["covinv"])
    w, mean, cov = h.train_target(k["x"], k["seg_begin"], k["seg_len"], world, nb_it=1, mean_reg=float(k["reg_factor"]))
    diff = np.abs(mean - k["mean_expected"])
    assert np.median(diff) < float(k["median_tol"]) and diff.max() < float(k["max_tol"])
    assert np.array_equal(w, k["w"]) and np.allclose(cov, 1.0 / k["covinv"], rtol=1e-15)   # mean-only adaptation


def test_iv_extractor_and_tv_training():
    from lia_ral_amd import host_capi as h
    C, D, R, U = 32, 20, 24, 40
    rng = np.random.default_rng(3)
    w, mean, iv = make_gmm(C, D, seed=8)
    lens = rng.integers(80, 160, U)
    ub = np.concatenate([[0], np.cumsum(lens)])
    x = make_frames(w, mean, iv, int(ub[-1]), seed=9)
    Tm = rng.normal(0, 0.05, (R, C * D))
    ubm = (w, mean, 1.0 / iv)
    W, N, F = h.iv_extract(x, ub, ubm, Tm, return_stats=True)
    og = orc.Gmm(w, mean, iv)
    utt = np.repeat(np.arange(U), lens)
    No, Fo = orc.tv_stats(og, x.astype(np.float64), utt, U)
    assert relerr(N, No) < 1e-9 and relerr(F, Fo) < 1e-9
    F0 = orc.tv_subtract_m(No, Fo, mean.ravel())
    te = orc.tv_tett(Tm, iv.ravel(), C, D)
    Wo = orc.tv_estimate_w(No, F0, Tm, iv.ravel(), te)
    assert relerr(W, Wo) < 1e-8                      # i-vectors: north_star bar 1e-6
    # two T-matrix EM iterations with minimum divergence (TotalVariability.cpp:118-169)
    Tg, mg = h.tv_train(No, Fo, ubm, Tm, 2, min_div=True)
    To, mo = Tm.copy(), mean.ravel().copy()
    for _ in range(2):
        F0 = orc.tv_subtract_m(No, Fo, mo)
        te = orc.tv_tett(To, iv.ravel(), C, D)
        o = orc.tv_estimate_a_and_c(No, F0, To, iv.ravel(), te)
        To = orc.tv_update_t(o["A"], o["Cmx"], C, D)
        mo, To = orc.tv_min_divergence(o["Rm"], o["r"], o["meanW"], mo, To, U, C, D)
    assert relerr(Tg, To) < 1e-6 and relerr(mg, mo) < 1e-8


def test_host_layer_reports_errors_like_the_reference():
    from lia_ral_amd import host_capi as h
    w, mean, iv = make_gmm(8, 12, seed=1)
    x = make_frames(w, mean, iv, 100, seed=2)
    with pytest.raises(h.HostError, match="segment ends after"):
        h.train_world(x, [50], [100], w, mean, 1.0 / iv, 1)      # verifyClusterFile-style failure


def test_computetest_from_files_reproduces_validate_res(tmp_path, golden_dir):
    """End to end from files: RAW models (written by our writer from the repaired arrays), the genuine
    test1.prm / test1.lbl, mask 0-15,17-32 -> the lines of ComputeTest/test/test1.validate.res."""
    import struct
    from lia_ral_amd import host_capi as h
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_files")
    k = np.load(os.path.join(golden_dir, "kat1_computetest.npz"))

    def write_raw(path, w, mean, covinv):
        C, D = mean.shape
        with open(path, "wb") as f:
            f.write(struct.pack("<II", C, D)); f.write(w.astype("<f8").tobytes())
            for c in range(C):
                det = float(np.prod(1.0 / covinv[c])); cst = (2 * np.pi) ** (-D / 2) / np.sqrt(det)
                f.write(struct.pack("<ddB", cst, det, 0)); f.write(covinv[c].astype("<f8").tobytes()); f.write(mean[c].astype("<f8").tobytes())

    wld, t1 = str(tmp_path / "wld"), str(tmp_path / "test1")
    write_raw(wld, k["w"], k["mean_world"], k["covinv"])
    write_raw(t1, k["w_client"], k["mean_client"], k["covinv_client"])
    llr, lines = h.compute_test_files(wld, [t1, wld], ["test1", "test2"], os.path.join(ref, "test1.prm"),
                                      os.path.join(ref, "computetest_test1.lbl"), mask="0-15,17-32", label="male",
                                      top_c=10, complete=True, gender="M", test_name="test3")
    assert np.allclose(llr[:, 0], k["expected_llr"], atol=float(k["abs_tol"]), rtol=0)
    # "M test1 1 test3 0 0.26 5.06601" / "M test1 1 test3 0.3 0.41 4.26793" (test1.validate.res:1,3)
    f = lines[0].split(); g = lines[2].split()
    assert f[:6] == ["M", "test1", "1", "test3", "0", "0.26"] and abs(float(f[6]) - 5.06601) < 5e-5
    assert g[:6] == ["M", "test1", "1", "test3", "0.3", "0.41"] and abs(float(g[6]) - 4.26793) < 5e-5
    assert lines[1].split()[1] == "test2" and abs(float(lines[1].split()[6])) < 1e-12


@pytest.mark.parametrize("mode", [1, 2])
def test_iv_extractor_approximate_modes(mode):
    """IvExtractorUbmWeigth / IvExtractorEigenDecomposition (IvExtractor.cpp:150-360) through the C++ host layer
    against the oracle chain; the eigenbasis of mode 2 comes from the host Jacobi solver (any orthonormal
    eigenbasis of W gives the same D-approximation up to the ordering of its columns)."""
    from lia_ral_amd import host_capi as host
    C, D, R, U = 64, 20, 12, 6
    w, mean, iv = make_gmm(C, D, seed=3)
    lens = [150, 80, 0, 200, 33, 64]
    ub = np.concatenate([[0], np.cumsum(lens)])
    x = make_frames(w, mean, iv, int(ub[-1]), seed=4)
    rng = np.random.default_rng(9)
    T = 0.02 * rng.normal(size=(R, C * D))
    res = host.iv_extract_approx(x, ub, (w, mean, 1.0 / iv), T, mode)
    utt = np.repeat(np.arange(U), lens)
    N, F = orc.tv_stats(orc.Gmm(w, mean, iv), x.astype(np.float64), utt, U)
    Tn = orc.tv_norm_t(T, iv.ravel(), C)
    Wc = orc.tv_weighted_cov(Tn, w)
    assert np.max(np.abs(res["Wcov"] - Wc)) < 1e-12 * np.max(np.abs(Wc))
    Fn = orc.tv_norm_statistics(N, F, mean.ravel(), iv.ravel())
    if mode == 1:
        ref = orc.tv_estimate_w_ubm_weight(N, Fn, Tn, Wc)
    else:
        Q = res["Q"]
        assert np.max(np.abs(Q.T @ Q - np.eye(R))) < 1e-12                       # orthonormal
        lam = np.diag(Q.T @ Wc @ Q)
        assert np.max(np.abs(Q.T @ Wc @ Q - np.diag(lam))) < 1e-12 * lam.max()   # diagonalises W
        assert np.all(np.diff(lam) <= 1e-15)                                     # sorted descending
        Dm = orc.tv_approximate_tctc(Tn, Q, C)
        assert np.max(np.abs(res["D"] - Dm)) < 1e-11 * np.max(np.abs(Dm))
        ref = orc.tv_estimate_w_eigen(N, Fn, Tn, Dm, Q)
    assert np.max(np.abs(res["W"] - ref)) < 1e-9 * max(np.max(np.abs(ref)), 1e-30)


@pytest.mark.parametrize("sph_norm", [False, True])
def test_backend_training_chain(sph_norm):
    """PldaDev::sphericalNuisanceNormalization (PldaTools.cpp:1822-1929) + WCCN / Mahalanobis / LDA through the C++
    host layer against a numpy restatement of the same loop built from the oracle pieces."""
    from lia_ral_amd import host_capi as host
    rng = np.random.default_rng(4)
    dim, sps = 16, rng.integers(3, 8, 30)
    k, n = len(sps), int(sps.sum())
    cls = np.repeat(np.arange(k), sps)
    X = (rng.normal(size=(dim, k)) * 1.2)[:, cls] + rng.normal(size=(dim, n)) * (1 + np.arange(dim))[:, None] * 0.2
    res = host.backend_train(X, sps, nb_it=2, sph_norm=sph_norm, lda_rank=3)
    Y = X.copy()
    for it in range(2):
        S, W, B = orc.dev_cov_mat(Y, sps)
        M = orc.dev_efr_matrix(W if sph_norm else S)
        assert relerr(np.abs(res["mats"][it]), np.abs(M)) < 1e-7 and relerr(res["means"][it], Y.mean(1)) < 1e-10
        M = res["mats"][it]                    # eigenvector signs are the solver's: continue with the product's matrix
        Y = M @ (Y - Y.mean(1)[:, None])
        Y /= np.linalg.norm(Y, axis=0)
    assert relerr(res["X"], Y) < 1e-9
    S, W, B = orc.dev_cov_mat(Y, sps)
    assert relerr(res["mahalanobis"], np.linalg.inv(W)) < 1e-7
    assert relerr(res["wccn"], orc.dev_wccn_chol(Y, sps)) < 1e-7
    oL, lam = orc.dev_lda(W, B, 3)
    assert relerr(np.abs(res["lda"]), np.abs(oL)) < 1e-6


@pytest.mark.parametrize("rg", [4, 0])        # 0: pldaEigenChannelNumber 0, the simplified model
def test_plda_training_loop(rg):
    """PLDA.cpp:80-95 through liagpu::PldaModel: 4 EM iterations against the oracle loop."""
    from lia_ral_amd import host_capi as host
    rng = np.random.default_rng(8)
    dim, rf, nspk = 24, 6, 60
    sps = rng.integers(2, 6, nspk)
    cls = np.repeat(np.arange(nspk), sps); n = int(sps.sum())
    X = rng.normal(size=(dim, rf)) @ rng.normal(size=(rf, nspk))[:, cls] + 0.5 * rng.normal(size=(dim, n)) + 1.0
    F = rng.normal(size=(dim, rf)); G = 0.3 * rng.normal(size=(dim, rg)); Sigma = np.cov(X) + 0.1 * np.eye(dim)
    res = host.plda_train(X, sps, F, G, Sigma, nb_it=4)
    ref = (X, F, G, Sigma, np.zeros(dim))
    for it in range(4):
        ref = orc.plda_em_iteration(ref[0], sps, *ref[1:])
    assert relerr(res["F"], ref[1]) < 1e-7 and (rg == 0 or relerr(res["G"], ref[2]) < 1e-7) and relerr(res["Sigma"], ref[3]) < 1e-7
    assert relerr(res["Delta"], ref[4]) < 1e-7 and relerr(res["X"], ref[0]) < 1e-8
    assert relerr(res["original_mean"], X.mean(1)) < 1e-12


def _window_llr_reference(size, dec, frame_idx, llr):
    """UnsupervisedTools.cpp:119-145 in python: llr [n, nClient]; returns rows [idxBegin, idxEnd, llr...]."""
    n, nc = llr.shape
    idx = np.zeros(size, np.int64); acc = np.zeros(nc); buf = np.zeros((size, nc))
    b = cnt = 0
    rows = []
    for t in range(n):
        if cnt < size:
            cnt += 1
            idx[(b + cnt - 1) % size] = frame_idx[t]
        else:
            for _ in range(dec):
                acc -= buf[b]
                b = (b + 1) % size
            cnt -= dec - 1
            idx[(b + cnt - 1) % size] = frame_idx[t]
        buf[(b + cnt - 1) % size] = llr[t]
        acc += llr[t]
        if cnt == size:
            rows.append([idx[b], idx[(b + cnt - 1) % size]] + list(acc / size))
    return np.array(rows).reshape(-1, 2 + nc)


@pytest.mark.parametrize("decime,wsize,wdec", [(1, 0, 0), (3, 0, 0), (1, 30, 30), (4, 20, 5)])
def test_compute_test_world_decime_and_window_llr(decime, wsize, wdec):
    """ComputeTest.cpp:154-207: DETERMINE_TOP_DISTRIBS every worldDecime-th frame of a segment, USE_TOP_DISTRIBS with the
    kept top set in between; WindowLLR over the per-frame LLRs.  Expected values from the oracle's per-frame functions."""
    from lia_ral_amd import host_capi as host
    C, D, topc = 64, 24, 6
    w, mean, iv = make_gmm(C, D, seed=5)
    rng = np.random.default_rng(1)
    cl_means = [mean + 0.3 * rng.normal(size=mean.shape) for _ in range(2)]
    x = make_frames(w, mean, iv, 400, seed=6)
    seg_begin, seg_len = np.array([10, 150, 300]), np.array([70, 90, 41])
    llr, wins = host.compute_test_ex(x, seg_begin, seg_len, (w, mean, 1.0 / iv), [(w, m, 1.0 / iv) for m in cl_means], top_c=topc,
                                     complete=True, segmental=True, world_decime=decime, window_size=wsize, window_dec=wdec)
    sel = np.concatenate([np.arange(b, b + n) for b, n in zip(seg_begin, seg_len)])
    xs = x[sel].astype(np.float64)
    gw = orc.Gmm(w, mean, iv)
    det = orc.llk_determine_top(gw, xs, topc, True, -200.0, 200.0)
    idx, nllk, llkw = det["idx"].copy(), det["nontop_lk"].copy(), det["llk"].copy()
    determined = np.ones(len(sel), bool)
    t = 0
    for n in seg_len:
        for f in range(n):
            if f % decime:
                last = t - (f % decime)
                idx[t] = idx[last]; nllk[t] = nllk[last]; determined[t] = False
            t += 1
    w2 = orc.llk_use_top(gw, xs, idx, nllk, True, -200.0, 200.0)
    llkw = np.where(determined, llkw, w2)
    llkc = np.stack([orc.llk_use_top(orc.Gmm(w, m, iv), xs, idx, nllk, True, -200.0, 200.0) for m in cl_means], 1)
    off = 0
    for s, n in enumerate(seg_len):
        ref = llkc[off:off + n].mean(0) - llkw[off:off + n].mean()
        assert np.max(np.abs(llr[s] - ref)) < 1e-9
        off += n
    if wsize:
        per_frame = llkc - np.where(determined, llkw, 0.0)[:, None]
        ref = _window_llr_reference(wsize, wdec or wsize, sel, per_frame)
        assert wins.shape == ref.shape and np.array_equal(wins[:, :2], ref[:, :2])
        assert np.max(np.abs(wins[:, 2:] - ref[:, 2:])) < 1e-9
    else:
        assert wins.shape[0] == 0


def _ivtest_data(seed=2, dim=20, nspk=40):
    rng = np.random.default_rng(seed)
    sps = rng.integers(3, 7, nspk)
    cls = np.repeat(np.arange(nspk), sps)
    A = rng.normal(size=(dim, dim)) * 0.4 + np.eye(dim)
    dev = A @ ((rng.normal(size=(dim, nspk)) * 1.3)[:, cls] + rng.normal(size=(dim, int(sps.sum())))) + 0.5
    epm = np.array([1, 2, 1, 3, 2])
    enrol = A @ rng.normal(size=(dim, int(epm.sum()))) * 1.5 + 0.5
    test = A @ rng.normal(size=(dim, 9)) * 1.5 + 0.5
    return dev, sps, enrol, epm, test


@pytest.mark.parametrize("scoring", ["cosine", "mahalanobis", "2cov", "plda"])
def test_iv_test_end_to_end(scoring):
    """IvTest.cpp:73-471 through liagpu::ivTest (EFR x 2, LDA, WCCN for cosine, back-end matrices or PLDA EM, scoring)
    against the same chain assembled from the oracle pieces."""
    from lia_ral_amd import host_capi as host
    dev, sps, enrol, epm, test = _ivtest_data()
    dim = dev.shape[0]
    lda_rank = 12
    rng = np.random.default_rng(7)
    rf, rg = 5, 3
    plda0 = (rng.normal(size=(lda_rank, rf)), 0.3 * rng.normal(size=(lda_rank, rg)), np.eye(lda_rank) * 0.5)
    got = host.iv_test(dev, sps, enrol, epm, test, scoring=scoring, iv_norm=True, iv_norm_it=2, lda_rank=lda_rank, wccn=(scoring == "cosine"),
                       plda=plda0 if scoring == "plda" else None, plda_it=3)
    # the same chain from oracle pieces (eigenvector signs of the EFR / LDA matrices are the solver's: take |.|-insensitive route by
    # recomputing with numpy's eigh-based restatement through the oracle functions, then fixing signs to the oracle's own)
    D, E, Tt = dev.copy(), enrol.copy(), test.copy()
    for it in range(2):
        S, W, B = orc.dev_cov_mat(D, sps)
        M = orc.dev_efr_matrix(S); mu = D.mean(1)
        f = lambda X: (lambda Y: Y / np.linalg.norm(Y, axis=0))(M @ (X - mu[:, None]))
        D, E, Tt = f(D), f(E), f(Tt)
    S, W, B = orc.dev_cov_mat(D, sps)
    L, _ = orc.dev_lda(W, B, lda_rank)
    D, E, Tt = L @ D, L @ E, L @ Tt
    starts = np.concatenate([[0], np.cumsum(epm)])
    if scoring == "cosine":
        U = orc.dev_wccn_chol(D, sps)
        E, Tt = U @ E, U @ Tt
        models = np.stack([E[:, a:b].mean(1) for a, b in zip(starts[:-1], starts[1:])], 1)
        ref = orc.score_cosine(models, Tt)
    elif scoring == "mahalanobis":
        S, W, B = orc.dev_cov_mat(D, sps)
        models = np.stack([E[:, a:b].mean(1) for a, b in zip(starts[:-1], starts[1:])], 1)
        ref = orc.score_mahalanobis(models, Tt, np.linalg.inv(W))
    elif scoring == "2cov":
        S, W, B = orc.dev_cov_mat(D, sps)
        models = np.stack([E[:, a:b].mean(1) for a, b in zip(starts[:-1], starts[1:])], 1)
        G, H = orc.twocov_model(W, B)
        ref = orc.score_twocov(models, Tt, G, H)
    else:
        st = (D - D.mean(1)[:, None],) + plda0 + (np.zeros(lda_rank),)
        for it in range(3):
            st = orc.plda_em_iteration(st[0], sps, *st[1:])
        FTJ, FTJF = orc.plda_precompute(st[1], st[2], st[3])
        sums = np.stack([(FTJ @ E[:, a:b]).sum(1) for a, b in zip(starts[:-1], starts[1:])], 1)
        ref = orc.score_plda(sums, epm, FTJ @ Tt, FTJF)
    # EFR / LDA eigenvectors are defined up to sign: cosine, Mahalanobis, 2cov and PLDA scores are invariant to a sign flip of a
    # whitening ROW only through the later steps -- all of them are (every step is linear or quadratic in the rotated vectors)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) < 1e-6 * max(1.0, np.max(np.abs(ref)))


def _jfa_problem(seed=5, C=8, D=6, rv=4, ru=3, nspk=12):
    rng = np.random.default_rng(seed)
    w, mean, iv = make_gmm(C, D, seed=seed)
    SV = C * D
    sps = rng.integers(1, 4, nspk); sb = np.concatenate([[0], np.cumsum(sps)]); nsess = int(sb[-1])
    owner = np.repeat(np.arange(nspk), sps)
    # statistics of a generative JFA model so that the EM steps have something to find
    V0 = rng.normal(size=(rv, SV)) * 0.5; U0 = rng.normal(size=(ru, SV)) * 0.3; D0 = rng.uniform(0.05, 0.3, SV)
    y = rng.normal(size=(nspk, rv)); xh = rng.normal(size=(nsess, ru)); z = rng.normal(size=(nspk, SV))
    Nh = rng.uniform(5, 60, (nsess, C))
    Ms = mean.ravel() + y[owner] @ V0 + xh @ U0 + D0 * z[owner]
    Fh = np.repeat(Nh, D, axis=1) * (Ms + rng.normal(size=(nsess, SV)) * 0.05)
    N = np.zeros((nspk, C)); F = np.zeros((nspk, SV))
    np.add.at(N, owner, Nh); np.add.at(F, owner, Fh)
    return dict(w=w, mean=mean, iv=iv, sps=sps, sb=sb, owner=owner, N=N, Nh=Nh, F=F, Fh=Fh,
                V=rng.normal(size=(rv, SV)) * 0.1, U=rng.normal(size=(ru, SV)) * 0.1, D=np.sqrt(1.0 / (iv.ravel() * 14.0)), C=C, Dm=D)


def _orc_y(p, V, U, Dv, X, Z):
    """the 'speaker factors first' block shared by EigenChannel.cpp:120-128 and EstimateDMatrix.cpp:143-151"""
    m, iv = p["mean"].ravel(), p["iv"].ravel()
    F = orc.jfa_subtract(p["N"], p["F"], None, m, None, None, Dv, Z)
    F = orc.jfa_subtract_sessions(p["sb"], p["Nh"], F, U, X)
    return orc.tv_estimate_w(p["N"], F, V, iv, orc.tv_tett(V, iv, p["C"], p["Dm"]))


def test_jfa_training_tools_match_the_oracle_loops():
    """EigenVoice / EigenChannel / EstimateDMatrix driver loops through liagpu::JFAAcc against the same loops assembled from
    the oracle's restatement of the JFAAcc methods (tests the orchestration: store / restore, which statistics feed which step)."""
    from lia_ral_amd import host_capi as hc
    p = _jfa_problem()
    C, D = p["C"], p["Dm"]; SV = C * D
    m, iv = p["mean"].ravel(), p["iv"].ravel()
    ubm = (p["w"], p["mean"], 1.0 / p["iv"])
    nspk, nsess = len(p["sps"]), len(p["owner"])
    rv, ru = p["V"].shape[0], p["U"].shape[0]
    il_v = np.tril_indices(rv); il_u = np.tril_indices(ru)
    full = lambda A, R: A.reshape(C, R * R)

    # ---- EigenVoice, 3 iterations (X = 0, Z = 0 like a first pass of the recipe)
    V = p["V"].copy(); X0 = np.zeros((nsess, ru)); Z0 = np.zeros((nspk, SV))
    for _ in range(3):
        te = orc.tv_tett(V, iv, C, D)
        F = orc.jfa_subtract(p["N"], p["F"], None, m, None, None, p["D"], Z0)
        F = orc.jfa_subtract_sessions(p["sb"], p["Nh"], F, p["U"], X0)
        Yo, Ao, Co = orc.jfa_estimate_y_and_v(p["N"], F, V, iv, te)
        V = orc.tv_update_t(full(Ao, rv), Co, C, D)
    g = hc.jfa_train(0, p["sps"], ubm, p["N"], p["Nh"], p["F"], p["Fh"], p["V"], p["U"], p["D"], 3)
    assert relerr(g["V"], V) < 1e-8 and relerr(g["Y"], Yo) < 1e-8
    assert np.array_equal(g["U"], p["U"]) and np.array_equal(g["D"], p["D"])
    V_tr = g["V"]

    # ---- EigenChannel, 3 iterations on top of the trained V
    Yo = _orc_y(p, V_tr, p["U"], p["D"], X0, Z0)
    U = p["U"].copy()
    for _ in range(3):
        te = orc.tv_tett(U, iv, C, D)
        Fh = orc.jfa_subtract(p["Nh"], p["Fh"], p["owner"], m, V_tr, Yo, p["D"], Z0)
        Xo, Ao, Co = orc.jfa_estimate_y_and_v(p["Nh"], Fh, U, iv, te)
        U = orc.tv_update_t(full(Ao, ru), Co, C, D)
    g = hc.jfa_train(1, p["sps"], ubm, p["N"], p["Nh"], p["F"], p["Fh"], V_tr, p["U"], p["D"], 3)
    assert relerr(g["U"], U) < 1e-8 and relerr(g["X"], Xo) < 1e-8 and relerr(g["Y"], Yo) < 1e-8
    U_tr = g["U"]

    # ---- EstimateDMatrix, 2 iterations on top of V and U
    Yo = _orc_y(p, V_tr, U_tr, p["D"], X0, Z0)
    Fh = orc.jfa_subtract(p["Nh"], p["Fh"], p["owner"], m, V_tr, Yo, p["D"], Z0)
    Xo = orc.tv_estimate_w(p["Nh"], Fh, U_tr, iv, orc.tv_tett(U_tr, iv, C, D))
    Dv = p["D"].copy()
    for _ in range(2):
        F = orc.jfa_subtract(p["N"], p["F"], None, m, V_tr, Yo)
        F = orc.jfa_subtract_sessions(p["sb"], p["Nh"], F, U_tr, Xo)
        Zo, Dv = orc.jfa_estimate_z_and_d(p["N"], F, iv, Dv)
    g = hc.jfa_train(2, p["sps"], ubm, p["N"], p["Nh"], p["F"], p["Fh"], V_tr, U_tr, p["D"], 2)
    assert relerr(g["D"], Dv) < 1e-8 and relerr(g["Z"], Zo) < 1e-8 and relerr(g["X"], Xo) < 1e-8
    # the estimated D moved towards the generating one (sanity of the recipe, not of the arithmetic)
    assert np.all(np.isfinite(g["D"])) and g["D"].min() > 0


def test_jfa_statistics_from_frames():
    """computeAndAccumulateJFAStat (AccumulateJFAStat.cpp:515-577): per-session Baum-Welch statistics on the device, speaker rows
    = sums over the speaker's sessions."""
    from lia_ral_amd import host_capi as hc
    C, D = 16, 12
    w, mean, iv = make_gmm(C, D, seed=4)
    x = make_frames(w, mean, iv, 900, seed=8).astype(np.float32)
    sps = np.array([2, 1, 3]); sb = np.array([0, 100, 250, 400, 520, 700, 900])
    N, Nh, FX, FXh = hc.jfa_stats(x, sb, sps, (w, mean, 1.0 / iv))
    g = orc.Gmm(w, mean, iv)
    utt = np.repeat(np.arange(len(sb) - 1), np.diff(sb))
    No, Fo = orc.tv_stats(g, x.astype(np.float64), utt, len(sb) - 1)
    assert relerr(Nh, No) < 1e-9 and relerr(FXh, Fo) < 1e-9
    own = np.repeat(np.arange(3), sps)
    Ns = np.zeros((3, C)); Fs = np.zeros((3, C * D))
    np.add.at(Ns, own, No); np.add.at(Fs, own, Fo)
    assert relerr(N, Ns) < 1e-9 and relerr(FX, Fs) < 1e-9


def test_jfa_dot_product_scoring():
    """ComputeTest.cpp:303-358: x of every test segment (y = z = 0), SPEAKER statistics minus N_h (m + U x), normalised by the
    occupancy, dotted with the client supervectors -- against the same steps assembled from the oracle's JFAAcc restatement
    (pins substractMplusUX to the reference's semantics: it modifies _F_X, not _F_X_h)."""
    from lia_ral_amd import host_capi as hc
    p = _jfa_problem(seed=9, nspk=7)
    C, D = p["C"], p["Dm"]; SV = C * D
    m, iv = p["mean"].ravel(), p["iv"].ravel()
    N, F = p["Nh"][:9], p["Fh"][:9]                       # 9 test segments
    nT = N.shape[0]
    rng = np.random.default_rng(2)
    clients = rng.normal(size=(5, SV))
    sb = np.arange(nT + 1)
    Fh = orc.jfa_subtract(N, F, None, m, None, None, None, None)                      # substractMplusVYplusDZ with y = z = 0
    X = orc.tv_estimate_w(N, Fh, p["U"], iv, orc.tv_tett(p["U"], iv, C, D))           # estimateX
    Fx = orc.jfa_subtract_m_plus_ux(sb, N, F, m, p["U"], X)                           # substractMplusUX on the speaker statistics
    ref = (Fx / N.sum(1, keepdims=True)) @ clients.T
    got = hc.jfa_dot_product((p["w"], p["mean"], 1.0 / p["iv"]), N, F, p["V"], p["U"], p["D"], clients)
    assert relerr(got, ref) < 1e-9



def _tv_case(seed=3):
    C, D, R, U = 7, 12, 10, 60                       # 7 Gaussians: blocks of 4 + 3 on two ranks
    rng = np.random.default_rng(seed)
    w, mean, iv = make_gmm(C, D, seed=seed)
    N = rng.gamma(0.8, 3.0, (U, C)); F = rng.normal(size=(U, C * D)) * 3 + np.repeat(N, D, 1) * mean.ravel()
    Tm = rng.normal(0, 0.05, (R, C * D))
    return C, D, R, U, w, mean, iv, N, F, Tm


def _oracle_tv_loop(C, D, U, mean, iv, N, F, Tm, nb_it):
    To, mo = Tm.copy(), mean.ravel().copy()
    for _ in range(nb_it):
        F0 = orc.tv_subtract_m(N, F, mo)
        o = orc.tv_estimate_a_and_c(N, F0, To, iv.ravel(), orc.tv_tett(To, iv.ravel(), C, D))
        To = orc.tv_update_t(o["A"], o["Cmx"], C, D)
        mo, To = orc.tv_min_divergence(o["Rm"], o["r"], o["meanW"], mo, To, U, C, D)
    return To, mo


def test_tv_train_dist_single_rank_is_the_plain_loop():
    """liagpu_tv_train_dist with one rank (device-resident TVAcc, no RCCL) == TotalVariability's loop from oracle pieces."""
    from lia_ral_amd import host_capi as h
    C, D, R, U, w, mean, iv, N, F, Tm = _tv_case()
    Tg, mg, times = h.tv_train_dist(N, F, (w, mean, 1.0 / iv), Tm, 2)
    To, mo = _oracle_tv_loop(C, D, U, mean, iv, N, F, Tm, 2)
    assert relerr(Tg, To) < 1e-6 and relerr(mg, mo) < 1e-8
    assert times.shape == (2, 4) and np.all(times > 0)


def _tv_rank(rank, world, idfile, q, transport="rccl", overlap=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["GMMIV_COMM_TRANSPORT"] = transport        # read by rank 0 when it draws the id (gmmiv_comm_exchange_id_file)
    from lia_ral_amd import host_capi as h
    from lia_ral_amd.dist import shard_range
    C, D, R, U, w, mean, iv, N, F, Tm = _tv_case()
    b, e = shard_range(U, rank, world)
    try:
        Tg, mg, _ = h.tv_train_dist(N[b:e], F[b:e], (w, mean, 1.0 / iv), Tm, 2, world=world, rank=rank, id_file=idfile, n_total=U,
                                    device=rank if transport == "rccl" else 0, overlap=overlap)
        q.put((rank, Tg, mg))
    except Exception as ex:      # noqa: BLE001 - the parent fails the test with the message
        q.put((rank, repr(ex), None))


def test_tv_train_dist_two_ranks_rccl(tmp_path):
    """Utterances sharded over two GPUs, reduce-scatter / all-gather through gmmiv_comm inside the C++ host layer: same T and
    means on both ranks, equal to the single-process loop."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_tv_rank, args=(r, 2, str(tmp_path / "id"), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    C, D, R, U, w, mean, iv, N, F, Tm = _tv_case()
    To, mo = _oracle_tv_loop(C, D, U, mean, iv, N, F, Tm, 2)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert relerr(res[0][1], To) < 1e-6 and relerr(res[0][2], mo) < 1e-8


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_tv_train_dist_ranks_share_gpu0_shm(world, overlap, tmp_path):
    """overlap: TVAcc::setOverlap (reduce-scatter of A begun from the "tv_a_ready" hook into the padded send buffer A lives in, the
    all-gather of T joined from the "md_factored" hook) -- the same assertions hold, in particular equality with the single-rank loop.
    The C++ host layer's multi-rank TotalVariability loop (liagpu_tv_train_dist: TVAcc::updateTestimate(comm) = reduce-scatter
    by padded Gaussian blocks, sharded solve, all-gather; AccumulateTVStat.cpp:974-1005, 1920-1937) with 2 and 3 processes on
    GPU 0 over the C ABI's shm transport: identical T / means on every rank, equal to the oracle's single-process loop."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_tv_rank, args=(r, world, str(tmp_path / "id"), q, "shm", overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r][2] is not None, res[r][1]
    C, D, R, U, w, mean, iv, N, F, Tm = _tv_case()
    To, mo = _oracle_tv_loop(C, D, U, mean, iv, N, F, Tm, 2)
    for r in range(1, world):
        assert np.array_equal(res[0][1], res[r][1]) and np.array_equal(res[0][2], res[r][2])
    assert relerr(res[0][1], To) < 1e-6 and relerr(res[0][2], mo) < 1e-8
    from lia_ral_amd import host_capi as h
    T1, m1, _ = h.tv_train_dist(N, F, (w, mean, 1.0 / iv), Tm, 2)           # the single-rank HIP loop
    assert relerr(res[0][1], T1) < 1e-10 and relerr(res[0][2], m1) < 1e-11
    other = _TV_SHM_RESULTS.get((world, not overlap))                       # serial and overlapped order: bitwise the same T and means
    if other is not None:
        assert np.array_equal(other[0], res[0][1]) and np.array_equal(other[1], res[0][2])
    _TV_SHM_RESULTS[(world, overlap)] = (res[0][1], res[0][2])


_TV_SHM_RESULTS = {}


def _tw_rank(rank, world, idfile, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["GMMIV_COMM_TRANSPORT"] = "shm"
    from lia_ral_amd import host_capi as h
    from lia_ral_amd.dist import shard_range
    C, D, T, x, w0, mean0, cov0, gcov = _tw_case()
    b, e = shard_range(T, rank, world)
    try:
        r = h.train_world_dist(x[b:e], [0], [e - b], w0, mean0, cov0, 2, gcov, world=world, rank=rank, id_file=idfile, init_floor=0.1, final_floor=0.1, device=0)
        q.put((rank, r["w"], r["mean"], r["cov"], r["llk"]))
    except Exception as ex:      # noqa: BLE001
        q.put((rank, repr(ex), None, None, None))


def _tw_case():
    C, D, T = 16, 12, 3001
    w, mean, iv = make_gmm(C, D, seed=5)
    x = make_frames(w, mean, iv, T, seed=6)
    gcov = x.astype(np.float64).var(0)
    return C, D, T, x, np.full(C, 1.0 / C), mean + 0.3, np.ones((C, D)) * 2.0, gcov


def test_train_world_dist_ranks_share_gpu0_shm(tmp_path):
    """TrainWorld's EM over frame-sharded ranks (trainModelStream(..., comm): ONE gmmiv_allreduce_f64 of the flat accumulator per
    iteration, AccumulateStat.cpp:286-292) with two processes on GPU 0 over shm == the single-rank run to summation-order accuracy."""
    import torch.multiprocessing as mp
    from lia_ral_amd import host_capi as h
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_tw_rank, args=(r, 2, str(tmp_path / "id"), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][2] is not None and res[1][2] is not None, (res[0][1], res[1][1])
    C, D, T, x, w0, mean0, cov0, gcov = _tw_case()
    one = h.train_world_dist(x, [0], [T], w0, mean0, cov0, 2, gcov, init_floor=0.1, final_floor=0.1)
    for k, name in ((1, "w"), (2, "mean"), (3, "cov")):
        assert np.array_equal(res[0][k], res[1][k])
        assert relerr(res[0][k], one[name]) < 1e-10, name
    assert relerr(res[0][4], one["llk"]) < 1e-10


def test_train_world_dist_single_rank_matches_train_world():
    from lia_ral_amd import host_capi as h
    C, D, T = 16, 12, 3000
    w, mean, iv = make_gmm(C, D, seed=5)
    x = make_frames(w, mean, iv, T, seed=6)
    w0 = np.full(C, 1.0 / C); mean0 = mean + 0.3; cov0 = np.ones((C, D)) * 2.0
    a = h.train_world(x, [0], [T], w0, mean0, cov0, nb_it=2, init_floor=0.1, final_floor=0.1)
    b = h.train_world_dist(x, [0], [T], w0, mean0, cov0, 2, a["global_cov"], init_floor=0.1, final_floor=0.1)
    assert np.array_equal(a["mean"], b["mean"]) and np.array_equal(a["cov"], b["cov"]) and np.array_equal(a["llk"], b["llk"])


def test_accumulate_stat_llk_wrapper_and_kat4_through_frame_moments(golden_dir):
    """KAT-4 (LIA_SpkDet/NormFeat/test/test1.validate.prm: FrameAccGD mean / biased std, AccumulateStat.cpp:387-396) through
    gmmiv_frame_moments on the device, and liagpu::accumulateStatLLK (AccumulateStat.cpp:69-94) through the host layer."""
    from lia_ral_amd import capi, host_capi as h
    k = np.load(os.path.join(golden_dir, "kat4_normfeat.npz"))
    rows = np.concatenate([np.arange(b, b + n) for b, n in zip(k["seg_begin"], k["seg_len"])])
    x = np.ascontiguousarray(k["x"][rows], np.float32)
    D = x.shape[1]
    ctx = capi.Context(0)
    acc = ctx.frame_moments(x)
    ctx.close()
    n = acc[2 * D]
    mean = acc[:D] / n
    cov = acc[D:2 * D] / n - mean * mean                    # biased (ddof = 0), FrameAccGD::getCovVect
    diff = np.abs((x.astype(np.float64) - mean) / np.sqrt(cov) - k["x_norm"][rows].astype(np.float64))
    assert n == len(rows) and np.median(diff) < float(k["median_tol"]) and diff.max() < float(k["max_tol"])
    s_, ss_, cnt = orc.frame_acc(x.astype(np.float64))
    assert relerr(acc[:D], s_) < 1e-13 and relerr(acc[D:2 * D], ss_) < 1e-13 and n == cnt
    # accumulateStatLLK: mean clamped log-likelihood over label segments
    C, T = 16, 900
    w, m, iv = make_gmm(C, D, seed=2)
    xx = make_frames(w, m, iv, T, seed=3)
    segs_b, segs_l = [10, 400], [200, 333]
    got = h.mean_llk(xx, segs_b, segs_l, (w, m, 1.0 / iv), -200.0, 200.0)
    sel = np.concatenate([np.arange(b, b + n_) for b, n_ in zip(segs_b, segs_l)])
    ref = orc.llk(orc.Gmm(w, m, iv), xx[sel].astype(np.float64)).mean()
    assert abs(got - ref) < 1e-10
    floor = h.mean_llk(xx, segs_b, segs_l, (w, m, 1.0 / iv), ref + 50.0, 400.0)      # every frame clamped to minLLK
    assert abs(floor - (ref + 50.0)) < 1.0 and floor >= ref + 50.0 - 1e-12


def test_init_t_and_statistics_with_a_file_on_several_lines():
    """TVAcc::initT (AccumulateTVStat.cpp:701-757, Box-Muller chain on glibc rand(), ScoreWarp.cpp:68-81) and
    computeAndAccumulateTVStat with the file -> ndx-line map (:318-346): a file listed on two lines counts for both, a line with
    two files sums them, an empty line stays zero."""
    from lia_ral_amd import capi, host_capi as h
    C, D, R = 16, 12, 5
    w, mean, iv = make_gmm(C, D, seed=4)
    ubm = (w, mean, 1.0 / iv)
    for seed in (1, 77):
        Tg = h.tv_init_t(ubm, R, seed)
        To = orc.tv_init_t(R, iv.ravel(), seed)
        assert np.array_equal(Tg, To) and np.all(np.isfinite(Tg)) and abs(Tg.std() / (iv.sum() * 0.001) - 1.0) < 0.2
    lens = [120, 75, 200, 33]
    fb = np.concatenate([[0], np.cumsum(lens)])
    x = make_frames(w, mean, iv, int(fb[-1]), seed=5)
    lines = [[0], [1, 2], [], [2], [3, 0]]
    og = orc.Gmm(w, mean, iv)
    Nf, Ff = orc.tv_stats(og, x.astype(np.float64), np.repeat(np.arange(4), lens), 4)
    Nref = np.array([Nf[l].sum(0) if l else np.zeros(C) for l in lines])
    Fref = np.array([Ff[l].sum(0) if l else np.zeros(C * D) for l in lines])
    N, F = h.tv_stats_lines(x, fb, lines, ubm)                        # C++ host layer
    assert relerr(N, Nref) < 1e-10 and relerr(F, Fref) < 1e-10 and not N[2].any() and not F[2].any()
    ctx = capi.Context(0)
    N2, F2 = ctx.gmm(w, mean, iv).tv_stats_lines(x, fb, lines)        # C ABI directly
    assert np.array_equal(N2, N) and np.array_equal(F2, F)
    ctx.close()


@pytest.mark.parametrize("select", [50.0, 2000.0])
def test_mixture_init_and_train_world_from_scratch(select):
    """mixtureInit (TrainTools.cpp:674-766, the form TrainWorld.cpp:177 calls): per-component random picking of 3..7-frame chunks
    with the reference's srand / rand() order (multi-selection baggedSegments, GeneralTools.cpp:330-390), means of the picked
    frames through gmmiv_frame_moments, covariances = globalCov, equal weights -- against the oracle restatement; then TrainWorld
    runs FROM SCRATCH (no initial model given) and agrees with the oracle's EM loop started from the oracle's init.  select = 2000 on
    300 frames drives the bagging probability to 6.67: three bagging passes of p = 0.9 (:700-708); a probability in (1, 4.9) makes the
    reference's fold loop spin forever -- the host layer reports it instead."""
    from lia_ral_amd import host_capi as h
    C, D = 12, 10
    w, mean, iv = make_gmm(C, D, seed=8)
    T = 6000 if select == 50.0 else 330
    x = make_frames(w, mean, iv, T, seed=9)
    seg_begin, seg_len = ([0, 2600], [2500, 3400]) if select == 50.0 else ([0, 180], [150, 150])
    sel = np.concatenate([np.arange(b, b + n) for b, n in zip(seg_begin, seg_len)])
    s, ss, n = orc.frame_acc(x[sel].astype(np.float64))
    _, gcov = orc.frame_mean_cov(s, ss, n)
    got = h.mixture_init(x, seg_begin, seg_len, C, gcov, nb_frame_to_select=select)
    m_o, cnt_o = orc.mixture_init(C, x.astype(np.float64), seg_begin, seg_len, nb_frame_to_select=select)
    assert np.array_equal(got["counts"], cnt_o.astype(np.int64)) and got["counts"].min() > 0     # the same frames were drawn
    assert relerr(got["mean"], m_o) < 1e-12
    assert np.array_equal(got["cov"], np.tile(gcov, (C, 1))) and np.array_equal(got["w"], np.full(C, 1.0 / C))
    if select != 50.0:
        with pytest.raises(h.HostError, match="does not terminate"):
            h.mixture_init(x, seg_begin, seg_len, C, gcov, nb_frame_to_select=400.0)
    if select == 50.0:
        tw = h.train_world(x, seg_begin, seg_len, got["w"], got["mean"], got["cov"], nb_it=3, init_floor=0.5, final_floor=0.05, init_ceil=5.0,
                           final_ceil=10.0)
        ref = oracle_train_world(x, np.array(seg_begin), np.array(seg_len), np.full(C, 1.0 / C), m_o, np.tile(gcov, (C, 1)), 3, 1.0, 0.5, 0.05,
                                 5.0, 10.0)
        assert np.max(np.abs(tw["llk"] - ref["llk"])) < 1e-9 and np.all(np.diff(tw["llk"]) > 0)   # EM from scratch: llk rises
        assert relerr(tw["mean"], ref["mean"]) < 1e-8 and relerr(tw["cov"], ref["cov"]) < 1e-8
        one = h.train_world_scratch(x, seg_begin, seg_len, C, 3, nb_frame_to_select=select, init_floor=0.5, final_floor=0.05, init_ceil=5.0,
                                    final_ceil=10.0)                                            # the whole tool in one call
        assert relerr(one["mean"], tw["mean"]) < 1e-9 and relerr(one["cov"], tw["cov"]) < 1e-9 and np.max(np.abs(one["llk"] - tw["llk"])) < 1e-9
        assert relerr(one["global_cov"], gcov) < 1e-12


def test_mixture_init_single_stream_form():
    """The single-stream mixtureInit (TrainTools.cpp:619-672): one plain baggedSegments pass per component, seed (c + 1)(it + 1),
    p = baggedFrameProbabilityInit / distribCount -- against the same selection assembled from oracle pieces."""
    from lia_ral_amd import host_capi as h
    C, D, T = 6, 8, 4000
    w, mean, iv = make_gmm(C, D, seed=4)
    x = make_frames(w, mean, iv, T, seed=5)
    seg_begin, seg_len = [10, 2100], [1900, 1800]
    gcov = np.full(D, 1.5)
    for pinit, nb_it in ((0.3, 1), (7.5, 2)):          # 7.5 / 6 > 1: two bagging passes of p = 0.625
        got = h.mixture_init(x, seg_begin, seg_len, C, gcov, single_stream_proba=pinit)
        p = pinit / C
        if p > 1:
            assert nb_it == int(p) + 1
            p /= nb_it
        for c in range(C):
            fr = []
            for it in range(nb_it):
                bb, bl, _ = orc.bagged_segments((c + 1) * (it + 1), seg_begin, seg_len, p, 3, 7)
                fr += [np.arange(b, b + n) for b, n in zip(bb, bl)]
            fr = np.concatenate(fr)
            assert got["counts"][c] == len(fr)
            assert relerr(got["mean"][c], x[fr].astype(np.float64).mean(0)) < 1e-12


def test_tv_verify_emlk_and_speaker_models():
    """TVAcc::getMplusTW / getSpeakerModel / getLLK / verifyEMLK (AccumulateTVStat.cpp:964-971, 1533-1545, 1626-1688): the
    supervector m + T^T w of each file's statistics row, the mean log-likelihood of the file under the UBM with those means, the
    total over the first `computeLLK` files."""
    from lia_ral_amd import host_capi as h
    C, D, R, U = 8, 12, 5, 3
    w, mean, iv = make_gmm(C, D, seed=11)
    rng = np.random.default_rng(12)
    Tm = rng.normal(0, 0.3, (R, C * D)); W = rng.normal(size=(U, R))
    x = make_frames(w, mean, iv, 1700, seed=13)
    file_begin = [0, 400, 900, 1000, 1700]; rows = [0, 1, 1, 2]      # two files on statistics row 1
    llk, total, sv = h.tv_verify_emlk(x, file_begin, rows, (w, mean, 1.0 / iv), Tm, W)
    sv_ref = mean.ravel()[None, :] + W[rows] @ Tm
    assert relerr(sv, sv_ref) < 1e-13
    ref = []
    for f in range(4):
        gm = orc.Gmm(w, sv_ref[f].reshape(C, D), iv)
        ref.append(orc.llk(gm, x[file_begin[f]:file_begin[f + 1]].astype(np.float64)).mean())
    assert np.max(np.abs(llk - np.array(ref))) < 1e-9 and abs(total - sum(ref)) < 1e-8
    llk2, total2, _ = h.tv_verify_emlk(x, file_begin, rows, (w, mean, 1.0 / iv), Tm, W, max_llk_computed=2)   # config "computeLLK 2"
    assert len(llk2) == 2 and abs(total2 - sum(ref[:2])) < 1e-8


def oracle_train_world_streams(xs, segs, weights, w, mean, cov, nb_it, p, fl0, fl1, ce0, ce1, init_rand=0, target=None, normalize=False,
                               mean_only=False, norm_it=1):
    """The stream form of trainModelStream (TrainTools.cpp:1030-1110) from oracle pieces: per iteration and stream a bagging pass
    (or several when the stream's probability exceeds 1), ONE accumulator, then getEM / varianceControl / componentReduction /
    normalizeModel."""
    xs = [x.astype(np.float64) for x in xs]
    sel = [np.concatenate([np.arange(b, b + n) for b, n in zip(sb, sl)]) for sb, sl in segs]
    s = ss = n = 0
    for x, se in zip(xs, sel):
        a, b, c = orc.frame_acc(x[se]); s = s + a; ss = ss + b; n = n + c
    gmean, gcov = orc.frame_mean_cov(s, ss, n)
    total = [int(np.sum(sl)) for _, sl in segs]
    C0 = len(w)
    llks, frames = [], []
    for it in range(nb_it):
        floor = orc.set_it_parameter(fl0, fl1, nb_it, it); ceil = orc.set_it_parameter(ce0, ce1, nb_it, it)
        nb_total = sum(int(float(t) * wt) for t, wt in zip(total, weights))
        nb_sel = p * nb_total
        acc = None
        g = orc.Gmm(w, mean, 1.0 / cov)
        for st, (x, (sb, sl)) in enumerate(zip(xs, segs)):
            proba = nb_sel * weights[st] / float(total[st]); nb_bag = 1
            if proba > 1:
                nb_bag = int(proba) + 1; proba /= nb_bag
            for b in range(nb_bag):
                seed = ((it + 1 + init_rand) * 200) + ((st + 1) * 20) + (b + 1)
                bb, bl, _ = orc.bagged_segments(seed, sb, sl, proba, 3, 7)
                fr = np.concatenate([np.arange(q, q + m) for q, m in zip(bb, bl)]) if len(bb) else np.zeros(0, int)
                acc = orc.em_accumulate(g, x[fr], acc=acc)
        llks.append(acc["llk"] / acc["count"]); frames.append(acc["count"])
        w, mean, cov = orc.em_get(acc, mean, cov)
        cov, _, _ = orc.variance_control(cov, floor, ceil, gcov)
        if target is not None:
            diff = (C0 - target) / float(nb_it)
            nb_top = C0 - int(float(it + 1) * diff)
            if it == nb_it - 1:
                nb_top = target
            if nb_top < len(w):
                w, mean, cov = orc.reduce_model(w, mean, cov, nb_top)
        if normalize:
            mean, cov = orc.normalize_mixture(w, mean, cov, norm_it if mean_only else 1, mean_only)
    return dict(w=w, mean=mean, cov=cov, llk=np.array(llks), frames=np.array(frames), global_mean=gmean, global_cov=gcov)


@pytest.mark.parametrize("case", ["two_streams", "weights_fold", "reduction", "normalize", "normalize_mean_only"])
def test_train_model_stream_streams_reduction_normalisation(case):
    """trainModelStream over nbStream input streams with weightTab (seed term (stream + 1) * 20, probability nbFrameToSelect *
    weight / totalFrame(stream), folded into several passes when > 1), componentReduction (selectComponent / reduceModel /
    normalizeWeights) and normalizeModel (TrainTools.cpp:1059-1099) through liagpu::trainModelStream, against the oracle loop."""
    from lia_ral_amd import host_capi as h
    C, D = 24, 12
    w, mean, iv = make_gmm(C, D, seed=15)
    rng = np.random.default_rng(1)
    xs = [make_frames(w, mean, iv, 5000, seed=16), make_frames(w, mean, iv, 1800, seed=17)]
    segs = [(np.array([0, 2600]), np.array([2400, 2300])), (np.array([100, 900]), np.array([700, 800]))]
    w0 = rng.dirichlet(np.full(C, 5.0)); mean0 = mean + rng.normal(0, 0.5, mean.shape); cov0 = np.ones((C, D)) * 2.0
    kw = dict(nb_it=3, bagged_p=0.6, init_floor=0.5, final_floor=0.05, init_ceil=5.0, final_ceil=10.0)
    okw = dict(target=None)
    weights = [0.5, 0.5]
    if case == "weights_fold":
        weights = [0.2, 0.8]        # stream 1: 0.6 * (940 + 1200) * 0.8 / 1500 = 0.68; with p = 1.5 below it exceeds 1 -> 2 passes
        kw["bagged_p"] = 1.5
    if case == "reduction":
        kw.update(component_reduction=True, target_distrib_count=10); okw["target"] = 10
    if case.startswith("normalize"):
        mo = case.endswith("mean_only")
        kw.update(normalize_model=True, normalize_mean_only=mo, normalize_nb_it=2); okw.update(normalize=True, mean_only=mo, norm_it=2)
    got = h.train_world_streams(xs, [s[0] for s in segs], [s[1] for s in segs], w0, mean0, cov0, weights=weights, **kw)
    ref = oracle_train_world_streams(xs, segs, weights, w0, mean0, cov0, 3, kw["bagged_p"], 0.5, 0.05, 5.0, 10.0, **okw)
    assert relerr(got["global_cov"], ref["global_cov"]) < 1e-12 and relerr(got["global_mean"], ref["global_mean"]) < 1e-12
    assert got["w"].shape == ref["w"].shape and (case != "reduction" or len(got["w"]) == 10)
    assert np.max(np.abs(got["llk"] - ref["llk"])) < 1e-9
    assert relerr(got["w"], ref["w"]) < 1e-8 and relerr(got["mean"], ref["mean"]) < 1e-8 and relerr(got["cov"], ref["cov"]) < 1e-8
    if case == "two_streams":       # default weights = 1 / nbStream (reserveMem, TrainWorld.cpp:85)
        dflt = h.train_world_streams(xs, [s[0] for s in segs], [s[1] for s in segs], w0, mean0, cov0, **kw)
        assert np.array_equal(dflt["mean"], got["mean"]) and np.array_equal(dflt["llk"], got["llk"])
        # one stream through the stream form == the single-stream entry point, bit for bit
        a = h.train_world_streams(xs[:1], [segs[0][0]], [segs[0][1]], w0, mean0, cov0, **kw)
        b = h.train_world(xs[0], segs[0][0], segs[0][1], w0, mean0, cov0, **kw)
        assert np.array_equal(a["mean"], b["mean"]) and np.array_equal(a["cov"], b["cov"]) and np.array_equal(a["llk"], b["llk"])


def test_feature_selection_by_runs_matches_the_frame_list():
    """gmmiv_gather_runs (run table: source frame, output row, length) against gmmiv_gather_frames (one index per frame) and numpy:
    bagged 3..7-frame chunks, a long label segment cut into 64-frame pieces, f32 and f64, a row-strided source."""
    import torch
    from lia_ral_amd import capi
    ctx = capi.Context(0)
    rng = np.random.default_rng(4)
    for dt, D, ld in [(torch.float32, 60, 60), (torch.float64, 60, 60), (torch.float32, 19, 19), (torch.float32, 33, 40)]:
        T = 5000
        xfull = torch.randn(T, ld, dtype=dt, device="cuda")
        x = xfull[:, :D]
        begins = np.sort(rng.choice(np.arange(0, T - 8, 8), 300, replace=False)); lens = rng.integers(3, 8, 300)
        runs = [(int(b), 0, int(l)) for b, l in zip(begins, lens)] + [(1000 + 64 * i, 0, 64) for i in range(20)] + [(4000, 0, 37)]
        dst = 0; table = []
        for b, _, l in runs:
            table.append((b, dst, l)); dst += l
        table = np.array(table, np.int64)
        idx = np.concatenate([np.arange(b, b + l) for b, _, l in table])
        out_r = torch.full((dst, D), -1.0, dtype=dt, device="cuda"); out_f = torch.empty_like(out_r); out_d = torch.empty_like(out_r)
        ctx.gather_runs(x, table, out_r)
        ctx.gather_frames(x, idx, out_f)
        ctx.gather_runs(x, torch.from_numpy(table).cuda(), out_d)     # device table: enqueue only
        ctx.sync()
        ref = x.cpu().numpy()[idx]
        assert np.array_equal(out_r.cpu().numpy(), ref) and np.array_equal(out_f.cpu().numpy(), ref) and np.array_equal(out_d.cpu().numpy(), ref)
    ctx.close()


@pytest.mark.parametrize("top_gauss,cap", [(0.9, 64), (0.999, 64), (5.0, 64), (0.99999999, 100), (90.0, 100)])   # lists longer than 64: the any-shape kernels (round 5)
def test_topgauss_mass_threshold_cache_file_and_get(top_gauss, cap, tmp_path):
    """TopGauss (LIA_SpkTools/src/TopGauss.cpp): compute with topGauss < 1 -- Gaussians until the cumulative likelihood passes
    topGauss * exp(llk): a VARIABLE count per frame (:162-167) -- and with a fixed count (:170); sumNonSelectedWeights / LLK with
    the EPS_LK floor (:183-192); the nbGaussian cache file in the reference's binary layout (:200-224), read back (:76-98) and
    parsed here independently; get() on the stored selection for the UBM and for a mean-adapted model (:275-316).  Against the
    oracle restatement: counts and indices exact."""
    from lia_ral_amd import host_capi as h
    C, D, T = 128, 20, 3000
    w, mean, iv = make_gmm(C, D, seed=21, spread=0.7)          # overlapping Gaussians: several carry mass per frame
    x = make_frames(w, mean, iv, T, seed=22)
    seg_begin, seg_len = np.array([10, 1500]), np.array([1200, 1400])
    sel = np.concatenate([np.arange(b, b + n) for b, n in zip(seg_begin, seg_len)])
    rng = np.random.default_rng(3)
    mean2 = mean + rng.normal(0, 0.1, mean.shape)
    path = str(tmp_path / "utt.nbg")
    got = h.topgauss(x, seg_begin, seg_len, (w, mean, 1.0 / iv), top_gauss, path, top_distribs_count=cap, model2_mean=mean2)
    og = orc.Gmm(w, mean, iv)
    xs = x[sel].astype(np.float64)
    ref = orc.topgauss_compute(og, xs, cap, top_gauss)
    assert np.array_equal(got["nbg"], ref["nbg"]) and np.array_equal(got["idx"], ref["idx"])
    if top_gauss < 1:
        assert got["nbg"].min() >= 1 and len(np.unique(got["nbg"])) > 3         # the count really varies
        assert got["capped"] <= int(np.sum(ref["nbg"] == cap))                   # capped frames have cap entries
    else:
        assert np.all(got["nbg"] == int(top_gauss))
    assert np.max(np.abs(got["snsw"] - ref["snsw"])) < 1e-12
    assert np.max(np.abs(got["snsl"] - ref["snsl"]) / np.maximum(np.exp(ref["llk"]), 1e-300)) < 1e-9   # relative to the frame likelihood
    assert abs(got["llk_compute"] - ref["llk"].mean()) < 1e-9
    llk_o = orc.topgauss_get(og, xs, ref["nbg"], ref["idx"], ref["snsl"])
    assert abs(got["llk_get"] - llk_o.mean()) < 1e-9
    llk_o2 = orc.topgauss_get(orc.Gmm(w, mean2, iv), xs, ref["nbg"], ref["idx"], ref["snsl"])
    assert abs(got["llk_get_model2"] - llk_o2.mean()) < 1e-9
    # COMPLETE mode on the UBM itself: selection + remainder = the full likelihood
    assert abs(got["llk_get"] - got["llk_compute"]) < 1e-9
    # the file, parsed without the library: _nt, _nbgcnt, _nbg, _idx (8-byte unsigned), _snsw, _snsl (double)
    raw = open(path, "rb").read()
    nt, cnt = np.frombuffer(raw, np.uint64, 2)
    assert nt == len(sel) and cnt == got["nbg"].sum() and len(raw) == 8 * (2 + 3 * int(nt) + int(cnt))
    nbg_f = np.frombuffer(raw, np.uint64, int(nt), 16)
    idx_f = np.frombuffer(raw, np.uint64, int(cnt), 16 + 8 * int(nt))
    snsw_f = np.frombuffer(raw, np.float64, int(nt), 16 + 8 * int(nt + cnt))
    snsl_f = np.frombuffer(raw, np.float64, int(nt), 16 + 8 * int(2 * nt + cnt))
    assert np.array_equal(nbg_f.astype(np.int64), got["nbg"]) and np.array_equal(idx_f.astype(np.int64), got["idx"])
    assert np.array_equal(snsw_f, got["snsw"]) and np.array_equal(snsl_f, got["snsl"])
    # an unwritable path is reported like the reference does (TopGauss.cpp:204)
    with pytest.raises(h.HostError, match="Cannot find nbGaussian file"):
        h.topgauss(x, seg_begin, seg_len, (w, mean, 1.0 / iv), top_gauss, str(tmp_path / "no" / "dir.nbg"), top_distribs_count=cap)


def test_segment_means_on_the_device():
    """gmmiv_segment_means: per-segment means of rows of per-frame values (what ComputeTest prints per segment), ragged and empty
    segments, a segment longer than one 8192-frame piece; against numpy."""
    import torch
    from lia_ral_amd import capi
    ctx = capi.Context(0)
    rng = np.random.default_rng(6)
    n = 40000
    v = rng.normal(-90, 5, (5, n))
    vd = torch.from_numpy(v).cuda()
    sb = np.array([0, 1, 1, 700, 9000, 9000, 30000, n])
    got = ctx.segment_means(vd, sb)
    ref = np.array([[v[r, sb[s]:sb[s + 1]].mean() if sb[s + 1] > sb[s] else 0.0 for s in range(len(sb) - 1)] for r in range(5)])
    assert np.max(np.abs(got - ref)) < 1e-11
    one = ctx.segment_means(vd[2], np.array([0, n]))
    assert abs(one[0, 0] - v[2].mean()) < 1e-11
    again = ctx.segment_means(vd, sb)
    assert np.array_equal(got, again)           # fixed summation order
    ctx.close()


def test_mixture_init_over_several_streams():
    """mixtureInit(ms, fsTab, segTab, weightTab, nbStream, ...) (TrainTools.cpp:674-766, the call of TrainWorld.cpp:177): per stream its own
    probability nbFrameToSelect * weight / totalFrame and its own seeds (stream + 1) * 100 + baggedIt + 1; ONE frame accumulator per
    component over all streams -- against the oracle, stream by stream."""
    from lia_ral_amd import host_capi as h
    C, D = 10, 8
    w, mean, iv = make_gmm(C, D, seed=31)
    xs = [make_frames(w, mean, iv, 5000, seed=32), make_frames(w, mean, iv, 2200, seed=33)]
    segs = [(np.array([0, 2600]), np.array([2400, 2300])), (np.array([100, 1200]), np.array([900, 950]))]
    gcov = np.linspace(0.5, 1.5, D)
    for weights in ([0.5, 0.5], [0.2, 0.8]):
        got = h.mixture_init_streams(xs, [s[0] for s in segs], [s[1] for s in segs], C, gcov, weights=weights, nb_frame_to_select=60.0)
        m_o, cnt_o = orc.mixture_init_streams(C, [x.astype(np.float64) for x in xs], segs, weights, nb_frame_to_select=60.0)
        assert np.array_equal(got["counts"], cnt_o.astype(np.int64)) and got["counts"].min() > 0     # the same frames were drawn in every stream
        assert relerr(got["mean"], m_o) < 1e-12
        assert np.array_equal(got["cov"], np.tile(gcov, (C, 1))) and np.array_equal(got["w"], np.full(C, 1.0 / C))
    one = h.mixture_init_streams(xs[:1], [segs[0][0]], [segs[0][1]], C, gcov, weights=[1.0], nb_frame_to_select=60.0)
    ref = h.mixture_init(xs[0], segs[0][0], segs[0][1], C, gcov, nb_frame_to_select=60.0)                # the one-stream entry point
    assert np.array_equal(one["mean"], ref["mean"]) and np.array_equal(one["counts"], ref["counts"])


@pytest.mark.parametrize("method,opts", [("MAPOccDep", dict(var=True, weight=True)), ("MAPModelBased", dict()), ("MAPConst", dict()),
                                         ("MAPConst2", dict()), ("MAPOccDep", dict(normalize=True))])
def test_adapt_model_methods_and_normalisation(method, opts):
    """adaptModel (TrainTools.cpp:871-904): per iteration baggedSegments (before srand(trainIt), as written), EM statistics on the HIP path,
    computeMAP with the configured method, optional normalizeMixture -- against the same loop assembled from oracle pieces."""
    from lia_ral_amd import host_capi as h
    C, D, T = 16, 10, 5000
    w, mean, iv = make_gmm(C, D, seed=41)
    x = make_frames(w, mean, iv, T, seed=42)
    rng = np.random.default_rng(5)
    world = (w, mean + rng.normal(0, 0.2, mean.shape), 1.0 / iv)
    seg_begin, seg_len = np.array([0, 2600]), np.array([2400, 2300])
    nb_it, p, reg, alpha = 2, 0.7, (12.0, 8.0, 30.0), 0.6
    opts = dict(opts)
    normalize = opts.pop("normalize", False)
    kw = dict(mean=True, var=False, weight=False); kw.update(opts)
    import ctypes as ct
    libc = ct.CDLL("libc.so.6")
    # adaptModel bags BEFORE it seeds (baggedSegments, then srand(trainIt), :880-883): iteration 0 draws from the state the process is in
    # -- set here through the same libc --, iteration it > 0 from srand(it - 1)
    libc.srand(777)
    got = h.train_target_ex(x, seg_begin, seg_len, world, method=method, nb_it=nb_it, bagged_p=p, reg=reg, alpha_mean=alpha, normalize=normalize, **kw)
    xd = x.astype(np.float64)
    cw, cm, cc = [np.array(a, np.float64) for a in world]
    for it in range(nb_it):
        bb, bl, _ = orc.bagged_segments(777 if it == 0 else it - 1, seg_begin, seg_len, p, 3, 7)
        fr = np.concatenate([np.arange(b, b + n) for b, n in zip(bb, bl)])
        acc = orc.em_accumulate(orc.Gmm(cw, cm, 1.0 / cc), xd[fr])
        mw, mm, mc = orc.em_get(acc, cm, cc)
        cw, cm, cc = orc.compute_map(method, world, (mw, mm, mc), float(int(acc["count"])), reg=reg, alpha_mean=alpha, **kw)
        if normalize:
            cm, cc = orc.normalize_mixture(cw, cm, cc, 1, False)
    assert relerr(got[0], cw) < 1e-9 and relerr(got[1], cm) < 1e-9 and relerr(got[2], cc) < 1e-9
    if normalize:       # global moments of the adapted mixture: mean 0, variance 1
        m1 = (got[0][:, None] * got[1]).sum(0); m2 = (got[0][:, None] * (got[2] + got[1] ** 2)).sum(0)
        assert np.max(np.abs(m1)) < 1e-10 and np.max(np.abs(m2 - 1.0)) < 1e-10


def test_mean_likelihood_over_streams_and_decision_weights():
    """meanLikelihood on a set of input streams (GeneralTools.cpp:599-607) and with one decision weight per feature server
    (computeAndAccumulateLLK(f, weight): :610-624, AccumulateStat.cpp:344-379): sum w llk / sum w, against the oracle."""
    from lia_ral_amd import host_capi as h
    C, D = 16, 10
    w, mean, iv = make_gmm(C, D, seed=51)
    xs = [make_frames(w, mean, iv, 1500, seed=52), make_frames(w, mean, iv, 900, seed=53) * 1.5]
    segs = [(np.array([0, 800]), np.array([700, 650])), (np.array([50]), np.array([800]))]
    og = orc.Gmm(w, mean, iv)
    llk = []
    for x, (sb, sl) in zip(xs, segs):
        fr = np.concatenate([np.arange(b, b + n) for b, n in zip(sb, sl)])
        llk.append(orc.llk(og, x[fr].astype(np.float64)))
    model = (w, mean, 1.0 / iv)
    got = h.mean_llk_streams(xs, [s[0] for s in segs], [s[1] for s in segs], model)
    assert abs(got - np.concatenate(llk).mean()) < 1e-9
    dec = [0.25, 2.0]
    got = h.mean_llk_streams(xs, [s[0] for s in segs], [s[1] for s in segs], model, decision=dec)
    ref = sum(d * l.sum() for d, l in zip(dec, llk)) / sum(d * len(l) for d, l in zip(dec, llk))
    assert abs(got - ref) < 1e-9
