"""Degenerate inputs through every frame-consuming entry point of the C ABI, on every kernel path (include/gmmiv.h, "DEGENERATE
INPUTS"): non-finite and absurd feature values, frames further from every Gaussian than the arithmetic can resolve, a Gaussian
of weight 0, occupancy 0 in the M-step, identical Gaussians (ties over a whole row), T = 0.  The rule: such a frame is a
zero-likelihood frame -- llk = min_llk, lowest indices selected, a row of zero posteriors, and it adds nothing to any
statistic.  No out-of-range index, no fault, no hang (every case runs under pytest-timeout)."""
import numpy as np
import pytest

from conftest import make_frames, make_gmm
from oracle import oracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]

C, D, T = 256, 60, 900
BAD = {5: np.nan, 9: np.inf, 11: -np.inf, 13: 1e30, 300: np.nan, 899: np.inf}   # frame -> value put into one of its dimensions
FAR = 17                                                                         # a frame of finite, absurd values (kind 2)
PATHS = [
    {},                                            # defaults: fused top-C, stored-likelihood statistics, 8-wave shapes
    {"topc_fused": 0},                             # top-C from the stored likelihoods
    {"topc_fused": 0, "topc_z": 0},                # direct-form top-C kernel
    {"stats_z": 0},                                # recomputing statistics kernel
    {"wg_waves": 4},                               # 4-wave shapes of the two MFMA kernels
    {"short_calls": 0},
]


def _case():
    w, mean, iv = make_gmm(C, D, seed=1)
    x = make_frames(w, mean, iv, T, seed=2)
    for t, v in BAD.items():
        x[t, (7 * t) % D] = v
    x[FAR, :] = 3e17                                # finite and inside the screening bound: every logit is about -1e37
    zero = sorted(list(BAD) + [FAR])
    good = np.setdiff1d(np.arange(T), zero)
    return w, mean, iv, x, zero, good


def _ctx(opts):
    from lia_ral_amd import capi
    ctx = capi.Context(0)
    for k, v in opts.items():
        ctx.set_option(k, v)
    return ctx


@pytest.mark.parametrize("opts", PATHS, ids=lambda o: ",".join("%s=%s" % kv for kv in o.items()) or "default")
def test_zero_likelihood_frames_follow_the_rule_on_every_path(opts):
    w, mean, iv, x, zero, good = _case()
    ctx = _ctx(opts)
    g = ctx.gmm(w, mean, iv)
    og = orc.Gmm(w, mean, iv)
    xg = x[good].astype(np.float64)
    ctop = 10
    # --- gmmiv_llk: min_llk for the zero-likelihood frames, the oracle's values elsewhere; sums count every frame
    sums = np.zeros(2)
    l = g.llk(x, sums=sums)
    assert np.all(l[zero] == -200.0)
    assert np.max(np.abs(l[good] - orc.llk(og, xg))) < 1e-9
    assert sums[1] == T and abs(sums[0] - l.sum()) < 1e-6
    # --- DETERMINE_TOP_DISTRIBS
    d = g.llk_determine_top(x, ctop)
    assert d["idx"].min() >= 0 and d["idx"].max() < C
    do = orc.llk_determine_top(og, xg, ctop, True)
    assert np.array_equal(d["idx"][good], do["idx"]) and np.max(np.abs(d["llk"][good] - do["llk"])) < 1e-9
    for t in zero:
        assert d["idx"][t].tolist() == list(range(ctop)) and np.all(d["lk"][t] == 0.0)
        assert d["nontop_lk"][t] == 0.0 and d["nontop_llk"][t] == -np.inf and d["llk"][t] == -200.0
        assert abs(d["nontop_w"][t] - (1.0 - w[:ctop].sum())) < 1e-12
    # --- USE_TOP_DISTRIBS (one client, several clients)
    rng = np.random.default_rng(0)
    cl = [ctx.gmm(w, mean + rng.normal(0, 0.1, mean.shape), iv) for _ in range(3)]
    u = cl[0].llk_use_top(x, d["idx"], d["nontop_llk"])
    um = type(cl[0]).llk_use_top_multi(cl, x, d["idx"], d["nontop_llk"])
    assert np.all(u[zero] == -200.0) and np.all(um[:, zero] == -200.0) and np.isfinite(u).all() and np.isfinite(um).all()
    assert np.array_equal(um[0], u)
    # --- posteriors: zero rows; every other row sums to 1
    o = g.occ(x[:40])
    for t in range(40):
        if t in zero:
            assert np.all(o[t] == 0.0)
        else:
            assert abs(o[t].sum() - 1.0) < 1e-9
    # --- EM statistics: the zero-likelihood frames add nothing, anywhere
    a = g.split_acc(g.em_accumulate(x))
    ref = orc.em_accumulate(og, xg)
    rel = lambda p, q: np.max(np.abs(p - q)) / np.max(np.abs(q))
    assert a["count"] == len(good) and abs(a["occ"].sum() - len(good)) < 1e-6
    assert rel(a["occ"], ref["occ"]) < 1e-9 and rel(a["sx"], ref["sx"]) < 1e-9 and rel(a["sxx"], ref["sxx"]) < 1e-9
    assert abs(a["llk"] - ref["llk"]) < 1e-6 * abs(ref["llk"])
    # --- Baum-Welch N / F: utterance bounds in the middle of the bad frames, one empty utterance
    ub = np.array([0, 6, 6, 301, T])
    N = np.zeros((4, C)); F = np.zeros((4, C * D))
    g.tv_stats(x, ub, N, F)
    utt = np.searchsorted(ub, good, side="right") - 1
    No, Fo = orc.tv_stats(og, xg, utt, 4)
    assert np.isfinite(N).all() and np.isfinite(F).all()
    assert rel(N, No) < 1e-9 and rel(F, Fo) < 1e-9
    if not opts:
        assert ctx.set_option("screened_frames", 0) >= len(BAD)          # the screening saw them (the far frame is kind 2)
    for m in cl:
        m.close()
    g.close(); ctx.close()


def _logits(w, mean, iv, x):
    """log(w_c lk_c(x_t)) in the log domain (numpy): the reference of the boundary test -- the linear-domain oracle holds a
    likelihood of exp(-744) in one or two denormal bits and cannot serve there."""
    x = np.asarray(x, np.float64)
    a = np.log(w) + 0.5 * np.sum(np.log(iv), 1) - 0.5 * D * np.log(2 * np.pi)
    d = x[:, None, :] - mean[None]
    return a[None] - 0.5 * np.einsum("tcd,cd->tc", d * d, iv)


def _frame_at(w, mean, iv, target, seed):
    """A float32 frame whose LARGEST logit is `target` (+- 1e-2): mu_0 + s u, s by bisection."""
    u = np.random.default_rng(seed).normal(0, 1, D) / np.sqrt(iv[0])
    lo, hi = 0.0, 1e3
    for _ in range(200):
        s = 0.5 * (lo + hi)
        z = _logits(w, mean, iv, (mean[0] + s * u).astype(np.float32)[None]).max()
        lo, hi = (s, hi) if z > target else (lo, s)
    xf = (mean[0] + lo * u).astype(np.float32)
    assert abs(_logits(w, mean, iv, xf[None]).max() - target) < 1e-2
    return xf


@pytest.mark.parametrize("opts", PATHS, ids=lambda o: ",".join("%s=%s" % kv for kv in o.items()) or "default")
def test_kind_2_threshold_is_log_2_pow_minus_1075_on_every_path(opts):
    """include/gmmiv.h: a frame is a zero-likelihood frame of kind (2) when its largest w_c lk_c lies below 2^-1075 = exp(-745.13).
    One frame at -744 (ordinary: it counts everywhere, its values are the log-domain ones) and one at -746.5 (dropped, and COUNTED
    by "zero_llk_frames") through every statistics / likelihood entry point (the header said -6.9e5)."""
    w, mean, iv = make_gmm(C, D, seed=1)
    x = make_frames(w, mean, iv, 200, seed=3)
    A, B = 50, 120
    x[A] = _frame_at(w, mean, iv, -744.0, 11)
    x[B] = _frame_at(w, mean, iv, -746.5, 12)
    z = _logits(w, mean, iv, x)
    m = z.max(1, keepdims=True)
    lse = (m + np.log(np.exp(z - m).sum(1, keepdims=True)))[:, 0]
    live = np.setdiff1d(np.arange(200), [B])
    ctx = _ctx(opts)
    g = ctx.gmm(w, mean, iv)
    assert ctx.set_option("zero_llk_frames", 0) == 0
    # gmmiv_llk
    l = g.llk(x, min_llk=-1e4, max_llk=1e4)
    assert l[B] == -1e4 and np.max(np.abs(l[live] - lse[live])) < 1e-9 and abs(l[A] - lse[A]) < 1e-9 and lse[A] > -745.0
    assert ctx.set_option("zero_llk_frames", 0) == 1
    # posteriors
    o = g.occ(x)
    gam = np.exp(z - lse[:, None])
    assert np.all(o[B] == 0.0) and abs(o[A].sum() - 1.0) < 1e-9 and np.max(np.abs(o[live] - gam[live])) < 1e-9
    assert ctx.set_option("zero_llk_frames", 0) == 1
    # EM statistics: frame A is in (count, occupancy, sums), frame B adds nothing
    a = g.split_acc(g.em_accumulate(x))
    xl = x[live].astype(np.float64)
    rel = lambda p, q: np.max(np.abs(p - q)) / np.max(np.abs(q))
    assert a["count"] == 199 and abs(a["occ"].sum() - 199) < 1e-6
    assert rel(a["occ"], gam[live].sum(0)) < 1e-9 and rel(a["sx"], gam[live].T @ xl) < 1e-9 and rel(a["sxx"], gam[live].T @ (xl * xl)) < 1e-9
    assert abs(a["llk"] - lse[live].sum()) < 1e-6
    assert ctx.set_option("zero_llk_frames", 0) == 1
    # Baum-Welch N / F, A and B in different utterances
    ub = np.array([0, 100, 200])
    N = np.zeros((2, C)); F = np.zeros((2, C * D))
    g.tv_stats(x, ub, N, F)
    No = np.stack([gam[:100].sum(0), gam[live][live >= 100].sum(0)])
    Fo = np.stack([(gam[:100].T @ x[:100].astype(np.float64)).ravel(), (gam[live][live >= 100].T @ x[live][live >= 100].astype(np.float64)).ravel()])
    assert rel(N, No) < 1e-9 and rel(F, Fo) < 1e-9 and abs(N[0].sum() - 100) < 1e-6 and abs(N[1].sum() - 99) < 1e-6
    assert ctx.set_option("zero_llk_frames", 0) == 1
    # top-C: B gets the lowest indices and min_llk, A its true selection
    d = g.llk_determine_top(x, 5, min_llk=-1e4, max_llk=1e4)
    assert d["idx"][B].tolist() == [0, 1, 2, 3, 4] and d["llk"][B] == -1e4 and np.all(d["lk"][B] == 0.0)
    assert d["idx"][A].tolist() == np.argsort(-z[A], kind="stable")[:5].tolist() and abs(d["llk"][A] - lse[A]) < 1e-9
    assert ctx.set_option("screened_frames", 0) == 0
    g.close(); ctx.close()


def test_call_made_of_unusable_frames_only_and_empty_calls():
    from lia_ral_amd import capi
    w, mean, iv, x, zero, good = _case()
    ctx = capi.Context(0)
    g = ctx.gmm(w, mean, iv)
    xb = x[[5, 9, 11]]                               # every frame unusable
    assert np.all(g.llk(xb) == -200.0)
    d = g.llk_determine_top(xb, 4)
    assert np.array_equal(d["idx"], np.tile(np.arange(4), (3, 1))) and np.all(d["llk"] == -200.0)
    acc = g.em_accumulate(xb)
    assert np.all(acc == 0.0)
    N = np.ones((1, C)); F = np.ones((1, C * D))
    g.tv_stats(xb, np.array([0, 3]), N, F)
    assert np.all(N == 0.0) and np.all(F == 0.0)     # rows are overwritten: an utterance without a usable frame has empty statistics
    assert np.all(g.occ(xb) == 0.0)
    e = np.zeros((0, D), np.float32)                 # T = 0
    assert g.llk(e).shape == (0,) and g.llk_determine_top(e, 4)["idx"].shape == (0, 4) and g.occ(e).shape == (0, C)
    acc = np.full(g.em_acc_len(), 3.0)
    assert np.all(g.em_accumulate(e, acc=acc) == 3.0)                   # accumulators unchanged
    import ctypes as ct
    n = ct.c_int64(-1)
    capi._chk(capi.lib.gmmiv_count_unusable_frames(ctx._h, capi._ptr(x), capi.F32, ct.c_int64(T), ct.c_int64(D), D, ct.byref(n)))
    assert n.value == len(BAD)
    g.close(); ctx.close()


def test_assume_finite_skips_the_screening_and_clean_data_is_unchanged():
    """Clean features: identical results with and without the screening pass (bitwise), nothing counted as screened."""
    from lia_ral_amd import capi
    w, mean, iv = make_gmm(C, D, seed=1)
    x = make_frames(w, mean, iv, 3000, seed=4)
    res = []
    for af in (0, 1):
        ctx = capi.Context(0)
        ctx.set_option("assume_finite", af)
        g = ctx.gmm(w, mean, iv)
        res.append((g.llk(x), g.llk_determine_top(x, 10)["idx"], g.em_accumulate(x)))
        assert ctx.set_option("screened_frames", 0) == 0
        g.close(); ctx.close()
    assert all(np.array_equal(p, q) for p, q in zip(res[0], res[1]))


@pytest.mark.parametrize("opts", [{}, {"topc_fused": 0}, {"topc_fused": 0, "topc_z": 0}, {"wg_waves": 4}], ids=str)
def test_weight_zero_gaussian_ties_and_dead_components(opts):
    w, mean, iv = make_gmm(C, D, seed=1)
    xs = make_frames(w, mean, iv, 500, seed=3)
    ctx = _ctx(opts)
    # a Gaussian of weight 0: likelihood 0 -- never selected, occupancy 0, the M-step keeps its mean / covariance and gives weight 0
    w0 = w.copy(); w0[3] = 0.0; w0 /= w0.sum()
    g = ctx.gmm(w0, mean, iv)
    og = orc.Gmm(w0, mean, iv)
    l = g.llk(xs)
    assert np.max(np.abs(l - orc.llk(og, xs.astype(np.float64)))) < 1e-9
    d = g.llk_determine_top(xs, 10)
    assert not (d["idx"] == 3).any() and np.array_equal(d["idx"], orc.llk_determine_top(og, xs.astype(np.float64), 10, True)["idx"])
    acc = g.em_accumulate(xs)
    a = g.split_acc(acc)
    assert a["occ"][3] == 0.0 and np.all(a["sx"][3] == 0.0) and np.all(a["sxx"][3] == 0.0) and abs(a["occ"].sum() - 500) < 1e-9
    wm, mm, cc = g.em_get(acc, mean, 1.0 / iv)
    assert wm[3] == 0.0 and np.array_equal(mm[3], mean[3]) and np.array_equal(cc[3], (1.0 / iv)[3]) and abs(wm.sum() - 1.0) < 1e-12
    g.close()
    # identical Gaussians: every logit of a row is the same -- the lowest indices win, in order; posteriors uniform
    we = np.full(C, 1.0 / C); me = np.tile(mean[0], (C, 1)); ive = np.tile(iv[0], (C, 1))
    g = ctx.gmm(we, me, ive)
    d = g.llk_determine_top(xs, 10)
    assert np.array_equal(d["idx"], np.tile(np.arange(10), (500, 1))) and np.isfinite(d["llk"]).all()
    assert np.allclose(d["lk"], d["lk"][:, :1], rtol=1e-12)
    a = g.split_acc(g.em_accumulate(xs))
    assert np.allclose(a["occ"], 500.0 / C, rtol=1e-9)
    o = g.occ(xs[:8])
    assert np.allclose(o, 1.0 / C, rtol=1e-9)
    g.close(); ctx.close()


def test_host_layer_checks_a_feature_buffer_once_and_trains_past_bad_frames():
    """liagpu::FeatureBuffer counts the unusable frames once, at upload: the per-call screening of the C ABI is then off for a clean
    buffer and stays on for a dirty one.  TrainWorld on a stream with NaN / Inf frames inside the selected segment equals TrainWorld
    on the stream without them (those frames add nothing, not even to the frame count); the global mean / covariance of
    FrameAccGD is NOT screened -- a NaN goes into the sums like in the reference."""
    from lia_ral_amd import host_capi as h
    Cs, Ds, Ts = 16, 12, 4000
    w, mean, iv = make_gmm(Cs, Ds, seed=5)
    x = make_frames(w, mean, iv, Ts, seed=6)
    w0 = np.full(Cs, 1.0 / Cs); mean0 = mean + np.random.default_rng(0).normal(0, 0.3, mean.shape); cov0 = np.ones((Cs, Ds)) * 2.0
    # floors 0 / ceilings huge: varianceControl never clamps, so the (poisoned) global covariance of the dirty stream is not used
    kw = dict(nb_it=3, init_floor=0.0, final_floor=0.0, init_ceil=1e30, final_ceil=1e30)
    a = h.train_world(x, [0], [Ts], w0, mean0, cov0, **kw)
    xd = np.insert(x, [100, 100, 2500], 0.0, axis=0)                   # three extra frames ...
    xd[100, 3] = np.nan; xd[101, 0] = np.inf; xd[2502, 5] = -np.inf    # ... that are unusable
    b = h.train_world(xd, [0], [Ts + 3], w0, mean0, cov0, **kw)
    assert np.isfinite(a["global_cov"]).all() and not np.isfinite(b["global_cov"]).all()
    assert np.isfinite(b["mean"]).all() and np.isfinite(b["cov"]).all() and np.isfinite(b["llk"]).all()
    assert np.allclose(a["mean"], b["mean"], rtol=1e-9, atol=1e-11) and np.allclose(a["cov"], b["cov"], rtol=1e-9, atol=1e-11)
    assert np.allclose(a["w"], b["w"], rtol=1e-9) and np.allclose(a["llk"], b["llk"], rtol=1e-11)


def test_screening_follows_the_buffer_that_is_read_not_the_accumulators_server():
    """a TVAcc on server A (which owns only CLEAN buffers) handed the frames of a DIRTY buffer on server B must
    still screen them -- `assume_finite` is decided per call from the FeatureBuffer actually read (liagpu::FiniteScope), not from
    the buffers registered with the accumulator's context, and the context's own option is untouched afterwards."""
    from lia_ral_amd import host_capi as h
    Cs, Ds = 32, 20
    w, mean, iv = make_gmm(Cs, Ds, seed=7)
    own = make_frames(w, mean, iv, 300, seed=8)                  # clean, lives on the accumulator's server
    x = make_frames(w, mean, iv, 1200, seed=9)
    bad = [3, 400, 401, 1199]
    xd = x.copy()
    xd[3, 0] = np.nan; xd[400, 5] = np.inf; xd[401, 7] = -np.inf; xd[1199, 2] = 1e30
    ub = np.array([0, 300, 300, 800, 1200])
    N, F, cnt, af = h.tv_stats_cross_server(own, xd, ub, (w, mean, 1.0 / iv))
    assert cnt == (0, len(bad)) and af == 0
    assert np.isfinite(N).all() and np.isfinite(F).all()         # before the fix: 0 x NaN poisoned F
    good = np.setdiff1d(np.arange(1200), bad)
    utt = np.searchsorted(ub, good, side="right") - 1
    No, Fo = orc.tv_stats(orc.Gmm(w, mean, iv), x[good].astype(np.float64), utt, 4)
    rel = lambda p, q: np.max(np.abs(p - q)) / np.max(np.abs(q))
    assert rel(N, No) < 1e-9 and rel(F, Fo) < 1e-9
    # and a clean foreign buffer gives the plain statistics (the screening pass is skipped for it)
    N2, F2, cnt2, af2 = h.tv_stats_cross_server(own, x, ub, (w, mean, 1.0 / iv))
    utt2 = np.searchsorted(ub, np.arange(1200), side="right") - 1
    No2, Fo2 = orc.tv_stats(orc.Gmm(w, mean, iv), x.astype(np.float64), utt2, 4)
    assert cnt2 == (0, 0) and af2 == 0 and rel(N2, No2) < 1e-9 and rel(F2, Fo2) < 1e-9


def capi_determine_too_many(g, x):
    """gmmiv_llk_determine_top with topDistribsCount > mixtureDistribCount through the raw ABI (the Python wrapper clamps): ERR_ARG"""
    import ctypes as ct
    from lia_ral_amd import capi
    ctop = g.C + 1
    idx = np.empty((len(x), ctop), np.int32)
    capi._chk(capi.lib.gmmiv_llk_determine_top(g.ctx._h, g._h, capi._ptr(x), capi.F32, ct.c_int64(len(x)), ct.c_int64(x.shape[1]), ctop, 1,
                                               ct.c_double(-200.0), ct.c_double(200.0), capi._ptr(idx), None, None, None, None, None))


def test_generic_paths_follow_the_same_rule():
    """vectSize > 80 (no MFMA instantiation: VALU logits, statistics on the fp64 GEMM), topDistribsCount > 64 (the any-shape selection
    kernel): the degenerate-input rule holds there too -- unusable frames are screened out, a far frame is a zero-likelihood frame
    (counted), everything else equals the oracle."""
    Cg, Dg, Tg = 96, 100, 310
    w, mean, iv = make_gmm(Cg, Dg, seed=31, spread=0.5)
    x = make_frames(w, mean, iv, Tg, seed=32)
    bad = [4, 150, 309]
    x[4, 3] = np.nan; x[150, 99] = np.inf; x[309, 50] = -1e25
    x[77, :] = 2e17                                   # kind 2
    zero = sorted(bad + [77])
    good = np.setdiff1d(np.arange(Tg), zero)
    ctx = _ctx({})
    g = ctx.gmm(w, mean, iv)
    og = orc.Gmm(w, mean, iv)
    xg = x[good].astype(np.float64)
    l = g.llk(x)
    assert np.all(l[zero] == -200.0) and np.max(np.abs(l[good] - orc.llk(og, xg, -200.0, 200.0))) < 1e-9
    for ctop in (7, 80):                               # 80 > 64: k_topc_determine_big for the 60-dim case below too
        d = g.llk_determine_top(x, ctop)
        do = orc.llk_determine_top(og, xg, ctop, True)
        assert np.array_equal(d["idx"][good], do["idx"]) and np.max(np.abs(d["llk"][good] - do["llk"])) < 1e-9
        for t in zero:
            assert d["idx"][t].tolist() == list(range(ctop)) and np.all(d["lk"][t] == 0.0) and d["llk"][t] == -200.0
        u = g.llk_use_top(x, d["idx"], d["nontop_llk"])
        uo = orc.llk_use_top(og, xg, do["idx"], do["nontop_lk"], True)
        assert np.all(u[zero] == -200.0) and np.max(np.abs(u[good] - uo)) < 1e-9
    a = g.split_acc(g.em_accumulate(x))
    ref = orc.em_accumulate(og, xg)
    rel = lambda p, q: np.max(np.abs(p - q)) / np.max(np.abs(q))
    assert a["count"] == len(good) and rel(a["occ"], ref["occ"]) < 1e-9 and rel(a["sx"], ref["sx"]) < 1e-9 and rel(a["sxx"], ref["sxx"]) < 1e-9
    ub = np.array([0, 5, 5, 200, Tg])
    N = np.zeros((4, Cg)); F = np.zeros((4, Cg * Dg))
    g.tv_stats(x, ub, N, F)
    utt = np.searchsorted(ub, good, side="right") - 1
    No, Fo = orc.tv_stats(og, xg, utt, 4)
    assert rel(N, No) < 1e-9 and rel(F, Fo) < 1e-9 and not N[1].any()
    o = g.occ(x[:100])
    assert np.all(o[4] == 0.0) and np.all(o[77] == 0.0) and abs(o[5].sum() - 1.0) < 1e-9
    assert ctx.set_option("screened_frames", 0) >= len(bad) and ctx.set_option("zero_llk_frames", 0) >= 1
    g.close()
    # a Gaussian of weight 0 on the generic paths (a logit that is -inf or NaN is a term of likelihood 0 for the selection
    # kernel AND for the posteriors that feed the statistics GEMM -- nothing of it may reach S): occupancy 0, finite everywhere
    w0 = w.copy(); w0[3] = 0.0; w0 /= w0.sum()
    g0 = ctx.gmm(w0, mean, iv)
    a0 = g0.split_acc(g0.em_accumulate(x))
    ref0 = orc.em_accumulate(orc.Gmm(w0, mean, iv), xg)
    assert a0["occ"][3] == 0.0 and not a0["sx"][3].any() and not a0["sxx"][3].any()
    assert np.isfinite(a0["occ"]).all() and np.isfinite(a0["sx"]).all() and np.isfinite(a0["sxx"]).all()
    assert rel(a0["occ"], ref0["occ"]) < 1e-9 and rel(a0["sx"], ref0["sx"]) < 1e-9 and rel(a0["sxx"], ref0["sxx"]) < 1e-9
    o0 = g0.occ(x[:20])
    assert np.all(o0[:, 3] == 0.0) and np.isfinite(o0).all()
    d0 = g0.llk_determine_top(x, 80)
    assert np.array_equal(d0["idx"][good], orc.llk_determine_top(orc.Gmm(w0, mean, iv), xg, 80, True)["idx"])
    with pytest.raises(Exception, match="topDistribsCount|exceeds"):
        capi_determine_too_many(g0, x)
    g0.close()
    # topDistribsCount > 64 on an MFMA-served model (60 dims): the selection falls through to the any-shape kernel
    w6, m6, iv6 = make_gmm(200, 60, seed=33, spread=0.5)
    x6 = make_frames(w6, m6, iv6, 90, seed=34)
    x6[10, 0] = np.nan
    g6 = ctx.gmm(w6, m6, iv6)
    d = g6.llk_determine_top(x6, 70)
    keep = np.setdiff1d(np.arange(90), [10])
    do = orc.llk_determine_top(orc.Gmm(w6, m6, iv6), x6[keep].astype(np.float64), 70, True)
    assert np.array_equal(d["idx"][keep], do["idx"]) and d["idx"][10].tolist() == list(range(70)) and d["llk"][10] == -200.0
    g6.close(); ctx.close()


def test_calls_on_device_pointers_only_enqueue_with_the_screening_on():
    """include/gmmiv.h, DEGENERATE INPUTS: unusable frames are handled inside the kernels and only COUNTED by the screening pass, on the
    device -- so a frame-consuming call whose arrays are all device pointers returns before the stream has drained, with `assume_finite`
    at its default 0 and unusable frames in the call.  The stream is kept busy by a one-second spin kernel enqueued first: the calls
    behind it must come back at once (they waited for the stream once per call in rounds 1-5), and their results -- read after the
    synchronisation -- follow the rule."""
    import ctypes as ct
    import time
    import torch
    from lia_ral_amd import capi
    w, mean, iv, x, zero, good = _case()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx = capi.Context(0, s.cuda_stream)
        assert ctx.set_option("assume_finite", 0) == 0
        g = ctx.gmm(w, mean, iv)
        xd = torch.from_numpy(x).cuda()
        acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device="cuda")
        llk = torch.empty(T, dtype=torch.float64, device="cuda"); sums = torch.zeros(2, dtype=torch.float64, device="cuda")
        gam = torch.empty((T, C), dtype=torch.float64, device="cuda")
        g.em_accumulate(xd, acc=acc); g.llk(xd, out=llk, sums=sums); capi._chk(capi.lib.gmmiv_occ(ctx._h, g._h, capi._ptr(xd), capi.F32, ct.c_int64(T), ct.c_int64(D), capi._ptr(gam)))
        torch.cuda.synchronize()                                  # warm-up: the workspaces exist now (hipMalloc synchronises)
        acc.zero_(); sums.zero_()
        ctx.set_option("screened_frames", 0); ctx.set_option("zero_llk_frames", 0)
        torch.cuda._sleep(int(2.0e9))                             # ~1 s of spinning at ~2 GHz, on this stream
        t0 = time.perf_counter()
        g.em_accumulate(xd, acc=acc)
        g.llk(xd, out=llk, sums=sums)
        capi._chk(capi.lib.gmmiv_occ(ctx._h, g._h, capi._ptr(xd), capi.F32, ct.c_int64(T), ct.c_int64(D), capi._ptr(gam)))
        dt = time.perf_counter() - t0
        still_busy = not s.query()
        torch.cuda.synchronize()
        assert still_busy and dt < 0.25, (still_busy, dt)         # three calls enqueued while the spin kernel was still running
        a = g.split_acc(acc.cpu().numpy())
        ref = orc.em_accumulate(orc.Gmm(w, mean, iv), x[good].astype(np.float64))
        rel = lambda p, q: np.max(np.abs(p - q)) / np.max(np.abs(q))
        assert a["count"] == len(good) and rel(a["occ"], ref["occ"]) < 1e-9 and rel(a["sx"], ref["sx"]) < 1e-9 and rel(a["sxx"], ref["sxx"]) < 1e-9
        l = llk.cpu().numpy()
        assert np.all(l[zero] == -200.0) and sums.cpu().numpy()[1] == T
        o = gam.cpu().numpy()
        assert np.all(o[zero] == 0.0) and np.max(np.abs(o[good].sum(1) - 1.0)) < 1e-9
        assert ctx.set_option("screened_frames", 0) == 3 * len(BAD)          # counted by each of the three calls, on the device
        assert ctx.set_option("zero_llk_frames", 0) == 3 * len(zero)         # kind (1) frames are evaluated as kind (2) frames: counted here too
        g.close(); ctx.close()
