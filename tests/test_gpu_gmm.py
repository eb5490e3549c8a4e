"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerances (fp64 path; the reference itself is built -ffast-math so relative, not bitwise):
  per-frame llk            abs 1e-9   (|llk| ~ 1e2 -> ~1e-11 relative)
  posterior statistics     1e-9 relative to the largest magnitude of the compared array
  top-C indices            exact
"""
import os

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


CASES = [(8, 60, 1), (8, 60, 64), (128, 60, 1000), (2048, 60, 300), (37, 13, 257), (1024, 32, 50), (300, 75, 130),
         (256, 128, 203), (40, 97, 66),      # vectSize > 80: no MFMA instantiation, the generic paths (VALU logits, statistics on the fp64 GEMM)
         # the smallest shapes (round 6: EnergyDetector's model is 2 Gaussians x ONE dimension -- every statistic of a vectSize-1 model was
         # garbage, the frame-staging plan divided by vectSize with a reciprocal that does not exist for 1): vectSize 1 / 2 / 3, 1 - 3 Gaussians
         (2, 1, 26), (1, 1, 10), (3, 1, 50), (16, 1, 1000), (2, 2, 26), (1, 3, 10), (5, 2, 33), (2048, 1, 300)]


@pytest.mark.parametrize("C,D,T", CASES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_llk_matches_oracle(ctx, C, D, T, dtype):
    w, mean, iv = make_gmm(C, D, seed=C + D)
    x = make_frames(w, mean, iv, T, seed=T, dtype=dtype)
    g = ctx.gmm(w, mean, iv)
    sums = np.zeros(2)
    got = g.llk(x, -1e9, 1e9, sums=sums)
    ref = orc.llk(orc.Gmm(w, mean, iv), x.astype(np.float64), -1e9, 1e9)
    assert np.max(np.abs(got - ref)) < 1e-9
    assert sums[1] == T and abs(sums[0] - ref.sum()) < 1e-8 * T
    # both staging paths of the LLK kernel (LDS-DMA and through registers) give the same bits
    ctx.set_option("glds", 0)
    got2 = g.llk(x, -1e9, 1e9)
    ctx.set_option("glds", 1)
    assert np.array_equal(got, got2)


@pytest.mark.parametrize("spread", [0.05, 0.3, 1.0])
def test_overlapping_mixture_dense_paths(ctx, spread):
    """Heavily overlapping Gaussians: every logit matters (no exp can be skipped), posteriors are
    spread over hundreds of components -- exercises the dense branches of both MFMA kernels."""
    C, D, T = 512, 60, 700
    w, mean, iv = make_gmm(C, D, seed=11, spread=spread)
    x = make_frames(w, mean, iv, T, seed=12)
    g = ctx.gmm(w, mean, iv)
    og = orc.Gmm(w, mean, iv)
    got = g.llk(x, -1e9, 1e9)
    ref = orc.llk(og, x.astype(np.float64), -1e9, 1e9)
    assert np.max(np.abs(got - ref)) < 1e-9
    a = g.split_acc(g.em_accumulate(x))
    r = orc.em_accumulate(og, x.astype(np.float64))
    assert relerr(a["occ"], r["occ"]) < 1e-9 and relerr(a["sx"], r["sx"]) < 1e-9 and relerr(a["sxx"], r["sxx"]) < 1e-9


def test_llk_clamp_and_device_pointers(ctx):
    import torch
    w, mean, iv = make_gmm(128, 60, seed=3)
    x = make_frames(w, mean, iv, 500, seed=4)
    x[7] += 40.0   # an outlier far from every Gaussian: llk << -200 -> clamped like minLLK
    g = ctx.gmm(w, mean, iv)
    ref = orc.llk(orc.Gmm(w, mean, iv), x.astype(np.float64), -200.0, 200.0)
    got = g.llk(x, -200.0, 200.0)
    assert got[7] == -200.0 == ref[7]
    assert np.max(np.abs(got - ref)) < 1e-9
    xd = torch.from_numpy(x).cuda()
    out = torch.empty(500, dtype=torch.float64, device="cuda")
    g.llk(xd, -200.0, 200.0, out=out)
    torch.cuda.synchronize(); ctx.sync()
    assert np.array_equal(out.cpu().numpy(), got)


@pytest.mark.parametrize("C,D,T,ctop", [(128, 60, 1000, 10), (2048, 60, 64, 10), (1024, 32, 50, 10), (8, 60, 33, 20), (300, 13, 100, 5),
                                        (2, 1, 26, 2), (2, 1, 26, 1), (1, 1, 9, 1), (16, 1, 100, 5), (3, 2, 40, 3),
                                        # shapes the LDS / MFMA selection kernels do not serve (round-4 verdict: free keys of the reference):
                                        (256, 128, 70, 10), (8192, 60, 41, 10), (512, 60, 37, 100), (8192, 60, 13, 100), (200, 128, 21, 70)])
@pytest.mark.parametrize("complete", [True, False])
def test_top_c_matches_oracle(ctx, C, D, T, ctop, complete):
    # (vectSize > 80: Gaussians 2 sigma apart in 128 dimensions leave a frame ONE Gaussian above exp(-745) -- the oracle's linear-domain
    #  likelihoods of all others are 0 and tie, lowest index first, where libgmmiv ranks their logits ("linear-domain
    #  underflow"): the parity cases stay away from that floor)
    w, mean, iv = make_gmm(C, D, seed=C, spread=0.5 if D > 80 else 2.0)
    x = make_frames(w, mean, iv, T, seed=T + 1)
    wc, meanc, ivc = w, mean + np.random.default_rng(5).normal(0, 0.1, mean.shape), iv   # client: shifted means
    world, client = ctx.gmm(w, mean, iv), ctx.gmm(wc, meanc, ivc)
    d = world.llk_determine_top(x, ctop, complete)
    xo = x.astype(np.float64)
    do = orc.llk_determine_top(orc.Gmm(w, mean, iv), xo, ctop, complete)
    assert np.array_equal(d["idx"], do["idx"])
    assert relerr(d["lk"], do["lk"]) < 1e-10
    assert np.max(np.abs(d["llk"] - do["llk"])) < 1e-9
    assert np.max(np.abs(d["nontop_w"] - do["nontop_w"])) < 1e-12
    big = do["nontop_lk"] > 1e-250
    assert relerr(d["nontop_lk"][big], do["nontop_lk"][big]) < 1e-9 if big.any() else True
    lc = client.llk_use_top(x, d["idx"], d["nontop_llk"], complete)
    lco = orc.llk_use_top(orc.Gmm(wc, meanc, ivc), xo, do["idx"], do["nontop_lk"], complete)
    assert np.max(np.abs(lc - lco)) < 1e-9


@pytest.mark.parametrize("C,D,T,ctop,dtype", [(64, 60, 1001, 10, np.float32), (64, 62, 37, 16, np.float64), (16, 2, 9, 3, np.float32),
                                              (256, 64, 130, 1, np.float32), (32, 34, 6, 7, np.float64)])
def test_use_top_lane_layouts_agree(ctx, C, D, T, ctop, dtype):
    """USE_TOP_DISTRIBS: k_topc_use4 (four lanes per candidate, one frame per wave; the default) and k_topc_use16 (one lane, four
    frames) against each other and the oracle -- dimension counts with a masked last batch, frame counts that leave waves without
    a frame, and index vectors with entries outside the model (skipped, never dereferenced, by both)."""
    w, mean, iv = make_gmm(C, D, seed=C + D)
    x = make_frames(w, mean, iv, T, seed=T).astype(dtype)
    world = ctx.gmm(w, mean, iv)
    client = ctx.gmm(w, mean + np.random.default_rng(7).normal(0, 0.2, mean.shape), iv * 1.1)
    d = world.llk_determine_top(x, ctop, True)
    res = {}
    for lanes in (4, 1):
        ctx.set_option("topc_use_lanes", lanes)
        res[lanes] = client.llk_use_top(x, d["idx"], d["nontop_llk"], True)
    ctx.set_option("topc_use_lanes", 4)
    assert np.max(np.abs(res[4] - res[1])) < 1e-12
    do = orc.llk_determine_top(orc.Gmm(w, mean, iv), x.astype(np.float64), ctop, True)
    oc = orc.Gmm(w, mean + np.random.default_rng(7).normal(0, 0.2, mean.shape), iv * 1.1)
    assert np.max(np.abs(res[4] - orc.llk_use_top(oc, x.astype(np.float64), do["idx"], do["nontop_lk"], True))) < 1e-9
    if ctop > 1:
        bad = d["idx"].copy()
        bad[::3, 0] = -1
        bad[1::3, ctop - 1] = C + 5
        out = {}
        for lanes in (4, 1):
            ctx.set_option("topc_use_lanes", lanes)
            out[lanes] = client.llk_use_top(x, bad, d["nontop_llk"], False)
        ctx.set_option("topc_use_lanes", 4)
        assert np.isfinite(out[4]).all() and np.max(np.abs(out[4] - out[1])) < 1e-12


@pytest.mark.parametrize("C,D,T,ctop,n_clients", [(256, 60, 3001, 10, 7), (64, 13, 50, 5, 3), (64, 60, 40, 20, 2), (128, 32, 1, 16, 1)])
def test_use_top_for_several_clients_in_one_call(ctx, C, D, T, ctop, n_clients):
    """gmmiv_llk_use_top_multi (ComputeTest's client loop as one call) returns, row by row, exactly what gmmiv_llk_use_top returns
    for each client -- on the batched kernel (ctop <= 16, even D) and on the client-by-client fallback (odd D, ctop > 16), COMPLETE
    and PARTIAL; a client with another dimension count is refused."""
    from lia_ral_amd import capi
    w, mean, iv = make_gmm(C, D, seed=3 * C + D)
    x = make_frames(w, mean, iv, T, seed=T + 9).astype(np.float32)
    world = ctx.gmm(w, mean, iv)
    rng = np.random.default_rng(11)
    clients = [ctx.gmm(w, mean + rng.normal(0, 0.15, mean.shape), iv * (1.0 + 0.05 * i)) for i in range(n_clients)]
    for complete in (True, False):
        d = world.llk_determine_top(x, ctop, complete)
        one = np.stack([g.llk_use_top(x, d["idx"], d["nontop_llk"], complete) for g in clients])
        many = capi.Gmm.llk_use_top_multi(clients, x, d["idx"], d["nontop_llk"], complete)
        assert many.shape == (n_clients, T) and np.array_equal(many, one)
    if n_clients > 1 and D > 2:
        w2, m2, iv2 = make_gmm(C, D - 1, seed=1)
        with pytest.raises(capi.GmmivError):
            capi.Gmm.llk_use_top_multi([clients[0], ctx.gmm(w2, m2, iv2)], x, d["idx"], d["nontop_llk"], True)


@pytest.mark.parametrize("T", [1, 129, 3000, 32768])
def test_short_calls_run_other_workgroup_shapes_with_the_same_results(ctx, T):
    """Calls of at most 32 768 frames run the log-likelihood kernels on 4-wave workgroups (option short_calls, default 1); with the
    option off they run the 8-wave shape of long calls.  Per-frame values, EM accumulators and top-C selections are bitwise the
    same."""
    C, D, ctop = 256, 60, 10
    w, mean, iv = make_gmm(C, D, seed=77)
    x = make_frames(w, mean, iv, T, seed=T + 3).astype(np.float32)
    g = ctx.gmm(w, mean, iv)
    res = {}
    for short in (1, 0):
        ctx.set_option("short_calls", short)
        d = g.llk_determine_top(x, ctop, True)
        res[short] = (g.llk(x, -1e9, 1e9), g.em_accumulate(x), d["idx"], d["llk"], d["nontop_llk"], d["lk"])
    ctx.set_option("short_calls", 1)
    for a, b in zip(res[1], res[0]):
        assert np.array_equal(a, b)


def test_kat1_on_gpu(ctx, golden_dir):
    """ComputeTest golden LLRs (test1.validate.res) through the HIP path."""
    k = np.load(os.path.join(golden_dir, "kat1_computetest.npz"))
    world = ctx.gmm(k["w"], k["mean_world"], k["covinv"])
    client = ctx.gmm(k["w_client"], k["mean_client"], k["covinv_client"])
    got = []
    for b, n in zip(k["seg_begin"], k["seg_len"]):
        xs = k["x"][b:b + n]
        d = world.llk_determine_top(xs, int(k["top_c"]), True)
        lc = client.llk_use_top(xs, d["idx"], d["nontop_llk"], True)
        got.append(lc.mean() - d["llk"].mean())
    assert np.allclose(got, k["expected_llr"], atol=float(k["abs_tol"]), rtol=0), got


@pytest.mark.parametrize("C,D,T", CASES)
def test_em_stats_match_oracle(ctx, C, D, T):
    w, mean, iv = make_gmm(C, D, seed=C + 1)
    x = make_frames(w, mean, iv, T, seed=T + 2)
    g = ctx.gmm(w, mean, iv)
    acc = g.em_accumulate(x)
    a = g.split_acc(acc)
    ref = orc.em_accumulate(orc.Gmm(w, mean, iv), x.astype(np.float64))
    assert a["count"] == T
    assert abs(a["llk"] - ref["llk"]) < 1e-9 * max(1, T)
    assert abs(a["occ"].sum() - T) < 1e-9 * T
    assert relerr(a["occ"], ref["occ"]) < 1e-9
    assert relerr(a["sx"], ref["sx"]) < 1e-9
    assert relerr(a["sxx"], ref["sxx"]) < 1e-9


def test_em_accumulates_weights_and_chunks(ctx):
    w, mean, iv = make_gmm(256, 60, seed=9)
    x = make_frames(w, mean, iv, 9000, seed=10)
    g = ctx.gmm(w, mean, iv)
    og = orc.Gmm(w, mean, iv)
    acc = g.em_accumulate(x[:5000])
    acc = g.em_accumulate(x[5000:], weight=0.5, acc=acc)       # weighted EM (AccumulateStat.cpp:143-152)
    ref = orc.em_accumulate(og, x[:5000].astype(np.float64))
    ref2 = orc.em_accumulate(og, x[5000:].astype(np.float64), weight=0.5)
    a = g.split_acc(acc)
    assert abs(a["count"] - (5000 + 0.5 * 4000)) < 1e-9
    assert relerr(a["sx"], ref["sx"] + ref2["sx"]) < 1e-9
    assert abs(a["llk"] - (ref["llk"] + 0.5 * ref2["llk"])) < 1e-6
    # different chunking of the frame stream -> same sums up to rounding
    ctx.set_option("em_chunks", 8)
    acc8 = g.em_accumulate(x)
    ctx.set_option("em_chunks", 1)
    acc1 = g.em_accumulate(x)
    ctx.set_option("em_chunks", 0)
    assert relerr(acc8, acc1) < 1e-12
    # M-step on the device copy == oracle getEM
    wn, mn, cn = g.em_get(acc1, mean, 1.0 / iv)
    refa = orc.em_accumulate(og, x.astype(np.float64))
    wo, mo, co = orc.em_get(refa, mean, 1.0 / iv)
    assert relerr(wn, wo) < 1e-9 and relerr(mn, mo) < 1e-9 and relerr(cn, co) < 1e-8


def test_workgroup_shapes_agree(ctx):
    """4-wave and 8-wave workgroup variants of the two MFMA kernels compute the same sums."""
    w, mean, iv = make_gmm(2048, 60, seed=21)
    x = make_frames(w, mean, iv, 3000, seed=22)
    g = ctx.gmm(w, mean, iv)
    res = {}
    for nw in (4, 8):
        ctx.set_option("wg_waves", nw)
        res[nw] = (g.llk(x, -1e9, 1e9), g.em_accumulate(x))
    ctx.set_option("wg_waves", 8)
    assert np.array_equal(res[4][0], res[8][0])
    assert relerr(res[4][1], res[8][1]) < 1e-12


@pytest.mark.parametrize("spread", [2.0, 0.3])
def test_posterior_pruning_is_invisible(ctx, spread):
    """Opt-in pruning: groups whose posteriors are all below 2^-100 are skipped by the statistics
    kernels; against the default (every pair accumulated) the sums agree to rounding."""
    w, mean, iv = make_gmm(1024, 60, seed=31, spread=spread)
    x = make_frames(w, mean, iv, 5000, seed=32)
    g = ctx.gmm(w, mean, iv)
    ub = np.array([0, 1234, 5000])
    a_off = g.em_accumulate(x)
    n_off, f_off = g.tv_stats(x, ub)
    ctx.set_option("prune_log2", 100)
    a_on = g.em_accumulate(x)
    n_on, f_on = g.tv_stats(x, ub)
    ctx.set_option("prune_log2", 0)
    assert relerr(a_on, a_off) < 1e-14 and relerr(n_on, n_off) < 1e-14 and relerr(f_on, f_off) < 1e-14


def test_em_zero_frames_and_ragged_edges(ctx):
    w, mean, iv = make_gmm(64, 60, seed=2)
    g = ctx.gmm(w, mean, iv)
    acc = g.em_accumulate(np.zeros((0, 60), np.float32))
    assert not acc.any()
    for T in (1, 63, 64, 65, 255, 256, 257):
        x = make_frames(w, mean, iv, T, seed=T)
        a = g.split_acc(g.em_accumulate(x))
        ref = orc.em_accumulate(orc.Gmm(w, mean, iv), x.astype(np.float64))
        assert relerr(a["sxx"], ref["sxx"]) < 1e-9, T


@pytest.mark.parametrize("C,D", [(128, 60), (2048, 60), (37, 13), (32, 20), (96, 60),   # 8, 128, 4 (padded), 2 and 6 Gaussian tiles: both wave shapes
                                 (96, 128), (33, 101),                                     # vectSize > 80: the generic path (gamma^T [x | 1] per utterance)
                                 (2, 1), (1, 1), (16, 1), (2, 2), (3, 3)])                 # the smallest models (vectSize 1: see CASES)
def test_tv_stats_match_oracle(ctx, C, D):
    w, mean, iv = make_gmm(C, D, seed=C + 7)
    lens = [70, 0, 131, 64, 1]
    ub = np.concatenate([[0], np.cumsum(lens)])
    x = make_frames(w, mean, iv, int(ub[-1]), seed=11)
    g = ctx.gmm(w, mean, iv)
    N, F = g.tv_stats(x, ub)
    utt = np.repeat(np.arange(len(lens)), lens)
    No, Fo = orc.tv_stats(orc.Gmm(w, mean, iv), x.astype(np.float64), utt, len(lens))
    assert relerr(N, No) < 1e-9 and relerr(F, Fo) < 1e-9
    assert not N[1].any() and not F[1].any()


@pytest.mark.parametrize("lens", [[3000], [1], [70, 0, 5000], [257] * 16, [64, 65, 1000, 333, 2], [40000]])
def test_tv_stats_of_a_few_utterances_in_pieces_match_one_segment_per_utterance(ctx, lens):
    """gmmiv_tv_stats on at most 16 utterances cuts them into pieces of whole tiles (more workgroups; option tv_stats_split) and
    sums the pieces back: N / F against the one-segment-per-utterance form (1e-13: another summation order) and the oracle --
    one utterance, an empty one in the middle, lengths around the tile and piece sizes."""
    C, D = 128, 60
    w, mean, iv = make_gmm(C, D, seed=len(lens) + sum(lens) % 97)
    T = int(sum(lens))
    x = make_frames(w, mean, iv, max(T, 1), seed=5)[:T].astype(np.float32)
    ub = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    g = ctx.gmm(w, mean, iv)
    out = {}
    for split in (1, 0):
        ctx.set_option("tv_stats_split", split)
        out[split] = g.tv_stats(x, ub)
    ctx.set_option("tv_stats_split", 1)
    assert relerr(out[1][0], out[0][0]) < 1e-13 and relerr(out[1][1], out[0][1]) < 1e-13
    utt = np.repeat(np.arange(len(lens)), lens)
    No, Fo = orc.tv_stats(orc.Gmm(w, mean, iv), x.astype(np.float64), utt, len(lens))
    assert relerr(out[1][0], No) < 1e-9 and relerr(out[1][1], Fo) < 1e-9
    for u, n in enumerate(lens):
        if n == 0:
            assert not out[1][0][u].any() and not out[1][1][u].any()


@pytest.mark.parametrize("C,D,T,mb", [(128, 60, 20000, 8), (2048, 60, 9000, 100), (37, 13, 7001, 4), (300, 24, 15000, 16)])
def test_stored_logit_path_chunked_matches_recompute(ctx, C, D, T, mb):
    """Default EM path (k_llk_mfma<WZ> + k_stats_z, frames in chunks that fit the logit scratch) against
    the recomputing kernel (stats_z = 0) and the oracle."""
    w, mean, iv = make_gmm(C, D, seed=C + 21)
    x = make_frames(w, mean, iv, T, seed=T + 1)
    g = ctx.gmm(w, mean, iv)
    ctx.set_option("stats_z", 0)
    try:
        ref = g.em_accumulate(x, weight=0.5)
    finally:
        ctx.set_option("stats_z", 1)
    one = g.em_accumulate(x, weight=0.5)
    prev = ctx.set_option("z_scratch_mb", mb)
    try:
        ctx.set_option("timing", 1)
        got = g.em_accumulate(x, weight=0.5)
        nl = ctx.kernel_launches("k_stats_z")
        ctx.set_option("timing", 0)
    finally:
        ctx.set_option("z_scratch_mb", prev)
    assert nl >= 2, nl                      # really chunked
    assert relerr(one, ref) < 1e-12 and relerr(got, ref) < 1e-12
    a = g.split_acc(got)
    o = orc.em_accumulate(orc.Gmm(w, mean, iv), x.astype(np.float64))
    assert relerr(a["occ"], 0.5 * o["occ"]) < 1e-9 and relerr(a["sxx"], 0.5 * o["sxx"]) < 1e-9


@pytest.mark.parametrize("C,D,U,mb", [(128, 60, 30, 8), (2048, 60, 12, 100), (37, 13, 20, 4)])
def test_tv_stats_stored_logit_path_chunks_of_utterances(ctx, C, D, U, mb):
    rng = np.random.default_rng(U + C)
    lens = rng.integers(0, 1500, U)
    lens[2] = 0
    lens[-1] = 1
    ub = np.concatenate([[0], np.cumsum(lens)])
    w, mean, iv = make_gmm(C, D, seed=C + 17)
    x = make_frames(w, mean, iv, int(ub[-1]), seed=3)
    g = ctx.gmm(w, mean, iv)
    ctx.set_option("stats_z", 0)
    try:
        N0, F0 = g.tv_stats(x, ub)
    finally:
        ctx.set_option("stats_z", 1)
    N1, F1 = g.tv_stats(x, ub)
    prev = ctx.set_option("z_scratch_mb", mb)
    try:
        ctx.set_option("timing", 1)
        N2, F2 = g.tv_stats(x, ub)
        nl = ctx.kernel_launches("k_stats_z")
        ctx.set_option("timing", 0)
    finally:
        ctx.set_option("z_scratch_mb", prev)
    assert nl >= 2, nl
    for N, F in ((N1, F1), (N2, F2)):
        assert relerr(N, N0) < 1e-12 and relerr(F, F0) < 1e-12
        assert not N[2].any() and not F[2].any()
        assert np.allclose(N.sum(1), lens, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("C,D,T", [(128, 60, 300), (37, 13, 65), (2048, 60, 40), (64, 100, 33)])
def test_posterior_vectors_match_oracle(ctx, C, D, T):
    """gmmiv_occ = computeAndAccumulateOcc + getOccVect: gamma[t][c], rows sum to 1."""
    w, mean, iv = make_gmm(C, D, seed=C + 31)
    x = make_frames(w, mean, iv, T, seed=T + 2)
    g = ctx.gmm(w, mean, iv)
    got = g.occ(x)
    ref = orc.occ(orc.Gmm(w, mean, iv), x.astype(np.float64))
    assert np.max(np.abs(got - ref)) < 1e-12
    assert np.allclose(got.sum(1), 1.0, atol=1e-12)
    assert g.occ(np.zeros((0, D), np.float32)).shape == (0, C)


@pytest.mark.parametrize("C,D", [(128, 60), (2048, 60), (37, 13)])
def test_row_strided_device_features(ctx, C, D):
    """Frames as rows of a wider device matrix (ldx > D): every entry point that takes features, on the default
    (stored-likelihood) path and on the recomputing one."""
    import torch
    w, mean, iv = make_gmm(C, D, seed=C + 41)
    T = 1500
    x = make_frames(w, mean, iv, T, seed=3)
    wide = torch.zeros((T, D + 5), dtype=torch.float32, device="cuda")
    wide[:, :D] = torch.from_numpy(x).cuda()
    wide[:, D:] = 1e30                      # must never be read as a feature
    xv = wide[:, :D]
    assert xv.stride(0) == D + 5
    g = ctx.gmm(w, mean, iv)
    og = orc.Gmm(w, mean, iv)
    assert np.max(np.abs(g.llk(xv, -1e9, 1e9) - orc.llk(og, x.astype(np.float64), -1e9, 1e9))) < 1e-9
    ref = g.em_accumulate(x)
    for sz in (1, 0):
        ctx.set_option("stats_z", sz)
        try:
            assert relerr(g.em_accumulate(xv), ref) < 1e-12
            ub = np.array([0, 700, 700, 1500])
            N, F = g.tv_stats(xv, ub)
            N0, F0 = g.tv_stats(x, ub)
            assert relerr(N, N0) < 1e-12 and relerr(F, F0) < 1e-12
        finally:
            ctx.set_option("stats_z", 1)
    assert relerr(ctx.frame_moments(xv), ctx.frame_moments(x)) < 1e-13   # strided rows: scalar kernel; dense rows: flat-stream kernel


def test_tv_stats_utterance_longer_than_the_scratch_falls_back(ctx):
    """An utterance that does not fit the likelihood scratch sends tv_stats to the recomputing kernel."""
    C, D = 128, 60
    w, mean, iv = make_gmm(C, D, seed=77)
    lens = [300, 9000, 120]
    ub = np.concatenate([[0], np.cumsum(lens)])
    x = make_frames(w, mean, iv, int(ub[-1]), seed=5)
    g = ctx.gmm(w, mean, iv)
    N0, F0 = g.tv_stats(x, ub)
    prev = ctx.set_option("z_scratch_mb", 8)      # 6784 frames per chunk < 9000
    try:
        ctx.set_option("timing", 1)
        N1, F1 = g.tv_stats(x, ub)
        assert ctx.kernel_launches("k_stats_mfma") >= 1
        ctx.set_option("timing", 0)
    finally:
        ctx.set_option("z_scratch_mb", prev)
    assert relerr(N1, N0) < 1e-12 and relerr(F1, F0) < 1e-12


def test_frame_moments(ctx):
    rng = np.random.default_rng(0)
    for T, D in [(1, 60), (1000, 60), (4097, 34), (50, 130), (100003, 60), (7777, 20), (5000, 128)]:
        for dtype in (np.float32, np.float64):           # dense rows of 16-byte multiples take the flat-stream kernel
            x = rng.normal(1.0, 2.0, (T, D)).astype(dtype)
            acc = ctx.frame_moments(x)
            s, ss, n = orc.frame_acc(x.astype(np.float64))
            assert acc[2 * D] == T == n
            assert relerr(acc[:D], s) < 1e-12 and relerr(acc[D:2 * D], ss) < 1e-12


def test_linearity_property_full_size(ctx):
    """Size-independent property at the BASELINE model size (2048 x 60): statistics of a
    concatenation are the sum of the statistics of the parts; occupancies sum to T."""
    import torch
    w, mean, iv = make_gmm(2048, 60, seed=1)
    x = torch.from_numpy(make_frames(w, mean, iv, 200_000, seed=2)).cuda()
    g = ctx.gmm(w, mean, iv)
    n = g.em_acc_len()
    full = torch.zeros(n, dtype=torch.float64, device="cuda")
    parts = torch.zeros(n, dtype=torch.float64, device="cuda")
    g.em_accumulate(x, acc=full)
    g.em_accumulate(x[:70_001], acc=parts)
    g.em_accumulate(x[70_001:], acc=parts)
    ctx.sync()
    f, p = full.cpu().numpy(), parts.cpu().numpy()
    assert relerr(p, f) < 1e-11
    assert abs(f[:2048].sum() - 200_000) < 1e-6


@pytest.mark.parametrize("spread", [2.0, 0.3])
def test_top_c_paths_agree(ctx, spread):
    """DETERMINE_TOP_DISTRIBS: the stored-likelihood path (k_llk_mfma<WZ> + k_topc_from_z) against the direct-form kernel on the
    same frames -- identical indices, values to 1e-9 -- on well separated and on heavily overlapping mixtures."""
    C, D, T = 2048, 60, 3000
    w, mean, iv = make_gmm(C, D, seed=5, spread=spread)
    x = make_frames(w, mean, iv, T, seed=6).astype(np.float32)
    g = ctx.gmm(w, mean, iv)
    res = {}
    for z in (1, 0):
        ctx.set_option("topc_z", z)
        res[z] = g.llk_determine_top(x, 10, complete=True)
    ctx.set_option("topc_z", 1)
    assert np.array_equal(res[1]["idx"], res[0]["idx"])
    for k in ("lk", "nontop_lk", "llk", "nontop_w"):
        assert relerr(res[1][k], res[0][k]) < 1e-9, k
    fin = np.isfinite(res[0]["nontop_llk"])
    assert np.array_equal(fin, np.isfinite(res[1]["nontop_llk"]))
    assert np.max(np.abs(res[1]["nontop_llk"][fin] - res[0]["nontop_llk"][fin])) < 1e-8


def test_fused_top_c_redoes_piled_up_frames_exactly(ctx):
    """Fused selection (candidates collected in k_llk_mfma<TC>, ranked by k_topc_rank): a mixture with 100 identical Gaussians
    makes more than 64 logits tie at the threshold -- those frames go through the direct-form kernel and still come back in the
    reference's order (ties: lowest index first, TopGauss.cpp:167-193); separated frames stay on the fused path."""
    C, D, T, ctop = 256, 60, 700, 10
    w, mean, iv = make_gmm(C, D, seed=21)
    mean[100:200] = mean[100]; iv[100:200] = iv[100]; w[100:200] = w[100]; w /= w.sum()
    x = make_frames(w, mean, iv, T, seed=22)
    g = ctx.gmm(w, mean, iv)
    before = ctx.set_option("topc_fallbacks", 0)
    d = g.llk_determine_top(x, ctop, True)
    redone = ctx.set_option("topc_fallbacks", 0)
    do = orc.llk_determine_top(orc.Gmm(w, mean, iv), x.astype(np.float64), ctop, True)
    assert np.array_equal(d["idx"], do["idx"])
    assert np.max(np.abs(d["llk"] - do["llk"])) < 1e-9 and relerr(d["lk"], do["lk"]) < 1e-10
    assert 0 < redone < T            # the frames drawn from the duplicated Gaussians were redone, the others were not
    # the three selection paths agree
    for fused, z in ((0, 1), (0, 0)):
        ctx.set_option("topc_fused", fused); ctx.set_option("topc_z", z)
        e = g.llk_determine_top(x, ctop, True)
        assert np.array_equal(e["idx"], d["idx"]) and np.max(np.abs(e["llk"] - d["llk"])) < 1e-9
        assert np.max(np.abs(e["nontop_llk"] - d["nontop_llk"])) < 1e-9
    ctx.set_option("topc_fused", 1); ctx.set_option("topc_z", 1)


def test_fused_top_c_ranks_on_mfma_logits_and_resolves_ties_in_the_direct_form(ctx):
    """k_topc_rank ranks the survivors on their MFMA logits and re-evaluates them in the reference's direct form only when another
    survivor lies within 1e-6 of a selected one.  A mixture with three PAIRS of identical Gaussians makes exact ties inside the
    selection (reference order: lowest index first, TopGauss.cpp:167-193): indices must equal the oracle's on every frame, and the
    result must agree with the form that re-evaluates every frame (option topc_rank_direct, what round 2 did)."""
    C, D, T, ctop = 512, 60, 4000, 10
    w, mean, iv = make_gmm(C, D, seed=31, spread=0.5)
    for a, b in ((7, 300), (8, 9), (100, 511)):
        mean[b] = mean[a]; iv[b] = iv[a]; w[b] = w[a]
    w /= w.sum()
    x = make_frames(w, mean, iv, T, seed=32).astype(np.float32)
    g = ctx.gmm(w, mean, iv)
    do = orc.llk_determine_top(orc.Gmm(w, mean, iv), x.astype(np.float64), ctop, True)
    res = {}
    for direct in (0, 1):
        ctx.set_option("topc_rank_direct", direct)
        res[direct] = g.llk_determine_top(x, ctop, True)
    ctx.set_option("topc_rank_direct", 0)
    tied = sum(1 for t in range(T) if any(p in do["idx"][t] for p in (7, 8, 100)))
    assert tied > 50                                   # the tie-breaking rule was exercised
    for direct in (0, 1):
        d = res[direct]
        assert np.array_equal(d["idx"], do["idx"]), direct
        assert np.max(np.abs(d["llk"] - do["llk"])) < 1e-9 and relerr(d["lk"], do["lk"]) < 1e-10
        big = do["nontop_lk"] > 1e-250
        assert relerr(d["nontop_lk"][big], do["nontop_lk"][big]) < 1e-9
    assert relerr(res[0]["lk"], res[1]["lk"]) < 1e-11


@pytest.mark.parametrize("spread,ctop", [(2.0, 10), (0.3, 10), (0.1, 16), (2.0, 1)])
def test_fused_top_c_two_frames_per_wave_matches_one_frame_per_wave(ctx, spread, ctop):
    """k_topc_rank2 (two frames per wave; frames with more than 128 records or more than 32 survivors are passed to k_topc_rank in
    list mode without a host round trip) against k_topc_rank on every frame: identical indices, values to 1e-12 (the half-wave
    sums add in another order) -- on separated, overlapping and very flat mixtures (the flat one has wide frames) and an odd frame
    count (a dead half wave).  Both against the oracle."""
    C, D, T = 2048, 60, 3001
    w, mean, iv = make_gmm(C, D, seed=41, spread=spread)
    x = make_frames(w, mean, iv, T, seed=42).astype(np.float32)
    g = ctx.gmm(w, mean, iv)
    res = {}
    for two in (1, 0):
        ctx.set_option("topc_rank2", two)
        ctx.set_option("topc_fallbacks", 0)
        res[two] = g.llk_determine_top(x, ctop, complete=True)
        res[two]["fallbacks"] = ctx.set_option("topc_fallbacks", 0)
    ctx.set_option("topc_rank2", 1)
    assert np.array_equal(res[1]["idx"], res[0]["idx"])
    for k in ("lk", "nontop_lk", "llk", "nontop_w"):
        assert relerr(res[1][k], res[0][k]) < 1e-12, k
    fin = np.isfinite(res[0]["nontop_llk"])
    assert np.array_equal(fin, np.isfinite(res[1]["nontop_llk"]))
    assert np.max(np.abs(res[1]["nontop_llk"][fin] - res[0]["nontop_llk"][fin])) < 1e-10
    assert res[1]["fallbacks"] == res[0]["fallbacks"]     # passing a frame on to the one-frame kernel is not a fallback
    do = orc.llk_determine_top(orc.Gmm(w, mean, iv), x[:600].astype(np.float64), ctop, True)
    assert np.array_equal(res[1]["idx"][:600], do["idx"])
    assert np.max(np.abs(res[1]["llk"][:600] - do["llk"])) < 1e-9


def test_default_paths_are_bitwise_the_round_5_results(golden_dir):
    """tests/golden/r05_bitwise.json holds SHA-256 digests of 107 result arrays (EM statistics, log-likelihoods, top-10 / top-20 lists,
    posteriors, N / F, TETt, i-vectors, the T-matrix accumulators, updateTestimate, minDivergence, two scoring rules; three model shapes; the Cholesky family at orders 400 / 200 / 130)
    computed by the ROUND-5 library on an MI355X (tools/bitwise_fixture.py).  Round 6 took the measured-slower kernel variants out of
    libgmmiv.so, moved the handling of unusable feature values into the kernels and the chunk tables onto the device: none of it may
    move one bit of a default path on clean data."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bitwise_fixture as bf
    ref = json.load(open(os.path.join(golden_dir, "r05_bitwise.json")))["arrays"]
    got = bf.digests(bf.compute())
    bad = [k for k in ref if got.get(k) != ref[k]]
    assert len(ref) == 107 and not bad, bad
