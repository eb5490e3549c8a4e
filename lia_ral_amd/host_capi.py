"""ctypes binding of host/libliatools_gpu.so -- the C++ mirror of the LIA_SpkTools hot-path drivers
(trainModelStream, ComputeTest's LLR loop, IvExtractor, TotalVariability) over libgmmiv."""
import ctypes as ct
import os

import numpy as np

from . import capi  # noqa: F401  (loads torch's HIP runtime first, then libgmmiv.so)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "host", "libliatools_gpu.so")


class HostError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise HostError("libliatools_gpu.so is not built (%s); run __graft_entry__.build()" % LIB_PATH)
    lib = ct.CDLL(LIB_PATH)
    lib.liagpu_last_error.restype = ct.c_char_p
    return lib


lib = _load()
_dp = ct.POINTER(ct.c_double)
_lp = ct.POINTER(ct.c_long)
_fp = ct.POINTER(ct.c_float)


def _chk(rc):
    if rc != 0:
        raise HostError(lib.liagpu_last_error().decode())


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _segs(seg_begin, seg_len):
    b = np.ascontiguousarray(seg_begin, np.int64); l = np.ascontiguousarray(seg_len, np.int64)
    return b, l, b.ctypes.data_as(_lp), l.ctypes.data_as(_lp)


def train_world(x, seg_begin, seg_len, w, mean, cov, nb_it, bagged_p=1.0, init_floor=0.0, final_floor=0.0,
                init_ceil=10.0, final_ceil=10.0, init_rand=0, device=0):
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    w = np.array(w, np.float64); mean = np.array(mean, np.float64); cov = np.array(cov, np.float64)
    C = len(w)
    b, l, bp, lp = _segs(seg_begin, seg_len)
    gm = np.empty(D); gc = np.empty(D); llk = np.empty(nb_it); it_ms = np.empty(nb_it)
    _chk(lib.liagpu_train_world_timed(device, x.ctypes.data_as(_fp), ct.c_long(T), D, bp, lp, ct.c_long(len(b)), C, _d(w), _d(mean),
                                      _d(cov), nb_it, ct.c_double(bagged_p), ct.c_double(init_floor), ct.c_double(final_floor),
                                      ct.c_double(init_ceil), ct.c_double(final_ceil), ct.c_long(init_rand), _d(gm), _d(gc), _d(llk), _d(it_ms)))
    return dict(w=w, mean=mean, cov=cov, global_mean=gm, global_cov=gc, llk=llk, it_ms=it_ms)


def train_world_streams(xs, seg_begins, seg_lens, w, mean, cov, nb_it, weights=None, bagged_p=1.0, init_floor=0.0, final_floor=0.0,
                        init_ceil=10.0, final_ceil=10.0, init_rand=0, component_reduction=False, target_distrib_count=0,
                        normalize_model=False, normalize_mean_only=False, normalize_nb_it=1, device=0):
    """TrainWorld over several input streams (liagpu_train_world_streams): xs / seg_begins / seg_lens are lists, one entry per stream."""
    ns = len(xs)
    xs = [np.ascontiguousarray(x, np.float32) for x in xs]
    D = xs[0].shape[1]
    w = np.array(w, np.float64); mean = np.array(mean, np.float64); cov = np.array(cov, np.float64)
    C = len(w)
    segs = [_segs(b, l) for b, l in zip(seg_begins, seg_lens)]
    xp = (_fp * ns)(*[x.ctypes.data_as(_fp) for x in xs])
    Tp = (ct.c_long * ns)(*[x.shape[0] for x in xs])
    bp = (_lp * ns)(*[s[2] for s in segs]); lp = (_lp * ns)(*[s[3] for s in segs])
    np_ = (ct.c_long * ns)(*[len(s[0]) for s in segs])
    wt = None if weights is None else np.ascontiguousarray(weights, np.float64)
    opts = (ct.c_long * 5)(int(component_reduction), int(target_distrib_count), int(normalize_model), int(normalize_mean_only), int(normalize_nb_it))
    gm = np.empty(D); gc = np.empty(D); llk = np.empty(nb_it); cout = ct.c_long(0)
    _chk(lib.liagpu_train_world_streams(device, ns, xp, Tp, D, bp, lp, np_, _d(wt), C, _d(w), _d(mean), _d(cov), nb_it, ct.c_double(bagged_p),
                                        ct.c_double(init_floor), ct.c_double(final_floor), ct.c_double(init_ceil), ct.c_double(final_ceil),
                                        ct.c_long(init_rand), opts, ct.byref(cout), _d(gm), _d(gc), _d(llk)))
    Co = cout.value
    return dict(w=w[:Co].copy(), mean=mean[:Co].copy(), cov=cov[:Co].copy(), global_mean=gm, global_cov=gc, llk=llk)


def mixture_init_streams(xs, seg_begins, seg_lens, C, global_cov, weights=None, nb_frame_to_select=50.0, min_len=3, max_len=7, device=0):
    """mixtureInit over several input streams (liagpu_mixture_init_streams) -> dict(w, mean, cov, counts)."""
    ns = len(xs)
    xs = [np.ascontiguousarray(x, np.float32) for x in xs]
    D = xs[0].shape[1]
    segs = [_segs(b, l) for b, l in zip(seg_begins, seg_lens)]
    xp = (_fp * ns)(*[x.ctypes.data_as(_fp) for x in xs])
    Tp = (ct.c_long * ns)(*[x.shape[0] for x in xs])
    bp = (_lp * ns)(*[s[2] for s in segs]); lp = (_lp * ns)(*[s[3] for s in segs])
    np_ = (ct.c_long * ns)(*[len(s[0]) for s in segs])
    wt = None if weights is None else np.ascontiguousarray(weights, np.float64)
    gc = np.ascontiguousarray(global_cov, np.float64)
    w = np.empty(C); mean = np.empty((C, D)); cov = np.empty((C, D)); cnt = np.zeros(C, np.int64)
    _chk(lib.liagpu_mixture_init_streams(device, ns, xp, Tp, D, bp, lp, np_, _d(wt), C, _d(gc), ct.c_double(nb_frame_to_select), ct.c_long(min_len),
                                         ct.c_long(max_len), _d(w), _d(mean), _d(cov), cnt.ctypes.data_as(_lp)))
    return dict(w=w, mean=mean, cov=cov, counts=cnt)


def model_reduce_normalize(w, mean, cov, nb_top=0, normalize=False, mean_only=False, nb_it=1):
    """selectComponent(nbTop) + reduceModel + normalizeWeights, then normalizeMixture to N(0,1) (TrainTools.cpp:1078-1098); host only.
    Returns (w, mean, cov, order) with order = TabWeight's heaviest-first component order of the INPUT model."""
    w = np.array(w, np.float64); mean = np.array(mean, np.float64); cov = np.array(cov, np.float64)
    C, D = mean.shape
    order = np.empty(C, np.int64)
    _chk(lib.liagpu_model_reduce_normalize(C, D, _d(w), _d(mean), _d(cov), ct.c_long(nb_top), int(normalize), int(mean_only), ct.c_long(nb_it),
                                           order.ctypes.data_as(_lp)))
    Co = nb_top if 0 < nb_top < C else C
    return w[:Co].copy(), mean[:Co].copy(), cov[:Co].copy(), order


def train_world_scratch(x, seg_begin, seg_len, C, nb_it, nb_frame_to_select=50.0, use01=False, bagged_p=1.0, init_floor=0.0, final_floor=0.0,
                        init_ceil=10.0, final_ceil=10.0, init_rand=0, device=0):
    """TrainWorld with no initial model: computeMeanCov (or use01), mixtureInit, trainModelStream (liagpu_train_world_scratch)."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    b, l, bp, lp = _segs(seg_begin, seg_len)
    w = np.empty(C); mean = np.empty((C, D)); cov = np.empty((C, D)); gc = np.empty(D); llk = np.empty(nb_it)
    _chk(lib.liagpu_train_world_scratch(device, x.ctypes.data_as(_fp), ct.c_long(T), D, bp, lp, ct.c_long(len(b)), C, ct.c_double(nb_frame_to_select),
                                        int(use01), _d(w), _d(mean), _d(cov), nb_it, ct.c_double(bagged_p), ct.c_double(init_floor),
                                        ct.c_double(final_floor), ct.c_double(init_ceil), ct.c_double(final_ceil), ct.c_long(init_rand), _d(gc), _d(llk)))
    return dict(w=w, mean=mean, cov=cov, global_cov=gc, llk=llk)


def mixture_init(x, seg_begin, seg_len, C, global_cov, nb_frame_to_select=50.0, stream_weight=1.0, single_stream_proba=None, min_len=3,
                 max_len=7, device=0):
    """mixtureInit: TrainWorld's start-from-scratch model (multi-stream form with one stream; single_stream_proba = the
    baggedFrameProbabilityInit of the single-stream form).  Returns dict(w, mean, cov, counts)."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    b, l, bp, lp = _segs(seg_begin, seg_len)
    gc = np.ascontiguousarray(global_cov, np.float64)
    w = np.empty(C); mean = np.empty((C, D)); cov = np.empty((C, D)); cnt = np.zeros(C, np.int64)
    single = single_stream_proba is not None
    _chk(lib.liagpu_mixture_init(device, x.ctypes.data_as(_fp), ct.c_long(T), D, bp, lp, ct.c_long(len(b)), C, int(single),
                                 ct.c_double(single_stream_proba if single else nb_frame_to_select), ct.c_double(stream_weight),
                                 ct.c_long(min_len), ct.c_long(max_len), _d(gc), _d(w), _d(mean), _d(cov), cnt.ctypes.data_as(_lp)))
    return dict(w=w, mean=mean, cov=cov, counts=cnt)


def tv_verify_emlk(x, file_begin, row_of_file, ubm, Tmat, W, max_llk_computed=None, min_llk=-200.0, max_llk=200.0, device=0):
    """TVAcc::verifyEMLK: per-file mean log-likelihood under the speaker model m + T^T w_row; returns (llk [nfiles], total, supervectors)."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in ubm]
    C = len(w)
    fb = np.ascontiguousarray(file_begin, np.int64); rows = np.ascontiguousarray(row_of_file, np.int64)
    nf = len(fb) - 1
    Tm = np.ascontiguousarray(Tmat, np.float64); Wm = np.ascontiguousarray(W, np.float64)
    n = nf if max_llk_computed is None else min(nf, max_llk_computed)
    llk = np.zeros(nf); total = ct.c_double(0.0); sv = np.empty((nf, C * D))
    _chk(lib.liagpu_tv_verify_emlk(device, x.ctypes.data_as(_fp), ct.c_long(T), D, fb.ctypes.data_as(_lp), ct.c_long(nf), rows.ctypes.data_as(_lp),
                                   C, _d(w), _d(mean), _d(cov), Tm.shape[0], _d(Tm), ct.c_long(Wm.shape[0]), _d(Wm), ct.c_long(n),
                                   ct.c_double(min_llk), ct.c_double(max_llk), _d(llk), ct.byref(total), _d(sv)))
    return llk[:n], total.value, sv


def mean_llk(x, seg_begin, seg_len, model, min_llk=-200.0, max_llk=200.0, device=0):
    """accumulateStatLLK: mean clamped log-likelihood of the selected frames."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in model]
    b, l, bp, lp = _segs(seg_begin, seg_len)
    out = ct.c_double(0.0)
    _chk(lib.liagpu_mean_llk(device, x.ctypes.data_as(_fp), ct.c_long(T), D, bp, lp, ct.c_long(len(b)), len(w), _d(w), _d(mean), _d(cov),
                             ct.c_double(min_llk), ct.c_double(max_llk), ct.byref(out)))
    return out.value


def mean_llk_streams(xs, seg_begins, seg_lens, model, decision=None, min_llk=-200.0, max_llk=200.0, device=0):
    """meanLikelihood over several feature servers, optionally weighted by one decision value per server (GeneralTools.cpp:599-624)."""
    ns = len(xs)
    xs = [np.ascontiguousarray(x, np.float32) for x in xs]
    D = xs[0].shape[1]
    w, m, c = [np.ascontiguousarray(a, np.float64) for a in model]
    segs = [_segs(b, l) for b, l in zip(seg_begins, seg_lens)]
    xp = (_fp * ns)(*[x.ctypes.data_as(_fp) for x in xs])
    Tp = (ct.c_long * ns)(*[x.shape[0] for x in xs])
    bp = (_lp * ns)(*[s[2] for s in segs]); lp = (_lp * ns)(*[s[3] for s in segs])
    np_ = (ct.c_long * ns)(*[len(s[0]) for s in segs])
    dec = None if decision is None else np.ascontiguousarray(decision, np.float64)
    out = ct.c_double(0.0)
    _chk(lib.liagpu_mean_llk_streams(device, ns, xp, Tp, D, bp, lp, np_, _d(dec), len(w), _d(w), _d(m), _d(c), ct.c_double(min_llk),
                                     ct.c_double(max_llk), ct.byref(out)))
    return out.value


def train_target(x, seg_begin, seg_len, world, nb_it=1, mean_reg=16.0, device=0):
    """TrainTarget: mean-only MAPOccDep adaptation of the world model; returns (w, mean, cov)."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in world]
    C = len(w)
    b, l, bp, lp = _segs(seg_begin, seg_len)
    wo = np.empty(C); mo = np.empty((C, D)); co = np.empty((C, D))
    _chk(lib.liagpu_train_target(device, x.ctypes.data_as(_fp), ct.c_long(T), D, bp, lp, ct.c_long(len(b)), C, _d(w), _d(mean),
                                 _d(cov), nb_it, ct.c_double(mean_reg), _d(wo), _d(mo), _d(co)))
    return wo, mo, co


def _map_flags(mean, var, weight):
    return int(mean) | (int(var) << 1) | (int(weight) << 2)


def train_target_ex(x, seg_begin, seg_len, world, method="MAPOccDep", nb_it=1, bagged_p=1.0, mean=True, var=False, weight=False,
                    reg=(16.0, 16.0, 16.0), alpha_mean=0.75, normalize=False, normalize_mean_only=False, normalize_nb_it=1, device=0):
    """TrainTarget (adaptModel, TrainTools.cpp:871-904) with every MAPCfg parameter -> (w, mean, cov)."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    w, m, c = [np.ascontiguousarray(a, np.float64) for a in world]
    C = len(w)
    b, l, bp, lp = _segs(seg_begin, seg_len)
    r = np.ascontiguousarray(reg, np.float64)
    norm = (ct.c_long * 3)(int(normalize), int(normalize_mean_only), int(normalize_nb_it))
    wo = np.empty(C); mo = np.empty((C, D)); co = np.empty((C, D))
    _chk(lib.liagpu_train_target_ex(device, x.ctypes.data_as(_fp), ct.c_long(T), D, bp, lp, ct.c_long(len(b)), C, _d(w), _d(m), _d(c), method.encode(),
                                    nb_it, ct.c_double(bagged_p), _map_flags(mean, var, weight), _d(r), ct.c_double(alpha_mean), norm, _d(wo), _d(mo), _d(co)))
    return wo, mo, co


def compute_map(method, init, client, frame_count, mean=True, var=False, weight=False, reg=(16.0, 16.0, 16.0), alpha_mean=0.75):
    """computeMAP (TrainTools.cpp:543-556) on its own, host arithmetic only: init / client = (w, mean, cov) -> adapted (w, mean, cov)."""
    w0, m0, c0 = [np.ascontiguousarray(a, np.float64) for a in init]
    w, m, c = [np.array(a, np.float64, order="C", copy=True) for a in client]
    C, D = m0.shape
    r = np.ascontiguousarray(reg, np.float64)
    _chk(lib.liagpu_compute_map(C, D, _d(w0), _d(m0), _d(c0), _d(w), _d(m), _d(c), ct.c_double(frame_count), method.encode(),
                                _map_flags(mean, var, weight), _d(r), ct.c_double(alpha_mean)))
    return w, m, c


def topgauss(x, seg_begin, seg_len, ubm, top_gauss, path, top_distribs_count=64, model2_mean=None, complete=True, min_llk=-200.0,
             max_llk=200.0, device=0):
    """TopGauss::compute -> write(path) -> read(path) -> get (liagpu_topgauss).  Returns dict(llk_compute, llk_get, llk_get_model2,
    capped, nbg [T], idx (flat), snsw, snsl)."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in ubm]
    C = len(w)
    b, l, bp, lp = _segs(seg_begin, seg_len)
    n = int(l.sum())
    cap = min(top_distribs_count, C)
    out = np.zeros(4); cnt = np.zeros(n, np.int64); idx = np.zeros(n * cap, np.int64); sw = np.zeros(n); sl = np.zeros(n); tot = ct.c_long(0)
    m2 = None if model2_mean is None else np.ascontiguousarray(model2_mean, np.float64)
    _chk(lib.liagpu_topgauss(device, x.ctypes.data_as(_fp), ct.c_long(T), D, bp, lp, ct.c_long(len(b)), C, _d(w), _d(mean), _d(cov), _d(m2),
                             ct.c_double(top_gauss), int(top_distribs_count), int(complete), ct.c_double(min_llk), ct.c_double(max_llk),
                             path.encode(), _d(out), cnt.ctypes.data_as(_lp), idx.ctypes.data_as(_lp), _d(sw), _d(sl), ct.byref(tot)))
    return dict(llk_compute=out[0], llk_get=out[1], llk_get_model2=out[2], capped=int(out[3]), nbg=cnt, idx=idx[:tot.value].copy(), snsw=sw, snsl=sl)


def compute_test(x, seg_begin, seg_len, world, clients, top_c=10, complete=True, min_llk=-200.0, max_llk=200.0,
                 segmental=False, device=0, reps=0):
    """world = (w, mean, cov); clients = list of (w, mean, cov).  Returns LLR[n_seg_or_1, n_clients]; with reps > 0 the LLR loop
    is run that many times on the resident features and (llr, ms[reps]) is returned."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    ww, mw, cw = [np.ascontiguousarray(a, np.float64) for a in world]
    C = len(ww)
    wc = np.ascontiguousarray(np.stack([c[0] for c in clients]), np.float64)
    mc = np.ascontiguousarray(np.stack([c[1] for c in clients]), np.float64)
    cc = np.ascontiguousarray(np.stack([c[2] for c in clients]), np.float64)
    b, l, bp, lp = _segs(seg_begin, seg_len)
    nout = (len(b) if segmental else 1) * len(clients)
    out = np.empty(nout)
    ms = np.zeros(max(reps, 1))
    _chk(lib.liagpu_compute_test_timed(device, x.ctypes.data_as(_fp), ct.c_long(T), D, bp, lp, ct.c_long(len(b)), C, _d(ww), _d(mw),
                                       _d(cw), len(clients), _d(wc), _d(mc), _d(cc), top_c, int(complete), ct.c_double(min_llk),
                                       ct.c_double(max_llk), int(segmental), _d(out), max(reps, 1), _d(ms)))
    out = out.reshape(-1, len(clients))
    return (out, ms) if reps > 0 else out


def compute_test_ex(x, seg_begin, seg_len, world, clients, top_c=10, complete=True, min_llk=-200.0, max_llk=200.0, segmental=False,
                    world_decime=1, window_size=0, window_dec=0, device=0):
    """ComputeTest frame loop with worldDecime and the WindowLLR mode.  -> (llr [nseg or 1, nClients], windows [n, 2 + nClients])."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    sb = np.ascontiguousarray(seg_begin, np.int64); sl = np.ascontiguousarray(seg_len, np.int64)
    ww, wm, wc = [np.ascontiguousarray(a, np.float64) for a in world]
    C = len(ww)
    cw = np.ascontiguousarray(np.stack([c[0] for c in clients]), np.float64)
    cm = np.ascontiguousarray(np.stack([c[1] for c in clients]), np.float64)
    cc = np.ascontiguousarray(np.stack([c[2] for c in clients]), np.float64)
    nC = len(clients)
    nseg = len(sb) if segmental else 1
    out = np.empty((nseg, nC))
    max_win = int(sl.sum()) + 1
    win = np.zeros((max_win, 2 + nC)); nw = ct.c_long(0)
    _chk(lib.liagpu_compute_test_ex(device, x.ctypes.data_as(_fp), ct.c_long(T), D, sb.ctypes.data_as(_lp), sl.ctypes.data_as(_lp),
                                    ct.c_long(len(sb)), C, _d(ww), _d(wm), _d(wc), nC, _d(cw), _d(cm), _d(cc), top_c, int(complete),
                                    ct.c_double(min_llk), ct.c_double(max_llk), int(segmental), ct.c_long(world_decime),
                                    ct.c_long(window_size), ct.c_long(window_dec), _d(out), _d(win), ct.c_long(max_win), ct.byref(nw)))
    return out, win[:nw.value]


def iv_extract(x, utt_begin, ubm, Tmat, device=0, return_stats=False, reps=0):
    """IvExtractor.  reps > 0: the extraction is run that many times on the resident features; the wall times
    ms[reps, 4] = (statistics, substractM, estimateTETt [first run only], estimateW) are appended to the result."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in ubm]
    C = len(w)
    Tm = np.ascontiguousarray(Tmat, np.float64)
    R = Tm.shape[0]
    ub = np.ascontiguousarray(utt_begin, np.int64)
    U = len(ub) - 1
    W = np.empty((U, R))
    N = np.empty((U, C)) if return_stats else None
    F = np.empty((U, C * D)) if return_stats else None
    ms = np.zeros((max(reps, 1), 4))
    _chk(lib.liagpu_iv_extract_timed(device, x.ctypes.data_as(_fp), ct.c_long(T), D, ub.ctypes.data_as(_lp), ct.c_long(U), C, _d(w),
                                     _d(mean), _d(cov), R, _d(Tm), _d(W), _d(N), _d(F), max(reps, 1), _d(ms)))
    res = (W, N, F) if return_stats else W
    if reps > 0:
        return res + (ms,) if return_stats else (W, ms)
    return res


def iv_extract_approx(x, utt_begin, ubm, Tmat, mode, device=0):
    """IvExtractor --mode ubmWeight (mode=1) / eigenDecomposition (mode=2).  -> dict(W, Wcov, Q, D)."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in ubm]
    C = len(w)
    Tm = np.ascontiguousarray(Tmat, np.float64)
    R = Tm.shape[0]
    ub = np.ascontiguousarray(utt_begin, np.int64)
    U = len(ub) - 1
    W = np.empty((U, R)); Wc = np.empty((R, R)); Q = np.zeros((R, R)); Dm = np.zeros((C, R))
    _chk(lib.liagpu_iv_extract_approx(device, int(mode), x.ctypes.data_as(_fp), ct.c_long(T), D, ub.ctypes.data_as(_lp), ct.c_long(U), C,
                                      _d(w), _d(mean), _d(cov), R, _d(Tm), _d(W), _d(Wc), _d(Q), _d(Dm)))
    return dict(W=W, Wcov=Wc, Q=Q, D=Dm)


def backend_train(X, sps, nb_it=2, sph_norm=False, lda_rank=0, device=0):
    """PldaDev EFR / sphNorm training (+ WCCN, Mahalanobis, LDA on the normalised set). X[dim, n] -> dict."""
    X = np.array(X, np.float64, order="C", copy=True)
    dim, n = X.shape
    sps = np.ascontiguousarray(sps, np.int64)
    mats = np.empty((nb_it, dim, dim)); means = np.empty((nb_it, dim)); wccn = np.empty((dim, dim)); mah = np.empty((dim, dim))
    lda = np.empty((max(lda_rank, 1), dim))
    _chk(lib.liagpu_backend_train(device, dim, ct.c_long(n), _d(X), ct.c_long(len(sps)), sps.ctypes.data_as(_lp), int(sph_norm), nb_it,
                                  _d(mats), _d(means), _d(wccn), _d(mah), lda_rank, _d(lda)))
    return dict(X=X, mats=mats, means=means, wccn=wccn, mahalanobis=mah, lda=lda[:lda_rank])


def plda_train(X, sps, F, G, Sigma, nb_it, device=0):
    """PLDA.cpp training loop: nb_it x PldaModel::em_iteration.  -> dict(X, F, G, Sigma, Delta, original_mean)."""
    X = np.array(X, np.float64, order="C", copy=True); F = np.array(F, np.float64, order="C", copy=True)
    G = np.array(G, np.float64, order="C", copy=True); Sigma = np.array(Sigma, np.float64, order="C", copy=True)
    dim, n = X.shape
    sps = np.ascontiguousarray(sps, np.int64)
    Delta = np.zeros(dim); om = np.zeros(dim)
    _chk(lib.liagpu_plda_train(device, dim, ct.c_long(n), _d(X), ct.c_long(len(sps)), sps.ctypes.data_as(_lp), F.shape[1], G.shape[1], nb_it,
                               _d(F), _d(G), _d(Sigma), _d(Delta), _d(om)))
    return dict(X=X, F=F, G=G, Sigma=Sigma, Delta=Delta, original_mean=om)


def iv_test(dev, sps, enrol, enrol_per_model, test, scoring="cosine", iv_norm=False, iv_norm_it=1, sph_norm=False, lda_rank=0,
            wccn=False, plda=None, plda_it=0, device=0):
    """IvTest end to end.  plda = (F, G, Sigma) initial matrices in the normalised space.  -> scores [nModels, nTest]."""
    dev = np.ascontiguousarray(dev, np.float64); enrol = np.ascontiguousarray(enrol, np.float64); test = np.ascontiguousarray(test, np.float64)
    dim, n_dev = dev.shape
    sps = np.ascontiguousarray(sps, np.int64); epm = np.ascontiguousarray(enrol_per_model, np.int64)
    code = {"cosine": 0, "mahalanobis": 1, "2cov": 2, "plda": 3}[scoring]
    if plda is not None:
        F, G, S = [np.ascontiguousarray(a, np.float64) for a in plda]
        rf, rg = F.shape[1], G.shape[1]
    else:
        F = G = S = np.zeros(1); rf = rg = 0
    out = np.empty((len(epm), test.shape[1]))
    _chk(lib.liagpu_iv_test(device, dim, ct.c_long(n_dev), _d(dev), ct.c_long(len(sps)), sps.ctypes.data_as(_lp), ct.c_long(len(epm)),
                            epm.ctypes.data_as(_lp), _d(enrol), ct.c_long(test.shape[1]), _d(test), int(iv_norm), int(iv_norm_it), int(sph_norm),
                            int(lda_rank > 0), int(lda_rank), int(wccn), code, rf, rg, int(plda_it), _d(F), _d(G), _d(S), _d(out)))
    return out


def tv_train(N, F, ubm, Tmat, nb_it, min_div=True, device=0):
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in ubm]
    C, D = mean.shape
    N = np.ascontiguousarray(N, np.float64); F = np.ascontiguousarray(F, np.float64)
    Tm = np.array(Tmat, np.float64)
    R = Tm.shape[0]
    means = np.empty(C * D)
    _chk(lib.liagpu_tv_train(device, ct.c_long(N.shape[0]), C, D, _d(w), _d(mean), _d(cov), R, _d(N), _d(F), _d(Tm), nb_it,
                             int(min_div), _d(means)))
    return Tm, means


def io_xml(xml_in, xml_out, raw_out=""):
    dims = np.zeros(2, np.int64); first = np.zeros(3)
    _chk(lib.liagpu_io_xml(xml_in.encode(), xml_out.encode(), raw_out.encode(), dims.ctypes.data_as(_lp), _d(first)))
    return dims, first


def io_db_header_bytes(nbytes=0):
    """Width (4 | 8) of the rows / cols extents writeMatrixDB writes; returns the previous width (0 only queries)."""
    return int(lib.liagpu_io_db_header_bytes(int(nbytes)))


def io_matrix_convert(path_in, fmt_in, path_out, fmt_out):
    dims = np.zeros(2, np.int64)
    _chk(lib.liagpu_io_matrix_convert(path_in.encode(), fmt_in.encode(), path_out.encode(), fmt_out.encode(), dims.ctypes.data_as(_lp)))
    return tuple(int(v) for v in dims)


def io_vectors(directory, ids, ext, fmt, W):
    W = np.ascontiguousarray(W, np.float64)
    n, rank = W.shape
    arr = (ct.c_char_p * n)(*[i.encode() for i in ids])
    back = np.empty((rank, n))
    _chk(lib.liagpu_io_vectors(directory.encode(), n, arr, ext.encode(), fmt.encode(), rank, _d(W), _d(back)))
    return back


def tv_init_t(ubm, R, seed=1, device=0):
    """TVAcc::initT, randomInitLaw normal (glibc rand() Box-Muller chain); seed 1 = a process that never called srand."""
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in ubm]
    C, D = mean.shape
    Tm = np.empty((R, C * D))
    _chk(lib.liagpu_tv_init_t(device, C, D, _d(w), _d(mean), _d(cov), R, ct.c_uint(seed), _d(Tm)))
    return Tm


def tv_stats_lines(x, file_begin, lines, ubm, device=0):
    """Baum-Welch statistics per ndx line through liagpu::TVAcc (a file may be listed on several lines)."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in ubm]
    C = len(w)
    fb = np.ascontiguousarray(file_begin, np.int64)
    off = np.zeros(len(lines) + 1, np.int64); off[1:] = np.cumsum([len(l) for l in lines])
    files = np.ascontiguousarray([f for l in lines for f in l] or [0], np.int64)
    N = np.empty((len(lines), C)); F = np.empty((len(lines), C * D))
    _chk(lib.liagpu_tv_stats_lines(device, x.ctypes.data_as(_fp), ct.c_long(T), D, fb.ctypes.data_as(_lp), ct.c_long(len(fb) - 1),
                                   ct.c_long(len(lines)), off.ctypes.data_as(_lp), files.ctypes.data_as(_lp), C, _d(w), _d(mean), _d(cov),
                                   _d(N), _d(F)))
    return N, F


def tv_train_dist(N, F, ubm, Tmat, nb_it, world=1, rank=0, id_file="", n_total=None, min_div=True, device=0, overlap=False):
    """One rank of a multi-GPU TotalVariability run (liagpu_tv_train_dist2); returns (T, means, times_ms [nb_it x 4]).
    overlap: TVAcc::setOverlap -- the reduce-scatter of A starts inside estimateAandC, the all-gather of T is joined inside minDivergence."""
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in ubm]
    C, D = mean.shape
    N = np.ascontiguousarray(N, np.float64); F = np.ascontiguousarray(F, np.float64)
    Tm = np.array(Tmat, np.float64)
    R = Tm.shape[0]
    means = np.empty(C * D); times = np.zeros((nb_it, 4))
    _chk(lib.liagpu_tv_train_dist2(device, world, rank, id_file.encode(), ct.c_long(N.shape[0]), ct.c_long(n_total or N.shape[0]), C, D, _d(w),
                                   _d(mean), _d(cov), R, _d(N), _d(F), _d(Tm), nb_it, int(min_div), int(overlap), _d(means), _d(times)))
    return Tm, means, times


def train_world_dist(x, seg_begin, seg_len, w, mean, cov, nb_it, global_cov, world=1, rank=0, id_file="", init_floor=0.0, final_floor=0.0,
                     init_ceil=10.0, final_ceil=10.0, device=0):
    """One rank of a multi-GPU TrainWorld run (liagpu_train_world_dist)."""
    x = np.ascontiguousarray(x, np.float32)
    T, D = x.shape
    w = np.array(w, np.float64); mean = np.array(mean, np.float64); cov = np.array(cov, np.float64)
    b, l, bp, lp = _segs(seg_begin, seg_len)
    llk = np.empty(nb_it); gc = np.ascontiguousarray(global_cov, np.float64)
    _chk(lib.liagpu_train_world_dist(device, world, rank, id_file.encode(), x.ctypes.data_as(_fp), ct.c_long(T), D, bp, lp, ct.c_long(len(b)),
                                     len(w), _d(w), _d(mean), _d(cov), nb_it, ct.c_double(init_floor), ct.c_double(final_floor),
                                     ct.c_double(init_ceil), ct.c_double(final_ceil), _d(gc), _d(llk)))
    return dict(w=w, mean=mean, cov=cov, llk=llk)


def jfa_train(task, sess_per_spk, ubm, N, N_h, F_X, F_X_h, V, U, Dm, nb_it, Z0=None, ortho_v=False, device=0):
    """EigenVoice (task 0) / EigenChannel (1) / EstimateDMatrix (2) on given statistics; returns dict(V, U, D, Y, X, Z)."""
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in ubm]
    C, D = mean.shape
    sps = np.ascontiguousarray(sess_per_spk, np.int64)
    nspk, nsess = len(sps), int(sps.sum())
    N, N_h, F_X, F_X_h = [np.ascontiguousarray(a, np.float64) for a in (N, N_h, F_X, F_X_h)]
    V = np.array(V, np.float64); U = np.array(U, np.float64); Dm = np.array(Dm, np.float64)
    Y = np.zeros((nspk, V.shape[0])); X = np.zeros((nsess, U.shape[0])); Z = np.zeros((nspk, C * D))
    z0 = None if Z0 is None else np.ascontiguousarray(Z0, np.float64)
    _chk(lib.liagpu_jfa_train(device, task, ct.c_long(nspk), sps.ctypes.data_as(_lp), C, D, _d(w), _d(mean), _d(cov), V.shape[0], U.shape[0],
                              _d(N), _d(N_h), _d(F_X), _d(F_X_h), _d(V), _d(U), _d(Dm), None if z0 is None else _d(z0), nb_it, int(ortho_v),
                              _d(Y), _d(X), _d(Z)))
    return dict(V=V, U=U, D=Dm, Y=Y, X=X, Z=Z)


def jfa_dot_product(ubm, N, F, V, U, Dm, client_sv, device=0):
    """ComputeTest dot-product scoring in the JFA framework: scores[nTest, nClients]."""
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in ubm]
    C, D = mean.shape
    N, F, V, U, Dm, client_sv = [np.ascontiguousarray(a, np.float64) for a in (N, F, V, U, Dm, client_sv)]
    sc = np.empty((N.shape[0], client_sv.shape[0]))
    _chk(lib.liagpu_jfa_dot_product(device, ct.c_long(N.shape[0]), C, D, _d(w), _d(mean), _d(cov), V.shape[0], U.shape[0], _d(N), _d(F), _d(V),
                                    _d(U), _d(Dm), ct.c_long(client_sv.shape[0]), _d(client_sv), _d(sc)))
    return sc

def tv_stats_cross_server(x_own, x, utt_begin, ubm, device=0):
    """N / F of a TVAcc on one GpuServer from the frames of a FeatureBuffer on ANOTHER server; the accumulator's server holds
    the (clean) buffer x_own.  -> N, F, (unusable frames of x_own, of x), assume_finite of the accumulator's context afterwards."""
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in ubm]
    C, D = mean.shape
    x_own = np.ascontiguousarray(x_own, np.float32); x = np.ascontiguousarray(x, np.float32)
    ub = np.ascontiguousarray(utt_begin, np.int64)
    U = len(ub) - 1
    N = np.empty((U, C)); F = np.empty((U, C * D))
    bad = (ct.c_long * 2)(); af = ct.c_long(-7)
    _chk(lib.liagpu_tv_stats_cross_server(device, x_own.ctypes.data_as(_fp), ct.c_long(x_own.shape[0]), x.ctypes.data_as(_fp), ct.c_long(x.shape[0]), D,
                                          ub.ctypes.data_as(_lp), ct.c_long(U), C, _d(w), _d(mean), _d(cov), _d(N), _d(F), bad, ct.byref(af)))
    return N, F, (bad[0], bad[1]), af.value


def jfa_stats(x, sess_begin, sess_per_spk, ubm, device=0):
    """JFAAcc::computeAndAccumulateJFAStat on float32 frames: returns N, N_h, F_X, F_X_h."""
    w, mean, cov = [np.ascontiguousarray(a, np.float64) for a in ubm]
    C, D = mean.shape
    x = np.ascontiguousarray(x, np.float32)
    sb = np.ascontiguousarray(sess_begin, np.int64); sps = np.ascontiguousarray(sess_per_spk, np.int64)
    nspk, nsess = len(sps), len(sb) - 1
    N = np.empty((nspk, C)); Nh = np.empty((nsess, C)); FX = np.empty((nspk, C * D)); FXh = np.empty((nsess, C * D))
    _chk(lib.liagpu_jfa_stats(device, x.ctypes.data_as(ct.POINTER(ct.c_float)), ct.c_long(x.shape[0]), D, sb.ctypes.data_as(_lp), ct.c_long(nspk),
                              sps.ctypes.data_as(_lp), C, _d(w), _d(mean), _d(cov), _d(N), _d(Nh), _d(FX), _d(FXh)))
    return N, Nh, FX, FXh

def compute_test_files(world_path, client_paths, client_names, prm_path, lbl_path, mask="", label="male", frame_length=0.01,
                       top_c=10, complete=True, min_llk=-200.0, max_llk=200.0, gender="M", test_name="test", threshold=0.0,
                       device=0):
    """ComputeTest (segmental mode) driven from RAW model / .prm / .lbl files; returns (LLR[nseg, nclients], text)."""
    n = len(client_paths)
    cp = (ct.c_char_p * n)(*[p.encode() for p in client_paths])
    cn = (ct.c_char_p * n)(*[p.encode() for p in client_names])
    llr = np.empty(4096)
    text = ct.create_string_buffer(1 << 16)
    _chk(lib.liagpu_compute_test_files(device, world_path.encode(), n, cp, cn, prm_path.encode(), lbl_path.encode(), mask.encode(),
                                       label.encode(), ct.c_double(frame_length), top_c, int(complete), ct.c_double(min_llk),
                                       ct.c_double(max_llk), gender.encode(), test_name.encode(), ct.c_double(threshold),
                                       llr.ctypes.data_as(_dp), ct.c_long(len(llr)), text, ct.c_long(len(text))))
    lines = text.value.decode().strip().split("\n")
    return llr[:len(lines)].reshape(-1, n), lines


def energy_detector(energy, seg_begin, seg_len, C=2, nb_train_it=10, variance_flooring=0.5, variance_ceiling=10.0, alpha=0.25, device=0):
    """EnergyDetector (meanStd mode) on one energy column -> dict(begin, length, w, mean, cov, threshold)."""
    e = np.ascontiguousarray(energy, np.float32)
    b, l, bp, lp = _segs(seg_begin, seg_len)
    ob = np.zeros(4096, np.int64); ol = np.zeros(4096, np.int64); n = ct.c_long(0)
    model = np.zeros(3 * C); th = ct.c_double(0.0)
    _chk(lib.liagpu_energy_detector(device, e.ctypes.data_as(_fp), ct.c_long(len(e)), bp, lp, ct.c_long(len(b)), int(C), int(nb_train_it),
                                    ct.c_double(variance_flooring), ct.c_double(variance_ceiling), ct.c_double(alpha),
                                    ob.ctypes.data_as(_lp), ol.ctypes.data_as(_lp), ct.c_long(len(ob)), ct.byref(n), _d(model), ct.byref(th)))
    return dict(begin=ob[:n.value].copy(), length=ol[:n.value].copy(), w=model[:C].copy(), mean=model[C:2 * C].copy(), cov=model[2 * C:].copy(),
                threshold=th.value)


def select_frames(energy, threshold, seg_begin, seg_len):
    """selectFrames (EnergyDetector.cpp:118-157) on the host: -> (begin[], length[], frames above the threshold)."""
    e = np.ascontiguousarray(energy, np.float32)
    b, l, bp, lp = _segs(seg_begin, seg_len)
    ob = np.zeros(max(16, len(e) + 1), np.int64); ol = np.zeros_like(ob); n = ct.c_long(0); cnt = ct.c_long(0)
    _chk(lib.liagpu_select_frames(e.ctypes.data_as(_fp), ct.c_long(len(e)), ct.c_double(threshold), bp, lp, ct.c_long(len(b)),
                                  ob.ctypes.data_as(_lp), ol.ctypes.data_as(_lp), ct.c_long(len(ob)), ct.byref(n), ct.byref(cnt)))
    return ob[:n.value].copy(), ol[:n.value].copy(), cnt.value


def gmm_tokenizer_files(world_path, prm_path, lbl_path, mask="", label="male", frame_length=0.01, top_c=1, min_llk=-200.0, max_llk=200.0,
                        matrix_path="", device=0):
    """GmmTokenizer from its files (GmmTokenizer.cpp:120-207): -> (symbols[n_selected], confusion[C, C]) with nBest = top_c."""
    dims = np.zeros(3, np.int64)
    sym = np.zeros(1 << 20, np.int64); n = ct.c_long(0)
    _chk(lib.liagpu_gmm_tokenizer_files(device, world_path.encode(), prm_path.encode(), lbl_path.encode(), mask.encode(), label.encode(),
                                        ct.c_double(frame_length), int(top_c), ct.c_double(min_llk), ct.c_double(max_llk),
                                        sym.ctypes.data_as(_lp), ct.c_long(len(sym)), ct.byref(n), None, dims.ctypes.data_as(_lp), b""))
    conf = np.zeros((int(dims[0]), int(dims[0])), np.int64)
    _chk(lib.liagpu_gmm_tokenizer_files(device, world_path.encode(), prm_path.encode(), lbl_path.encode(), mask.encode(), label.encode(),
                                        ct.c_double(frame_length), int(top_c), ct.c_double(min_llk), ct.c_double(max_llk),
                                        None, ct.c_long(0), None, conf.ctypes.data_as(_lp), dims.ctypes.data_as(_lp), matrix_path.encode()))
    return sym[:n.value].copy(), conf


def io_roundtrip(raw_in, raw_out, prm_in, prm_out, mask):
    dims = np.zeros(4, np.int64); mean0 = np.zeros(512); frame0 = np.zeros(512, np.float32)
    _chk(lib.liagpu_io_roundtrip(raw_in.encode(), raw_out.encode(), prm_in.encode(), prm_out.encode(), mask.encode(),
                                 dims.ctypes.data_as(_lp), mean0.ctypes.data_as(_dp), frame0.ctypes.data_as(_fp)))
    return dims, mean0[:dims[1]], frame0[:dims[3]]


def label_segments(lbl_path, label, frame_length=0.01):
    b = np.zeros(1024, np.int64); l = np.zeros(1024, np.int64); n = ct.c_long(0)
    _chk(lib.liagpu_label_segments(lbl_path.encode(), label.encode(), ct.c_double(frame_length), b.ctypes.data_as(_lp),
                                   l.ctypes.data_as(_lp), ct.c_long(1024), ct.byref(n)))
    return b[:n.value].copy(), l[:n.value].copy()
