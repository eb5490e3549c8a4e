"""ctypes front-end to oracle/liboracle.so (the C restatement of the reference arithmetic).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by lia_ral_amd (the product).
"""
import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

c_dp = ct.POINTER(ct.c_double)
c_lp = ct.POINTER(ct.c_long)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib(fast=False):
    name = "liboracle_fast.so" if fast else "liboracle.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        _LIBS[name] = ct.CDLL(path)
    return _LIBS[name]


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_dp)


def _l(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(c_lp)


class Gmm:
    """w[C], mean[C,D], covinv[C,D] (fp64)."""

    def __init__(self, w, mean, covinv):
        self.w = np.ascontiguousarray(w, np.float64)
        self.mean = np.ascontiguousarray(mean, np.float64)
        self.covinv = np.ascontiguousarray(covinv, np.float64)
        self.C, self.D = self.mean.shape

    def args(self):
        return (ct.c_int(self.C), ct.c_int(self.D), self.w.ctypes.data_as(c_dp),
                self.mean.ctypes.data_as(c_dp), self.covinv.ctypes.data_as(c_dp))


def llk(g, x, min_llk=-200.0, max_llk=200.0):
    x, xp = _d(x)
    T = x.shape[0]
    out = np.empty(T)
    _lib().orc_llk(*g.args(), xp, ct.c_long(T), ct.c_double(min_llk), ct.c_double(max_llk),
                   out.ctypes.data_as(c_dp))
    return out


def llk_determine_top(g, x, ctop, complete=True, min_llk=-200.0, max_llk=200.0):
    x, xp = _d(x)
    T = x.shape[0]
    ctop = min(ctop, g.C)
    idx = np.empty((T, ctop), np.int64)
    lk = np.empty((T, ctop))
    nlk = np.empty(T); nw = np.empty(T); out = np.empty(T)
    _lib().orc_llk_determine_top(*g.args(), xp, ct.c_long(T), ct.c_int(ctop), ct.c_int(int(complete)),
                                 ct.c_double(min_llk), ct.c_double(max_llk),
                                 idx.ctypes.data_as(c_lp), lk.ctypes.data_as(c_dp),
                                 nlk.ctypes.data_as(c_dp), nw.ctypes.data_as(c_dp), out.ctypes.data_as(c_dp))
    return dict(idx=idx, lk=lk, nontop_lk=nlk, nontop_w=nw, llk=out)


def llk_use_top(g, x, idx, nontop_lk, complete=True, min_llk=-200.0, max_llk=200.0):
    x, xp = _d(x)
    T = x.shape[0]
    idx, ip = _l(idx)
    nlk, nlp = _d(nontop_lk)
    out = np.empty(T)
    _lib().orc_llk_use_top(*g.args(), xp, ct.c_long(T), ct.c_int(idx.shape[1]), ip, nlp,
                           ct.c_int(int(complete)), ct.c_double(min_llk), ct.c_double(max_llk),
                           out.ctypes.data_as(c_dp))
    return out


def topgauss_compute(g, x, cap, top_gauss, complete=True, min_llk=-200.0, max_llk=200.0):
    """TopGauss::compute (TopGauss.cpp:136-198) -> dict(nbg [T], idx (flat, frame after frame), snsw, snsl, llk)."""
    x, xp = _d(x)
    T = x.shape[0]
    nbg = np.zeros(T, np.int64); idx = np.zeros(T * cap, np.int64)
    snsw = np.empty(T); snsl = np.empty(T); llk_ = np.empty(T)
    f = _lib().orc_topgauss_compute
    f.restype = ct.c_long
    n = f(*g.args(), xp, ct.c_long(T), ct.c_int(cap), ct.c_double(top_gauss), ct.c_int(int(complete)), ct.c_double(min_llk),
          ct.c_double(max_llk), nbg.ctypes.data_as(c_lp), idx.ctypes.data_as(c_lp), snsw.ctypes.data_as(c_dp),
          snsl.ctypes.data_as(c_dp), llk_.ctypes.data_as(c_dp))
    return dict(nbg=nbg, idx=idx[:n].copy(), snsw=snsw, snsl=snsl, llk=llk_)


def topgauss_get(g, x, nbg, idx, snsl, complete=True, min_llk=-200.0, max_llk=200.0):
    """TopGauss::get (TopGauss.cpp:275-316): per-frame llk on the stored selection."""
    x, xp = _d(x)
    T = x.shape[0]
    nbg, np_ = _l(nbg); idx, ip = _l(idx); snsl, sp = _d(snsl)
    out = np.empty(T)
    _lib().orc_topgauss_get(*g.args(), xp, ct.c_long(T), np_, ip, sp, ct.c_int(int(complete)), ct.c_double(min_llk), ct.c_double(max_llk),
                            out.ctypes.data_as(c_dp))
    return out


def occ(g, x):
    x, xp = _d(x)
    T = x.shape[0]
    out = np.empty((T, g.C))
    _lib().orc_occ(*g.args(), xp, ct.c_long(T), out.ctypes.data_as(c_dp))
    return out


def em_accumulate(g, x, weight=1.0, acc=None, fast=False, threads=0):
    """Returns dict(occ, sx, sxx, count, llk) -- accumulating into `acc` when given."""
    x, xp = _d(x)
    T = x.shape[0]
    if acc is None:
        acc = dict(occ=np.zeros(g.C), sx=np.zeros((g.C, g.D)), sxx=np.zeros((g.C, g.D)), count=0.0, llk=0.0)
    cnt = ct.c_double(acc["count"])
    lib = _lib(fast)
    if threads and fast:
        f = lib.orc_em_accumulate_mt
        f.restype = ct.c_double
        r = f(ct.c_int(threads), *g.args(), xp, ct.c_long(T), ct.c_double(weight),
              acc["occ"].ctypes.data_as(c_dp), acc["sx"].ctypes.data_as(c_dp),
              acc["sxx"].ctypes.data_as(c_dp), ct.byref(cnt))
    else:
        f = lib.orc_em_accumulate
        f.restype = ct.c_double
        r = f(*g.args(), xp, ct.c_long(T), ct.c_double(weight),
              acc["occ"].ctypes.data_as(c_dp), acc["sx"].ctypes.data_as(c_dp),
              acc["sxx"].ctypes.data_as(c_dp), ct.byref(cnt))
    acc["count"] = cnt.value
    acc["llk"] += r
    return acc


def em_get(acc, prev_mean, prev_cov):
    C, D = acc["sx"].shape
    w = np.empty(C); mean = np.array(prev_mean, np.float64); cov = np.array(prev_cov, np.float64)
    _lib().orc_em_get(ct.c_int(C), ct.c_int(D), acc["occ"].ctypes.data_as(c_dp),
                      acc["sx"].ctypes.data_as(c_dp), acc["sxx"].ctypes.data_as(c_dp),
                      ct.c_double(acc["count"]), w.ctypes.data_as(c_dp), mean.ctypes.data_as(c_dp),
                      cov.ctypes.data_as(c_dp))
    return w, mean, cov


def set_it_parameter(begin, end, nb_it, it):
    f = _lib().orc_set_it_parameter
    f.restype = ct.c_double
    return f(ct.c_double(begin), ct.c_double(end), ct.c_int(nb_it), ct.c_int(it))


def variance_control(cov, flooring, ceiling, cov_signal):
    cov = np.array(cov, np.float64)
    C, D = cov.shape
    cs, csp = _d(cov_signal)
    nf = ct.c_long(0); nc = ct.c_long(0)
    _lib().orc_variance_control(ct.c_int(C), ct.c_int(D), cov.ctypes.data_as(c_dp),
                                ct.c_double(flooring), ct.c_double(ceiling), csp, ct.byref(nf), ct.byref(nc))
    return cov, nf.value, nc.value


def map_occdep_mean(mean_world, w_ml, mean_ml, frame_count, reg):
    mw, mwp = _d(mean_world); wm, wmp = _d(w_ml); mm, mmp = _d(mean_ml)
    C, D = mw.shape
    out = np.empty((C, D))
    _lib().orc_map_occdep_mean(ct.c_int(C), ct.c_int(D), mwp, wmp, mmp, ct.c_double(frame_count),
                               ct.c_double(reg), out.ctypes.data_as(c_dp))
    return out


MAP_METHODS = {"MAPOccDep": 0, "MAPConst": 1, "MAPConst2": 2, "MAPModelBased": 3}


def compute_map(method, init, client, frame_count, mean=True, var=False, weight=False, reg=(16.0, 16.0, 16.0), alpha_mean=0.75):
    """computeMAP (TrainTools.cpp:543-556): init / client = (w, mean, cov); returns the adapted (w, mean, cov)."""
    w0, m0, c0 = [np.ascontiguousarray(a, np.float64) for a in init]
    w, m, c = [np.array(a, np.float64, order="C", copy=True) for a in client]
    C, D = m0.shape
    r = np.ascontiguousarray(reg, np.float64)
    _lib().orc_compute_map(ct.c_int(MAP_METHODS.get(method, -1)), ct.c_int(C), ct.c_int(D), w0.ctypes.data_as(c_dp), m0.ctypes.data_as(c_dp),
                           c0.ctypes.data_as(c_dp), w.ctypes.data_as(c_dp), m.ctypes.data_as(c_dp), c.ctypes.data_as(c_dp), ct.c_double(frame_count),
                           ct.c_int(int(mean) | (int(var) << 1) | (int(weight) << 2)), r.ctypes.data_as(c_dp), ct.c_double(alpha_mean))
    return w, m, c


def frame_acc(x):
    x, xp = _d(x)
    T, D = x.shape
    s = np.zeros(D); ss = np.zeros(D); n = ct.c_double(0)
    _lib().orc_frame_acc(ct.c_int(D), xp, ct.c_long(T), s.ctypes.data_as(c_dp), ss.ctypes.data_as(c_dp), ct.byref(n))
    return s, ss, n.value


def frame_mean_cov(s, ss, n):
    D = len(s)
    m = np.empty(D); c = np.empty(D)
    _lib().orc_frame_mean_cov(ct.c_int(D), _d(s)[1], _d(ss)[1], ct.c_double(n), m.ctypes.data_as(c_dp), c.ctypes.data_as(c_dp))
    return m, c


def bagged_segments(seed, seg_begin, seg_len, p, min_len=3, max_len=7):
    sb, sbp = _l(seg_begin); sl, slp = _l(seg_len)
    cap = int(sl.sum()) + len(sl) + 8
    ob = np.empty(cap, np.int64); ol = np.empty(cap, np.int64); os_ = np.empty(cap, np.int64)
    f = _lib().orc_bagged_segments
    f.restype = ct.c_long
    n = f(ct.c_uint(seed), sbp, slp, ct.c_long(len(sb)), ct.c_double(p), ct.c_long(min_len), ct.c_long(max_len),
          ob.ctypes.data_as(c_lp), ol.ctypes.data_as(c_lp), os_.ctypes.data_as(c_lp), ct.c_long(cap))
    return ob[:n].copy(), ol[:n].copy(), os_[:n].copy()


def mixture_init_streams(C, xs, segs, weights, nb_frame_to_select=50.0, min_len=3, max_len=7):
    """mixtureInit over several input streams (TrainTools.cpp:674-766): (mean [C x D], frames picked per component)."""
    D = np.asarray(xs[0]).shape[1]
    s = np.zeros((C, D)); cnt = np.zeros(C)
    f = _lib().orc_mixture_init_stream
    for st, (x, (sb, sl)) in enumerate(zip(xs, segs)):
        x, xp = _d(x)
        sb, sbp = _l(sb); sl, slp = _l(sl)
        rc = f(ct.c_long(st), ct.c_int(C), ct.c_int(D), xp, sbp, slp, ct.c_long(len(sb)), ct.c_double(weights[st]),
               ct.c_double(nb_frame_to_select), ct.c_long(min_len), ct.c_long(max_len), s.ctypes.data_as(c_dp), cnt.ctypes.data_as(c_dp))
        assert rc == 0
    return s / cnt[:, None], cnt


def sort_by_weight(w):
    """TabWeight::_sortByWeight (GeneralTools.h:157-164): component indices, heaviest first (libc qsort)."""
    w, wp = _d(w)
    order = np.empty(len(w), np.int64)
    _lib().orc_sort_by_weight(ct.c_int(len(w)), wp, order.ctypes.data_as(c_lp))
    return order


def reduce_model(w, mean, cov, nb_top):
    """selectComponent(nbTop) + reduceModel + normalizeWeights (TrainTools.cpp:197-227)."""
    w = np.array(w, np.float64); mean = np.array(mean, np.float64); cov = np.array(cov, np.float64)
    C, D = mean.shape
    f = _lib().orc_reduce_model
    f.restype = ct.c_int
    n = f(ct.c_int(C), ct.c_int(D), w.ctypes.data_as(c_dp), mean.ctypes.data_as(c_dp), cov.ctypes.data_as(c_dp), ct.c_int(nb_top))
    return w[:n].copy(), mean[:n].copy(), cov[:n].copy()


def normalize_mixture(w, mean, cov, nb_it=1, mean_only=False):
    """normalizeMixture to N(0, 1) (TrainTools.cpp:287-315)."""
    w, wp = _d(w)
    mean = np.array(mean, np.float64); cov = np.array(cov, np.float64)
    C, D = mean.shape
    _lib().orc_normalize_mixture(ct.c_int(C), ct.c_int(D), wp, mean.ctypes.data_as(c_dp), cov.ctypes.data_as(c_dp), ct.c_int(nb_it), ct.c_int(int(mean_only)))
    return mean, cov


def mixture_init(C, x, seg_begin, seg_len, nb_frame_to_select=50.0, stream_weight=1.0, min_len=3, max_len=7):
    """mixtureInit (TrainTools.cpp:674-766, one stream): (mean [C x D], frames picked per component)."""
    x, xp = _d(x)
    D = x.shape[1]
    sb, sbp = _l(seg_begin); sl, slp = _l(seg_len)
    mean = np.empty((C, D)); cnt = np.empty(C)
    rc = _lib().orc_mixture_init(ct.c_int(C), ct.c_int(D), xp, sbp, slp, ct.c_long(len(sb)), ct.c_double(stream_weight),
                                 ct.c_double(nb_frame_to_select), ct.c_long(min_len), ct.c_long(max_len),
                                 mean.ctypes.data_as(c_dp), cnt.ctypes.data_as(c_dp))
    if rc == -2:
        raise ValueError("mixtureInit: the reference's probability fold (TrainTools.cpp:703-706) does not terminate for this nbFrameToSelect")
    assert rc == 0
    return mean, cnt


# ---------------------------------------------------------------- total variability
def tv_stats(g, x, utt, U):
    x, xp = _d(x)
    T = x.shape[0]
    utt, up = _l(utt)
    N = np.zeros((U, g.C)); F = np.zeros((U, g.C * g.D))
    _lib().orc_tv_stats(*g.args(), xp, ct.c_long(T), up, N.ctypes.data_as(c_dp), F.ctypes.data_as(c_dp))
    return N, F


def tv_subtract_m(N, F, means):
    U, C = N.shape
    D = F.shape[1] // C
    F = np.array(F, np.float64)
    _lib().orc_tv_subtract_m(ct.c_long(U), ct.c_int(C), ct.c_int(D), _d(N)[1], F.ctypes.data_as(c_dp), _d(means)[1])
    return F


def tv_tett(Tm, invvar, C, D):
    Tm, tp = _d(Tm)
    R = Tm.shape[0]
    out = np.empty((C, R, R))
    _lib().orc_tv_tett(ct.c_int(C), ct.c_int(D), ct.c_int(R), tp, _d(invvar)[1], out.ctypes.data_as(c_dp))
    return out


def tv_estimate_w(N, F, Tm, invvar, TETt, fast=False, threads=0):
    N, Np = _d(N); F, Fp = _d(F); Tm, tp = _d(Tm); iv, ivp = _d(invvar); TE, tep = _d(TETt)
    U, C = N.shape
    R = Tm.shape[0]
    D = Tm.shape[1] // C
    W = np.zeros((U, R))
    lib = _lib(fast)
    if threads and fast:
        rc = lib.orc_tv_estimate_w_mt(ct.c_int(threads), ct.c_long(U), ct.c_int(C), ct.c_int(D), ct.c_int(R),
                                      Np, Fp, tp, ivp, tep, W.ctypes.data_as(c_dp))
    else:
        rc = lib.orc_tv_estimate_w(ct.c_long(U), ct.c_int(C), ct.c_int(D), ct.c_int(R), Np, Fp, tp, ivp, tep,
                                   W.ctypes.data_as(c_dp))
    assert rc == 0
    return W


def iv_extract_mt(g, x, utt_begin, Tm, invvar, TETt, threads=1):
    """IvExtractor end to end on `threads` threads, utterance ranges per thread like the reference (oracle_mt.c, -O3 -ffast-math
    build: bench.py's cpu_baseline of the i-vector metric).  x [T, D] fp64, utt_begin [U + 1] -> W [U, R]."""
    x, xp = _d(x)
    ub, ubp = _l(utt_begin)
    U = len(ub) - 1
    Tm, tp = _d(Tm); invvar, ip = _d(invvar); TETt, tep = _d(TETt)
    R = Tm.shape[0]
    W = np.empty((U, R))
    f = _lib(True).orc_iv_extract_mt
    f.restype = ct.c_int
    rc = f(ct.c_int(threads), ct.c_long(U), ct.c_int(g.C), ct.c_int(g.D), ct.c_int(R), *g.args()[2:], xp, ubp, tp, ip, tep,
           W.ctypes.data_as(c_dp))
    assert rc == 0
    return W


def tv_estimate_a_and_c(N, F, Tm, invvar, TETt):
    N, Np = _d(N); F, Fp = _d(F); Tm, tp = _d(Tm); iv, ivp = _d(invvar); TE, tep = _d(TETt)
    U, C = N.shape
    R = Tm.shape[0]
    SV = Tm.shape[1]
    D = SV // C
    W = np.zeros((U, R)); A = np.zeros((C, R * R)); Cmx = np.zeros((R, SV))
    Rm = np.zeros((R, R)); r = np.zeros(R); meanW = np.zeros(R)
    rc = _lib().orc_tv_estimate_a_and_c(ct.c_long(U), ct.c_int(C), ct.c_int(D), ct.c_int(R), Np, Fp, tp, ivp, tep,
                                        *[a.ctypes.data_as(c_dp) for a in (W, A, Cmx, Rm, r, meanW)])
    assert rc == 0
    return dict(W=W, A=A, Cmx=Cmx, Rm=Rm, r=r, meanW=meanW)


def tv_update_t(A, Cmx, C, D):
    A, Ap = _d(A); Cmx, Cp = _d(Cmx)
    R = Cmx.shape[0]
    Tm = np.zeros((R, C * D))
    rc = _lib().orc_tv_update_t(ct.c_int(C), ct.c_int(D), ct.c_int(R), Ap, Cp, Tm.ctypes.data_as(c_dp))
    assert rc == 0
    return Tm


def tv_min_divergence(Rm, r, meanW, means, Tm, n_sessions, C, D):
    Rm = np.array(Rm, np.float64); r = np.array(r, np.float64)
    means = np.array(means, np.float64); Tm = np.array(Tm, np.float64)
    R = Tm.shape[0]
    rc = _lib().orc_tv_min_divergence(ct.c_int(C), ct.c_int(D), ct.c_int(R), ct.c_double(n_sessions),
                                      Rm.ctypes.data_as(c_dp), r.ctypes.data_as(c_dp), _d(meanW)[1],
                                      means.ctypes.data_as(c_dp), Tm.ctypes.data_as(c_dp))
    assert rc == 0
    return means, Tm


def tv_em_iteration_mt(N, F, Tm, invvar, means, threads=1, upd_threads=1):
    """One TotalVariability iteration (estimateTETt, estimateAandC, updateTestimate, minDivergence) on `threads` threads with the
    reference's partition (oracle_mt.c, the -O3 -ffast-math build: bench.py's cpu_baseline of the T-matrix workload).  F centred.
    -> dict(T, means, W, phase_s [TETt, estimateAandC, updateTestimate, minDivergence])."""
    N, Np = _d(N); F, Fp = _d(F); iv, ivp = _d(invvar)
    Tm = np.array(Tm, np.float64); means = np.array(means, np.float64).ravel()
    U, C = N.shape
    R = Tm.shape[0]
    D = Tm.shape[1] // C
    W = np.zeros((U, R)); ph = np.zeros(4)
    f = _lib(True).orc_tv_em_iteration_mt
    f.restype = ct.c_int
    rc = f(ct.c_int(threads), ct.c_int(upd_threads), ct.c_long(U), ct.c_int(C), ct.c_int(D), ct.c_int(R), Np, Fp, Tm.ctypes.data_as(c_dp), ivp,
           means.ctypes.data_as(c_dp), W.ctypes.data_as(c_dp), ph.ctypes.data_as(c_dp))
    assert rc == 0, "orc_tv_em_iteration_mt rc %d" % rc
    return dict(T=Tm, means=means, W=W, phase_s=ph)


def tv_init_t(R, invvar, seed=1):
    iv, ivp = _d(invvar)
    Tm = np.empty((R, len(iv)))
    _lib().orc_tv_init_t(ct.c_int(R), ct.c_long(len(iv)), ivp, ct.c_uint(seed), Tm.ctypes.data_as(c_dp))
    return Tm


def tv_orthonormalize_t(Tm):
    Tm = np.array(Tm, np.float64)
    R, SV = Tm.shape
    _lib().orc_tv_orthonormalize_t(ct.c_int(R), ct.c_size_t(SV), Tm.ctypes.data_as(c_dp))
    return Tm


def invert(a):
    a, ap = _d(a)
    n = a.shape[0]
    out = np.empty((n, n))
    rc = _lib().orc_invert(ct.c_int(n), ap, out.ctypes.data_as(c_dp))
    assert rc == 0
    return out


def upper_cholesky(a):
    a, ap = _d(a)
    n = a.shape[0]
    out = np.empty((n, n))
    rc = _lib().orc_upper_cholesky(ct.c_int(n), ap, out.ctypes.data_as(c_dp))
    assert rc == 0
    return out


# ---------------------------------------------------------------- scoring (vectors as columns)
def _trials(tr):
    if tr is None:
        return None, None
    tr = np.ascontiguousarray(tr, np.uint8)
    return tr, tr.ctypes.data_as(ct.POINTER(ct.c_ubyte))


def iv_normalize(X, mean=None, M=None, length_norm=True):
    X, xp = _d(X)
    din, n = X.shape
    dout = M.shape[0] if M is not None else din
    Y = np.empty((dout, n))
    mp = _d(mean)[1] if mean is not None else None
    Mp = _d(M)[1] if M is not None else None
    _lib().orc_iv_normalize(ct.c_int(din), ct.c_int(dout), ct.c_long(n), xp, mp, Mp, ct.c_int(int(length_norm)),
                            Y.ctypes.data_as(c_dp))
    return Y


def score_cosine(models, segs, trials=None):
    m, mp = _d(models); s, sp = _d(segs)
    dim, M = m.shape; S = s.shape[1]
    out = np.zeros((M, S)); tr, trp = _trials(trials)
    _lib().orc_score_cosine(ct.c_int(dim), ct.c_long(M), ct.c_long(S), mp, sp, trp, out.ctypes.data_as(c_dp))
    return out


def score_mahalanobis(models, segs, Mah, trials=None):
    m, mp = _d(models); s, sp = _d(segs)
    dim, M = m.shape; S = s.shape[1]
    out = np.zeros((M, S)); tr, trp = _trials(trials)
    _lib().orc_score_mahalanobis(ct.c_int(dim), ct.c_long(M), ct.c_long(S), mp, sp, _d(Mah)[1], trp,
                                 out.ctypes.data_as(c_dp))
    return out


def score_mahalanobis_mt(models, segs, Mah, threads=1):
    """The per-pair form of mahalanobisDistance (liboracle_fast.so, -O3 -ffast-math) with the model rows split over `threads`
    pthreads -- bench.py's cpu_baseline leg for config 5 only."""
    m, mp = _d(models); s, sp = _d(segs)
    dim, M = m.shape; S = s.shape[1]
    out = np.zeros((M, S))
    f = _lib(True).orc_score_mahalanobis_mt
    f.restype = None
    f(ct.c_int(int(threads)), ct.c_int(dim), ct.c_long(M), ct.c_long(S), mp, sp, _d(Mah)[1], out.ctypes.data_as(c_dp))
    return out


def twocov_model(Wm, Bm):
    Wm, wp = _d(Wm); Bm, bp = _d(Bm)
    n = Wm.shape[0]
    G = np.empty((n, n)); H = np.empty((n, n))
    rc = _lib().orc_twocov_model(ct.c_int(n), wp, bp, G.ctypes.data_as(c_dp), H.ctypes.data_as(c_dp))
    assert rc == 0
    return G, H


def score_twocov(models, segs, G, H):
    m, mp = _d(models); s, sp = _d(segs)
    dim, M = m.shape; S = s.shape[1]
    out = np.zeros((M, S))
    _lib().orc_score_twocov(ct.c_int(dim), ct.c_long(M), ct.c_long(S), mp, sp, _d(G)[1], _d(H)[1],
                            out.ctypes.data_as(c_dp))
    return out


def score_plda(models_sum, nsess, segs, FTJF):
    m, mp = _d(models_sum); s, sp = _d(segs)
    rf, M = m.shape; S = s.shape[1]
    ns, nsp = _l(nsess)
    out = np.zeros((M, S))
    rc = _lib().orc_score_plda(ct.c_int(rf), ct.c_long(M), ct.c_long(S), mp, nsp, sp, _d(FTJF)[1],
                               out.ctypes.data_as(c_dp))
    assert rc == 0
    return out


# ---- approximate extractors / PLDA pre-computation ---------------------------------------------
def tv_norm_statistics(N, F, means, invvar):
    N, Np = _d(N); F = np.array(F, np.float64, order="C", copy=True); m, mp = _d(means); iv, ivp = _d(invvar)
    U, C = N.shape
    D = F.shape[1] // C
    _lib().orc_tv_norm_statistics(ct.c_long(U), ct.c_int(C), ct.c_int(D), Np, F.ctypes.data_as(c_dp), mp, ivp)
    return F


def tv_subtract_m_plus_tw(N, F, means, Tm, W):
    N, Np = _d(N); F = np.array(F, np.float64, order="C", copy=True); m, mp = _d(means); Tm, tp = _d(Tm); W, wp = _d(W)
    U, C = N.shape
    D = F.shape[1] // C
    _lib().orc_tv_subtract_m_plus_tw(ct.c_long(U), ct.c_int(C), ct.c_int(D), ct.c_int(Tm.shape[0]), Np, F.ctypes.data_as(c_dp), mp, tp, wp)
    return F


def tv_norm_t(Tm, invvar, C):
    Tm = np.array(Tm, np.float64, order="C", copy=True); iv, ivp = _d(invvar)
    _lib().orc_tv_norm_t(ct.c_int(C), ct.c_int(Tm.shape[1] // C), ct.c_int(Tm.shape[0]), Tm.ctypes.data_as(c_dp), ivp)
    return Tm


def tv_weighted_cov(Tm, weight):
    Tm, tp = _d(Tm); w, wp = _d(weight)
    R = Tm.shape[0]; C = w.shape[0]
    out = np.zeros((R, R))
    _lib().orc_tv_weighted_cov(ct.c_int(C), ct.c_int(Tm.shape[1] // C), ct.c_int(R), tp, wp, out.ctypes.data_as(c_dp))
    return out


def tv_approximate_tctc(Tm, Q, C):
    Tm, tp = _d(Tm); Q, qp = _d(Q)
    R = Tm.shape[0]
    out = np.zeros((C, R))
    _lib().orc_tv_approximate_tctc(ct.c_int(C), ct.c_int(Tm.shape[1] // C), ct.c_int(R), tp, qp, out.ctypes.data_as(c_dp))
    return out


def tv_estimate_w_ubm_weight(N, F, Tm, Wm):
    N, Np = _d(N); F, Fp = _d(F); Tm, tp = _d(Tm); Wm, wp = _d(Wm)
    U, C = N.shape
    R = Tm.shape[0]
    out = np.zeros((U, R))
    rc = _lib().orc_tv_estimate_w_ubm_weight(ct.c_long(U), ct.c_int(C), ct.c_int(Tm.shape[1] // C), ct.c_int(R), Np, Fp, tp, wp,
                                             out.ctypes.data_as(c_dp))
    assert rc == 0
    return out


def tv_estimate_w_eigen(N, F, Tm, Dm, Q):
    N, Np = _d(N); F, Fp = _d(F); Tm, tp = _d(Tm); Dm, dp = _d(Dm); Q, qp = _d(Q)
    U, C = N.shape
    R = Tm.shape[0]
    out = np.zeros((U, R))
    _lib().orc_tv_estimate_w_eigen(ct.c_long(U), ct.c_int(C), ct.c_int(Tm.shape[1] // C), ct.c_int(R), Np, Fp, tp, dp, qp,
                                   out.ctypes.data_as(c_dp))
    return out


def plda_precompute(F, G, Sigma):
    F, fp = _d(F); Sigma, sp = _d(Sigma)
    dim, rf = F.shape
    if G is None or G.size == 0:
        rg = 0; G = np.zeros((dim, 1)); gp = G.ctypes.data_as(c_dp)
    else:
        G, gp = _d(G); rg = G.shape[1]
    FTJ = np.zeros((rf, dim)); FTJF = np.zeros((rf, rf))
    rc = _lib().orc_plda_precompute(ct.c_int(dim), ct.c_int(rf), ct.c_int(rg), fp, gp, sp, FTJ.ctypes.data_as(c_dp), FTJF.ctypes.data_as(c_dp))
    assert rc == 0
    return FTJ, FTJF


# ---- i-vector back-end estimation (PldaDev) -----------------------------------------------------
def _dev(X, sps):
    X, xp = _d(X); sps, sp = _l(sps)
    return X, xp, sps, sp, X.shape[0], X.shape[1], len(sps)


def dev_means(X, sps):
    X, xp, sps, sp, dim, n, k = _dev(X, sps)
    mean = np.zeros(dim); sm = np.zeros((dim, k))
    _lib().orc_dev_means(ct.c_int(dim), ct.c_long(n), xp, ct.c_long(k), sp, mean.ctypes.data_as(c_dp), sm.ctypes.data_as(c_dp))
    return mean, sm


def dev_cov_mat(X, sps):
    X, xp, sps, sp, dim, n, k = _dev(X, sps)
    S = np.zeros((dim, dim)); W = np.zeros((dim, dim)); B = np.zeros((dim, dim))
    _lib().orc_dev_cov_mat(ct.c_int(dim), ct.c_long(n), xp, ct.c_long(k), sp, S.ctypes.data_as(c_dp), W.ctypes.data_as(c_dp), B.ctypes.data_as(c_dp))
    return S, W, B


def dev_wccn_chol(X, sps):
    X, xp, sps, sp, dim, n, k = _dev(X, sps)
    out = np.zeros((dim, dim))
    assert _lib().orc_dev_wccn_chol(ct.c_int(dim), ct.c_long(n), xp, ct.c_long(k), sp, out.ctypes.data_as(c_dp)) == 0
    return out


def dev_scatter_mat(X, sps):
    X, xp, sps, sp, dim, n, k = _dev(X, sps)
    SB = np.zeros((dim, dim)); SW = np.zeros((dim, dim))
    _lib().orc_dev_scatter_mat(ct.c_int(dim), ct.c_long(n), xp, ct.c_long(k), sp, SB.ctypes.data_as(c_dp), SW.ctypes.data_as(c_dp))
    return SB, SW


def sym_eigen(A, rank=None):
    A, ap = _d(A)
    n = A.shape[0]; rank = n if rank is None else rank
    vect = np.zeros((n, rank)); val = np.zeros(rank)
    _lib().orc_sym_eigen(ct.c_int(n), ap, ct.c_int(rank), vect.ctypes.data_as(c_dp), val.ctypes.data_as(c_dp))
    return vect, val


def dev_efr_matrix(Cov):
    Cov, cp = _d(Cov)
    out = np.zeros_like(Cov)
    _lib().orc_dev_efr_matrix(ct.c_int(Cov.shape[0]), cp, out.ctypes.data_as(c_dp))
    return out


def dev_lda(W, B, rank):
    W, wp = _d(W); B, bp = _d(B)
    dim = W.shape[0]
    out = np.zeros((rank, dim)); val = np.zeros(rank)
    assert _lib().orc_dev_lda(ct.c_int(dim), wp, bp, ct.c_int(rank), out.ctypes.data_as(c_dp), val.ctypes.data_as(c_dp)) == 0
    return out, val


def plda_em_iteration(X, sps, F, G, Sigma, Delta):
    """One PldaModel::em_iteration; returns updated copies (X centred by the incoming Delta)."""
    X = np.array(X, np.float64, order="C", copy=True); F = np.array(F, np.float64, order="C", copy=True)
    G = np.array(G, np.float64, order="C", copy=True); Sigma = np.array(Sigma, np.float64, order="C", copy=True)
    Delta = np.array(Delta, np.float64, order="C", copy=True)
    sps, sp = _l(sps)
    dim, n = X.shape
    rc = _lib().orc_plda_em_iteration(ct.c_int(dim), ct.c_long(n), X.ctypes.data_as(c_dp), ct.c_long(len(sps)), sp, ct.c_int(F.shape[1]),
                                      ct.c_int(G.shape[1]), F.ctypes.data_as(c_dp), G.ctypes.data_as(c_dp), Sigma.ctypes.data_as(c_dp),
                                      Delta.ctypes.data_as(c_dp))
    assert rc == 0
    return X, F, G, Sigma, Delta


# ---- JFA (AccumulateJFAStat.cpp) -----------------------------------------------------------------
def _opt(a):
    if a is None:
        return None, None
    return _d(a)


def jfa_estimate_y_and_v(N, F, V, invvar, VEVT):
    """estimateAndInverseL_E{V,C} + estimate{YandV,XandU}; VEVT [C, R, R] full. Returns Y, A [C, R, R], Cmx [R, SV]."""
    N, Np = _d(N); F, Fp = _d(F); V, Vp = _d(V); iv, ivp = _d(invvar); VEVT, vp = _d(VEVT)
    U, C = N.shape; R, SV = V.shape; D = SV // C
    Y = np.zeros((U, R)); A = np.zeros((C, R, R)); Cm = np.zeros((R, SV))
    f = _lib().orc_jfa_estimate_y_and_v; f.restype = ct.c_int
    rc = f(ct.c_long(U), ct.c_int(C), ct.c_int(D), ct.c_int(R), Np, Fp, Vp, ivp, vp, Y.ctypes.data_as(c_dp), A.ctypes.data_as(c_dp),
           Cm.ctypes.data_as(c_dp))
    if rc:
        raise RuntimeError("orc_jfa_estimate_y_and_v: singular L")
    return Y, A, Cm


def jfa_subtract(N, F, owner=None, means=None, T=None, W=None, Dm=None, Z=None):
    N, Np = _d(N); F = np.array(F, np.float64, order="C", copy=True)
    rows, C = N.shape; D = F.shape[1] // C
    m, mp = _opt(means); T_, tp = _opt(T); W_, wp = _opt(W); D_, dp = _opt(Dm); Z_, zp = _opt(Z)
    o = None if owner is None else np.ascontiguousarray(owner, np.int64)
    _lib().orc_jfa_subtract(ct.c_long(rows), ct.c_int(C), ct.c_int(D), Np, F.ctypes.data_as(c_dp),
                            None if o is None else o.ctypes.data_as(c_lp), mp, ct.c_int(0 if T_ is None else T_.shape[0]), tp, wp, dp, zp)
    return F


def jfa_subtract_sessions(sess_begin, N_h, F_X, U, X):
    N_h, np_ = _d(N_h); F_X = np.array(F_X, np.float64, order="C", copy=True); U, up = _d(U); X, xp = _d(X)
    sb = np.ascontiguousarray(sess_begin, np.int64)
    C = N_h.shape[1]; D = F_X.shape[1] // C
    _lib().orc_jfa_subtract_sessions(ct.c_long(len(sb) - 1), sb.ctypes.data_as(c_lp), ct.c_int(C), ct.c_int(D), np_,
                                     F_X.ctypes.data_as(c_dp), ct.c_int(U.shape[0]), up, xp)
    return F_X


def jfa_estimate_z(N, F, invvar, Dm, tau=-1.0):
    N, Np = _d(N); F, Fp = _d(F); iv, ivp = _d(invvar); Dm, dp = _d(Dm)
    U, C = N.shape; D = F.shape[1] // C
    Z = np.zeros_like(F)
    _lib().orc_jfa_estimate_z(ct.c_long(U), ct.c_int(C), ct.c_int(D), Np, Fp, ivp, dp, ct.c_double(tau), Z.ctypes.data_as(c_dp))
    return Z


def jfa_estimate_z_and_d(N, F, invvar, Dm):
    N, Np = _d(N); F, Fp = _d(F); iv, ivp = _d(invvar); Dn = np.array(Dm, np.float64, order="C", copy=True)
    U, C = N.shape; D = F.shape[1] // C
    Z = np.zeros_like(F)
    _lib().orc_jfa_estimate_z_and_d(ct.c_long(U), ct.c_int(C), ct.c_int(D), Np, Fp, ivp, Dn.ctypes.data_as(c_dp), Z.ctypes.data_as(c_dp))
    return Z, Dn


def jfa_subtract_m_plus_ux(sess_begin, N_h, F_X, means, U, X):
    N_h, np_ = _d(N_h); F_X = np.array(F_X, np.float64, order="C", copy=True); m, mp = _d(means); U, up = _d(U); X, xp = _d(X)
    sb = np.ascontiguousarray(sess_begin, np.int64)
    C = N_h.shape[1]; D = F_X.shape[1] // C
    _lib().orc_jfa_subtract_m_plus_ux(ct.c_long(len(sb) - 1), sb.ctypes.data_as(c_lp), ct.c_int(C), ct.c_int(D), np_,
                                      F_X.ctypes.data_as(c_dp), mp, ct.c_int(U.shape[0]), up, xp)
    return F_X

