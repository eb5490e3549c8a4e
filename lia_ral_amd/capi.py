"""ctypes binding of include/gmmiv.h.  Arrays may be numpy arrays (host) or torch CUDA tensors
(device, used in place).  Raises GmmivError on any non-zero status -- never falls back to a CPU path."""
import ctypes as ct
import os
import sys

import numpy as np

try:  # PyTorch-ROCm bundles its own HIP runtime: load it first so the process holds ONE runtime
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the binding itself
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
# GMMIV_LIB_PATH: development knob (tools/k1_ablate.sh times instrumented builds of the same library); still libgmmiv, never a fallback
LIB_PATH = os.environ.get("GMMIV_LIB_PATH") or os.path.join(_HERE, "csrc", "libgmmiv.so")

F32, F64 = 0, 1
TOP_PARTIAL, TOP_COMPLETE = 0, 1


class GmmivError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise GmmivError("libgmmiv.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "or `make -C lia_ral_amd/csrc`" % LIB_PATH)
    lib = ct.CDLL(LIB_PATH)
    lib.gmmiv_last_error.restype = ct.c_char_p
    lib.gmmiv_version.restype = ct.c_char_p
    lib.gmmiv_em_acc_len.restype = ct.c_size_t
    lib.gmmiv_tv_packed_len.restype = ct.c_size_t
    lib.gmmiv_ctx_last_kernel_ms.restype = ct.c_double
    lib.gmmiv_ctx_kernel_ms.restype = ct.c_double
    lib.gmmiv_ctx_kernel_launches.restype = ct.c_long
    lib.gmmiv_ctx_set_option.restype = ct.c_long
    lib.gmmiv_comm_backend.restype = ct.c_char_p
    lib.gmmiv_comm_take_bytes.restype = ct.c_double
    lib.gmmiv_ctx_stream.restype = ct.c_void_p
    return lib


lib = _load()


def _chk(rc):
    if rc != 0:
        raise GmmivError("gmmiv error %d: %s" % (rc, lib.gmmiv_last_error().decode()))


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _ptr(a):
    """void* of a numpy array or torch tensor (None -> NULL)."""
    if a is None:
        return ct.c_void_p(0)
    if _is_torch(a):
        assert a.is_contiguous() or (a.dim() == 2 and a.stride(1) == 1)  # feature matrices may be row-strided (ldx)
        return ct.c_void_p(a.data_ptr())
    assert a.flags["C_CONTIGUOUS"]
    return ct.c_void_p(a.ctypes.data)


def _f64(a):
    if a is None or _is_torch(a):
        return a
    return np.ascontiguousarray(a, dtype=np.float64)


def _feat(x):
    """-> (array, dtype code, T, ldx)"""
    if _is_torch(x):
        import torch
        assert x.dim() == 2 and x.stride(1) == 1
        dt = F64 if x.dtype == torch.float64 else F32
        assert x.dtype in (torch.float32, torch.float64)
        return x, dt, x.shape[0], x.stride(0) if x.shape[0] > 1 else x.shape[1]
    x = np.asarray(x)
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    x = np.ascontiguousarray(x)
    return x, (F64 if x.dtype == np.float64 else F32), x.shape[0], x.shape[1]


STREAM_DEFAULT = -1        # GMMIV_STREAM_DEFAULT: launch on the NULL (legacy default) stream, torch's default stream


class Context:
    """stream: a hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream) or None for a private non-blocking stream.
    The handle 0 -- what torch reports for its DEFAULT stream -- means "the stream torch is using", so it is passed on as
    GMMIV_STREAM_DEFAULT: every call of the context is then ordered with the torch kernels around it (a private stream would not be).
    torch_stream(): the same stream as a torch object, for `with torch.cuda.stream(ctx.torch_stream()):`."""

    def __init__(self, device=0, stream=None):
        self._h = ct.c_void_p()
        self.device = int(device)
        if stream is not None and int(stream) == 0:
            stream = STREAM_DEFAULT
        _chk(lib.gmmiv_ctx_create(ct.c_int(device), ct.c_void_p(stream or 0), ct.byref(self._h)))

    def stream(self):
        return int(lib.gmmiv_ctx_stream(self._h) or 0)

    def ordered(self):
        """Context manager: the context's stream waits for torch's current stream on entry, torch's current stream waits for
        the context's on exit -- calls made inside see every tensor torch has written and torch sees their results.  A no-op
        when both are the same stream."""
        import contextlib
        import torch
        cur = torch.cuda.current_stream(self.device)
        mine = self.torch_stream()
        if cur.cuda_stream == mine.cuda_stream:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def bracket():
            mine.wait_stream(cur)
            try:
                yield
            finally:
                cur.wait_stream(mine)
        return bracket()

    def torch_stream(self):
        import torch
        h = self.stream()
        if h == 0:         # the NULL stream
            return torch.cuda.default_stream(self.device)
        return torch.cuda.ExternalStream(h, device=self.device)

    def close(self):
        if self._h:
            lib.gmmiv_ctx_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            if sys is None or sys.is_finalizing():   # the HIP runtime may already be gone at interpreter exit
                return
            self.close()
        except Exception:
            pass

    def kernel_ms(self, name):
        return lib.gmmiv_ctx_kernel_ms(self._h, name.encode())

    def kernel_launches(self, name):
        return lib.gmmiv_ctx_kernel_launches(self._h, name.encode())

    def sync(self):
        _chk(lib.gmmiv_ctx_sync(self._h))

    def set_option(self, key, value):
        return lib.gmmiv_ctx_set_option(self._h, key.encode(), ct.c_long(value))

    def set_hook(self, point, fn):
        """gmmiv_ctx_set_hook: `fn()` is called on the host at `point` ("tv_a_ready": inside tv_estimate_a_and_c once A is
        complete and before the Cmx GEMM is enqueued; "md_factored": inside tv_min_divergence after R is factored, before T is
        read).  fn = None removes the hook.  Exceptions raised by fn are kept and re-raised by the next checked call."""
        if not hasattr(self, "_hooks"):
            self._hooks = {}
        if fn is None:
            self._hooks.pop(point, None)
            rc = lib.gmmiv_ctx_set_hook(self._h, point.encode(), None, None)
        else:
            def tramp(_user, fn=fn):
                try:
                    fn()
                except BaseException as e:      # noqa: BLE001 - a C frame is below us: park it
                    self._hook_error = e
            cb = _HOOK_T(tramp)
            self._hooks[point] = cb             # keep the trampoline alive as long as it is installed
            rc = lib.gmmiv_ctx_set_hook(self._h, point.encode(), ct.cast(cb, ct.c_void_p), None)
        if rc != 0:
            raise GmmivError("unknown hook point %r" % point)

    def _raise_hook_error(self):
        e = getattr(self, "_hook_error", None)
        if e is not None:
            self._hook_error = None
            raise e

    def last_kernel_ms(self):
        name = ct.c_char_p()
        ms = lib.gmmiv_ctx_last_kernel_ms(self._h, ct.byref(name))
        return ms, (name.value.decode() if name.value else "")

    # ---- model
    def gmm(self, w, mean, covinv):
        return Gmm(self, w, mean, covinv)

    # ---- FrameAccGD
    def frame_moments(self, x, acc=None):
        x, dt, T, ldx = _feat(x)
        D = x.shape[1]
        if acc is None:
            acc = np.zeros(2 * D + 1)
        _chk(lib.gmmiv_frame_moments(self._h, _ptr(x), dt, ct.c_int64(T), ct.c_int64(ldx), D, _ptr(acc)))
        return acc

    # ---- frame selection on the device (x, out: torch CUDA tensors)
    def gather_frames(self, x, frame_idx, out):
        x, dt, T, ldx = _feat(x)
        idx = frame_idx if _is_torch(frame_idx) else np.ascontiguousarray(frame_idx, np.int64)
        _chk(lib.gmmiv_gather_frames(self._h, _ptr(x), dt, ct.c_int64(ldx), x.shape[1], _ptr(idx), ct.c_int64(idx.shape[0]), _ptr(out)))
        return out

    def gather_runs(self, x, runs, out):
        """runs [nrun, 3] int64 (source frame, output row, length), host or device."""
        x, dt, T, ldx = _feat(x)
        r = runs if _is_torch(runs) else np.ascontiguousarray(runs, np.int64)
        _chk(lib.gmmiv_gather_runs(self._h, _ptr(x), dt, ct.c_int64(ldx), x.shape[1], _ptr(r), ct.c_int64(r.shape[0]), _ptr(out)))
        return out

    def segment_means(self, v, seg_begin, out=None):
        """v: torch CUDA float64 [nrows, ld] (or 1-D); seg_begin: nseg + 1 host offsets -> out [nrows, nseg]."""
        v2 = v if v.dim() == 2 else v.reshape(1, -1)
        sb = np.ascontiguousarray(seg_begin, np.int64)
        if out is None:
            out = np.empty((v2.shape[0], len(sb) - 1))
        _chk(lib.gmmiv_segment_means(self._h, _ptr(v2), ct.c_int64(v2.stride(0)), v2.shape[0], _ptr(sb), ct.c_int64(len(sb) - 1), _ptr(out)))
        return out

    def variance_control(self, cov, flooring, ceiling, cov_signal, C, D, count=True):
        counts = np.zeros(2, np.int64) if count else None
        _chk(lib.gmmiv_variance_control(self._h, C, D, _ptr(cov), ct.c_double(flooring), ct.c_double(ceiling),
                                        _ptr(_f64(cov_signal)), _ptr(counts)))
        return cov, counts

    # ---- TVAcc maths
    def tv_subtract_m(self, N, F, means, C, D):
        U = N.shape[0]
        _chk(lib.gmmiv_tv_subtract_m(self._h, ct.c_int64(U), C, D, _ptr(N), _ptr(F), _ptr(means)))
        return F

    def tv_subtract_m_to(self, N, F_src, F_dst, means, C, D):
        """F_dst = F_src - N means (restoreStats + substractM in one pass); F_dst may be F_src."""
        _chk(lib.gmmiv_tv_subtract_m_to(self._h, ct.c_int64(N.shape[0]), C, D, _ptr(N), _ptr(F_src), _ptr(F_dst), _ptr(means)))
        return F_dst

    def tv_tett(self, Tm, invvar, C, D, out=None):
        R = Tm.shape[0]
        if out is None:
            out = np.empty((C, lib.gmmiv_tv_packed_len(R)))
        _chk(lib.gmmiv_tv_tett(self._h, C, D, R, _ptr(Tm), _ptr(invvar), _ptr(out)))
        return out

    def tv_estimate_w(self, N, F, Tm, invvar, tett, C, D, out=None):
        U, R = N.shape[0], Tm.shape[0]
        if out is None:
            out = np.empty((U, R))
        _chk(lib.gmmiv_tv_estimate_w(self._h, ct.c_int64(U), C, D, R, _ptr(N), _ptr(F), _ptr(Tm), _ptr(invvar),
                                     _ptr(tett), _ptr(out)))
        return out

    # ---- approximate extractors (IvExtractor modes ubmWeight / eigenDecomposition); in-place on numpy / torch arrays
    def tv_norm_statistics(self, N, F, means, invvar, C, D):
        _chk(lib.gmmiv_tv_norm_statistics(self._h, ct.c_int64(N.shape[0]), C, D, _ptr(N), _ptr(F), _ptr(means), _ptr(invvar)))
        return F

    def tv_subtract_m_plus_tw(self, N, F, means, Tm, W, C, D):
        _chk(lib.gmmiv_tv_subtract_m_plus_tw(self._h, ct.c_int64(N.shape[0]), C, D, Tm.shape[0], _ptr(N), _ptr(F), _ptr(means),
                                             _ptr(Tm), _ptr(W)))
        return F

    # ---- JFA (gmmiv_jfa_*; the factor steps are tv_tett / tv_estimate_a_and_c / tv_estimate_w / tv_update_t) ----
    def jfa_subtract(self, N, F, C, D, owner=None, nfact=None, means=None, T=None, W=None, Dm=None, Z=None):
        rows = N.shape[0]
        if nfact is None:
            nfact = W.shape[0] if W is not None else (Z.shape[0] if Z is not None else rows)
        o = None if owner is None else (owner if _is_torch(owner) else np.ascontiguousarray(owner, np.int64))
        _chk(lib.gmmiv_jfa_subtract(self._h, ct.c_int64(rows), C, D, _ptr(N), _ptr(F), _ptr(o), ct.c_int64(nfact), _ptr(means),
                                    0 if T is None else T.shape[0], _ptr(T), _ptr(W), _ptr(Dm), _ptr(Z)))
        return F

    def jfa_subtract_sessions(self, sess_begin, N_h, F_X, U, X, C, D):
        sb = np.ascontiguousarray(sess_begin, np.int64)
        _chk(lib.gmmiv_jfa_subtract_sessions(self._h, ct.c_int64(len(sb) - 1), _ptr(sb), C, D, _ptr(N_h), _ptr(F_X), U.shape[0], _ptr(U), _ptr(X)))
        return F_X

    def jfa_estimate_z(self, N, F, invvar, Dm, C, D, tau=-1.0, out=None):
        if out is None:
            out = np.empty((N.shape[0], C * D))
        _chk(lib.gmmiv_jfa_estimate_z(self._h, ct.c_int64(N.shape[0]), C, D, _ptr(N), _ptr(F), _ptr(invvar), _ptr(Dm), ct.c_double(tau), _ptr(out)))
        return out

    def jfa_estimate_z_and_d(self, N, F, invvar, Dm, C, D, out=None):
        """Dm is updated in place; returns Z."""
        if out is None:
            out = np.empty((N.shape[0], C * D))
        _chk(lib.gmmiv_jfa_estimate_z_and_d(self._h, ct.c_int64(N.shape[0]), C, D, _ptr(N), _ptr(F), _ptr(invvar), _ptr(Dm), _ptr(out)))
        return out

    def tv_norm_t(self, Tm, invvar, C, D):
        _chk(lib.gmmiv_tv_norm_t(self._h, C, D, Tm.shape[0], _ptr(Tm), _ptr(invvar)))
        return Tm

    def tv_weighted_cov(self, Tm, weight, C, D, out=None):
        R = Tm.shape[0]
        if out is None:
            out = np.empty((R, R))
        _chk(lib.gmmiv_tv_weighted_cov(self._h, C, D, R, _ptr(Tm), _ptr(weight), _ptr(out)))
        return out

    def tv_approximate_tctc(self, Tm, Q, C, D, out=None):
        R = Tm.shape[0]
        if out is None:
            out = np.zeros((C, R))
        _chk(lib.gmmiv_tv_approximate_tctc(self._h, C, D, R, _ptr(Tm), _ptr(Q), _ptr(out)))
        return out

    def tv_estimate_w_ubm_weight(self, N, F, Tm, Wm, C, D, out=None):
        U, R = N.shape[0], Tm.shape[0]
        if out is None:
            out = np.zeros((U, R))
        _chk(lib.gmmiv_tv_estimate_w_ubm_weight(self._h, ct.c_int64(U), C, D, R, _ptr(N), _ptr(F), _ptr(Tm), _ptr(Wm), _ptr(out)))
        return out

    def tv_estimate_w_eigen(self, N, F, Tm, Dm, Q, C, D, out=None):
        U, R = N.shape[0], Tm.shape[0]
        if out is None:
            out = np.zeros((U, R))
        _chk(lib.gmmiv_tv_estimate_w_eigen(self._h, ct.c_int64(U), C, D, R, _ptr(N), _ptr(F), _ptr(Tm), _ptr(Dm), _ptr(Q), _ptr(out)))
        return out

    # ---- PldaDev: back-end estimation on a development set X[dim, n], sessions grouped by speaker
    def _dev_args(self, X, sps):
        sps = np.ascontiguousarray(sps, dtype=np.int64)
        return X.shape[0], ct.c_int64(X.shape[1]), _ptr(X), ct.c_int64(len(sps)), sps.ctypes.data_as(ct.c_void_p), sps

    def dev_means(self, X, sps):
        dim, n, xp, k, sp, keep = self._dev_args(X, sps)
        mean = np.empty(dim); sm = np.empty((dim, len(keep)))
        _chk(lib.gmmiv_dev_means(self._h, dim, n, xp, k, sp, _ptr(mean), _ptr(sm)))
        return mean, sm

    def dev_cov_mat(self, X, sps):
        dim, n, xp, k, sp, keep = self._dev_args(X, sps)
        S = np.empty((dim, dim)); W = np.empty((dim, dim)); B = np.empty((dim, dim))
        _chk(lib.gmmiv_dev_cov_mat(self._h, dim, n, xp, k, sp, _ptr(S), _ptr(W), _ptr(B)))
        return S, W, B

    def dev_wccn_chol(self, X, sps):
        dim, n, xp, k, sp, keep = self._dev_args(X, sps)
        out = np.empty((dim, dim))
        _chk(lib.gmmiv_dev_wccn_chol(self._h, dim, n, xp, k, sp, _ptr(out)))
        return out

    def dev_mahalanobis(self, X, sps):
        dim, n, xp, k, sp, keep = self._dev_args(X, sps)
        out = np.empty((dim, dim))
        _chk(lib.gmmiv_dev_mahalanobis(self._h, dim, n, xp, k, sp, _ptr(out)))
        return out

    def dev_scatter_mat(self, X, sps):
        dim, n, xp, k, sp, keep = self._dev_args(X, sps)
        SB = np.empty((dim, dim)); SW = np.empty((dim, dim))
        _chk(lib.gmmiv_dev_scatter_mat(self._h, dim, n, xp, k, sp, _ptr(SB), _ptr(SW)))
        return SB, SW

    def sym_eigen(self, A, rank=None):
        n = A.shape[0]; rank = n if rank is None else rank
        vect = np.empty((n, rank)); val = np.empty(rank)
        _chk(lib.gmmiv_sym_eigen(self._h, n, _ptr(_f64(A)), rank, _ptr(vect), _ptr(val)))
        return vect, val

    def dev_efr_matrix(self, Cov):
        out = np.empty_like(Cov)
        _chk(lib.gmmiv_dev_efr_matrix(self._h, Cov.shape[0], _ptr(_f64(Cov)), _ptr(out)))
        return out

    def dev_lda(self, W, B, rank):
        dim = W.shape[0]
        out = np.empty((rank, dim)); val = np.empty(rank)
        _chk(lib.gmmiv_dev_lda(self._h, dim, _ptr(_f64(W)), _ptr(_f64(B)), rank, _ptr(out), _ptr(val)))
        return out, val

    def plda_em_iteration(self, X, sps, F, G, Sigma, Delta):
        """One PldaModel::em_iteration, in place on X (centred by Delta), F, G, Sigma, Delta (numpy float64 arrays)."""
        dim, n, xp, k, sp, keep = self._dev_args(X, sps)
        _chk(lib.gmmiv_plda_em_iteration(self._h, dim, n, xp, k, sp, F.shape[1], G.shape[1], _ptr(F), _ptr(G), _ptr(Sigma), _ptr(Delta)))
        return X, F, G, Sigma, Delta

    def twocov_model(self, W, B):
        dim = W.shape[0]
        G = np.empty((dim, dim)); H = np.empty((dim, dim))
        _chk(lib.gmmiv_twocov_model(self._h, dim, _ptr(_f64(W)), _ptr(_f64(B)), _ptr(G), _ptr(H)))
        return G, H

    def plda_precompute(self, F, G, Sigma):
        """-> (FTJ [rf x dim], FTJF [rf x rf]); G may be None."""
        dim, rf = F.shape
        rg = 0 if G is None else G.shape[1]
        FTJ = np.empty((rf, dim)); FTJF = np.empty((rf, rf))
        _chk(lib.gmmiv_plda_precompute(self._h, dim, rf, rg, _ptr(_f64(F)), _ptr(_f64(G)), _ptr(_f64(Sigma)), _ptr(FTJ), _ptr(FTJF)))
        return FTJ, FTJF

    def tv_estimate_a_and_c(self, N, F, Tm, invvar, tett, C, D, acc=None):
        U, R = N.shape[0], Tm.shape[0]
        P = lib.gmmiv_tv_packed_len(R)
        if acc is None:
            acc = dict(A=np.zeros((C, P)), Cmx=np.zeros((R, C * D)), Rm=np.zeros((R, R)), r=np.zeros(R),
                       meanW=np.zeros(R))
        W = acc.get("W")
        if W is None or W.shape[0] != U:
            W = np.empty((U, R))
        self._hook_error = None
        rc = lib.gmmiv_tv_estimate_a_and_c(self._h, ct.c_int64(U), C, D, R, _ptr(N), _ptr(F), _ptr(Tm), _ptr(invvar),
                                           _ptr(tett), _ptr(W), _ptr(acc["A"]), _ptr(acc["Cmx"]), _ptr(acc["Rm"]),
                                           _ptr(acc["r"]), _ptr(acc["meanW"]))
        self._raise_hook_error()        # an exception parked by the hook comes first: it is the cause, and it never outlives this call
        _chk(rc)
        acc["W"] = W
        return acc

    def tv_update_t(self, A_packed, Cmx, C, D, out=None):
        R = Cmx.shape[0]
        if out is None:
            out = np.empty((R, C * D))
        _chk(lib.gmmiv_tv_update_t(self._h, C, D, R, _ptr(A_packed), _ptr(Cmx), _ptr(out)))
        return out

    def tv_min_divergence(self, Rm, r, meanW, means, Tm, n_sessions, C, D):
        R = Tm.shape[0]
        self._hook_error = None
        rc = lib.gmmiv_tv_min_divergence(self._h, C, D, R, ct.c_double(n_sessions), _ptr(Rm), _ptr(r), _ptr(meanW),
                                         _ptr(means), _ptr(Tm))
        self._raise_hook_error()
        _chk(rc)
        return means, Tm

    def tv_orthonormalize_t(self, Tm):
        R, SV = Tm.shape
        _chk(lib.gmmiv_tv_orthonormalize_t(self._h, R, ct.c_int64(SV), _ptr(Tm)))
        return Tm

    def iv_normalize(self, X, mean=None, M=None, length_norm=True, out=None):
        dim_in, n = X.shape
        dim_out = M.shape[0] if M is not None else dim_in
        if out is None:
            out = np.empty((dim_out, n))
        _chk(lib.gmmiv_iv_normalize(self._h, dim_in, dim_out, ct.c_int64(n), _ptr(X), _ptr(_f64(mean)), _ptr(_f64(M)),
                                    int(bool(length_norm)), _ptr(out)))
        return out

    # ---- scoring (vectors as columns: models[dim, M], segs[dim, S])
    def _score_out(self, models, segs, out):
        M, S = models.shape[1], segs.shape[1]
        if out is None:
            out = np.empty((M, S))
        return M, S, out

    def score_cosine(self, models, segs, out=None):
        M, S, out = self._score_out(models, segs, out)
        _chk(lib.gmmiv_score_cosine(self._h, models.shape[0], ct.c_int64(M), ct.c_int64(S), _ptr(models), _ptr(segs),
                                    _ptr(out)))
        return out

    def score_mahalanobis(self, models, segs, Mah, out=None):
        M, S, out = self._score_out(models, segs, out)
        _chk(lib.gmmiv_score_mahalanobis(self._h, models.shape[0], ct.c_int64(M), ct.c_int64(S), _ptr(models),
                                         _ptr(segs), _ptr(Mah), _ptr(out)))
        return out

    def score_twocov(self, models, segs, G, H, out=None):
        M, S, out = self._score_out(models, segs, out)
        _chk(lib.gmmiv_score_twocov(self._h, models.shape[0], ct.c_int64(M), ct.c_int64(S), _ptr(models), _ptr(segs),
                                    _ptr(G), _ptr(H), _ptr(out)))
        return out

    def score_twocov_mix_part(self, models, segs, G, scores):
        """scores += (m + s)^T G (m + s) (PldaTest::twoCovScoringMixPart); scores is updated in place."""
        M, S = models.shape[1], segs.shape[1]
        _chk(lib.gmmiv_score_twocov_mix_part(self._h, models.shape[0], ct.c_int64(M), ct.c_int64(S), _ptr(models), _ptr(segs), _ptr(G),
                                             _ptr(scores)))
        return scores

    def score_apply_trials(self, trials, scores, fill=0.0):
        """scores[m, s] = fill where trials[m, s] == 0 (PldaTest::_trials); in place."""
        M, S = scores.shape
        t = trials if _is_torch(trials) else np.ascontiguousarray(trials, np.uint8)
        _chk(lib.gmmiv_score_apply_trials(self._h, ct.c_int64(M), ct.c_int64(S), _ptr(t), ct.c_double(fill), _ptr(scores)))
        return scores

    def score_plda(self, models_sum, nsess, segs, FTJF, out=None):
        M, S, out = self._score_out(models_sum, segs, out)
        ns = np.ascontiguousarray(nsess, dtype=np.int64)
        _chk(lib.gmmiv_score_plda(self._h, models_sum.shape[0], ct.c_int64(M), ct.c_int64(S), _ptr(models_sum),
                                  ns.ctypes.data_as(ct.c_void_p), _ptr(segs), _ptr(FTJF), _ptr(out)))
        return out


COMM_ID_BYTES = 128
_HOOK_T = ct.CFUNCTYPE(None, ct.c_void_p)
lib.gmmiv_ctx_set_hook.argtypes = [ct.c_void_p, ct.c_char_p, ct.c_void_p, ct.c_void_p]


class Comm:
    """gmmiv_comm: the RCCL communicator of one context (one rank per GPU).  world == 1 needs no id and no RCCL.
    Buffers: torch CUDA tensors (float64, contiguous) are used in place; numpy arrays are accepted by allreduce / broadcast."""

    def __init__(self, ctx, world=1, rank=0, uid=None):
        self.ctx, self.world, self.rank = ctx, int(world), int(rank)
        self._h = ct.c_void_p()
        if world > 1 and (uid is None or len(uid) != COMM_ID_BYTES):
            raise GmmivError("Comm: world > 1 needs the %d-byte id of rank 0 (Comm.unique_id())" % COMM_ID_BYTES)
        buf = ct.create_string_buffer(bytes(uid), COMM_ID_BYTES) if uid is not None else None
        _chk(lib.gmmiv_comm_create(ctx._h, self.world, self.rank, buf, ct.byref(self._h)))

    @staticmethod
    def unique_id(transport=None):
        """transport: "rccl", "shm" (ranks sharing a GPU / no RCCL; see include/gmmiv.h) or None = $GMMIV_COMM_TRANSPORT, else rccl."""
        buf = ct.create_string_buffer(COMM_ID_BYTES)
        _chk(lib.gmmiv_comm_get_unique_id_for(transport.encode() if transport else None, buf))
        return buf.raw

    @staticmethod
    def exchange_id_file(path, rank, timeout_s=120.0):
        buf = ct.create_string_buffer(COMM_ID_BYTES)
        _chk(lib.gmmiv_comm_exchange_id_file(path.encode(), int(rank), buf, ct.c_double(timeout_s)))
        return buf.raw

    def close(self):
        if self._h:
            lib.gmmiv_comm_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            if sys is None or sys.is_finalizing():
                return
            self.close()
        except Exception:
            pass

    def backend(self):
        return lib.gmmiv_comm_backend(self._h).decode()

    def info(self):
        """{"rccl_version": ncclGetVersion code, "rccl_comm_count": ncclCommCount} -- zeros when no RCCL is behind the communicator."""
        v, n = ct.c_int(0), ct.c_int(0)
        _chk(lib.gmmiv_comm_info(self._h, ct.byref(v), ct.byref(n)))
        return {"rccl_version": v.value, "rccl_comm_count": n.value}

    def take_bytes(self):
        return lib.gmmiv_comm_take_bytes(self._h)

    @staticmethod
    def _n(a):
        return a.numel() if _is_torch(a) else a.size

    def allreduce(self, a):
        _chk(lib.gmmiv_allreduce_f64(self._h, _ptr(a), ct.c_size_t(self._n(a))))
        return a

    def broadcast(self, a, root=0):
        _chk(lib.gmmiv_broadcast_f64(self._h, _ptr(a), ct.c_size_t(self._n(a)), int(root)))
        return a

    def reduce_scatter(self, send, recv):
        assert self._n(send) == self.world * self._n(recv)
        _chk(lib.gmmiv_reduce_scatter_f64(self._h, _ptr(send), _ptr(recv), ct.c_size_t(self._n(recv))))
        return recv

    def allgather(self, send, recv):
        assert self._n(recv) == self.world * self._n(send)
        _chk(lib.gmmiv_allgather_f64(self._h, _ptr(send), _ptr(recv), ct.c_size_t(self._n(send))))
        return recv

    # overlapped forms (device tensors): the collective runs on the communicator's side stream behind what the context's stream
    # holds so far; join() orders the context's stream behind everything begun since the last join
    def allreduce_begin(self, a):
        _chk(lib.gmmiv_allreduce_f64_begin(self._h, _ptr(a), ct.c_size_t(self._n(a))))
        return a

    def reduce_scatter_begin(self, send, recv):
        assert self._n(send) == self.world * self._n(recv)
        _chk(lib.gmmiv_reduce_scatter_f64_begin(self._h, _ptr(send), _ptr(recv), ct.c_size_t(self._n(recv))))
        return recv

    def allgather_begin(self, send, recv):
        assert self._n(recv) == self.world * self._n(send)
        _chk(lib.gmmiv_allgather_f64_begin(self._h, _ptr(send), _ptr(recv), ct.c_size_t(self._n(send))))
        return recv

    def join(self):
        _chk(lib.gmmiv_comm_join(self._h))


def shard_range(n, rank, world):
    """gmmiv_shard_range: contiguous [begin, end) of rank's share of n items."""
    b, e = ct.c_int64(), ct.c_int64()
    lib.gmmiv_shard_range(ct.c_int64(n), int(rank), int(world), ct.byref(b), ct.byref(e))
    return b.value, e.value


class Gmm:
    """Device-resident MixtureGD: w[C], mean[C,D], covinv[C,D]."""

    def __init__(self, ctx, w, mean, covinv):
        self.ctx = ctx
        w, mean, covinv = _f64(w), _f64(mean), _f64(covinv)
        self.C, self.D = mean.shape
        self._h = ct.c_void_p()
        _chk(lib.gmmiv_gmm_create(ctx._h, self.C, self.D, _ptr(w), _ptr(mean), _ptr(covinv), ct.byref(self._h)))

    def set(self, w, mean, covinv):
        _chk(lib.gmmiv_gmm_set(self._h, _ptr(_f64(w)), _ptr(_f64(mean)), _ptr(_f64(covinv))))

    def set_cov(self, w, mean, cov):
        _chk(lib.gmmiv_gmm_set_cov(self._h, _ptr(_f64(w)), _ptr(_f64(mean)), _ptr(_f64(cov))))

    def close(self):
        if self._h:
            lib.gmmiv_gmm_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            if sys is None or sys.is_finalizing():
                return
            self.close()
        except Exception:
            pass

    def llk(self, x, min_llk=-200.0, max_llk=200.0, out=None, sums=None):
        x, dt, T, ldx = _feat(x)
        if out is None:
            out = np.empty(T)
        _chk(lib.gmmiv_llk(self.ctx._h, self._h, _ptr(x), dt, ct.c_int64(T), ct.c_int64(ldx), ct.c_double(min_llk),
                           ct.c_double(max_llk), _ptr(out), _ptr(sums)))
        return out

    def llk_determine_top(self, x, ctop, complete=True, min_llk=-200.0, max_llk=200.0):
        x, dt, T, ldx = _feat(x)
        ctop = min(ctop, self.C)
        idx = np.empty((T, ctop), np.int32)
        lk = np.empty((T, ctop)); nlk = np.empty(T); nllk = np.empty(T); nw = np.empty(T); out = np.empty(T)
        _chk(lib.gmmiv_llk_determine_top(self.ctx._h, self._h, _ptr(x), dt, ct.c_int64(T), ct.c_int64(ldx), ctop,
                                         TOP_COMPLETE if complete else TOP_PARTIAL, ct.c_double(min_llk),
                                         ct.c_double(max_llk), _ptr(idx), _ptr(lk), _ptr(nlk), _ptr(nllk), _ptr(nw),
                                         _ptr(out)))
        return dict(idx=idx, lk=lk, nontop_lk=nlk, nontop_llk=nllk, nontop_w=nw, llk=out)

    def llk_use_top(self, x, idx, nontop_llk, complete=True, min_llk=-200.0, max_llk=200.0):
        x, dt, T, ldx = _feat(x)
        idx = np.ascontiguousarray(idx, np.int32)
        out = np.empty(T)
        _chk(lib.gmmiv_llk_use_top(self.ctx._h, self._h, _ptr(x), dt, ct.c_int64(T), ct.c_int64(ldx), idx.shape[1],
                                   _ptr(idx), _ptr(_f64(nontop_llk)), TOP_COMPLETE if complete else TOP_PARTIAL,
                                   ct.c_double(min_llk), ct.c_double(max_llk), _ptr(out)))
        return out

    @staticmethod
    def llk_use_top_multi(clients, x, idx, nontop_llk, complete=True, min_llk=-200.0, max_llk=200.0):
        """USE_TOP_DISTRIBS for a list of client models on the same frames and world indices (ComputeTest's client loop) in one
        call: [len(clients), T]."""
        x, dt, T, ldx = _feat(x)
        idx = np.ascontiguousarray(idx, np.int32)
        n = len(clients)
        out = np.empty((n, T))
        if n == 0:
            return out
        arr = (ct.c_void_p * n)(*[g._h for g in clients])
        _chk(lib.gmmiv_llk_use_top_multi(clients[0].ctx._h, n, arr, _ptr(x), dt, ct.c_int64(T), ct.c_int64(ldx), idx.shape[1],
                                         _ptr(idx), _ptr(_f64(nontop_llk)), TOP_COMPLETE if complete else TOP_PARTIAL,
                                         ct.c_double(min_llk), ct.c_double(max_llk), _ptr(out)))
        return out

    def occ(self, x):
        """Posterior vectors [T x C] (computeAndAccumulateOcc / getOccVect)."""
        x, dt, T, ldx = _feat(x)
        out = np.empty((T, self.C))
        _chk(lib.gmmiv_occ(self.ctx._h, self._h, _ptr(x), dt, ct.c_int64(T), ct.c_int64(ldx), _ptr(out)))
        return out

    def em_acc_len(self):
        return lib.gmmiv_em_acc_len(self.C, self.D)

    def em_accumulate(self, x, weight=1.0, acc=None):
        x, dt, T, ldx = _feat(x)
        if acc is None:
            acc = np.zeros(self.em_acc_len())
        _chk(lib.gmmiv_em_accumulate(self.ctx._h, self._h, _ptr(x), dt, ct.c_int64(T), ct.c_int64(ldx),
                                     ct.c_double(weight), _ptr(acc)))
        return acc

    def em_get(self, acc, prev_mean, prev_cov):
        C, D = self.C, self.D
        w = np.empty(C); mean = np.empty((C, D)); cov = np.empty((C, D))
        _chk(lib.gmmiv_em_get(self.ctx._h, C, D, _ptr(acc), _ptr(_f64(prev_mean)), _ptr(_f64(prev_cov)), _ptr(w),
                              _ptr(mean), _ptr(cov)))
        return w, mean, cov

    def split_acc(self, acc):
        C, D = self.C, self.D
        a = np.asarray(acc)
        return dict(occ=a[:C], sx=a[C:C + C * D].reshape(C, D), sxx=a[C + C * D:C + 2 * C * D].reshape(C, D),
                    llk=a[-2], count=a[-1])

    def tv_stats_lines(self, x, file_begin, lines, N=None, F=None):
        """Baum-Welch statistics per ndx LINE: lines = list of lists of file indices (a file may appear on several lines)."""
        x, dt, T, ldx = _feat(x)
        fb = np.ascontiguousarray(file_begin, dtype=np.int64)
        off = np.zeros(len(lines) + 1, np.int64)
        off[1:] = np.cumsum([len(l) for l in lines])
        files = np.ascontiguousarray([f for l in lines for f in l], dtype=np.int64) if off[-1] else np.zeros(1, np.int64)
        if N is None:
            N = np.empty((len(lines), self.C)); F = np.empty((len(lines), self.C * self.D))
        _chk(lib.gmmiv_tv_stats_lines(self.ctx._h, self._h, _ptr(x), dt, ct.c_int64(T), ct.c_int64(ldx), fb.ctypes.data_as(ct.c_void_p),
                                      ct.c_int64(len(fb) - 1), ct.c_int64(len(lines)), off.ctypes.data_as(ct.c_void_p),
                                      files.ctypes.data_as(ct.c_void_p), _ptr(N), _ptr(F)))
        return N, F

    def tv_stats(self, x, utt_begin, N=None, F=None):
        x, dt, T, ldx = _feat(x)
        ub = np.ascontiguousarray(utt_begin, dtype=np.int64)
        U = len(ub) - 1
        if N is None:
            N = np.empty((U, self.C)); F = np.empty((U, self.C * self.D))
        _chk(lib.gmmiv_tv_stats(self.ctx._h, self._h, _ptr(x), dt, ct.c_int64(T), ct.c_int64(ldx),
                                ub.ctypes.data_as(ct.c_void_p), ct.c_int64(U), _ptr(N), _ptr(F)))
        return N, F
