"""Multi-GPU plumbing for the paths that shard: one process per GPU, frames or utterances split into
contiguous per-rank ranges exactly like the reference splits work across pthreads (LIA_SpkTools/src/AccumulateStat.cpp:234-299,
AccumulateTVStat.cpp:498-507), and the ranks' private accumulators merged by a collective where the reference merges its
threads' accumulators under a mutex (MixtureStat::addAccEM, AccumulateStat.cpp:286-292; the `+=` of A / Cmx / R / r in the
threaded estimateAandC, AccumulateTVStat.cpp:1920-1937, 2036-2044).

Collectives come in three interchangeable back ends with the same four operations:
  * GmmivCollectives -- the product's own C ABI (gmmiv_comm_*: RCCL over xGMI on the device buffers), what bench.py uses;
  * TorchCollectives -- torch.distributed: "nccl" (= RCCL) with reduce_scatter_tensor / all_gather_into_tensor, or "gloo" in
    the CPU tests (gloo has no reduce-scatter: it is emulated with one reduce per destination, same bytes on the wire) and in
    the multi-process GPU tests where the ranks share one device (device tensors are staged through host copies: gloo moves
    the bytes, every sum that matters -- the statistics -- is still computed by libgmmiv on the device);
  * LocalCollectives -- a single rank.
Everything here is orchestration; the arithmetic is done by the callbacks (libgmmiv on a GPU rank)."""
import time

import numpy as np


def shard_range(n, rank, world):
    """Contiguous [begin, end) of rank's share of n items (sizes differ by at most one)."""
    base, rem = divmod(int(n), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _as_tensor(a):
    """torch view of a numpy array (shared memory) or the tensor itself."""
    import torch
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return torch.from_numpy(a)
    return a


# ---------------------------------------------------------------------------------------- collectives
class LocalCollectives:
    world, rank, name = 1, 0, "single rank"

    def allreduce(self, a):
        return a

    def reduce_scatter(self, send, recv):
        _as_tensor(recv).copy_(_as_tensor(send).reshape(-1)[: _as_tensor(recv).numel()].view_as(_as_tensor(recv)))
        return recv

    def allgather(self, send, recv):
        _as_tensor(recv).reshape(-1).copy_(_as_tensor(send).reshape(-1))
        return recv

    def take_bytes(self):
        return 0.0


class TorchCollectives:
    """torch.distributed process group (must be initialised).  Tensors or numpy arrays, float64, contiguous."""

    def __init__(self):
        import torch.distributed as dist
        assert dist.is_available() and dist.is_initialized()
        self.dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.backend = dist.get_backend()
        self.name = "torch.distributed(%s)" % self.backend
        self._bytes = 0.0

    def allreduce(self, a):
        t = _as_tensor(a)
        self._bytes += t.numel() * 8
        if self.backend != "nccl" and t.is_cuda:      # gloo: device tensors travel as host copies
            h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h)
        else:
            self.dist.all_reduce(t)
        return a

    def reduce_scatter(self, send, recv):
        s, r = _as_tensor(send).reshape(-1), _as_tensor(recv)
        n = r.numel()
        assert s.numel() == self.world * n
        self._bytes += s.numel() * 8
        if self.backend == "nccl":
            self.dist.reduce_scatter_tensor(r.view(-1), s)
        else:  # gloo: one reduce per destination
            for g in range(self.world):
                chunk = s[g * n:(g + 1) * n].cpu() if s.is_cuda else s[g * n:(g + 1) * n].clone()
                self.dist.reduce(chunk, dst=g)
                if g == self.rank:
                    r.view(-1).copy_(chunk)
        return recv

    def allgather(self, send, recv):
        s, r = _as_tensor(send).reshape(-1), _as_tensor(recv).view(-1)
        n = s.numel()
        assert r.numel() == self.world * n
        self._bytes += r.numel() * 8
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(r, s.contiguous())
        else:
            sh = s.cpu() if s.is_cuda else s.contiguous()
            parts = [sh.new_empty(n) for _ in range(self.world)]
            self.dist.all_gather(parts, sh)
            for g in range(self.world):
                r[g * n:(g + 1) * n].copy_(parts[g])
        return recv

    def take_bytes(self):
        b, self._bytes = self._bytes, 0.0
        return b


class GmmivCollectives:
    """The C ABI's communicator (include/gmmiv.h, gmmiv_comm_*): RCCL (or the shm transport) called by libgmmiv on the
    context's stream.  Device tensors produced by torch kernels are safe to hand over from ANY torch stream: each call is
    bracketed by stream waits (capi.Context.ordered) unless torch's current stream already is the context's."""

    def __init__(self, comm):
        self.comm = comm
        self.world, self.rank = comm.world, comm.rank
        self.backend = comm.backend()
        self.name = "gmmiv_comm (%s)" % self.backend

    def _ordered(self, *tensors):
        import contextlib
        if any(getattr(t, "is_cuda", False) for t in tensors):
            return self.comm.ctx.ordered()
        return contextlib.nullcontext()

    def allreduce(self, a):
        with self._ordered(a):
            return self.comm.allreduce(a)

    def reduce_scatter(self, send, recv):
        with self._ordered(send, recv):
            return self.comm.reduce_scatter(send, recv)

    def allgather(self, send, recv):
        with self._ordered(send, recv):
            return self.comm.allgather(send, recv)

    # overlapped forms (include/gmmiv.h: gmmiv_*_begin / gmmiv_comm_join)
    supports_overlap = True

    def allreduce_begin(self, a):
        with self._ordered(a):
            return self.comm.allreduce_begin(a)

    def reduce_scatter_begin(self, send, recv):
        with self._ordered(send, recv):
            return self.comm.reduce_scatter_begin(send, recv)

    def allgather_begin(self, send, recv):
        with self._ordered(send, recv):
            return self.comm.allgather_begin(send, recv)

    def join(self):
        with self.comm.ctx.ordered():
            self.comm.join()

    def take_bytes(self):
        return self.comm.take_bytes()


def default_collectives():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return TorchCollectives()
    except Exception:  # pragma: no cover
        pass
    return LocalCollectives()


def gmmiv_collectives_from_torch(ctx, device=None, transport=None):
    """Bootstrap a gmmiv communicator inside an initialised torch.distributed job: rank 0 draws the id (transport "rccl", or
    "shm" for ranks that share a GPU; None = $GMMIV_COMM_TRANSPORT, else rccl), the id travels through the existing process
    group, every rank joins.  Returns GmmivCollectives (single-rank communicator for one rank)."""
    import torch
    import torch.distributed as dist
    from . import capi
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return GmmivCollectives(capi.Comm(ctx, 1, 0))
    world, rank = dist.get_world_size(), dist.get_rank()
    # rank 0 ALWAYS broadcasts (id + a validity byte): if it cannot draw an id (no RCCL to load) the other ranks must not be left
    # waiting in a broadcast that never comes while rank 0 moves on to the next collective of the launcher's process group
    uid, err = bytes(capi.COMM_ID_BYTES), None
    if rank == 0:
        try:
            uid = capi.Comm.unique_id(transport)
        except Exception as e:      # noqa: BLE001 - re-raised on every rank below
            err = e
    t = torch.tensor(list(uid) + [0 if err else 1], dtype=torch.uint8, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.broadcast(t, src=0)
    got = t.cpu().tolist()
    if got[-1] != 1:
        raise capi.GmmivError("rank 0 could not draw a communicator id%s" % (": %r" % err if err else ""))
    return GmmivCollectives(capi.Comm(ctx, world, rank, bytes(got[:-1])))


def all_reduce_sum(acc, coll=None):
    """In-place sum over ranks of a torch tensor or numpy array; no-op without a process group."""
    coll = coll or default_collectives()
    if coll.world == 1:
        return acc
    return coll.allreduce(acc)


# ---------------------------------------------------------------------------------------- UBM EM
def em_iteration(accumulate, n_frames, acc, rank=0, world=1, coll=None):
    """One distributed E-step: `accumulate(begin, end, acc)` adds the statistics of frames [begin, end) into the flat
    accumulator (gmmiv_em_accumulate on a GPU rank), then the ranks' accumulators are summed -- ONE all-reduce of
    C (1 + 2 D) + 2 doubles.  Every rank returns the same global statistics."""
    b, e = shard_range(n_frames, rank, world)
    accumulate(b, e, acc)
    return all_reduce_sum(acc, coll)


# ---------------------------------------------------------------------------------------- i-vector scoring
def score_model_block(score, models, segs, rank=0, world=1, nsess=None):
    """IvTest scoring on several GPUs: the M x S score matrix tiles by blocks of MODELS, models / segments are
    replicated (2 x 320 MB at 100 k x 400), there is no collective -- rank g scores the model columns [m0, m1) =
    shard_range(M, g, world) against every segment and keeps / writes its own rows, like the reference's worker threads take
    ranges of models (PldaTools.cpp:4175-4183 -> pldaScoringThreaded, :3912-3920).  `score(models_block, segs[, nsess_block])`
    is any gmmiv_score_* rule bound to a context.  Returns (m0, m1, scores[m1 - m0, S])."""
    M = models.shape[1]
    m0, m1 = shard_range(M, rank, world)
    blk = models[:, m0:m1]
    blk = np.ascontiguousarray(blk) if isinstance(blk, np.ndarray) else blk.contiguous()
    if nsess is not None:
        return m0, m1, score(blk, segs, nsess[m0:m1])
    return m0, m1, score(blk, segs)


# ---------------------------------------------------------------------------------------- T-matrix EM
def tv_estep(estimate, n_utt, acc, rank=0, world=1, coll=None):
    """One distributed E-step of the T-matrix EM (TVAcc::estimateAandC, AccumulateTVStat.cpp:1702-1795), all-reduce form:
    `estimate(begin, end, acc)` adds the utterances [begin, end) of this rank into acc = {"A", "Cmx", "Rm", "r", "meanW"}
    (gmmiv_tv_estimate_a_and_c; meanW is the SUM of the i-vectors), then every accumulator is summed over ranks and the
    M-step is replicated.  The sharded form below moves half the bytes and splits the M-step; this one is kept for
    callers that want the global A on every rank."""
    b, e = shard_range(n_utt, rank, world)
    estimate(b, e, acc)
    for k in ("A", "Cmx", "Rm", "r", "meanW"):
        all_reduce_sum(acc[k], coll)
    return acc


def block_size(C, world):
    """Gaussians per rank of the sharded M-step: equal blocks (a reduce-scatter needs them), the last may be padded."""
    return (int(C) + int(world) - 1) // int(world)


def tv_mstep_sharded(acc, update_t, C, D, rank=0, world=1, coll=None, phases=None, force_collectives=False):
    """Distributed M-step of the T-matrix EM, the layout SURVEY.md 8(e) prefers.  T_c = A_c^-1 Cmx_c is independent per
    Gaussian (TVAcc::updateTestimate, AccumulateTVStat.cpp:981-1000), so rank g OWNS the Gaussians [g Cb, (g+1) Cb),
    Cb = ceil(C / world):
      * ONE reduce-scatter of A (packed [C x P], 1.31 GB at C = 2048, R = 400; row blocks are contiguous) and ONE of Cmx
        (393 MB, re-laid out as [world][R][Cb D] so that a rank's column block is contiguous): every rank receives only the
        sum of its own block -- (world - 1) / world of the payload crosses each link once where an all-reduce moves it twice,
        and no rank ever holds the global A;
      * `update_t(A_block [cb x P], Cmx_block [R x cb*D], cb) -> T_block [R x cb*D]` on the rank's own cb <= Cb real
        Gaussians (gmmiv_tv_update_t on a GPU rank): the C factorisations are split over the ranks too;
      * ONE all-gather of the T blocks; TETt is recomputed locally by the caller;
      * R, r, meanW (a few KB, for the minimum-divergence step) travel in one small all-reduce.
    acc: the per-rank accumulators of gmmiv_tv_estimate_a_and_c (A [C, P], Cmx [R, C*D], Rm, r, meanW), numpy arrays or torch
    tensors; Rm / r / meanW are summed in place.  Returns the full T [R, C*D], identical on every rank.
    phases (dict, optional): receives wall-clock seconds of "reduce_scatter", "update_t", "allgather" (the caller's `sync`
    entry, if present, is called before each stamp)."""
    import torch
    coll = coll or default_collectives()
    is_np = isinstance(acc["A"], np.ndarray)
    A, Cmx = _as_tensor(acc["A"]), _as_tensor(acc["Cmx"])
    R, P = Cmx.shape[0], A.shape[1]
    sync = (phases or {}).get("sync") or (lambda: None)

    def stamp(key, t0):
        if phases is not None:
            sync()
            phases[key] = phases.get(key, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    t0 = time.perf_counter()
    # the small accumulators in one buffer
    small = torch.cat([_as_tensor(acc[k]).reshape(-1) for k in ("Rm", "r", "meanW")])
    coll.allreduce(small)
    o = 0
    for k in ("Rm", "r", "meanW"):
        t = _as_tensor(acc[k])
        t.copy_(small[o:o + t.numel()].view_as(t)); o += t.numel()
    if world == 1 and not force_collectives:   # (force_collectives: one rank still walks the exchange -- a one-rank RCCL communicator
        t0 = stamp("reduce_scatter", t0)       #  then executes every collective of the step on a one-GPU machine)
        Tn = update_t(acc["A"], acc["Cmx"], C)
        stamp("update_t", t0)
        return Tn
    Cb = block_size(C, world)
    Cpad = Cb * world
    if Cpad != C:   # pad with empty Gaussians so that the blocks are equal
        A_send = A.new_zeros((Cpad, P)); A_send[:C] = A
        Cp = Cmx.new_zeros((R, Cpad * D)); Cp[:, :C * D] = Cmx
    else:
        A_send, Cp = A, Cmx
    C_send = Cp.view(R, world, Cb * D).permute(1, 0, 2).contiguous()          # [world][R][Cb D]
    a_mine = A.new_empty((Cb, P)); c_mine = Cmx.new_empty((R, Cb * D))
    coll.reduce_scatter(A_send, a_mine)
    coll.reduce_scatter(C_send, c_mine)
    t0 = stamp("reduce_scatter", t0)
    t_mine = _update_own_block(update_t, a_mine, c_mine, C, D, Cb, rank, is_np)
    t0 = stamp("update_t", t0)
    gathered = Cmx.new_empty((world, R, Cb * D))
    coll.allgather(t_mine, gathered)
    Tn = gathered.permute(1, 0, 2).reshape(R, Cpad * D)[:, :C * D].contiguous()
    stamp("allgather", t0)
    return Tn.numpy() if is_np else Tn


def _update_own_block(update_t, a_mine, c_mine, C, D, Cb, rank, is_np):
    """updateTestimate on the rank's own cb <= Cb real Gaussians (the rest of a padded block stays zero) -> [R x Cb D]."""
    R = c_mine.shape[0]
    cb = max(0, min(Cb, C - rank * Cb))
    t_mine = c_mine.new_zeros((R, Cb * D))
    if cb > 0:
        a_in = a_mine[:cb]
        c_in = c_mine if cb == Cb else c_mine[:, :cb * D].contiguous()
        tb = _as_tensor(update_t(a_in.numpy() if is_np else a_in, c_in.numpy() if is_np else c_in, cb))
        if cb == Cb:
            t_mine = tb.contiguous()
        else:
            t_mine[:, :cb * D] = tb
    return t_mine


def _fatal_between_collectives(e, rank, world, where):
    """A rank that fails BETWEEN the begin and the join of an exchange cannot leave the iteration by raising: its peers are already
    waiting in the collective it will never reach, and they would wait for ever.  With more than one rank the failure is therefore
    fatal for the job: the error is printed and the process exits (the launcher takes the other ranks down).  One rank: the caller
    re-raises."""
    if world <= 1:
        return
    import os
    import sys
    import traceback
    sys.stderr.write("[gmmiv] rank %d: failure in %s with collectives in flight -- aborting the job:\n" % (rank, where))
    traceback.print_exception(type(e), e, e.__traceback__, file=sys.stderr)
    sys.stderr.flush()
    os._exit(70)


def _tv_overlapped(ops, n_sessions_total, C, D, rank, world, coll, phases, lap, t0):
    """E-step + sharded M-step + minDivergence of one iteration with the exchange started EARLY (option `overlap`):
      * the reduce-scatter of A (1.31 GB at 2048 x 400) begins from the "tv_a_ready" hook of gmmiv_tv_estimate_a_and_c, i.e. as
        soon as the last `A += N^T E` GEMM is enqueued, and runs on the communicator's side stream under `Cmx += W^T F` and the
        batch sums (the reference merges A, Cmx, R, r under one mutex after all workers are done, AccumulateTVStat.cpp:1920-1937);
        this needs the E-step to accumulate straight into the send buffer: ops.acc["A_pad"] ([Cb * world, P], rows >= C zero,
        acc["A"] = its first C rows) when C does not divide -- without it the exchange starts after the E-step like the serial order;
      * the all-gather of the T blocks begins after the rank's own solve and is joined from the "md_factored" hook of
        gmmiv_tv_min_divergence: it runs under the normalisation and factorisation of R.
    Same collectives on the same buffers as tv_mstep_sharded, same arithmetic: T and the means are bitwise those of the serial order."""
    import torch
    ctx = ops.ctx
    acc0 = ops.acc
    A, Cmx = acc0["A"], acc0["Cmx"]
    R, P = Cmx.shape[0], A.shape[1]
    Cb = block_size(C, world)
    Cpad = Cb * world
    A_send = A if Cpad == C else acc0.get("A_pad")
    a_mine = A.new_empty((Cb, P)); c_mine = Cmx.new_empty((R, Cb * D))
    begun = []

    def a_ready():
        if A_send is not None:
            coll.reduce_scatter_begin(A_send, a_mine)
            begun.append("A")
    ctx.set_hook("tv_a_ready", a_ready)
    try:
        acc = ops.estep()
    except BaseException as e:          # noqa: BLE001
        _fatal_between_collectives(e, rank, world, "the tv_a_ready hook / E-step")
        raise
    finally:
        ctx.set_hook("tv_a_ready", None)
    t0 = lap("estep", t0)
    small = torch.cat([acc[k].reshape(-1) for k in ("Rm", "r", "meanW")])
    if not begun:                                   # no send buffer the E-step could write into: pad now, exchange now
        A_send = A.new_zeros((Cpad, P)); A_send[:C] = A
        coll.reduce_scatter_begin(A_send, a_mine)
    coll.allreduce_begin(small)
    if Cpad != C:
        Cp = Cmx.new_zeros((R, Cpad * D)); Cp[:, :C * D] = Cmx
    else:
        Cp = Cmx
    C_send = Cp.view(R, world, Cb * D).permute(1, 0, 2).contiguous()
    coll.reduce_scatter_begin(C_send, c_mine)
    coll.join()
    o = 0
    for k in ("Rm", "r", "meanW"):
        t = acc[k]
        t.copy_(small[o:o + t.numel()].view_as(t)); o += t.numel()
    t0 = lap("reduce_scatter", t0)
    t_mine = _update_own_block(ops.update_t, a_mine, c_mine, C, D, Cb, rank, False)
    t0 = lap("update_t", t0)
    gathered = Cmx.new_empty((world, R, Cb * D))
    coll.allgather_begin(t_mine, gathered)
    Tn = Cmx.new_empty((R, C * D))

    def md_factored():                              # R is factored, T is about to be read: T must have arrived
        coll.join()
        Tn.copy_(gathered.permute(1, 0, 2).reshape(R, Cpad * D)[:, :C * D])
    t0 = lap("allgather", t0)
    ctx.set_hook("md_factored", md_factored)
    try:
        Tn = ops.min_divergence(acc, Tn, n_sessions_total)
    except BaseException as e:          # noqa: BLE001
        _fatal_between_collectives(e, rank, world, "the md_factored hook / minDivergence")
        raise
    finally:
        ctx.set_hook("md_factored", None)
    lap("min_divergence", t0)
    return Tn


def tv_em_iteration(ops, n_sessions_total, C, D, rank=0, world=1, coll=None, phases=None, overlap=False, force_collectives=False):
    """One full iteration of TotalVariability's EM on utterance-sharded statistics (TotalVariability.cpp:118-169 around
    TVAcc, AccumulateTVStat.cpp): every rank holds the statistics N / F of its own utterances and the full T.
      ops.recentre()                      (optional) restore the raw first-order statistics and substractM with the CURRENT UBM
                                          means -- TotalVariability.cpp:123-124 does it at the top of every iteration,
                                          minDivergence moves the means in between
      ops.tett()                          estimateTETt from the current T (replicated: 3 ms at 2048 x 400)
      ops.estep() -> acc                  resetTmpAcc + estimateAandC over the rank's utterances
      [reduce-scatter / update_t / all-gather: tv_mstep_sharded]
      ops.update_t(A_blk, Cmx_blk, cb)    updateTestimate on the rank's own Gaussians
      ops.min_divergence(acc, T, n)       minDivergence with the all-reduced R, r, meanW (replicated; T and the UBM
                                          means are updated in place on every rank identically)
    ops.stream_context() (optional) returns a context manager under which the whole iteration runs -- a GPU rank returns
    `torch.cuda.stream(ctx.torch_stream())` so that the torch kernels of the re-layout (pad, permute, cat) are enqueued on the
    stream libgmmiv launches on, in program order with its kernels and collectives.
    overlap = True (GmmivCollectives only): the reduce-scatter of A starts inside the E-step and the all-gather of T is joined
    inside minDivergence (_tv_overlapped); bitwise the same T and means as the serial order.
    force_collectives = True: a single rank walks the sharded exchange too (one block of all Gaussians) instead of calling
    update_t directly -- with a one-rank RCCL communicator (GMMIV_COMM_FORCE_RCCL=1) the whole step, overlapped or not, then runs its
    RCCL calls on a one-GPU machine.
    Returns the new T.  phases collects per-phase seconds ("recentre", "tett", "estep", "reduce_scatter", "update_t",
    "allgather", "min_divergence")."""
    import contextlib
    coll = coll or default_collectives()
    sync = (phases or {}).get("sync") or (lambda: None)

    def lap(key, t0):
        if phases is not None:
            sync()
            phases[key] = phases.get(key, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    cm = ops.stream_context() if hasattr(ops, "stream_context") else contextlib.nullcontext()
    with cm:
        t0 = time.perf_counter()
        if hasattr(ops, "recentre"):
            ops.recentre()
            t0 = lap("recentre", t0)
        ops.tett()
        t0 = lap("tett", t0)
        if overlap and (world > 1 or force_collectives) and getattr(coll, "supports_overlap", False) and hasattr(ops, "ctx") and hasattr(ops, "acc"):
            return _tv_overlapped(ops, n_sessions_total, C, D, rank, world, coll, phases, lap, t0)
        acc = ops.estep()
        t0 = lap("estep", t0)
        Tn = tv_mstep_sharded(acc, ops.update_t, C, D, rank, world, coll, phases, force_collectives)
        t0 = time.perf_counter()
        Tn = ops.min_divergence(acc, Tn, n_sessions_total)
        lap("min_divergence", t0)
    return Tn
