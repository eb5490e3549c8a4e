"""Multi-GPU plumbing for the paths that shard: one process per GPU, frames or
utterances split into contiguous per-rank ranges exactly like the reference splits work across
pthreads (LIA_SpkTools/src/AccumulateStat.cpp:234-299, AccumulateTVStat.cpp:498-507), and ONE
all-reduce (RCCL over xGMI with the "nccl" backend; gloo in the CPU tests) of the flat
sufficient-statistics array per EM iteration -- the collective twin of MixtureStat::addAccEM."""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous [begin, end) of rank's share of n items (sizes differ by at most one)."""
    base, rem = divmod(int(n), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def all_reduce_sum(acc):
    """In-place sum over ranks of a torch tensor or numpy array; no-op without a process group."""
    try:
        import torch
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return acc
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return acc
    if isinstance(acc, np.ndarray):
        t = torch.from_numpy(acc)
        dist.all_reduce(t)
        return acc
    dist.all_reduce(acc)
    return acc


def em_iteration(accumulate, n_frames, acc, rank=0, world=1):
    """One distributed E-step: `accumulate(begin, end, acc)` adds the statistics of frames
    [begin, end) into the flat accumulator (gmmiv_em_accumulate on a GPU rank), then the ranks'
    accumulators are summed.  Every rank returns the same global statistics."""
    b, e = shard_range(n_frames, rank, world)
    accumulate(b, e, acc)
    return all_reduce_sum(acc)


def tv_estep(estimate, n_utt, acc, rank=0, world=1):
    """One distributed E-step of the T-matrix EM (TVAcc::estimateAandC, AccumulateTVStat.cpp:1702-1795):
    `estimate(begin, end, acc)` adds the utterances [begin, end) of this rank into the accumulators
    acc = {"A", "Cmx", "Rm", "r", "meanW"} (gmmiv_tv_estimate_a_and_c on a GPU rank; meanW is the SUM
    of the i-vectors), then every accumulator is summed over ranks -- the collective twin of the
    mutex-guarded `+=` of the reference's threads (:1920-1937, :2036-2044).  A (packed, 1.31 GB at
    C=2048, R=400) and Cmx (393 MB) dominate the payload; the M-step is replicated on every rank."""
    b, e = shard_range(n_utt, rank, world)
    estimate(b, e, acc)
    for k in ("A", "Cmx", "Rm", "r", "meanW"):
        all_reduce_sum(acc[k])
    return acc


def _reduce_to_owner(chunks, rank, world):
    """chunks[g]: this rank's contribution to the block owned by rank g (numpy arrays or torch tensors).  Returns the SUM over
    ranks of chunks[rank] -- a reduce-scatter written as one reduce per destination (RCCL has reduce_scatter, gloo does not;
    the traffic is the same: every rank ships (world - 1) / world of its data once)."""
    import torch
    import torch.distributed as dist
    mine = None
    for g in range(world):
        c = chunks[g]
        t = torch.from_numpy(np.ascontiguousarray(c)) if isinstance(c, np.ndarray) else c.contiguous()
        dist.reduce(t, dst=g)
        if g == rank:
            mine = t
    return mine.numpy() if isinstance(chunks[rank], np.ndarray) else mine


def tv_mstep_sharded(acc, update_t, C, D, rank=0, world=1):
    """Distributed M-step of the T-matrix EM, the layout SURVEY.md 8(e) prefers: T_c = A_c^-1 Cmx_c is independent per Gaussian
    (TVAcc::updateTestimate, AccumulateTVStat.cpp:981-1000), so rank g OWNS the Gaussians [c0_g, c1_g) = shard_range(C, g, world):
      * reduce-scatter of A (packed, 1.31 GB at C = 2048, R = 400) and of the matching column blocks of Cmx (393 MB): every rank
        receives only the sum of its own blocks -- (world - 1) / world of the payload crosses each link once, where the all-reduce of
        tv_estep moves it twice, and no rank ever holds the global A;
      * `update_t(A_block [Cb x P], Cmx_block [R x Cb*D], Cb) -> T_block [R x Cb*D]` solves the rank's own Gaussians
        (gmmiv_tv_update_t on a GPU rank): the 2048 factorisations are split over the ranks too;
      * all-gather of the T blocks (393 MB / world each); TETt is recomputed locally by the caller.
    R, r, meanW (a few KB, for the minimum-divergence step) are all-reduced in place.  acc: the per-rank accumulators of
    gmmiv_tv_estimate_a_and_c (A [C, P], Cmx [R, C*D], Rm, r, meanW).  Returns the full T [R, C*D], identical on every rank."""
    A, Cmx = acc["A"], acc["Cmx"]
    R = Cmx.shape[0]
    if world == 1:
        return update_t(A, Cmx, C)
    import torch
    import torch.distributed as dist
    is_np = isinstance(A, np.ndarray)
    ranges = [shard_range(C, g, world) for g in range(world)]
    a_mine = _reduce_to_owner([A[c0:c1] for c0, c1 in ranges], rank, world)
    c_mine = _reduce_to_owner([Cmx[:, c0 * D:c1 * D] for c0, c1 in ranges], rank, world)
    for k in ("Rm", "r", "meanW"):
        all_reduce_sum(acc[k])
    c0, c1 = ranges[rank]
    t_mine = update_t(a_mine, c_mine, c1 - c0) if c1 > c0 else (np.zeros((R, 0)) if is_np else Cmx.new_zeros((R, 0)))
    # all-gather of unequal blocks: broadcast from each owner
    if is_np:
        T = np.empty((R, C * D))
    else:
        T = Cmx.new_empty((R, C * D))
    for g, (g0, g1) in enumerate(ranges):
        if g1 == g0:
            continue
        if is_np:
            blk = torch.from_numpy(np.ascontiguousarray(t_mine)) if g == rank else torch.empty((R, (g1 - g0) * D), dtype=torch.float64)
        else:
            blk = t_mine.contiguous() if g == rank else Cmx.new_empty((R, (g1 - g0) * D))
        dist.broadcast(blk, src=g)
        if is_np:
            T[:, g0 * D:g1 * D] = blk.numpy()
        else:
            T[:, g0 * D:g1 * D] = blk
    return T
