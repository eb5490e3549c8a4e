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
