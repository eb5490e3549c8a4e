"""config 4's multi-rank leg ON THE DEVICE with the hardware at hand: 2 and 3 processes share GPU 0, libgmmiv (HIP) is the
per-rank compute, and the bytes between the ranks travel over
  * "gloo": torch.distributed's gloo process group (lia_ral_amd.dist.TorchCollectives; device tensors staged through host copies), or
  * "shm" : the C ABI's own communicator (gmmiv_comm_*, include/gmmiv.h) on its shared-memory transport -- the same entry points,
            device buffers and stream ordering the RCCL transport uses, only the wire differs.
What runs: two whole TotalVariability iterations (restore + substractM, estimateTETt, estimateAandC, reduce-scatter of A / Cmx by
PADDED Gaussian blocks (C = 7 is divisible by neither world size), sharded updateTestimate, all-gather of T, minDivergence) and one
UBM EM statistics pass with its all-reduce (AccumulateTVStat.cpp:1831-2052, 974-1005; AccumulateStat.cpp:234-299).  Every rank's
result is compared with the single-rank HIP run (1e-11) and with the TotalVariability loop assembled from oracle pieces.  What this
leaves unexecuted on a one-GPU box is the RCCL transport itself (tests/test_gpu_comm.py::test_two_rank_rccl_through_the_c_abi)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(kind="small"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import make_frames, make_gmm
    if kind == "north_star":
        # BASELINE.json north_star's own split: the 2048-Gaussian x 60-dim UBM over 8 ranks = blocks of 256 Gaussians, no padding;
        # the EM all-reduce is the real 247 810-double accumulator (rank 40 and 320 utterances keep the oracle loop in seconds)
        C, D, R, U, nfr = 2048, 60, 40, 320, 8003
    else:
        C, D, R, U, nfr = 7, 12, 10, 60, 5003        # 7 Gaussians: blocks of 4 + 3 on two ranks, 3 + 3 + 1 on three
    rng = np.random.default_rng(3)
    w, mean, iv = make_gmm(C, D, seed=3)
    N = rng.gamma(0.8, 3.0, (U, C)); F = rng.normal(size=(U, C * D)) * 3 + np.repeat(N, D, 1) * mean.ravel()
    Tm = rng.normal(0, 0.05, (R, C * D))
    x = make_frames(w, mean, iv, nfr, seed=9)        # the EM leg: a frame count no world size divides
    return C, D, R, U, w, mean, iv, N, F, Tm, x


class DeviceTvOps:
    """The per-rank compute of lia_ral_amd.dist.tv_em_iteration on device-resident statistics (libgmmiv)."""

    def __init__(self, ctx, C, D, R, N, F, Tm, iv, means, world=1):
        import torch
        self.ctx, self.C, self.D, self.R = ctx, C, D, R
        dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        self.N, self.F_raw, self.T, self.iv, self.means = dv(N), dv(F), dv(Tm), dv(iv.ravel()), dv(means.ravel())
        self.F = torch.empty_like(self.F_raw)
        P = R * (R + 1) // 2
        z = lambda *s: torch.zeros(s, dtype=torch.float64, device="cuda")
        self.te = z(C, P)
        A_pad = z((C + world - 1) // world * world, P)     # the reduce-scatter's send buffer: equal Gaussian blocks, zero padded
        self.acc = dict(A=A_pad[:C], A_pad=A_pad, Cmx=z(R, C * D), Rm=z(R, R), r=z(R), meanW=z(R), W=z(N.shape[0], R))

    def stream_context(self):
        import torch
        return torch.cuda.stream(self.ctx.torch_stream())

    def recentre(self):
        self.F.copy_(self.F_raw)
        self.ctx.tv_subtract_m(self.N, self.F, self.means, self.C, self.D)

    def tett(self):
        self.ctx.tv_tett(self.T, self.iv, self.C, self.D, out=self.te)

    def estep(self):
        for k in ("A", "Cmx", "Rm", "r", "meanW"):
            self.acc[k].zero_()
        return self.ctx.tv_estimate_a_and_c(self.N, self.F, self.T, self.iv, self.te, self.C, self.D, acc=self.acc)

    def update_t(self, A_blk, C_blk, cb):
        import torch
        return self.ctx.tv_update_t(A_blk, C_blk, cb, self.D, out=torch.empty((self.R, cb * self.D), dtype=torch.float64, device="cuda"))

    def min_divergence(self, acc, Tn, n):
        self.ctx.tv_min_divergence(acc["Rm"], acc["r"], acc["meanW"] / n, self.means, Tn, n, self.C, self.D)
        self.T = Tn
        return Tn


def _run_rank(rank, world, transport, port, idfile, overlap=False, kind="small", nb_it=2):
    """One rank: returns (T, means, em_acc, backend name) as numpy.  overlap: the whole thing twice -- serial order, then with the
    exchange started early (gmmiv_*_begin / hooks) from the same initial state -- and the bitwise comparison instead of the EM leg."""
    import torch
    from lia_ral_amd import capi
    from lia_ral_amd import dist as gd
    torch.cuda.set_device(0)                               # every rank on GPU 0
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    ctx = capi.Context(0, side.cuda_stream)
    if world == 1:
        coll = gd.GmmivCollectives(capi.Comm(ctx, 1, 0))
    elif transport == "gloo":
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        coll = gd.TorchCollectives()
    else:
        os.environ["GMMIV_COMM_TRANSPORT"] = "shm"         # rank 0 draws a shm id
        os.environ["GMMIV_COMM_SHM_SLOT_MB"] = "1"         # (payloads larger than a slot: test_shm_collectives_chunked)
        uid = capi.Comm.exchange_id_file(idfile, rank, 300.0)
        coll = gd.GmmivCollectives(capi.Comm(ctx, world, rank, uid))
    C, D, R, U, w, mean, iv, N, F, Tm, x = _case(kind)
    b, e = gd.shard_range(U, rank, world)
    ops = DeviceTvOps(ctx, C, D, R, N[b:e], F[b:e], Tm, iv, mean, world)
    for _ in range(nb_it):
        Tg = gd.tv_em_iteration(ops, U, C, D, rank, world, coll)
    if overlap:
        ops2 = DeviceTvOps(ctx, C, D, R, N[b:e], F[b:e], Tm, iv, mean, world)
        fired = []
        for _ in range(2):
            ph = {}
            T2 = gd.tv_em_iteration(ops2, U, C, D, rank, world, coll, ph, overlap=True)
            fired.append(sorted(ph))
        torch.cuda.synchronize()
        same = bool(torch.equal(T2, Tg) and torch.equal(ops2.means, ops.means))
        out = (Tg.cpu().numpy(), ops.means.cpu().numpy(), np.array([float(same)]), coll.name + " " + repr(fired[-1]))
        ctx.close()
        return out
    # UBM EM statistics of this rank's frames + the all-reduce of the flat accumulator
    g = ctx.gmm(w, mean, iv)
    xd = torch.from_numpy(x).cuda()
    acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device="cuda")

    def accumulate(fb, fe, a):
        g.em_accumulate(xd[fb:fe], acc=a)
    gd.em_iteration(accumulate, x.shape[0], acc, rank, world, coll)
    torch.cuda.synchronize()
    out = (Tg.cpu().numpy(), ops.means.cpu().numpy(), acc.cpu().numpy(), coll.name)
    if world > 1 and transport == "gloo":
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    g.close(); ctx.close()
    return out


def _rank_main(rank, world, transport, port, idfile, q, overlap=False, kind="small", nb_it=2):
    sys.path.insert(0, ROOT)
    try:
        q.put((rank, _run_rank(rank, world, transport, port, idfile, overlap, kind, nb_it)))
    except Exception as ex:      # noqa: BLE001 - reported to the parent, which fails the test
        import traceback
        q.put((rank, "rank %d: %r\n%s" % (rank, ex, traceback.format_exc())))


def _oracle_loop(C, D, U, mean, iv, N, F, Tm, nb_it):
    To, mo = Tm.copy(), mean.ravel().copy()
    for _ in range(nb_it):
        F0 = orc.tv_subtract_m(N, F, mo)
        o = orc.tv_estimate_a_and_c(N, F0, To, iv.ravel(), orc.tv_tett(To, iv.ravel(), C, D))
        To = orc.tv_update_t(o["A"], o["Cmx"], C, D)
        mo, To = orc.tv_min_divergence(o["Rm"], o["r"], o["meanW"], mo, To, U, C, D)
    return To, mo


@pytest.fixture(scope="module")
def single_rank():
    import torch
    try:
        return _run_rank(0, 1, None, 0, "")
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())     # _run_rank made a side stream current: not for the tests that follow


@pytest.mark.parametrize("transport", ["gloo", "shm"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_tv_em_and_ubm_em_on_one_gpu(world, transport, tmp_path, single_rank):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank_main, args=(r, world, transport, port, str(tmp_path / "comm.id"), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, payload = q.get(timeout=600)
        assert not isinstance(payload, str), payload
        res[r] = payload
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    C, D, R, U, w, mean, iv, N, F, Tm, x = _case()
    T1, m1, acc1, _ = single_rank
    To, mo = _oracle_loop(C, D, U, mean, iv, N, F, Tm, 2)
    for r in range(world):
        Tg, mg, accg, name = res[r]
        assert ("shm" in name) if transport == "shm" else ("gloo" in name)
        # every rank holds the same T / means / statistics, bit for bit
        assert np.array_equal(Tg, res[0][0]) and np.array_equal(mg, res[0][1]) and np.array_equal(accg, res[0][2])
        assert relerr(Tg, T1) < 1e-11 and relerr(mg, m1) < 1e-11          # == the single-rank HIP iteration
        assert relerr(Tg, To) < 1e-6 and relerr(mg, mo) < 1e-8             # == the oracle's TotalVariability loop
        assert relerr(accg[:-2], acc1[:-2]) < 1e-11 and accg[-1] == x.shape[0] and abs(accg[-2] - acc1[-2]) < 1e-9 * abs(acc1[-2])
    og = orc.em_accumulate(orc.Gmm(w, mean, iv), x.astype(np.float64))
    a = res[0][2]
    assert relerr(a[:C], og["occ"]) < 1e-9 and relerr(a[C:C + C * D], og["sx"].ravel()) < 1e-9


@pytest.mark.timeout(1500)
def test_north_star_eight_way_split_of_the_2048_gaussian_ubm_on_one_gpu(tmp_path):
    """BASELINE.json north_star's exact partitioning, executed: EIGHT ranks (processes) on GPU 0 over the C ABI's shm transport, the
    2048-Gaussian x 60-dim UBM in blocks of 256 Gaussians (no padding) -- reduce-scatter of A_packed / Cmx by those blocks, rank g's
    updateTestimate on ITS 256 Gaussians (AccumulateTVStat.cpp:981-1000), all-gather of T, the all-reduce of R / r / meanW,
    minDivergence -- and the TrainWorld all-reduce of the real 247 810-double EM accumulator (AccumulateStat.cpp:286-292).  Every
    rank ends bitwise equal; equal to the single-rank HIP run (summation order differs: 1e-10) and to the oracle's loop."""
    import torch
    import torch.multiprocessing as mp
    world, kind, nb_it = 8, "north_star", 1
    try:
        T1, m1, acc1, _ = _run_rank(0, 1, None, 0, "", False, kind, nb_it)
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rank_main, args=(r, world, "shm", 0, str(tmp_path / "comm.id"), q, False, kind, nb_it)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, payload = q.get(timeout=1200)
        assert not isinstance(payload, str), payload
        res[r] = payload
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    C, D, R, U, w, mean, iv, N, F, Tm, x = _case(kind)
    assert C % world == 0 and C // world == 256 and acc1.size == 247810
    To, mo = _oracle_loop(C, D, U, mean, iv, N, F, Tm, nb_it)
    for r in range(world):
        Tg, mg, accg, name = res[r]
        assert "shm" in name and "8 ranks" in name
        assert np.array_equal(Tg, res[0][0]) and np.array_equal(mg, res[0][1]) and np.array_equal(accg, res[0][2])
        assert relerr(Tg, T1) < 1e-10 and relerr(mg, m1) < 1e-10
        assert relerr(Tg, To) < 1e-6 and relerr(mg, mo) < 1e-8
        assert relerr(accg[:-2], acc1[:-2]) < 1e-10 and accg[-1] == x.shape[0] and abs(accg[-2] - acc1[-2]) < 1e-9 * abs(acc1[-2])
    og = orc.em_accumulate(orc.Gmm(w, mean, iv), x.astype(np.float64))
    a = res[0][2]
    assert relerr(a[:C], og["occ"]) < 1e-9 and relerr(a[C:C + C * D], og["sx"].ravel()) < 1e-9


@pytest.mark.parametrize("world", [2, 3])
def test_overlapped_exchange_is_bitwise_the_serial_order(world, tmp_path, single_rank):
    """tv_em_iteration(overlap=True): the reduce-scatter of A begins from the "tv_a_ready" hook inside gmmiv_tv_estimate_a_and_c
    (into the padded send buffer the E-step accumulates in), the all-gather of T is joined from the "md_factored" hook inside
    gmmiv_tv_min_divergence -- through gmmiv_*_begin / gmmiv_comm_join of the C ABI, 2 and 3 ranks on GPU 0 over shm.  Two iterations
    from the same initial state give bitwise the T and means of the serial order (and the single-rank result to 1e-11)."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rank_main, args=(r, world, "shm", 0, str(tmp_path / "comm.id"), q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, payload = q.get(timeout=600)
        assert not isinstance(payload, str), payload
        res[r] = payload
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    T1, m1, _, _ = single_rank
    for r in range(world):
        Tg, mg, same, name = res[r]
        assert same[0] == 1.0, name
        assert "reduce_scatter" in name and "allgather" in name and "min_divergence" in name
        assert relerr(Tg, T1) < 1e-11 and relerr(mg, m1) < 1e-11


def _shm_rank(rank, world, idfile, n, q):
    sys.path.insert(0, ROOT)
    try:
        import torch
        from lia_ral_amd import capi
        os.environ["GMMIV_COMM_TRANSPORT"] = "shm"
        os.environ["GMMIV_COMM_SHM_SLOT_MB"] = "1"         # 131 072 doubles per slot: n = 300 001 needs 3 chunks, a reduce-scatter more
        torch.cuda.set_device(0)
        ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
        comm = capi.Comm(ctx, world, rank, capi.Comm.exchange_id_file(idfile, rank, 120.0))
        rng = np.random.default_rng(100 + rank)
        a_h = rng.normal(size=n)
        a = torch.from_numpy(a_h.copy()).cuda()
        comm.allreduce(a)
        blk = 100_003
        send_h = rng.normal(size=world * blk)
        send = torch.from_numpy(send_h.copy()).cuda()
        mine = torch.empty(blk, dtype=torch.float64, device="cuda")
        comm.reduce_scatter(send, mine)
        gathered = torch.empty(world * blk, dtype=torch.float64, device="cuda")
        comm.allgather(mine, gathered)
        b = torch.from_numpy(a_h.copy()).cuda()
        comm.broadcast(b, world - 1)
        h = a_h[:1000].copy()
        comm.allreduce(h)                                  # host buffer
        torch.cuda.synchronize()
        q.put((rank, (a.cpu().numpy(), mine.cpu().numpy(), gathered.cpu().numpy(), b.cpu().numpy(), h, comm.backend(), comm.take_bytes())))
        comm.close(); ctx.close()
    except Exception as ex:      # noqa: BLE001
        import traceback
        q.put((rank, "rank %d: %r\n%s" % (rank, ex, traceback.format_exc())))


def test_shm_collectives_chunked(tmp_path):
    """Every collective of the C ABI on the shm transport, payloads larger than a slot (several chunks), three ranks on GPU 0:
    sums in rank order (bitwise equal on every rank, equal to the host sum taken in the same order)."""
    import torch.multiprocessing as mp
    world, n, blk = 3, 300_001, 100_003
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    idfile = str(tmp_path / "shm.id")
    procs = [mpc.Process(target=_shm_rank, args=(r, world, idfile, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, payload = q.get(timeout=600)
        assert not isinstance(payload, str), payload
        res[r] = payload
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert not os.path.exists(idfile)                      # rank 0 retired the id file once the communicator existed
    src = [np.random.default_rng(100 + r) for r in range(world)]
    a_h = [g.normal(size=n) for g in src]
    send_h = [g.normal(size=world * blk) for g in src]
    tot = a_h[0].copy()
    for r in range(1, world):
        tot += a_h[r]
    rs = send_h[0].copy()
    for r in range(1, world):
        rs += send_h[r]
    for r in range(world):
        a, mine, gathered, b, h, backend, nbytes = res[r]
        assert "shm" in backend
        assert np.array_equal(a, tot)
        assert np.array_equal(mine, rs[r * blk:(r + 1) * blk])
        assert np.array_equal(gathered, rs)
        assert np.array_equal(b, a_h[world - 1])
        assert np.array_equal(h, tot[:1000])
        assert nbytes == (n + world * blk + world * blk + n + 1000) * 8


def _run_bench(args, timeout=1200):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` on a box with fewer GPUs fails loudly (no JSON line, non-zero status) instead of measuring one
    GPU and printing n_gpus = 1; so does a launcher whose world size is not --gpus."""
    import torch
    n = torch.cuda.device_count() + 1
    out = _run_bench(["--gpus", str(n), "--frames", "100000", "--steps", "1", "--warmup", "0", "--no-secondary", "--no-cpu-baseline"])
    assert out.returncode != 0 and "one rank per GPU" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "100000", "--steps", "1"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0 and "refusing" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("extra", [["--frames", "300000", "--steps", "1", "--warmup", "1", "--no-secondary"],
                                   ["--workload", "tv", "--tv-utterances", "150", "--steps", "2", "--warmup", "1"]])
def test_bench_self_launches_its_ranks(extra):
    """`python bench.py --gpus 2 --share-gpu`: bench.py starts two ranks itself (torch.distributed.run), they share GPU 0 over the
    C ABI's shm transport; ONE JSON line with n_gpus = 2, the communicator's world and back end, and a green parity block."""
    import json
    out = _run_bench(["--gpus", "2", "--share-gpu"] + extra)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["comm"]["world"] == 2 and "shm" in d["comm"]["backend"] and d["comm"]["gpu_sharing"] is True
    assert d["parity"]["ok"] is True and d["parity"]["max_rel_err"] < d["parity"]["tolerance"]
    assert "gmmiv_comm" in d["collectives"] and d["value"] > 0
    if "tv" in extra:
        assert d["finite"] and set(d["phases_ms"]) >= {"recentre", "tett", "estep", "reduce_scatter", "update_t", "allgather", "min_divergence"}
        assert d["collective_bytes_per_step_per_rank"] > 0
    _check_multi_rank_block(d, 2)


def _check_multi_rank_block(d, world):
    """the fields the first run on real multi-GPU hardware is read by (bench.py multi_rank_report): every rank's own ms_per_step (the
    line's is their MAX), the collectives' times per rank, and the replicated results behind each collective bitwise equal on all ranks"""
    m = d["multi_rank"]
    assert len(m["ms_per_step_per_rank"]) == world and abs(max(m["ms_per_step_per_rank"]) - d["ms_per_step"]) < 1e-6 * d["ms_per_step"] + 1e-9
    assert m["bitwise_equal_across_ranks"] and all(m["bitwise_equal_across_ranks"].values())
    assert len(m["collective_ms_per_step_per_rank"]) == world and all(v is not None for e in m["collective_ms_per_step_per_rank"] for v in e.values())
    assert "rccl_comm_count" in d["comm"] and d["comm"]["world"] == world


@pytest.mark.parametrize("extra", [["--frames", "200000", "--steps", "2", "--warmup", "1", "--no-secondary"],
                                   ["--workload", "tv", "--tv-utterances", "40", "--steps", "1", "--warmup", "1"]])
def test_bench_eight_ranks_on_one_gpu_carry_the_multi_rank_report(extra):
    """`python bench.py --gpus 8 --share-gpu`: exactly the code path the driver's 8-GPU run takes (same launcher, same sharding, same
    collective calls -- over the shm transport, since RCCL refuses two ranks on one device), so that the first run on an 8-GPU node is
    informative from its first line: per-rank step times, per-rank collective times, rccl_comm_count, bitwise-equal replicated results."""
    import json
    out = _run_bench(["--gpus", "8", "--share-gpu"] + extra, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["comm"]["gpu_sharing"] is True and d["parity"]["ok"] is True
    _check_multi_rank_block(d, 8)
    if "tv" in extra:
        assert set(d["multi_rank"]["bitwise_equal_across_ranks"]) == {"T_after_iteration", "ubm_means_after_min_divergence"}
    else:
        assert "em_accumulator_after_allreduce" in d["multi_rank"]["bitwise_equal_across_ranks"]
