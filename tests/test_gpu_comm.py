"""The C ABI's collectives (gmmiv_comm_*, RCCL inside libgmmiv) and the sharded TotalVariability iteration on HIP.

One MI355X per gpurun box: the single-rank forms are checked here (every entry point, device and host buffers, the whole
iteration against the oracle's loop); the 2-rank RCCL test runs when the box shows two devices, the N-rank orchestration
itself is covered on CPU with gloo (tests/test_cpu_plumbing.py)."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def test_single_rank_collectives_are_identities():
    import torch
    from lia_ral_amd import capi
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    comm = capi.Comm(ctx, 1, 0)
    assert comm.world == 1 and comm.rank == 0 and "single rank" in comm.backend()
    a = torch.arange(1000, dtype=torch.float64, device="cuda")
    b = a.clone()
    comm.allreduce(b); comm.broadcast(b, 0)
    out = torch.empty_like(a)
    comm.reduce_scatter(a, out)
    assert torch.equal(out, a) and torch.equal(b, a)
    out.zero_()
    comm.allgather(a, out)
    torch.cuda.synchronize()
    assert torch.equal(out, a)
    h = np.arange(10.0)
    comm.allreduce(h)
    assert np.array_equal(h, np.arange(10.0))
    assert comm.take_bytes() == (1000 * 4 + 10) * 8 and comm.take_bytes() == 0
    with pytest.raises(capi.GmmivError):
        capi.Comm(ctx, 2, 0)                      # world > 1 needs the id of rank 0
    assert len(capi.Comm.unique_id()) == capi.COMM_ID_BYTES     # RCCL resolves at run time (dlopen)
    comm.close(); ctx.close()


def test_tv_em_iterations_on_device_match_oracle_loop():
    """lia_ral_amd.dist.tv_em_iteration with libgmmiv as the compute (device-resident N, F, T, accumulators) == the same
    TotalVariability loop assembled from oracle pieces (TotalVariability.cpp:118-169), two iterations."""
    import torch
    from lia_ral_amd import capi
    from lia_ral_amd import dist as gd
    C, D, R, U = 32, 20, 40, 200
    rng = np.random.default_rng(3)
    N = rng.gamma(0.8, 3.0, (U, C)); F = rng.normal(size=(U, C * D)) * np.sqrt(np.repeat(N, D, 1) + 0.1)
    T0 = rng.normal(0, 0.05, (R, C * D)); iv = rng.uniform(0.5, 2, C * D); means = rng.normal(size=C * D)
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    coll = gd.GmmivCollectives(capi.Comm(ctx, 1, 0))
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    P = R * (R + 1) // 2

    class Ops:
        def __init__(self):
            self.N, self.F, self.T, self.iv, self.means = dv(N), dv(F), dv(T0), dv(iv), dv(means)
            z = lambda *s: torch.zeros(s, dtype=torch.float64, device="cuda")
            self.te = z(C, P)
            self.acc = dict(A=z(C, P), Cmx=z(R, C * D), Rm=z(R, R), r=z(R), meanW=z(R), W=z(U, R))

        def tett(self):
            ctx.tv_tett(self.T, self.iv, C, D, out=self.te)

        def estep(self):
            for k in ("A", "Cmx", "Rm", "r", "meanW"):
                self.acc[k].zero_()
            return ctx.tv_estimate_a_and_c(self.N, self.F, self.T, self.iv, self.te, C, D, acc=self.acc)

        def update_t(self, A_blk, C_blk, cb):
            return ctx.tv_update_t(A_blk, C_blk, cb, D, out=torch.empty((R, cb * D), dtype=torch.float64, device="cuda"))

        def min_divergence(self, acc, Tn, n):
            ctx.tv_min_divergence(acc["Rm"], acc["r"], acc["meanW"] / n, self.means, Tn, n, C, D)
            self.T = Tn
            return Tn

    ops = Ops()
    To, mo = T0.copy(), means.copy()
    for it in range(2):
        phases = {"sync": torch.cuda.synchronize}
        Tg = gd.tv_em_iteration(ops, U, C, D, 0, 1, coll, phases)
        o = orc.tv_estimate_a_and_c(N, F, To, iv, orc.tv_tett(To, iv, C, D))
        To = orc.tv_update_t(o["A"], o["Cmx"], C, D)
        mo, To = orc.tv_min_divergence(o["Rm"], o["r"], o["meanW"], mo, To, U, C, D)
        assert relerr(Tg.cpu().numpy(), To) < 1e-7, it
        assert relerr(ops.means.cpu().numpy(), mo) < 1e-9
        assert set(phases) >= {"tett", "estep", "reduce_scatter", "update_t", "min_divergence"}
    ctx.close()


def test_iteration_hooks_and_begin_join_single_rank():
    """gmmiv_ctx_set_hook: "tv_a_ready" fires once inside tv_estimate_a_and_c when the accumulators are device arrays (never for
    host accumulators: there is nothing on the device to exchange), "md_factored" once inside tv_min_divergence; an exception
    raised in a hook surfaces from the call; unknown points are refused.  gmmiv_*_begin / gmmiv_comm_join on one rank are the
    plain calls."""
    import torch
    from lia_ral_amd import capi
    C, D, R, U = 8, 12, 10, 20
    rng = np.random.default_rng(1)
    N = rng.gamma(0.8, 3.0, (U, C)); F = rng.normal(size=(U, C * D)); Tm = rng.normal(0, 0.1, (R, C * D)); iv = rng.uniform(0.5, 2, C * D)
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    te = ctx.tv_tett(Tm, iv, C, D)
    calls = []
    ctx.set_hook("tv_a_ready", lambda: calls.append("a"))
    ctx.set_hook("md_factored", lambda: calls.append("md"))
    P = R * (R + 1) // 2
    z = lambda *s: torch.zeros(s, dtype=torch.float64, device="cuda")
    acc = dict(A=z(C, P), Cmx=z(R, C * D), Rm=z(R, R), r=z(R), meanW=z(R), W=z(U, R))
    ctx.tv_estimate_a_and_c(dv(N), dv(F), dv(Tm), dv(iv), dv(te), C, D, acc=acc)
    assert calls == ["a"]
    host = ctx.tv_estimate_a_and_c(N, F, Tm, iv, te, C, D)                  # host accumulators: no hook
    assert calls == ["a"] and relerr(acc["A"].cpu().numpy(), host["A"]) < 1e-13
    Td, means = dv(Tm), dv(rng.normal(size=C * D))
    ctx.tv_min_divergence(acc["Rm"].clone(), acc["r"].clone(), acc["meanW"] / U, means, Td, U, C, D)
    assert calls == ["a", "md"]
    ctx.set_hook("md_factored", None)
    ctx.tv_min_divergence(acc["Rm"].clone(), acc["r"].clone(), acc["meanW"] / U, means, dv(Tm), U, C, D)
    assert calls == ["a", "md"]

    def boom():
        raise ValueError("from the hook")
    ctx.set_hook("tv_a_ready", boom)
    with pytest.raises(ValueError, match="from the hook"):
        ctx.tv_estimate_a_and_c(dv(N), dv(F), dv(Tm), dv(iv), dv(te), C, D, acc=acc)
    ctx.set_hook("tv_a_ready", None)
    with pytest.raises(capi.GmmivError):
        ctx.set_hook("no_such_point", lambda: None)
    comm = capi.Comm(ctx, 1, 0)
    a = torch.arange(100, dtype=torch.float64, device="cuda"); out = torch.empty_like(a)
    comm.allreduce_begin(a); comm.reduce_scatter_begin(a, out); comm.join()
    assert torch.equal(out, a)
    out.zero_(); comm.allgather_begin(a, out); comm.join(); comm.join()
    torch.cuda.synchronize()
    assert torch.equal(out, a)
    comm.close(); ctx.close()


def test_every_rccl_entry_point_executes_with_one_rank(monkeypatch):
    """GMMIV_COMM_FORCE_RCCL=1: a ONE-rank communicator is built through RCCL (ncclGetUniqueId, ncclCommInitRank) and every collective
    of the ABI -- plain and overlapped (side stream, fork / join events), device and host buffers -- executes its RCCL call on the one
    GPU of this box: the dlopen'ed symbols, data-type / reduction enums and stream arguments are exercised before a multi-GPU node ever
    sees them.  With one rank every collective must return its input."""
    import torch
    from lia_ral_amd import capi
    monkeypatch.setenv("GMMIV_COMM_FORCE_RCCL", "1")
    ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
    comm = capi.Comm(ctx, 1, 0)
    assert comm.world == 1 and "rccl" in comm.backend(), comm.backend()
    n = 1 << 20
    a = torch.randn(n, dtype=torch.float64, device="cuda")
    ref = a.clone()
    comm.allreduce(a)
    out = torch.zeros_like(a)
    comm.reduce_scatter(a, out)
    g = torch.zeros_like(a)
    comm.allgather(out, g)
    comm.broadcast(g, 0)
    h = np.arange(1000.0)
    comm.allreduce(h); comm.broadcast(h, 0)
    torch.cuda.synchronize(); ctx.sync()
    assert torch.equal(a, ref) and torch.equal(out, ref) and torch.equal(g, ref) and np.array_equal(h, np.arange(1000.0))
    # overlapped forms: the collective runs on the side stream behind the kernel enqueued before it, the join orders the rest behind it
    b = torch.zeros(n, dtype=torch.float64, device="cuda")
    b.add_(ref)                                   # enqueued BEFORE the begin: the collective must see it
    out2 = torch.zeros_like(b); g2 = torch.zeros_like(b)
    comm.allreduce_begin(b)
    comm.reduce_scatter_begin(b, out2)
    comm.allgather_begin(out2, g2)
    comm.join()
    s = g2.sum()                                  # enqueued AFTER the join: must see the gathered data
    torch.cuda.synchronize()
    assert torch.equal(g2, ref) and float(s) == float(ref.sum())
    assert comm.take_bytes() == (4 * n + 2 * 1000 + 3 * n) * 8
    comm.close(); ctx.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_tv_iteration_through_one_rank_rccl(overlap, monkeypatch):
    """The sharded TotalVariability iteration of lia_ral_amd.dist with a ONE-rank RCCL communicator (GMMIV_COMM_FORCE_RCCL=1,
    force_collectives): small all-reduce, reduce-scatter of A and Cmx, all-gather of T all go through RCCL on this GPU -- in the
    overlapped order the reduce-scatter of A really runs on the communicator's side stream while `Cmx += W^T F` runs on the context's,
    started from the "tv_a_ready" hook, and the all-gather is joined from "md_factored".  Two iterations: bitwise the T and means of
    the plain single-rank iteration (which calls update_t directly) -- for a single rank the exchange is the identity."""
    import torch
    from lia_ral_amd import capi
    from lia_ral_amd import dist as gd
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_multirank import DeviceTvOps, _case
    C, D, R, U, w, mean, iv, N, F, Tm, x = _case()
    side = torch.cuda.Stream()
    try:
        with torch.cuda.stream(side):
            ctx = capi.Context(0, side.cuda_stream)
            plain = DeviceTvOps(ctx, C, D, R, N, F, Tm, iv, mean)
            local = gd.GmmivCollectives(capi.Comm(ctx, 1, 0))
            for _ in range(2):
                T0 = gd.tv_em_iteration(plain, U, C, D, 0, 1, local)
            monkeypatch.setenv("GMMIV_COMM_FORCE_RCCL", "1")
            coll = gd.GmmivCollectives(capi.Comm(ctx, 1, 0))
            assert "rccl" in coll.name
            ops = DeviceTvOps(ctx, C, D, R, N, F, Tm, iv, mean)
            for _ in range(2):
                ph = {}
                T1 = gd.tv_em_iteration(ops, U, C, D, 0, 1, coll, ph, overlap=overlap, force_collectives=True)
            torch.cuda.synchronize()
            assert {"reduce_scatter", "update_t", "allgather", "min_divergence"} <= set(ph)
            assert coll.take_bytes() > 0
            assert torch.equal(T1, T0) and torch.equal(ops.means, plain.means)
            ctx.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())


def _rank_main(rank, world, idfile, q):
    sys.path.insert(0, ROOT)
    import torch
    from lia_ral_amd import capi
    torch.cuda.set_device(rank)
    ctx = capi.Context(rank, torch.cuda.current_stream().cuda_stream)
    uid = capi.Comm.exchange_id_file(idfile, rank, 60.0)
    comm = capi.Comm(ctx, world, rank, uid)
    dev = torch.device("cuda", rank)
    a = torch.full((1 << 16,), float(rank + 1), dtype=torch.float64, device=dev)
    comm.allreduce(a)
    send = torch.arange(world * 1000, dtype=torch.float64, device=dev) * (rank + 1)
    mine = torch.empty(1000, dtype=torch.float64, device=dev)
    comm.reduce_scatter(send, mine)
    gathered = torch.empty(world * 1000, dtype=torch.float64, device=dev)
    comm.allgather(mine, gathered)
    h = np.full(7, rank + 1.0)
    comm.allreduce(h)
    torch.cuda.synchronize()
    q.put((rank, float(a[0].item()), mine.cpu().numpy(), gathered.cpu().numpy(), h, comm.backend()))
    comm.close(); ctx.close()


def test_two_rank_rccl_through_the_c_abi(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box: the multi-rank RCCL path needs two devices")
    import torch.multiprocessing as mp
    world = 2
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rank_main, args=(r, world, str(tmp_path / "rccl.id"), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tot = sum(range(1, world + 1))
    full = np.arange(world * 1000, dtype=np.float64) * tot
    for r in range(world):
        assert res[r][1] == tot
        assert np.array_equal(res[r][2], full[r * 1000:(r + 1) * 1000])
        assert np.array_equal(res[r][3], full)
        assert np.array_equal(res[r][4], np.full(7, float(tot)))
        assert "rccl" in res[r][5]


@pytest.mark.parametrize("extra", [["--workload", "tv", "--tv-utterances", "300", "--steps", "2", "--warmup", "1"],
                                   ["--frames", "300000", "--steps", "1", "--warmup", "1", "--no-secondary"]])
def test_bench_cli_emits_the_contract_line(extra):
    """bench.py as the driver launches it (one rank): one JSON line with the contract's keys, roofline and cpu_baseline objects;
    the T-matrix workload stays finite over consecutive EM iterations."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "collectives", "comm", "parity"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["dtype"] == "f64" and "workload" in d["config"]
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert "gmmiv_comm" in d["collectives"] and d["comm"]["world"] == 1
    assert d["parity"]["ok"] is True and d["parity"]["max_rel_err"] < d["parity"]["tolerance"] <= 1e-6, d["parity"]   # the oracle check inside the timed number
    if "tv" in extra:
        assert d["finite"] and set(d["phases_ms"]) >= {"tett", "estep", "update_t", "min_divergence"}
