"""CPU-only checks: the C-ABI library loads and exports every symbol include/gmmiv.h declares, the
product fails loudly without a GPU (no CPU fallback), sharding helpers, and the world_size-2 gloo
path of the distributed E-step."""
import os
import re
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "gmmiv.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gmmiv_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from lia_ral_amd import capi
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(capi.lib, n)]
    assert not missing, missing
    assert capi.lib.gmmiv_em_acc_len(2048, 60) == 2048 * 121 + 2       # the 1.98 MB all-reduce payload
    assert capi.lib.gmmiv_tv_packed_len(400) == 80200


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lia_ral_amd import capi
    with pytest.raises(capi.GmmivError, match="no HIP device"):
        capi.Context(0)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "lia_ral_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.lower(), os.path.join(dp, f)


def test_lds_attribute_is_raised_in_one_place_per_device_and_kernel():
    """A launch with more than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize raised on EVERY device
    it runs on.  csrc/lds_attr.h is the one helper that does it, keyed by (device, kernel); a call anywhere else -- round 4
    had two sites behind a process-wide `static` -- would break the one-context-per-GPU-from-one-process mode gmmiv.h promises
    (the reference gives each worker thread private servers: AccumulateTVStat.cpp:392-393, 422-423)."""
    csrc = os.path.join(ROOT, "lia_ral_amd", "csrc")
    users = 0
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")):
            continue
        txt = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, f)).read())
        if f == "lds_attr.h":
            assert txt.count("hipFuncSetAttribute(") == 1 and "GMMIV_LDS_ATTR_MAX_DEV" in txt
            continue
        assert "hipFuncSetAttribute" not in txt, "%s sets a function attribute outside lds_attr.h" % f
        users += len(re.findall(r"gmmiv_lds_attr<", txt))
        # every launch that passes a run-time LDS size is preceded by the helper in the same function: no `static` size caches left
        assert not re.search(r"static\s+(size_t|int|bool)\s+(attr_\w+|blocks_per_cu)\b", txt), f
    assert users >= 10      # k_llk_mfma, k_stats_mfma, k_topc_determine, k_stats_z, k_tett_packed, the Cholesky family


def test_missing_rccl_is_err_unsupported_not_a_crash():
    """include/gmmiv.h: GMMIV_ERR_UNSUPPORTED when RCCL cannot be loaded (GMMIV_RCCL_LIB names the only library tried).  The
    message is built from ONE dlerror() call -- two calls appended NULL to a std::string and crashed the process."""
    import subprocess
    code = ("import ctypes as ct, sys; sys.path.insert(0, %r); from lia_ral_amd import capi; b = ct.create_string_buffer(128);"
            "rc = capi.lib.gmmiv_comm_get_unique_id_for(b'rccl', b); print(rc); print(capi.lib.gmmiv_last_error().decode())") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, GMMIV_RCCL_LIB="/nonexistent/librccl_not_here.so"))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "-3" and "RCCL not found" in lines[1] and "librccl_not_here" in lines[1]


def test_comm_id_transports():
    """The id rank 0 draws names the transport: "shm" ids carry the magic and a path under GMMIV_COMM_SHM_DIR; an unknown
    transport is an argument error; world > 1 without an id is refused before anything is touched."""
    import ctypes as ct
    from lia_ral_amd import capi
    uid = capi.Comm.unique_id("shm")
    assert len(uid) == capi.COMM_ID_BYTES and uid[:8] == b"GMMIVSHM" and b"/gmmiv_comm_" in uid
    assert capi.Comm.unique_id("shm") != uid                   # unique per draw
    with pytest.raises(capi.GmmivError, match="unknown transport"):
        capi.Comm.unique_id("carrier-pigeon")
    b = ct.create_string_buffer(capi.COMM_ID_BYTES)
    os.environ["GMMIV_COMM_TRANSPORT"] = "shm"
    try:
        assert capi.lib.gmmiv_comm_get_unique_id(b) == 0 and b.raw[:8] == b"GMMIVSHM"      # the environment default
    finally:
        del os.environ["GMMIV_COMM_TRANSPORT"]


def test_shard_range_partitions():
    from lia_ral_amd.dist import shard_range
    for n in (0, 1, 7, 8, 10_000_001):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from conftest import make_frames, make_gmm
    from lia_ral_amd.dist import em_iteration
    from oracle import oracle as orc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    w, mean, iv = make_gmm(16, 12, seed=0)
    x = make_frames(w, mean, iv, 501, seed=1).astype(np.float64)
    g = orc.Gmm(w, mean, iv)
    C, D = 16, 12
    acc = np.zeros(C * (1 + 2 * D) + 2)

    def accumulate(b, e, flat):      # the CPU stand-in for gmmiv_em_accumulate on this rank's frames
        a = orc.em_accumulate(g, x[b:e])
        flat[:C] += a["occ"]; flat[C:C + C * D] += a["sx"].ravel(); flat[C + C * D:C + 2 * C * D] += a["sxx"].ravel()
        flat[-2] += a["llk"]; flat[-1] += a["count"]

    em_iteration(accumulate, len(x), acc, rank, world)
    q.put((rank, acc))
    dist.barrier()
    dist.destroy_process_group()


def _tv_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from lia_ral_amd.dist import tv_estep
    from oracle import oracle as orc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    C, D, R, U = 6, 5, 7, 11
    N = rng.uniform(0.5, 20, (U, C)); F = rng.normal(size=(U, C * D)); T = rng.normal(size=(R, C * D)) * 0.3
    iv = rng.uniform(0.5, 2, C * D)
    te = orc.tv_tett(T, iv, C, D)
    acc = dict(A=np.zeros((C, R * R)), Cmx=np.zeros((R, C * D)), Rm=np.zeros((R, R)), r=np.zeros(R), meanW=np.zeros(R))

    def estimate(b, e, a):      # CPU stand-in for gmmiv_tv_estimate_a_and_c on this rank's utterances
        o = orc.tv_estimate_a_and_c(N[b:e], F[b:e], T, iv, te)
        a["A"] += o["A"]; a["Cmx"] += o["Cmx"]; a["Rm"] += o["Rm"]; a["r"] += o["r"]; a["meanW"] += o["meanW"] * (e - b)

    tv_estep(estimate, U, acc, rank, world)
    q.put((rank, acc))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tv_estep_allreduce_gloo():
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tv_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    C, D, R, U = 6, 5, 7, 11
    N = rng.uniform(0.5, 20, (U, C)); F = rng.normal(size=(U, C * D)); T = rng.normal(size=(R, C * D)) * 0.3
    iv = rng.uniform(0.5, 2, C * D)
    o = orc.tv_estimate_a_and_c(N, F, T, iv, orc.tv_tett(T, iv, C, D))
    for k in ("A", "Cmx", "Rm", "r"):
        assert np.array_equal(res[0][k], res[1][k])
        assert np.allclose(res[0][k], o[k], rtol=1e-12, atol=1e-12), k
    assert np.allclose(res[0]["meanW"] / U, o["meanW"], rtol=1e-12)


def test_two_rank_em_statistics_allreduce_gloo():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import make_frames, make_gmm
    from oracle import oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w, mean, iv = make_gmm(16, 12, seed=0)
    x = make_frames(w, mean, iv, 501, seed=1).astype(np.float64)
    a = orc.em_accumulate(orc.Gmm(w, mean, iv), x)
    ref = np.concatenate([a["occ"], a["sx"].ravel(), a["sxx"].ravel(), [a["llk"], a["count"]]])
    assert np.array_equal(res[0], res[1])                    # every rank holds the same global statistics
    assert np.allclose(res[0], ref, rtol=1e-12, atol=1e-12)  # == single-process accumulation


def _tv_mstep_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from lia_ral_amd.dist import shard_range, tv_mstep_sharded
    from oracle import oracle as orc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rng = np.random.default_rng(1)
    C, D, R, U = 7, 4, 5, 13            # 7 Gaussians over 2 ranks: blocks of 4 and 3
    N = rng.uniform(0.5, 20, (U, C)); F = rng.normal(size=(U, C * D)); T = rng.normal(size=(R, C * D)) * 0.3
    iv = rng.uniform(0.5, 2, C * D)
    te = orc.tv_tett(T, iv, C, D)
    b, e = shard_range(U, rank, world)
    o = orc.tv_estimate_a_and_c(N[b:e], F[b:e], T, iv, te)          # this rank's utterances
    acc = dict(A=o["A"], Cmx=o["Cmx"], Rm=o["Rm"], r=o["r"], meanW=o["meanW"] * (e - b))

    def update_t(A_blk, C_blk, Cb):     # CPU stand-in for gmmiv_tv_update_t on the rank's own Gaussians
        return orc.tv_update_t(A_blk, C_blk, Cb, D)

    Tn = tv_mstep_sharded(acc, update_t, C, D, rank, world)
    q.put((rank, dict(T=Tn, Rm=acc["Rm"], r=acc["r"])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tv_mstep_reduce_scatter_gloo():
    """rank g owns a block of Gaussians -- reduce-scatter of A / Cmx, local T_c = A_c^-1 Cmx_c, all-gather of T."""
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tv_mstep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(1)
    C, D, R, U = 7, 4, 5, 13
    N = rng.uniform(0.5, 20, (U, C)); F = rng.normal(size=(U, C * D)); T = rng.normal(size=(R, C * D)) * 0.3
    iv = rng.uniform(0.5, 2, C * D)
    o = orc.tv_estimate_a_and_c(N, F, T, iv, orc.tv_tett(T, iv, C, D))
    Tref = orc.tv_update_t(o["A"], o["Cmx"], C, D)
    assert np.array_equal(res[0]["T"], res[1]["T"])                      # every rank ends with the same T
    assert np.allclose(res[0]["T"], Tref, rtol=1e-9, atol=1e-12)         # == the single-process M-step
    assert np.allclose(res[0]["Rm"], o["Rm"], rtol=1e-12, atol=1e-12) and np.array_equal(res[0]["Rm"], res[1]["Rm"])


def _tv_iter_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from lia_ral_amd.dist import shard_range, tv_em_iteration
    from oracle import oracle as orc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    C, D, R, U = 7, 4, 5, 13            # 7 Gaussians over 3 ranks: blocks of 3, 3 and 1 (+ 2 of padding)
    rng = np.random.default_rng(1)
    N = rng.uniform(0.5, 20, (U, C)); F = rng.normal(size=(U, C * D)); T = rng.normal(size=(R, C * D)) * 0.3
    iv = rng.uniform(0.5, 2, C * D); means = rng.normal(size=C * D)
    b, e = shard_range(U, rank, world)

    class Ops:                           # CPU stand-ins for the gmmiv_tv_* calls of a GPU rank
        def tett(self):
            self.te = orc.tv_tett(T, iv, C, D)

        def estep(self):
            o = orc.tv_estimate_a_and_c(N[b:e], F[b:e], T, iv, self.te)
            return dict(A=o["A"], Cmx=o["Cmx"], Rm=o["Rm"], r=o["r"], meanW=o["meanW"] * (e - b))

        def update_t(self, A_blk, C_blk, cb):
            return orc.tv_update_t(A_blk, C_blk, cb, D)

        def min_divergence(self, acc, Tn, n):
            self.means, Tm = orc.tv_min_divergence(acc["Rm"], acc["r"], acc["meanW"] / n, means, Tn, n, C, D)
            return Tm

    ops = Ops()
    phases = {}
    Tn = tv_em_iteration(ops, U, C, D, rank, world, phases=phases)
    q.put((rank, dict(T=Tn, means=ops.means, phases=sorted(phases))))
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_tv_em_iteration_padded_blocks_gloo():
    """A whole TotalVariability iteration (estimateTETt, estimateAandC, reduce-scatter, sharded updateTestimate, all-gather,
    minDivergence) on 3 ranks with a Gaussian count that does not divide: same T and means as one process."""
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_tv_iter_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    C, D, R, U = 7, 4, 5, 13
    rng = np.random.default_rng(1)
    N = rng.uniform(0.5, 20, (U, C)); F = rng.normal(size=(U, C * D)); T = rng.normal(size=(R, C * D)) * 0.3
    iv = rng.uniform(0.5, 2, C * D); means = rng.normal(size=C * D)
    o = orc.tv_estimate_a_and_c(N, F, T, iv, orc.tv_tett(T, iv, C, D))
    Tref = orc.tv_update_t(o["A"], o["Cmx"], C, D)
    mref, Tref = orc.tv_min_divergence(o["Rm"], o["r"], o["meanW"], means, Tref, U, C, D)
    for r in range(1, world):
        assert np.array_equal(res[0]["T"], res[r]["T"]) and np.array_equal(res[0]["means"], res[r]["means"])
    assert np.allclose(res[0]["T"], Tref, rtol=1e-9, atol=1e-12)
    assert np.allclose(res[0]["means"], mref, rtol=1e-9, atol=1e-12)
    assert res[0]["phases"] == ["allgather", "estep", "min_divergence", "reduce_scatter", "tett", "update_t"]


def test_component_selection_and_model_normalisation_host_logic():
    """selectComponent / reduceModel / normalizeWeights / normalizeMixture of the C++ host layer (TrainTools.cpp:186-227, :240-315;
    host arithmetic, no device) against the oracle restatement; TabWeight's order is libc qsort's on both sides, ties included."""
    from lia_ral_amd import host_capi as h
    from oracle import oracle as orc
    rng = np.random.default_rng(2)
    C, D = 40, 7
    w = rng.dirichlet(np.full(C, 2.0)); w[5] = w[17] = w[30]; w /= w.sum()          # three equal weights
    mean = rng.normal(size=(C, D)); cov = rng.uniform(0.5, 2.0, (C, D))
    _, _, _, order = h.model_reduce_normalize(w, mean, cov)
    assert np.array_equal(order, orc.sort_by_weight(w)) and np.all(np.diff(w[order]) <= 0)
    for nb_top in (1, 13, 39):
        gw, gm, gc, _ = h.model_reduce_normalize(w, mean, cov, nb_top=nb_top)
        ow, om, oc = orc.reduce_model(w, mean, cov, nb_top)
        assert len(gw) == nb_top and np.array_equal(gw, ow) and np.array_equal(gm, om) and np.array_equal(gc, oc)
        assert abs(gw.sum() - 1.0) < 1e-14
    for mean_only, nb_it in ((False, 1), (True, 3)):
        gw, gm, gc, _ = h.model_reduce_normalize(w, mean, cov, normalize=True, mean_only=mean_only, nb_it=nb_it)
        om, oc = orc.normalize_mixture(w, mean, cov, nb_it, mean_only)
        assert np.allclose(gm, om, rtol=0, atol=1e-13) and np.allclose(gc, oc, rtol=1e-13, atol=0)
        if not mean_only:   # the mixture now has global mean 0 and variance 1
            m1 = (w[:, None] * gm).sum(0); m2 = (w[:, None] * (gc + gm * gm)).sum(0)
            assert np.max(np.abs(m1)) < 1e-12 and np.max(np.abs(m2 - 1.0)) < 1e-12


def test_id_file_of_another_job_is_never_accepted(tmp_path, monkeypatch):
    """gmmiv_comm_exchange_id_file: the published id carries the job's nonce (GMMIV_COMM_JOB, else the launcher's rendezvous
    address / port / run id).  A file left at the same path by a job that died before its communicator existed is recent, but it
    is another job's: a non-root rank must keep waiting for ITS rank 0 instead of joining with the dead id."""
    from lia_ral_amd import capi
    path = str(tmp_path / "id")
    monkeypatch.setenv("GMMIV_COMM_TRANSPORT", "shm")        # the id is drawn without RCCL (no GPU here)
    monkeypatch.setenv("GMMIV_COMM_JOB", "job-A")
    uid_a = capi.Comm.exchange_id_file(path, 0, 5.0)          # rank 0 of job A publishes ... and the job dies
    assert os.path.exists(path)
    assert capi.Comm.exchange_id_file(path, 1, 5.0) == uid_a   # a rank of job A reads it
    monkeypatch.setenv("GMMIV_COMM_JOB", "job-B")
    with pytest.raises(capi.GmmivError, match="waited"):
        capi.Comm.exchange_id_file(path, 1, 0.3)              # a rank of job B, arriving before its rank 0: not fooled
    uid_b = capi.Comm.exchange_id_file(path, 0, 5.0)          # job B's rank 0 replaces the file
    assert uid_b != uid_a and capi.Comm.exchange_id_file(path, 1, 5.0) == uid_b
    # no job variable: the launcher's rendezvous settings are the nonce
    monkeypatch.delenv("GMMIV_COMM_JOB")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1"); monkeypatch.setenv("MASTER_PORT", "29512")
    uid_c = capi.Comm.exchange_id_file(path, 0, 5.0)
    assert capi.Comm.exchange_id_file(path, 1, 5.0) == uid_c
    monkeypatch.setenv("MASTER_PORT", "29513")
    with pytest.raises(capi.GmmivError, match="ANOTHER job's tag"):          # the timeout says WHY the file that was there did not count
        capi.Comm.exchange_id_file(path, 1, 0.3)
    # a nonce of any length works: the file carries a fixed-length tag of it (512 raw bytes were the limit, silently)
    monkeypatch.setenv("GMMIV_COMM_JOB", "J" * 3000)
    uid_d = capi.Comm.exchange_id_file(path, 0, 5.0)
    assert os.path.getsize(path) == 128 + 19 and capi.Comm.exchange_id_file(path, 1, 5.0) == uid_d
    monkeypatch.setenv("GMMIV_COMM_JOB", "J" * 2999 + "K")
    with pytest.raises(capi.GmmivError, match="waited"):
        capi.Comm.exchange_id_file(path, 1, 0.3)
    with open(path, "wb") as f:                                              # a file of a build that wrote no tag
        f.write(b"x" * 128)
    with pytest.raises(capi.GmmivError, match="without this build's job tag"):
        capi.Comm.exchange_id_file(path, 1, 0.3)


def test_compute_map_methods_host_logic():
    """computeMAP (TrainTools.cpp:543-556) and the four methods it dispatches to -- MAPOccDep, MAPModelBased (the same statements),
    MAPConst, MAPConst2 -- with mean / variance / weight adaptation, in the C++ host layer against the oracle restatement (host
    arithmetic, no device).  An unknown method leaves the ML estimate as it is, like the reference ("No adaptation will be perform")."""
    from lia_ral_amd import host_capi as h
    from oracle import oracle as orc
    rng = np.random.default_rng(4)
    C, D = 12, 6
    init = (rng.dirichlet(np.ones(C)), rng.normal(size=(C, D)), rng.uniform(0.5, 2.0, (C, D)))
    client = (rng.dirichlet(np.ones(C)), init[1] + rng.normal(0, 0.3, (C, D)), rng.uniform(0.5, 2.0, (C, D)))
    for method in ("MAPOccDep", "MAPModelBased", "MAPConst", "MAPConst2"):
        for mean, var, weight in ((True, False, False), (True, True, True), (False, True, False)):
            kw = dict(mean=mean, var=var, weight=weight, reg=(14.0, 9.0, 20.0), alpha_mean=0.6)
            got = h.compute_map(method, init, client, 731.0, **kw)
            ref = orc.compute_map(method, init, client, 731.0, **kw)
            for g, r in zip(got, ref):
                assert np.allclose(g, r, rtol=1e-14, atol=0), method
            if method.startswith("MAPConst"):       # means only: variances and weights are the init model's
                assert np.array_equal(got[0], init[0]) and np.array_equal(got[2], init[2])
                assert mean == (not np.array_equal(got[1], init[1]))
    same = h.compute_map("MLLR?", init, client, 731.0)
    assert all(np.array_equal(a, b) for a, b in zip(same, client))
    # MAPOccDep, mean only, one component by hand
    w, m, c = h.compute_map("MAPOccDep", init, client, 100.0, reg=(10.0, 10.0, 10.0))
    a = client[0][3] * 100.0 / (client[0][3] * 100.0 + 10.0)
    assert np.allclose(m[3], (1 - a) * init[1][3] + a * client[1][3], rtol=1e-15) and np.array_equal(c, init[2]) and np.array_equal(w, init[0])


def test_bench_quotes_pmc_traffic_only_for_the_library_it_was_taken_on(tmp_path, monkeypatch):
    """bench.load_traffic (round-4 verdict): profiles/traffic.json records the sha256 of the libgmmiv.so its PMC passes ran on; the
    bench line quotes its figures only when the library that is loaded now has that sha256 -- another build gets `traffic: null` and
    a note that names both builds."""
    import hashlib
    import json
    sys.path.insert(0, ROOT)
    import bench
    so = os.path.join(ROOT, "lia_ral_amd", "csrc", "libgmmiv.so")
    sha = hashlib.sha256(open(so, "rb").read()).hexdigest()
    assert bench.lib_identity()["libgmmiv_sha256"] == sha
    real = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert len(real.get("libgmmiv_sha256", "")) == 64 and real["iv_extractor"]["hbm_bytes_per_utterance"] > 0 and real["tv_em"]["estep_hbm_bytes_per_utterance"] > 0
    fake_root = tmp_path / "repo"
    (fake_root / "profiles").mkdir(parents=True)
    (fake_root / "lia_ral_amd" / "csrc").mkdir(parents=True)
    os.symlink(so, fake_root / "lia_ral_amd" / "csrc" / "libgmmiv.so")
    monkeypatch.setattr(bench, "ROOT", str(fake_root))
    tj, note, me = bench.load_traffic()
    assert tj is None and "not found" in note and me["libgmmiv_sha256"] == sha
    json.dump(dict(real, libgmmiv_sha256="0" * 64), open(fake_root / "profiles" / "traffic.json", "w"))
    tj, note, _ = bench.load_traffic()
    assert tj is None and "PMC figures of another build are not quoted" in note and sha[:16] in note
    json.dump(dict(real, libgmmiv_sha256=sha), open(fake_root / "profiles" / "traffic.json", "w"))
    tj, note, _ = bench.load_traffic()
    assert note is None and tj["k_llk_mfma_hbm_bytes_per_launch"] == real["k_llk_mfma_hbm_bytes_per_launch"]
    d = {k: v for k, v in real.items() if k != "libgmmiv_sha256"}
    json.dump(d, open(fake_root / "profiles" / "traffic.json", "w"))
    assert bench.load_traffic()[0] is None


def _multi_rank_worker(rank, world, port, q, disagree):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    acc = torch.arange(10, dtype=torch.float64) * (1.0 + (1e-16 if False else 0.0))
    if disagree and rank == 1:
        acc = acc.clone(); acc[3] = np.nextafter(3.0, 4.0)        # ONE bit off on one rank
    code = 0
    try:
        rep = bench.multi_rank_report(world, rank, 0.5 + 0.1 * rank, 5, {"acc": bench.tensor_digest(acc)}, {"allreduce": 0.25 * (rank + 1)})
        q.put((rank, rep))
    except SystemExit as e:                                         # ranks whose replicated results differ end the run with status 4
        code = e.code
        q.put((rank, {"exit": code}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("disagree", [False, True])
def test_bench_multi_rank_report_two_ranks_gloo(disagree):
    """bench.py's report for world > 1 (what the first run on multi-GPU hardware is read by): every rank's own ms_per_step and collective
    times in rank order, and the replicated results behind a collective compared BIT FOR BIT over the launcher's group -- one differing bit
    on one rank ends every rank with status 4 instead of a number."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_multi_rank_worker, args=(r, 2, port, q, disagree)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if disagree:
        assert res[0] == {"exit": 4} and res[1] == {"exit": 4}
        return
    for r in (0, 1):
        m = res[r]
        assert m["ms_per_step_per_rank"] == pytest.approx([100.0, 120.0]) and m["bitwise_equal_across_ranks"] == {"acc": True}
        assert m["collective_ms_per_step_per_rank"] == [{"allreduce": 0.25}, {"allreduce": 0.5}]


def test_bench_summary_has_the_five_configs():
    """the LAST key of the bench line: one compact entry per BASELINE config, built from the blocks (a truncated tail of the stored record
    still shows i-vectors/s)"""
    sys.path.insert(0, ROOT)
    import bench
    blk = lambda v, u, key, err: {"value": v, "unit": u, "roofline": {"frac": 0.7}, "cpu_baseline": {"value": 1.0, "unit": u, "cores": 32},
                                  "parity": {key: err, "ok": True}}
    out = dict(blk(126.0, "Gframe-Gaussian/s", "max_rel_err", 5e-16), step_roofline={"frac": 0.78}, roofline={"frac": 0.74},
               secondary=blk(20500.0, "i-vectors/s", "max_rel_err_vs_oracle", 1.6e-14),
               tv_em=dict(blk(55000.0, "utterances/s", "max_rel_err", 1.8e-14)),
               scoring=blk(7.7e10, "trials/s", "max_rel_err_vs_oracle", 6e-15),
               computetest=dict(blk(177.0, "Gframe-Gaussian/s (top-10 world pass)", "max_abs_err_llk", 2e-13), config0={"abs_err": 1e-14}))
    out["tv_em"]["parity"]["whole_iteration"] = {"max_rel_err_T": 2e-12}
    s = bench.summarize(out)
    assert list(s) == ["configs[1] TrainWorld EM", "configs[2] IvExtractor", "configs[3] TotalVariability", "configs[4] IvTest scoring", "configs[0] ComputeTest"]
    assert s["configs[2] IvExtractor"]["value"] == 20500.0 and s["configs[2] IvExtractor"]["cpu"] == [1.0, "i-vectors/s", 32]
    assert s["configs[1] TrainWorld EM"]["frac"] == 0.78 and s["configs[1] TrainWorld EM"]["k1_frac"] == 0.74
    assert s["configs[3] TotalVariability"]["parity_err_T_whole_iteration"] == 2e-12 and s["configs[0] ComputeTest"]["literal_config_llr_abs_err"] == 1e-14
    assert all(e["ok"] for e in s.values()) and len(__import__("json").dumps(s)) < 2500
