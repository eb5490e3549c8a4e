#!/usr/bin/env python3
"""bench.py -- ONE JSON line: the headline (a UBM EM pass, TrainWorld's hot path) and every other single-GPU BASELINE config at its stated size.

Headline (BASELINE.json configs[1]): 2048-Gaussian diagonal UBM, 60-dim float32 frames, 10 M synthetic frames resident in HBM per GPU.
One step = one EM iteration: log-likelihood pass + full-posterior sufficient statistics over every frame (HIP, fp64 MFMA), the RCCL
all-reduce of the 1.98 MB statistics when N > 1, the M-step, variance control and the model re-pack.  Metric: Gframe-Gaussian
evaluations/s, whole job (all ranks).  Scaling is weak (10 M frames per GPU).

Beside it, in the same line (DESIGN.md section 6 has the table):
  `secondary`   configs[2] AT FULL SIZE: IvExtractor end to end on 10 000 utterances x 3000 frames per GPU -> i-vectors/s, with its own
                `roofline` (2.67 GFLOP per i-vector), `cpu_baseline` (oracle IvExtractor, 1 and 32 threads), parity, and the opt-in paths
                (pruned posteriors, single-pass statistics) as A/B legs on a 512-utterance slice (`slice_ab`);
  `computetest` configs[0]'s path at scale: top-10 world pass on 10^6 frames + client pass for 4 client models;
  `host_layer`  the same three workloads through the C++ host layer (libliatools_gpu.so: liagpu::trainModelStream at
                baggedFrameProbability 1.0 and 0.4, IvExtractor, computeTestLLR), each with its time, the ratio to the torch-driven number
                and parity against it (one rank only);
  `tv_em`       configs[3]: one TotalVariability T-matrix EM iteration per step on utterance-sharded statistics, 6250 utterances x 3000
                frames per GPU (= 50 k over 8 GPUs) -- at world 1 too; `--workload tv` runs this block alone;
  `scoring`     configs[4] AT FULL SIZE: 100 k x 100 k trials of dimension 400, all four rules, 80 GB of scores resident (one rank only);
  `dense_data`, `kernels`, `roofline`, `step_roofline`, `cpu_baseline`, `screening`, `library` (sha256 of the libgmmiv.so that ran:
  `roofline.traffic` figures are quoted from profiles/traffic.json only when they were collected on that build).
The collectives are the C ABI's own (gmmiv_comm_*: RCCL called by libgmmiv on the device buffers); torch.distributed only launches
the ranks and carries the 128-byte RCCL id.

Launch: python bench.py --gpus N (N > 1 without a launcher's WORLD_SIZE: bench.py starts its N ranks itself, like the reference
tools start their own worker threads, AccumulateStat.cpp:234-299) | python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N.  The number of ranks must equal --gpus and every rank needs its own GPU -- anything else exits non-zero, unless
--share-gpu is given: a correctness mode in which the ranks share the visible device(s) over the C ABI's "shm" transport
(gloo carries the launcher's small messages); its throughput says nothing about scaling and the JSON line says so.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

C, D = 2048, 60
FLOP_PER_PAIR_LLK = 240.0     # logit of one frame-Gaussian pair, 2 flop x 2D
FLOP_PER_PAIR_ACC = 242.0     # statistics of one pair: 2 flop x (1 + D + D)
FLOP_PER_PAIR_STATS = 480.0   # the recomputing k_stats_mfma (stats_z = 0): logits again + statistics
KERNEL_FLOP = {"k_llk_mfma": FLOP_PER_PAIR_LLK, "k_stats_z": FLOP_PER_PAIR_ACC, "k_stats_mfma": FLOP_PER_PAIR_STATS,
               }
PEAK_F64_TFLOPS = 78.6        # MI355X fp64 matrix = vector peak (AMD datasheet; measured ceiling in DESIGN.md)


def lib_identity():
    """sha256 of the libgmmiv.so this process runs (the build is deterministic: the library built from a revision is the same
    file here and on the GPU box) and the git revision when the tree has one (the gpurun snapshot has none)."""
    import hashlib
    import subprocess
    so = os.path.join(ROOT, "lia_ral_amd", "csrc", "libgmmiv.so")
    sha = hashlib.sha256(open(so, "rb").read()).hexdigest() if os.path.exists(so) else None
    try:
        rev = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        rev = None
    return {"libgmmiv_sha256": sha, "git_rev": rev}


def load_traffic():
    """profiles/traffic.json = HBM bytes from the committed rocprofv3 PMC passes (tools/profile_r05.sh + tools/make_traffic.py).
    PMC counters can only be read by a rocprofv3 run, so the line quotes the figures of those passes -- but ONLY when they were
    collected on the library that is running now (the file records its sha256): with another build `traffic` is null and
    `traffic_note` says why (round-4 verdict: a stale file would have gone unnoticed)."""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    me = lib_identity()
    if not os.path.exists(tf):
        return None, "profiles/traffic.json not found", me
    try:
        tj = json.load(open(tf))
    except Exception as e:      # noqa: BLE001
        return None, "profiles/traffic.json unreadable: %r" % (e,), me
    if not tj.get("libgmmiv_sha256"):
        return None, "profiles/traffic.json does not record the library it was collected on", me
    if tj["libgmmiv_sha256"] != me["libgmmiv_sha256"]:
        return None, ("profiles/traffic.json was collected on libgmmiv.so %s (git %s), this run loads %s: PMC figures of another build are "
                      "not quoted -- re-run tools/profile_r05.sh" % (tj["libgmmiv_sha256"][:16], tj.get("git_rev"), (me["libgmmiv_sha256"] or "?")[:16])), me
    return tj, None, me


def screen_once(ctx, x, what):
    """A one-time check that the synthetic frames of a block hold no unusable value (include/gmmiv.h "DEGENERATE INPUTS").  It is a check
    of the DATA, not a switch: every timed call below runs with the library's defaults -- "assume_finite" 0, i.e. with the pass that
    counts unusable frames on the device at the top of every frame-consuming call (it never waits for the stream; round 6)."""
    import ctypes as ct
    from lia_ral_amd import capi
    n = ct.c_int64(-1)
    capi._chk(capi.lib.gmmiv_count_unusable_frames(ctx._h, capi._ptr(x), capi.F32, ct.c_int64(x.shape[0]), ct.c_int64(x.shape[1]), x.shape[1], ct.byref(n)))
    if n.value != 0:
        raise RuntimeError("%s: %d unusable frames in synthetic data" % (what, n.value))
    return {"frames": int(x.shape[0]), "unusable": 0, "what": what}


def synth_frames(w, mean, iv, T, device, seed):
    """x = mu_c + sqrt(var_c) N(0,1), component ~ weights, float32 (generated on the GPU)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    wt = torch.from_numpy(w).to(device)
    mt = torch.from_numpy(mean).to(device=device, dtype=torch.float32)
    st = torch.from_numpy(1.0 / np.sqrt(iv)).to(device=device, dtype=torch.float32)
    x = torch.empty((T, D), dtype=torch.float32, device=device)
    step = 1 << 20
    for b in range(0, T, step):
        n = min(step, T - b)
        comp = torch.multinomial(wt, n, replacement=True, generator=g)
        x[b:b + n] = mt[comp] + st[comp] * torch.randn((n, D), device=device, dtype=torch.float32, generator=g)
    return x


def physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo; falls back to the logical count."""
    try:
        seen, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                seen.add((phys, line.split(":")[1].strip()))
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_baseline(w, mean, iv, seed):
    """Oracle port (-O3 -ffast-math, pthreads with the reference's frame partitioning and one private accumulator per thread,
    AccumulateStat.cpp:170-299) on the host cores: 1 thread, then a sweep up to the logical core count.  The private
    accumulators are 2 MB per thread (C (1 + 2 D) doubles, like the reference's per-thread MixtureStat), so past the point
    where they stop fitting the last-level cache the loop is DRAM-bound: the best thread count is reported as `value`, the
    single-thread figure and the whole sweep beside it."""
    from conftest import make_frames
    from oracle import oracle as orc
    logical, phys = os.cpu_count() or 1, physical_cores()
    # the sweep stops at half the physical cores: rounds 2-4 measured the 128- and 256-thread points SLOWER than the 32-thread one on
    # the 128-core host (the per-thread 2 MB accumulators leave the last-level cache) -- 20 s of driver time that said nothing new
    counts = sorted({1, max(1, phys // 8), max(1, phys // 4), max(1, phys // 2)})
    g = orc.Gmm(w, mean, iv)
    # Every point runs for about TARGET_S seconds: its frame count is sized from the aggregate rate the PREVIOUS point measured (the
    # first from a short single-thread probe), at least 4000 frames per thread -- the many-thread points then time the steady loop
    # (private 2 MB accumulators that no longer fit the last-level cache: the rate FALLS past ~32 threads on a 128-core host) and
    # not thread start-up (round 2 gave every thread 4000 frames whatever the rate).
    TARGET_S, MAX_FRAMES = 2.5, 1_500_000
    probe = make_frames(w, mean, iv, 6000, seed=seed).astype(np.float64)
    orc.em_accumulate(g, probe[:2000], fast=True, threads=1)                   # warm-up / page-in
    t = time.time()
    orc.em_accumulate(g, probe[2000:], fast=True, threads=1)
    rate = 4000 / max(time.time() - t, 1e-6)                                   # frames per second, all threads of the last point together
    base = make_frames(w, mean, iv, 150_000, seed=seed + 1).astype(np.float64)  # 150 k distinct frames, repeated: the loop's cost does not depend on them
    x = np.tile(base, (MAX_FRAMES // base.shape[0], 1))
    sweep = []
    for th in counts:
        frames = int(min(MAX_FRAMES, max(4000 * th, rate * TARGET_S)))
        t = time.time()
        orc.em_accumulate(g, x[:frames], fast=True, threads=th)
        dt = time.time() - t
        rate = frames / dt
        sweep.append({"threads": th, "frames": frames, "seconds": dt, "gpairs_per_s": frames * C / dt / 1e9})
    best = max(sweep, key=lambda r: r["gpairs_per_s"])
    return {"value": best["gpairs_per_s"], "unit": "Gframe-Gaussian/s", "cores": best["threads"], "kind": "port",
            "single_thread": sweep[0]["gpairs_per_s"], "logical_cores": logical, "physical_cores": phys, "sweep": sweep,
            "sample": "one EM statistics pass per thread count over %s frames x %d Gaussians (each point sized for ~%.1f s from the "
                      "rate of the point before; oracle/oracle_mt.c, gcc -O3 -ffast-math like the reference; a restatement of the reference "
                      "loops, not the original binary), %.1f s in all"
                      % ("/".join(str(r["frames"]) for r in sweep), C, TARGET_S, sum(r["seconds"] for r in sweep))}


def ivector_secondary(ctx, g, w, mean, iv, dev, rank, world, U=10_000, frames=3000, R=400, check=True, slice_u=512, traffic=None):
    """BASELINE.json configs[2] AT ITS STATED SIZE: IvExtractor end-to-end (Baum-Welch N/F statistics, substractM,
    L = I + sum N TETt, SPD inverse, w = L^-1 T Sigma^-1 F) for U = 10 000 utterances x 3000 frames per GPU, everything resident
    in HBM (features 7.2 GB float32, F 9.8 GB, N 0.16 GB, TETt 1.3 GB); TETt is precomputed once (T is fixed during extraction,
    IvExtractor.cpp:136).  The two opt-in variants (single-pass statistics kernel, pruned posteriors) are A/B legs on the first
    `slice_u` utterances, next to the default path timed on the same slice."""
    T = U * frames
    x = synth_frames(w, mean, iv, T, dev, seed=777 + rank)
    screened = screen_once(ctx, x, "i-vector secondary")
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    Tm = 0.01 * torch.randn((R, C * D), dtype=torch.float64, device=dev, generator=gen)
    invvar = torch.from_numpy(iv.ravel().copy()).to(dev)
    means = torch.from_numpy(mean.ravel().copy()).to(dev)
    P = R * (R + 1) // 2
    tett = torch.empty((C, P), dtype=torch.float64, device=dev)
    ctx.tv_tett(Tm, invvar, C, D, out=tett)
    N = torch.empty((U, C), dtype=torch.float64, device=dev)
    F = torch.empty((U, C * D), dtype=torch.float64, device=dev)
    W = torch.empty((U, R), dtype=torch.float64, device=dev)
    ub = np.arange(U + 1, dtype=np.int64) * frames
    times = {}

    def run(n=U):
        t0 = time.perf_counter()
        g.tv_stats(x[:n * frames], ub[:n + 1], N[:n], F[:n])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        times["k1_ms"] = ctx.kernel_ms("k_llk_mfma"); times["k3_ms"] = ctx.kernel_ms("k_stats_z")
        times["k1_launches"] = ctx.kernel_launches("k_llk_mfma")
        ctx.tv_subtract_m(N[:n], F[:n], means, C, D)
        ctx.tv_estimate_w(N[:n], F[:n], Tm, invvar, tett, C, D, out=W[:n])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        times["stats_ms"] = (t1 - t0) * 1e3
        times["solve_ms"] = (t2 - t1) * 1e3
        return t2 - t0

    run(); run()               # two warm-ups (workspace allocation, then the allocator's second look at it)
    NRUN = 3
    runs = []
    stats_ms = solve_ms = 0.0
    k1_ms = k3_ms = 0.0
    for _ in range(NRUN):
        runs.append(run())
        stats_ms += times["stats_ms"] / NRUN; solve_ms += times["solve_ms"] / NRUN
        k1_ms += times["k1_ms"] / NRUN; k3_ms += times["k3_ms"] / NRUN
    k1_launches = times["k1_launches"]
    dt = float(np.mean(runs))  # the MEAN of the timed runs
    dt = max_over_ranks(dt, world, dev)
    parity = ivector_parity(x, frames, w, mean, iv, Tm, W, [0, 1, U // 2, U - 1]) if rank == 0 and check else None
    finite = bool(torch.isfinite(W).all().item())
    W_all = W.clone()
    # ---- A/B legs on the first slice_u utterances (opt-in paths; never `value`) ----
    S = min(slice_u, U)
    run(S); run(S)
    sl = [run(S) for _ in range(NRUN)]
    slice_default = {"utterances": S, "ms": float(np.mean(sl)) * 1e3, "i-vectors/s": S / float(np.mean(sl)), "stats_ms": times["stats_ms"], "solve_ms": times["solve_ms"]}
    W_slice = W[:S].clone()
    # OPT-IN (not the default, not `value`): the N / F statistics with posteriors below 2^-100 skipped in groups (option "prune_log2":
    # at most 2^-100 of posterior mass per pair is dropped -- 1e-30 absolute on N / F, invisible in L, aux and the i-vector, but a
    # Gaussian whose whole occupancy is that small gets different statistics than the reference's sum of denormal-scale terms)
    pruned = {}
    try:
        ctx.set_option("prune_log2", 100)
        run(S)
        pr = [run(S) for _ in range(NRUN)]
        pdt = float(np.mean(pr))
        pruned = {"prune_log2": 100, "utterances": S, "value": S / pdt, "unit": "i-vectors/s", "stats_ms": times["stats_ms"], "k_stats_z_ms": times["k3_ms"],
                  "max_rel_diff_ivectors_vs_default": float(((W[:S] - W_slice).abs().max() / W_slice.abs().max()).item()),
                  "parity": ivector_parity(x, frames, w, mean, iv, Tm, W, [0, 1, S // 2, S - 1]) if rank == 0 and check else None}
    finally:
        ctx.set_option("prune_log2", 0)
    W.copy_(W_all)
    del W_all
    # 2.67 GFLOP per i-vector end to end = 3000 frames x 2048 x 362 (K1 logits 240 + K3 N / F statistics 122) + 448.5 M (solve)
    flop_stats = float(frames) * C * (FLOP_PER_PAIR_LLK + 2.0 * (1 + D)); flop_solve = 448.5e6
    tf = (flop_stats + flop_solve) * U / (dt / 1.0) / 1e12           # this rank's U i-vectors in dt (the slowest rank's time)
    tr = tr_note = None
    if traffic and traffic[0] and traffic[0].get("iv_extractor"):
        e = traffic[0]["iv_extractor"]      # HBM bytes of ONE pass over `utterances` utterances, all kernels (fetch x 2 + write, PMC)
        tr = e["hbm_bytes_per_utterance"] * U
        tr_note = e.get("source")
    elif traffic:
        tr_note = traffic[1] or "no iv_extractor entry in profiles/traffic.json"
    roof = {"bound": "mfma", "kernel": "IvExtractor end to end: k_llk_mfma<WZ> + k_stats_z<SQ=0> + L / aux GEMMs + chol_fused",
            "achieved": tf, "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F64_TFLOPS, "traffic": tr, "traffic_source": tr_note,
            "traffic_unit": "HBM bytes of one pass over the %d utterances (all kernels)" % U,
            "algorithmic_flop_per_ivector": flop_stats + flop_solve,
            "algorithmic_bytes": float(T) * D * 4 + float(U) * R * 8,      # the feature stream read once + the i-vectors written
            "parts": {"k_llk_mfma": {"ms": k1_ms, "launches": k1_launches, "tflops": FLOP_PER_PAIR_LLK * T * C / (k1_ms * 1e-3) / 1e12 if k1_ms > 0 else None},
                      "k_stats_z(N,F)": {"ms": k3_ms, "tflops": 2.0 * (1 + D) * T * C / (k3_ms * 1e-3) / 1e12 if k3_ms > 0 else None},
                      "solve (substractM + estimateW)": {"ms": solve_ms, "tflops": flop_solve * U / (solve_ms * 1e-3) / 1e12}},
            "time_fractions": {"k_llk_mfma": k1_ms / (dt * 1e3), "k_stats_z": k3_ms / (dt * 1e3), "solve": solve_ms / (dt * 1e3)}}
    return {"metric": "i-vectors/s (IvExtractor end-to-end, 2048-g UBM, rank 400, 3000-frame utterances)",
            "config": "BASELINE.json configs[2] at its stated size: %d utterances x %d frames per GPU" % (U, frames),
            "value": U * world / dt, "unit": "i-vectors/s", "utterances_per_gpu": U, "frames_per_gpu": T, "stats_ms": stats_ms,
            "solve_ms": solve_ms, "timed_runs_ms": [r * 1e3 for r in runs], "timing": "mean of %d runs after two warm-ups" % NRUN,
            "hbm_resident_gb": {"features_f32": T * D * 4 / 1e9, "F": U * C * D * 8 / 1e9, "N": U * C * 8 / 1e9, "tett_packed": C * P * 8 / 1e9},
            "screening": screened, "finite": finite, "parity": parity, "roofline": roof,
            "slice_ab": {"default": slice_default, "pruned_posteriors": pruned},
            "_W": W[:S].clone(), "_Tm": Tm, "_x_slice": x[:S * frames].clone(), "_slice": S}


def ivector_cpu_baseline(w, mean, iv, R=400, frames=3000):
    """The oracle IvExtractor (-O3 -ffast-math like the reference's default build) on the host cores: 1 thread, then a sweep with
    the reference's split -- contiguous utterance ranges per thread (AccumulateTVStat.cpp:498-507, :2282-2300) -- one utterance
    per thread (a 3000-frame utterance costs a scalar core several seconds).  TETt is precomputed, like estimateTETt before the threads."""
    from conftest import make_frames
    from oracle import oracle as orc
    logical, phys = os.cpu_count() or 1, physical_cores()
    g = orc.Gmm(w, mean, iv)
    rng = np.random.default_rng(5)
    Tm = rng.normal(0, 0.01, (R, C * D))
    t = time.time()
    Tc = np.ascontiguousarray(Tm.reshape(R, C, D).transpose(1, 0, 2))          # TETt_c = T_c Sigma_c^-1 T_c^T: an INPUT of the timed loop
    te = np.matmul(Tc * iv[:, None, :], Tc.transpose(0, 2, 1)).reshape(C, R * R)  # (numpy / BLAS here; the oracle's own estimateTETt is a 20 GFLOP scalar loop)
    del Tc
    t_tett = time.time() - t
    base = make_frames(w, mean, iv, frames * 4, seed=31).astype(np.float64)
    sweep = []
    counts = sorted({1, max(1, phys // 4)})     # one utterance per thread, ~4 s each: 1 thread and a quarter of the cores (the best point of rounds 3-4)
    budget = 25.0
    for th in counts:
        if sweep and budget < 1.5 * sweep[-1]["seconds"]:
            break
        U = th                                            # one utterance per thread
        x = np.tile(base, ((U + 3) // 4, 1))[:U * frames]
        ub = np.arange(U + 1, dtype=np.int64) * frames
        t = time.time()
        orc.iv_extract_mt(g, x, ub, Tm, iv.ravel(), te, threads=th)
        dt = time.time() - t
        budget -= dt
        sweep.append({"threads": th, "utterances": U, "seconds": dt, "ivectors_per_s": U / dt})
    best = max(sweep, key=lambda r: r["ivectors_per_s"])
    return {"value": best["ivectors_per_s"], "unit": "i-vectors/s", "cores": best["threads"], "kind": "port",
            "single_thread": sweep[0]["ivectors_per_s"], "logical_cores": logical, "physical_cores": phys, "sweep": sweep,
            "tett_once_s": t_tett,
            "sample": "oracle IvExtractor (statistics + substractM + estimateW, scalar fp64, gcc -O3 -ffast-math; a restatement of the "
                      "reference loops, not the original binary) on one %d-frame utterance per thread at C=%d, R=%d; TETt precomputed (numpy, "
                      "%.1f s) and excluded like estimateTETt in the GPU number; %.1f s in all" % (frames, C, R, t_tett, sum(r["seconds"] for r in sweep))}


def computetest_secondary(ctx, g, w, mean, iv, dev, rank, world, T=1_000_000, ctop=10, n_clients=4, check=True):
    """BASELINE.json configs[0]'s path at scale (ComputeTest.cpp:154-207): DETERMINE_TOP_DISTRIBS on the 2048-Gaussian world model
    for T frames per GPU, then USE_TOP_DISTRIBS for a few mean-adapted client models in one call.  The checker leg compares the first
    256 frames with the CPU oracle: exact indices, per-frame log-likelihoods of world and clients."""
    from lia_ral_amd import capi
    import ctypes as ct
    x = synth_frames(w, mean, iv, T, dev, seed=991 + rank)
    screen_once(ctx, x, "ComputeTest frames")
    rng = np.random.default_rng(17)
    cm = [mean + rng.normal(0, 0.1, mean.shape) for _ in range(n_clients)]
    clients = [ctx.gmm(w, m, iv) for m in cm]
    idx = torch.empty((T, ctop), dtype=torch.int32, device=dev)
    nllk = torch.empty(T, dtype=torch.float64, device=dev)
    llkw = torch.empty(T, dtype=torch.float64, device=dev)
    llkc = torch.empty((n_clients, T), dtype=torch.float64, device=dev)
    handles = (ct.c_void_p * n_clients)(*[c._h for c in clients])

    def world_pass():
        capi._chk(capi.lib.gmmiv_llk_determine_top(ctx._h, g._h, capi._ptr(x), capi.F32, ct.c_int64(T), ct.c_int64(D), ctop, capi.TOP_COMPLETE,
                                                   ct.c_double(-200.0), ct.c_double(200.0), capi._ptr(idx), None, None, capi._ptr(nllk), None, capi._ptr(llkw)))

    def client_pass():
        capi._chk(capi.lib.gmmiv_llk_use_top_multi(ctx._h, n_clients, handles, capi._ptr(x), capi.F32, ct.c_int64(T), ct.c_int64(D), ctop, capi._ptr(idx),
                                                   capi._ptr(nllk), capi.TOP_COMPLETE, ct.c_double(-200.0), ct.c_double(200.0), capi._ptr(llkc)))

    def timed(f):
        f(); torch.cuda.synchronize()
        best = 1e30
        for _ in range(2):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best
    tw = max_over_ranks(timed(world_pass), world, dev)
    tc = max_over_ranks(timed(client_pass), world, dev)
    parity = None
    if rank == 0 and check:
        from oracle import oracle as orc
        n = 256
        xo = x[:n].cpu().numpy().astype(np.float64)
        do = orc.llk_determine_top(orc.Gmm(w, mean, iv), xo, ctop, True)
        same_idx = bool(np.array_equal(idx[:n].cpu().numpy(), do["idx"]))
        err = float(np.max(np.abs(llkw[:n].cpu().numpy() - do["llk"])))
        for i in range(n_clients):
            lco = orc.llk_use_top(orc.Gmm(w, cm[i], iv), xo, do["idx"], do["nontop_lk"], True)
            err = max(err, float(np.max(np.abs(llkc[i, :n].cpu().numpy() - lco))))
        parity = {"frames_checked": n, "indices_identical": same_idx, "max_abs_err_llk": err, "tolerance": 1e-9, "ok": same_idx and err < 1e-9,
                  "what": "top-%d indices of the world pass (exact) and per-frame log-likelihoods of world and %d client models vs the CPU oracle" % (ctop, n_clients)}
    llr = (llkc.mean(1) - llkw.mean()).cpu().numpy()       # ComputeTest's file-mode score per client (mean llk_client - mean llk_world)
    world_tf = FLOP_PER_PAIR_LLK * T * C / tw / 1e12
    extra = {"roofline": {"bound": "mfma", "kernel": "k_llk_mfma<TC> + k_topc_rank2 (world pass)", "achieved": world_tf, "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s",
                          "frac": world_tf / PEAK_F64_TFLOPS, "traffic": None, "algorithmic_flop_per_pair": FLOP_PER_PAIR_LLK, "kernel_ms": tw * 1e3,
                          "algorithmic_bytes": float(T) * (D * 4 + ctop * 4 + 16)}}
    if rank == 0 and check and world == 1:
        extra["cpu_baseline"], extra["config0"] = computetest_cpu_and_config0(ctx, w, mean, iv, cm[0], ctop)
    return {"metric": "ComputeTest: top-%d world pass, frame-Gaussian pairs/s; client pass, frames/s per client model" % ctop, **extra,
            "value": T * C * world / tw / 1e9, "unit": "Gframe-Gaussian/s (top-%d world pass)" % ctop,
            "_llr": llr, "_client_means": cm, "_seed": 991 + rank,
            "world_pass_gpairs_per_s": T * C * world / tw / 1e9, "world_pass_ms": tw * 1e3,
            "client_pass_mframes_per_s_per_client": T * n_clients * world / tc / 1e6, "client_pass_ms": tc * 1e3,
            "frames_per_gpu": T, "clients": n_clients, "parity": parity}


def computetest_cpu_and_config0(ctx, w, mean, iv, client_mean, ctop):
    """(1) cpu_baseline of the ComputeTest block: the oracle's DETERMINE_TOP_DISTRIBS + USE_TOP_DISTRIBS loop (ComputeTest.cpp:154-207 restated,
    the checker's strict -O2 build, ONE thread like the reference tool) on a bounded sample of the same 2048-Gaussian workload.
    (2) BASELINE.json configs[0] LITERALLY: 128-Gaussian UBM, 60 dims, 1000 frames, top-10 COMPLETE, one client -- the LLR through libgmmiv
    and through the oracle (value parity, microseconds per frame on one CPU thread; no throughput claim)."""
    from conftest import make_frames, make_gmm
    from oracle import oracle as orc
    n = 50_000                                                              # ~10 s of one CPU thread
    xs = make_frames(w, mean, iv, n, seed=4242).astype(np.float64)
    og, oc = orc.Gmm(w, mean, iv), orc.Gmm(w, client_mean, iv)
    t0 = time.time()
    do = orc.llk_determine_top(og, xs, ctop, True)
    t1 = time.time()
    orc.llk_use_top(oc, xs, do["idx"], do["nontop_lk"], True)
    t2 = time.time()
    cpu = {"value": n * C / (t1 - t0) / 1e9, "unit": "Gframe-Gaussian/s (top-%d world pass)" % ctop, "cores": 1, "kind": "port",
           "client_pass_mframes_per_s_per_client": n / (t2 - t1) / 1e6,
           "sample": "oracle DETERMINE_TOP_DISTRIBS on %d frames x %d Gaussians (%.1f s) + USE_TOP_DISTRIBS of one client (%.3f s), one thread, gcc -O2 "
                     "(the checker's build; a restatement of the reference loops, not the original binary)" % (n, C, t1 - t0, t2 - t1)}
    w0, m0, iv0 = make_gmm(128, 60, seed=0)
    x0 = make_frames(w0, m0, iv0, 1000, seed=1)
    rng = np.random.default_rng(5)
    mc = m0 + rng.normal(0, 0.1, m0.shape)
    gw, gc = ctx.gmm(w0, m0, iv0), ctx.gmm(w0, mc, iv0)
    gw.llk_determine_top(x0, 10, True)                                       # warm-up
    t0 = time.perf_counter()
    d = gw.llk_determine_top(x0, 10, True)
    lc = gc.llk_use_top(x0, d["idx"], d["nontop_llk"], True)
    dt_gpu = time.perf_counter() - t0
    llr = float(lc.mean() - d["llk"].mean())
    xo = x0.astype(np.float64)
    t0 = time.time()
    do0 = orc.llk_determine_top(orc.Gmm(w0, m0, iv0), xo, 10, True)
    lco = orc.llk_use_top(orc.Gmm(w0, mc, iv0), xo, do0["idx"], do0["nontop_lk"], True)
    dt_cpu = time.time() - t0
    llr_o = float(lco.mean() - do0["llk"].mean())
    gw.close(); gc.close()
    cfg0 = {"workload": "BASELINE.json configs[0]: ComputeTest LLR, 128-Gaussian diag UBM, 60-dim, 1000 synthetic frames, top-10 COMPLETE, one client",
            "llr": llr, "llr_oracle": llr_o, "abs_err": abs(llr - llr_o), "indices_identical": bool(np.array_equal(d["idx"], do0["idx"])),
            "ok": bool(abs(llr - llr_o) < 1e-9 and np.array_equal(d["idx"], do0["idx"])),
            "cpu_us_per_frame_1thread": dt_cpu / 1000 * 1e6, "gpu_call_ms_host_arrays": dt_gpu * 1e3,
            "note": "plumbing + parity configuration (host arrays in and out: the GPU figure is a call latency, not a rate)"}
    return cpu, cfg0


SCORE_FLOP = 800.0          # SURVEY 8(d) / BASELINE.md: one score of dim 400 as a GEMM element: 2 x 400 flop, 8 B written


def scoring_block(ctx, dev, M=100_000, S=100_000, R=400, rf=200, check=True, traffic=None):
    """BASELINE.json configs[4] AT ITS STATED SIZE: IvTest scoring of 100 k enrolment x 100 k test i-vectors of dimension 400 on one
    MI355X, all four rules (PldaTest::cosineDistance / mahalanobisDistance / twoCovScoring / pldaScoringUnThreaded,
    PldaTools.cpp:3842-3909, 4083-4271).  The 100 000 x 100 000 fp64 score matrix (80 GB) STAYS in HBM -- what leaves the device is a
    checksum and the sampled trials the oracle checks (shipping it over PCIe would take 80 GB / 63 GB/s = 1.3 s against 0.13 s of
    compute); it is freed before the next block.  Returns the JSON block (value = Mahalanobis trials/s: the rule the reference
    spends an O(dim^2) product per pair on)."""
    gen = torch.Generator(device=dev); gen.manual_seed(11)
    models = torch.randn((R, M), dtype=torch.float64, device=dev, generator=gen)
    segs = torch.randn((R, S), dtype=torch.float64, device=dev, generator=gen)
    models /= models.norm(dim=0, keepdim=True); segs /= segs.norm(dim=0, keepdim=True)
    scores = torch.empty((M, S), dtype=torch.float64, device=dev)
    Q = torch.randn((R, R), dtype=torch.float64, device=dev, generator=gen)
    Mah = (Q @ Q.T / R + torch.eye(R, dtype=torch.float64, device=dev)).contiguous()
    Gm = ((Q + Q.T) / R).contiguous(); Hm = ((Q @ Q.T) / (R * R)).contiguous()
    Fp = torch.randn((R, rf), dtype=torch.float64, device=dev, generator=gen) / np.sqrt(R)
    FTJF = (Fp.T @ Fp + 0.1 * torch.eye(rf, dtype=torch.float64, device=dev)).contiguous()
    nsess = np.sort(np.random.default_rng(3).integers(1, 4, M)).astype(np.int64)
    mp = torch.randn((rf, M), dtype=torch.float64, device=dev, generator=gen) * torch.from_numpy(nsess).to(dev)
    sp = torch.randn((rf, S), dtype=torch.float64, device=dev, generator=gen)
    rng = np.random.default_rng(7)
    rows = np.unique(np.concatenate([[0, 127, 128, M - 129, M - 1], rng.integers(0, M, 11)]))
    cols = np.unique(np.concatenate([[0, 63, 128, S - 130, S - 1], rng.integers(0, S, 11)]))
    ri = torch.from_numpy(rows).to(dev); ci = torch.from_numpy(cols).to(dev)
    sub = lambda t2, idx: np.ascontiguousarray(t2[:, idx].cpu().numpy())
    relerr = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))
    orc = None
    if check:
        from oracle import oracle as orc          # the CHECKER of the sampled trials; nothing that is timed touches it
    rules = {
        "cosine": (lambda: ctx.score_cosine(models, segs, out=scores), lambda: orc.score_cosine(sub(models, ri), sub(segs, ci)), R),
        "mahalanobis": (lambda: ctx.score_mahalanobis(models, segs, Mah, out=scores),
                        lambda: orc.score_mahalanobis(sub(models, ri), sub(segs, ci), Mah.cpu().numpy()), R),
        "twocov": (lambda: ctx.score_twocov(models, segs, Gm, Hm, out=scores),
                   lambda: orc.score_twocov(sub(models, ri), sub(segs, ci), Gm.cpu().numpy(), Hm.cpu().numpy()), R),
        "plda(rankF=%d)" % rf: (lambda: ctx.score_plda(mp, nsess, sp, FTJF, out=scores),
                                lambda: orc.score_plda(sub(mp, ri), nsess[rows], sub(sp, ci), FTJF.cpu().numpy()), rf),
    }
    res = {}
    worst = 0.0
    for name, (run, ref, k) in rules.items():
        run(); torch.cuda.synchronize()                       # warm-up: workspace, the PLDA normalisers of the three session counts
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        dt = float(np.mean(ts))
        e = {"ms": dt * 1e3, "timed_runs_ms": [t * 1e3 for t in ts], "Gtrials_per_s": M * S / dt / 1e9, "tflops": 2.0 * k * M * S / dt / 1e12,
             "frac_of_fp64_peak": 2.0 * k * M * S / dt / 1e12 / PEAK_F64_TFLOPS, "score_write_TBps": M * S * 8 / dt / 1e12,
             "k_dgemm_ms": ctx.kernel_ms("k_dgemm(score)") if ctx.kernel_launches("k_dgemm(score)") > 0 else None,
             "checksum": float(scores.sum().item()), "finite": bool(torch.isfinite(scores).all().item())}
        if check:
            err = relerr(scores[ri][:, ci].cpu().numpy(), ref())
            e["max_rel_err_vs_oracle"] = err; e["ok"] = bool(err < 1e-9)
            worst = max(worst, err)
        res[name] = e
    head = res["mahalanobis"]
    tr = tr_note = None
    if traffic and traffic[0] and traffic[0].get("scoring"):
        tr = traffic[0]["scoring"].get("mahalanobis_hbm_bytes_per_call"); tr_note = traffic[0]["scoring"].get("source")
    elif traffic:
        tr_note = traffic[1] or "no scoring entry in profiles/traffic.json"
    out = {"metric": "trials/s (IvTest scoring, %d enrol x %d test i-vectors of dimension %d, score matrix resident in HBM)" % (M, S, R),
           "config": "BASELINE.json configs[4] at its stated size", "value": head["Gtrials_per_s"] * 1e9, "unit": "trials/s",
           "value_rule": "mahalanobis (PldaTools.cpp:3882-3909)", "models": M, "segments": S, "dim": R,
           "scores_resident_gb": M * S * 8 / 1e9, "scores_location": "HBM (never copied to the host; checksum + sampled trials leave the device)",
           "rules": res,
           "roofline": {"bound": "mfma", "kernel": "k_dgemm (M^T (Q S) with the per-vector quadratic terms in the epilogue)", "achieved": head["tflops"],
                        "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s", "frac": head["tflops"] / PEAK_F64_TFLOPS, "traffic": tr, "traffic_source": tr_note,
                        "algorithmic_flop_per_trial": SCORE_FLOP, "algorithmic_bytes": float(M) * S * 8 + (M + S) * R * 8.0,
                        "hbm_write_TBps": head["score_write_TBps"], "kernel_ms": head["ms"]}}
    if check:
        out["parity"] = {"trials_checked_per_rule": int(len(rows) * len(cols)), "max_rel_err_vs_oracle": worst, "tolerance": 1e-9, "ok": bool(worst < 1e-9),
                         "what": "sampled trials (tile corners, last rows / columns, random) of all four rules at full size vs the oracle's per-pair loops "
                                 "(restatement, parity unpinned: IvTest ships no test vector)"}
    del scores, models, segs, mp, sp
    torch.cuda.empty_cache()
    return out


def scoring_cpu_baseline(R=400):
    """the reference's O(dim^2)-per-pair form of mahalanobisDistance (PldaTools.cpp:3897-3901 -- the
    oracle restatement, -O3 -ffast-math) on one thread and, rows split, on a quarter of the cores; and the GEMM formulation
    -1/2 (m - s)^T Q (m - s) = m^T Q s - 1/2 m^T Q m - 1/2 s^T Q s on a 2 k x 2 k subset (numpy / the host BLAS), checked against the
    per-pair form.  The reference itself runs this loop on ONE thread."""
    from oracle import oracle as orc
    logical, phys = os.cpu_count() or 1, physical_cores()
    rng = np.random.default_rng(21)
    Q = rng.normal(size=(R, R)); Mah = Q @ Q.T / R + np.eye(R)
    sweep = []
    rate = None
    for th in sorted({1, max(1, phys // 4)}):
        n = 96 if rate is None else int(min(2000, max(96, np.sqrt(rate * th * 2.5))))     # ~2.5 s per point, sized from the single-thread rate
        m = rng.normal(size=(R, n)); sg = rng.normal(size=(R, n))
        t = time.time(); sc = orc.score_mahalanobis_mt(m, sg, Mah, threads=th); dt = time.time() - t
        if rate is None:
            rate = n * n / dt
        sweep.append({"threads": th, "models": n, "segments": n, "seconds": dt, "trials_per_s": n * n / dt})
    n = 2000
    m = rng.normal(size=(R, n)); sg = rng.normal(size=(R, n))
    _ = (Mah @ sg[:, :256]).sum()      # the host BLAS starts its thread pool on its first product: not part of the timed one
    t = time.time()
    QS = Mah @ sg
    gemm = m.T @ QS - 0.5 * np.einsum("km,km->m", m, Mah @ m)[:, None] - 0.5 * np.einsum("ks,ks->s", sg, QS)[None, :]
    dtg = time.time() - t
    chk = orc.score_mahalanobis(m[:, :8], sg[:, :8], Mah)
    err = float(np.max(np.abs(gemm[:8, :8] - chk)) / np.max(np.abs(chk)))
    best = max(sweep, key=lambda r: r["trials_per_s"])
    return {"value": best["trials_per_s"], "unit": "trials/s", "cores": best["threads"], "kind": "port", "single_thread": sweep[0]["trials_per_s"],
            "logical_cores": logical, "physical_cores": phys, "sweep": sweep,
            "gemm_form": {"trials_per_s": n * n / dtg, "models": n, "segments": n, "seconds": dtg, "max_rel_err_vs_per_pair_form": err,
                          "what": "numpy (host BLAS, its own thread pool) on a 2 k x 2 k subset"},
            "sample": "mahalanobisDistance in the reference's per-pair form (an O(dim^2) product per trial; oracle restatement, gcc -O3 -ffast-math, "
                      "not the original binary) on %s trials at dim %d; the reference runs it on one thread, the threaded point splits the model "
                      "rows" % (" / ".join("%dx%d" % (r["models"], r["segments"]) for r in sweep), R)}


def host_layer(ctx, g, w, mean, iv, x, dev, T, headline_gpairs, secondary, computetest, check=True):
    """The path through the boundary the reference's tools would call: the C++ host layer (host/liatools_gpu.cpp, the mirror of
    LIA_SpkTools' driver functions over the C ABI), driver-timed on the same workloads as the torch-driven numbers above --
    liagpu::trainModelStream (TrainTools.cpp:1030-1110) on the headline's frames at baggedFrameProbability 1.0 and 0.4, IvExtractor
    (IvExtractor.cpp:70-148) on the secondary's 512-utterance slice, computeTestLLR (ComputeTest.cpp:129-215) on the ComputeTest
    slice -- each with its time, the ratio to the torch-driven number and parity against it."""
    from lia_ral_amd import capi
    from lia_ral_amd import host_capi as h
    out = {"what": "liagpu::* (libliatools_gpu.so) over the C ABI; features handed over as HOST arrays and uploaded once per tool run "
                   "(untimed), iterations timed inside the library on the resident buffers"}
    cov = 1.0 / iv
    xh = x.cpu().numpy()
    SEG = 3000
    begin = np.arange(0, T, SEG, dtype=np.int64); length = np.minimum(SEG, T - begin)
    FL, CE, NIT = 1e-3, 10.0, 4
    tw = {}
    relerr = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))
    for p in (1.0, 0.4):
        r = h.train_world(xh, begin, length, w, mean, cov, NIT, bagged_p=p, init_floor=FL, final_floor=FL, init_ceil=CE, final_ceil=CE)
        it = r["it_ms"]
        ms = float(np.median(it[1:]))                       # the first iteration allocates the workspace
        sel = T * p
        e = {"bagged_p": p, "frames": T, "iterations": NIT, "ms_per_iteration": ms, "ms_iterations": it.tolist(),
             "gpairs_per_s_selected_frames": sel * C / ms / 1e6, "ratio_to_headline": sel * C / ms / 1e6 / headline_gpairs,
             "mean_llk": r["llk"].tolist()}
        if check:
            # the same iterations driven from Python through the C ABI on the resident tensor (what the headline times): frames
            # selected by the checker's restatement of baggedSegments, gathered by torch
            from oracle import oracle as orc
            mom = ctx.frame_moments(x)
            gcov = mom[D:2 * D] / mom[2 * D] - (mom[:D] / mom[2 * D]) ** 2
            gcov_d = torch.from_numpy(gcov).to(dev)
            m_d = torch.from_numpy(mean).to(dev); c_d = torch.from_numpy(cov).to(dev); w_d = torch.from_numpy(w).to(dev)
            g2 = ctx.gmm(w, mean, iv)
            acc = torch.zeros(g2.em_acc_len(), dtype=torch.float64, device=dev)
            llks = []
            for k in range(NIT):
                if p == 1.0:
                    xs = x
                else:
                    bb, bl, _ = orc.bagged_segments(((k + 1) * 200) + 20 + 1, begin, length, p, 3, 7)
                    rep = torch.from_numpy(bl).to(dev)
                    starts = torch.from_numpy(bb).to(dev)
                    off = torch.cumsum(rep, 0) - rep
                    idx = torch.repeat_interleave(starts - off, rep) + torch.arange(int(bl.sum()), device=dev)
                    xs = x.index_select(0, idx)
                acc.zero_()
                g2.em_accumulate(xs, acc=acc)
                a = acc[-2:].cpu().numpy()
                llks.append(a[0] / a[1])
                nm = torch.empty_like(m_d); nc = torch.empty_like(c_d)
                capi._chk(capi.lib.gmmiv_em_get(ctx._h, C, D, capi._ptr(acc), capi._ptr(m_d), capi._ptr(c_d), capi._ptr(w_d), capi._ptr(nm), capi._ptr(nc)))
                ctx.variance_control(nc, FL, CE, gcov_d, C, D, count=False)
                g2.set_cov(w_d, nm, nc)
                m_d, c_d = nm, nc
            torch.cuda.synchronize()
            errs = {"w": relerr(r["w"], w_d.cpu().numpy()), "mean": relerr(r["mean"], m_d.cpu().numpy()), "cov": relerr(r["cov"], c_d.cpu().numpy()),
                    "mean_llk": float(np.max(np.abs(r["llk"] - np.array(llks))))}
            e["parity"] = {"max_rel_err_model": max(errs["w"], errs["mean"], errs["cov"]), "max_abs_err_mean_llk": errs["mean_llk"], "tolerance": 1e-9,
                           "ok": bool(max(errs["w"], errs["mean"], errs["cov"]) < 1e-9 and errs["mean_llk"] < 1e-9), "per_output": errs,
                           "what": "model after %d iterations and per-iteration mean llk: liagpu::trainModelStream vs the same loop driven from "
                                   "Python through the C ABI (bagging by the checker's baggedSegments restatement, frames gathered by torch)" % NIT}
            g2.close()
        tw["p%.1f" % p] = e
    out["train_world"] = tw
    del xh
    if secondary is not None:
        U, frames, R = secondary["_slice"], 3000, 400       # the first `_slice` utterances of the secondary (features cross PCIe as host arrays here)
        xs = secondary["_x_slice"]
        Tm = secondary["_Tm"]
        ub = np.arange(U + 1, dtype=np.int64) * frames
        Wh, ms = h.iv_extract(xs.cpu().numpy(), ub, (w, mean, cov), Tm.cpu().numpy(), reps=4)
        per = ms[1:, [0, 1, 3]].sum(1)                      # statistics + substractM + estimateW of the runs after the first
        rate = U / (float(np.mean(per)) * 1e-3)
        ref_rate = secondary["slice_ab"]["default"]["i-vectors/s"]
        e = {"utterances": U, "ms_per_run": float(np.mean(per)), "stages_ms": {"statistics": float(ms[1:, 0].mean()), "substractM": float(ms[1:, 1].mean()),
             "estimateTETt_once": float(ms[0, 2]), "estimateW": float(ms[1:, 3].mean())}, "ivectors_per_s": rate,
             "ratio_to_secondary_same_slice": rate / ref_rate}
        Wt = secondary["_W"].cpu().numpy()
        err = relerr(Wh, Wt)
        e["parity"] = {"max_rel_err_vs_torch_driven": err, "tolerance": 1e-9, "ok": bool(err < 1e-9),
                       "what": "i-vectors of the first %d utterances: liagpu::TVAcc (IvExtractor order) vs the torch-driven C-ABI calls of `secondary`" % U}
        out["iv_extractor"] = e
        del xs
    if computetest is not None:
        Tc, ncl = computetest["frames_per_gpu"], computetest["clients"]
        xs = synth_frames(w, mean, iv, Tc, dev, seed=computetest["_seed"])
        cl = [(w, m, cov) for m in computetest["_client_means"]]
        llr, ms = h.compute_test(xs.cpu().numpy(), [0], [Tc], (w, mean, cov), cl, top_c=10, complete=True, reps=4)
        t = float(np.mean(ms[1:]))
        ref_ms = computetest["world_pass_ms"] + computetest["client_pass_ms"]
        err = float(np.max(np.abs(llr[0] - computetest["_llr"])))
        out["compute_test"] = {"frames": Tc, "clients": ncl, "ms_per_file": t, "ratio_to_torch_driven": ref_ms / t,
                               "torch_driven_ms": ref_ms, "llr": llr[0].tolist(),
                               "note": "computeTestLLR keeps the per-frame log-likelihoods of world and clients on the device and reads back the segment means "
                                       "(gmmiv_segment_means); its buffers live in the server's grow-only workspace",
                               "parity": {"max_abs_err_llr_vs_torch_driven": err, "tolerance": 1e-9, "ok": bool(err < 1e-9)}}
        del xs
    return out


def ivector_parity(x, frames, w, mean, iv, Tm, W, rows):
    """The CHECKER leg of the secondary (untimed): the i-vectors of a few utterances recomputed end to end by the CPU oracle
    (Baum-Welch statistics, substractM, estimateTETt, estimateW) from the same frames; north_star tolerance 1e-6 relative."""
    from oracle import oracle as orc
    og = orc.Gmm(w, mean, iv)
    Tm_h = Tm.cpu().numpy()
    te = orc.tv_tett(Tm_h, iv.ravel(), C, D)
    worst = 0.0
    for u in rows:
        xu = x[u * frames:(u + 1) * frames].cpu().numpy().astype(np.float64)
        No, Fo = orc.tv_stats(og, xu, np.zeros(frames, np.int64), 1)
        Wo = orc.tv_estimate_w(No, orc.tv_subtract_m(No, Fo, mean.ravel()), Tm_h, iv.ravel(), te)[0]
        worst = max(worst, float(np.max(np.abs(W[u].cpu().numpy() - Wo)) / np.max(np.abs(Wo))))
    return {"utterances_checked": rows, "max_rel_err_vs_oracle": worst, "tolerance": 1e-6, "ok": worst < 1e-6}


def make_collectives(ctx, dev, world, rank, want, transport=None):
    """The product's collectives (gmmiv_comm_* = RCCL inside libgmmiv).  Every rank must end up with the SAME back end, so
    the outcome of the communicator set-up is agreed through the launcher's process group; if it failed anywhere, EVERY rank exits
    non-zero -- a line whose data path silently went through another transport would not measure the product.  torch.distributed's
    RCCL binding is used only when asked for with --collectives torch (and the JSON line says so)."""
    from lia_ral_amd import capi
    from lia_ral_amd import dist as gd
    if world == 1:
        return gd.GmmivCollectives(capi.Comm(ctx, 1, 0)), None
    import threading
    import torch.distributed as dist
    if want == "torch":
        return gd.TorchCollectives(), "requested with --collectives torch"
    box = {}

    def create():
        try:
            box["coll"] = gd.gmmiv_collectives_from_torch(ctx, dev, transport)
        except Exception as e:      # noqa: BLE001 - reported, never silent
            box["err"] = repr(e)
    th = threading.Thread(target=create, daemon=True)
    th.start()
    th.join(timeout=180.0)
    ok = torch.tensor([1 if "coll" in box else 0], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 1:
        coll = box["coll"]
        probe = torch.full((4096,), float(rank + 1), dtype=torch.float64, device=dev)
        coll.allreduce(probe)
        torch.cuda.synchronize()
        if bool((probe == world * (world + 1) / 2.0).all().item()):
            coll.take_bytes()
            return coll, None
        box["err"] = "probe all-reduce returned a wrong sum"
    print("bench.py: rank %d: the C ABI's communicator (gmmiv_comm) could not be set up on every rank (%s); refusing to run the data path on "
          "another transport -- pass --collectives torch to measure with torch.distributed's RCCL binding instead"
          % (rank, box.get("err", "timeout or failure on another rank")), file=sys.stderr, flush=True)
    sys.exit(3)


class GpuTvOps:
    """The per-rank compute of one TotalVariability iteration on device-resident statistics (lia_ral_amd.dist.tv_em_iteration)."""

    def __init__(self, ctx, N, F, Tm, invvar, means, R, F_raw=None, world=1):
        self.ctx, self.N, self.F, self.T, self.invvar, self.means, self.R = ctx, N, F, Tm, invvar, means, R
        self.F_raw = F_raw          # the uncentred first-order statistics (TVAcc::storeStats), or None: F is centred once and kept
        dev = N.device
        P = R * (R + 1) // 2
        z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=dev)
        self.tett_buf = torch.empty((C, P), dtype=torch.float64, device=dev)
        Cpad = (C + world - 1) // world * world       # A lives in the reduce-scatter's send buffer (equal Gaussian blocks, zero padded)
        A_pad = z(Cpad, P)
        self.acc = dict(A=A_pad[:C], A_pad=A_pad, Cmx=z(R, C * D), Rm=z(R, R), r=z(R), meanW=z(R),
                        W=torch.empty((N.shape[0], R), dtype=torch.float64, device=dev))

    def stream_context(self):
        return torch.cuda.stream(self.ctx.torch_stream())

    def recentre(self):
        """TotalVariability.cpp:123-124: the statistics are reloaded and substractM runs at the top of EVERY iteration --
        minDivergence has moved the UBM means since the last one."""
        if self.F_raw is not None:                                      # TVAcc::restoreStats + substractM with the current means, one pass
            self.ctx.tv_subtract_m_to(self.N, self.F_raw, self.F, self.means, C, D)

    def tett(self):
        self.ctx.tv_tett(self.T, self.invvar, C, D, out=self.tett_buf)

    def estep(self):
        for k in ("A", "Cmx", "Rm", "r", "meanW"):      # TVAcc::resetTmpAcc
            self.acc[k].zero_()
        self.ctx.tv_estimate_a_and_c(self.N, self.F, self.T, self.invvar, self.tett_buf, C, D, acc=self.acc)
        return self.acc

    def update_t(self, A_blk, C_blk, cb):
        return self.ctx.tv_update_t(A_blk, C_blk, cb, D, out=torch.empty((self.R, cb * D), dtype=torch.float64, device=A_blk.device))

    def min_divergence(self, acc, Tn, n):
        self.ctx.tv_min_divergence(acc["Rm"], acc["r"], acc["meanW"] / n, self.means, Tn, n, C, D)
        self.T = Tn
        return Tn


TV_FLOP_PER_UTT = 2.0 * C * (400 * 401 // 2) * 2 + 2.0 * C * D * 400 * 2 + 21.7e6    # L + A (packed), aux + Cmx, solve: 875 M


def tv_parity_and_cpu_baseline(ops, R, U_chk=4, want_cpu=True):
    """The CHECKER leg of the T-matrix workload (untimed, rank 0): estimateAandC of this rank's first U_chk utterances under the
    CURRENT T at full shape, HIP (a fresh call on those rows) against the oracle's scalar loop (AccumulateTVStat.cpp:1702-1795
    restated) on the same rows; every Gaussian of A, all of Cmx, W, R, r.  The oracle's run time on those utterances is the
    cpu_baseline of this workload (1 thread, TETt precomputed)."""
    from oracle import oracle as orc
    ctx = ops.ctx
    ops.tett()          # TETt of the CURRENT T (the buffer still holds the one the last iteration started from)
    Nd, Fd = ops.N[:U_chk].contiguous(), ops.F[:U_chk].contiguous()
    got = ctx.tv_estimate_a_and_c(Nd, Fd, ops.T, ops.invvar, ops.tett_buf, C, D, acc=dict(
        A=torch.zeros_like(ops.acc["A"]), Cmx=torch.zeros_like(ops.acc["Cmx"]), Rm=torch.zeros_like(ops.acc["Rm"]),
        r=torch.zeros_like(ops.acc["r"]), meanW=torch.zeros_like(ops.acc["meanW"]),
        W=torch.empty((U_chk, R), dtype=torch.float64, device=Nd.device)))
    torch.cuda.synchronize()
    Th, ivh = ops.T.cpu().numpy(), ops.invvar.cpu().numpy()
    te = orc.tv_tett(Th, ivh, C, D)
    t = time.time()
    ref = orc.tv_estimate_a_and_c(Nd.cpu().numpy(), Fd.cpu().numpy(), Th, ivh, te)
    dt = time.time() - t
    rel = lambda a, b: float(np.max(np.abs(a.cpu().numpy() - b)) / max(np.max(np.abs(b)), 1e-300))
    il = np.tril_indices(R)      # the oracle keeps full R x R blocks like the reference, libgmmiv the packed lower triangle

    def rel_packed(packed, full):
        full = full.reshape(C, R, R)
        num = den = 0.0
        for c0 in range(0, C, 256):      # blockwise: the gathered triangle of all 2048 blocks would be a 1.3 GB temporary
            blk = full[c0:c0 + 256][:, il[0], il[1]]
            num = max(num, float(np.max(np.abs(packed[c0:c0 + 256].cpu().numpy() - blk))))
            den = max(den, float(np.max(np.abs(blk))))
        return num / max(den, 1e-300)
    errs = {"TETt": rel_packed(ops.tett_buf, te), "A": rel_packed(got["A"], ref["A"]), "Cmx": rel(got["Cmx"], ref["Cmx"]),
            "W": rel(got["W"], ref["W"]), "R": rel(got["Rm"], ref["Rm"]), "r": rel(got["r"], ref["r"]),
            "meanW": rel(got["meanW"] / U_chk, ref["meanW"])}
    worst = max(errs.values())
    parity = {"max_rel_err": worst, "tolerance": 1e-9, "ok": bool(worst < 1e-9), "per_output": errs,
              "what": "estimateTETt + estimateAandC of %d utterances at C=%d, R=%d under the T of the last timed iteration: libgmmiv vs the "
                      "oracle's scalar loops (restatement, parity unpinned: no reference vector exists for this path)" % (U_chk, C, R)}
    cpu = {"value": U_chk / dt, "unit": "utterances/s (E-step only)", "cores": 1, "kind": "port",
           "sample": "estimateAandC ALONE on %d utterances at C=2048, R=%d, the checker's strict scalar fp64 oracle (-O2), %.1f s; TETt precomputed; "
                     "the whole-iteration threaded -O3 -ffast-math figure is cpu_baseline" % (U_chk, R, dt)}
    return parity, (cpu if want_cpu else None)


def tv_iteration_cpu_baseline(ops, R, U_cpu=64, threads=32):
    """ONE WHOLE T-matrix EM iteration (estimateTETt, estimateAandC, updateTestimate, minDivergence) of
    the first U_cpu utterances of this rank, on the host cores with the reference's thread partition (oracle/oracle_mt.c
    orc_tv_em_iteration_mt, gcc -O3 -ffast-math: AccumulateTVStat.cpp:826-950, :1831-2052 -- utterance ranges per thread, private
    R / r / meanW, A and C shared under two mutexes), and the SAME iteration on the same utterances through libgmmiv; the relative
    error of T and of the UBM means after it is the parity of the whole iteration.  Untimed for the GPU; the CPU time is the baseline."""
    from lia_ral_amd import dist as gd
    from oracle import oracle as orc
    U_cpu = int(min(U_cpu, ops.N.shape[0]))
    threads = int(max(1, min(threads, os.cpu_count() or 1, U_cpu)))
    T0, m0 = ops.T.clone(), ops.means.clone()
    sub = GpuTvOps(ops.ctx, ops.N[:U_cpu].contiguous(), torch.empty_like(ops.F_raw[:U_cpu]), T0.clone(), ops.invvar, m0.clone(), R,
                   F_raw=ops.F_raw[:U_cpu].contiguous())
    gd.tv_em_iteration(sub, U_cpu, C, D)                       # one rank, no exchange: recentre, tett, estep, update_t, min_divergence
    torch.cuda.synchronize()
    T_gpu, m_gpu = sub.T.cpu().numpy(), sub.means.cpu().numpy()
    Nh = sub.N.cpu().numpy()
    Fh = orc.tv_subtract_m(Nh, sub.F_raw.cpu().numpy(), m0.cpu().numpy())       # substractM under the means the iteration starts from
    del sub
    torch.cuda.empty_cache()
    t = time.time()
    ref = orc.tv_em_iteration_mt(Nh, Fh, T0.cpu().numpy(), ops.invvar.cpu().numpy(), m0.cpu().numpy(), threads=threads, upd_threads=threads)
    dt = time.time() - t
    relT = float(np.max(np.abs(T_gpu - ref["T"])) / np.max(np.abs(ref["T"])))
    relm = float(np.max(np.abs(m_gpu - ref["means"])) / np.max(np.abs(ref["means"])))
    ph = [float(v) for v in ref["phase_s"]]
    cpu = {"value": U_cpu / dt, "unit": "utterances/s", "cores": threads, "kind": "port", "seconds": dt, "utterances": U_cpu,
           "phases_s": {"estimateTETt": ph[0], "estimateAandC": ph[1], "updateTestimate": ph[2], "minDivergence": ph[3]},
           "per_iteration_fixed_s": ph[0] + ph[2] + ph[3],
           "sample": "ONE whole T-matrix EM iteration on %d utterances at C=%d, R=%d: estimateTETt + estimateAandC on %d threads with the "
                     "reference's partition (utterance ranges, private R / r / meanW, A and C under two mutexes), then updateTestimate and "
                     "minDivergence -- single-threaded in the reference, here ALSO on %d threads (Gaussian / column ranges: generous to the "
                     "CPU); gcc -O3 -ffast-math; a restatement of the reference loops, not the original binary.  TETt, updateTestimate and "
                     "minDivergence cost the same whatever the utterance count (per_iteration_fixed_s): utterances/s of this sample is NOT "
                     "the rate of a 6250-utterance iteration" % (U_cpu, C, R, threads, threads)}
    parity = {"max_rel_err_T": relT, "max_rel_err_means": relm, "tolerance": 1e-6, "ok": bool(relT < 1e-6 and relm < 1e-6),
              "what": "T and the UBM means after ONE whole iteration on %d utterances from the T of the last timed step: libgmmiv vs the threaded "
                      "-ffast-math oracle (restatement; parity unpinned -- no reference vector exists for this path)" % U_cpu}
    return parity, cpu


def em_parity(ctx, g, w, mean, iv, x, acc_last, frames_per_rank, world, nframes=2000):
    """The CHECKER leg of the headline (untimed): (1) the EM statistics of the first `nframes` frames of the timed block under
    the seed model, libgmmiv against the oracle (MixtureStat::computeAndAccumulateEM restated, pinned by KAT-2); (2) a
    size-independent property of the LAST TIMED step's all-reduced accumulator at full size: the posteriors of a frame sum to
    one, so sum_c occ_c = the frame count over all ranks = acc[-1]."""
    from oracle import oracle as orc
    g.set(w, mean, iv)
    xs = x[:nframes]
    acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device=x.device)
    prev = ctx.set_option("short_calls", 0)   # the checked call runs the kernel shapes of the timed (long) calls, not the short-call ones
    g.em_accumulate(xs, acc=acc)
    ctx.set_option("short_calls", prev)
    torch.cuda.synchronize()
    a = g.split_acc(acc.cpu().numpy())
    ref = orc.em_accumulate(orc.Gmm(w, mean, iv), xs.cpu().numpy().astype(np.float64))
    rel = lambda p, q: float(np.max(np.abs(p - q)) / np.max(np.abs(q)))
    errs = {"occ": rel(a["occ"], ref["occ"]), "sum_gamma_x": rel(a["sx"], ref["sx"]), "sum_gamma_x2": rel(a["sxx"], ref["sxx"]),
            "sum_log_lk": float(abs(a["llk"] - ref["llk"]) / abs(ref["llk"]))}
    last = acc_last.cpu().numpy()
    total = float(frames_per_rank) * world
    occ_sum_err = float(abs(last[:C].sum() - total) / total)
    count_ok = bool(last[-1] == total)
    worst = max(errs.values())
    return {"max_rel_err": worst, "tolerance": 1e-9, "ok": bool(worst < 1e-9 and occ_sum_err < 1e-9 and count_ok), "per_output": errs,
            "timed_step_invariant": {"sum_occ_vs_frames_rel_err": occ_sum_err, "frame_count_exact": count_ok, "frames_all_ranks": total},
            "what": "EM statistics of the first %d frames of the timed block under the seed model: libgmmiv vs the oracle "
                    "(computeAndAccumulateEM restated, pinned by KAT-2); plus sum_c occ_c == frames on the last timed step's accumulator" % nframes}


def max_over_ranks(dt, world, dev):
    """The slowest rank's time (the contract's MAX over ranks); on the gloo launcher group the value travels as a host tensor."""
    if world == 1:
        return dt
    import torch.distributed as dist
    tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def gather_over_ranks(obj, world):
    """every rank's value, in rank order (the launcher's process group carries it as a Python object)"""
    if world == 1:
        return [obj]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def tensor_digest(t):
    """SHA-256 of a device tensor's bytes: replicated results must be the SAME BITS on every rank after a collective"""
    import hashlib
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def multi_rank_report(world, rank, dt_rank, steps, digests, event_ms=None):
    """What the first run on real multi-GPU hardware should show beside the headline: every rank's own
    ms_per_step (the line's ms_per_step is their MAX), the collectives' times from HIP events on the stream they are enqueued on, and
    whether the replicated results behind a collective are bitwise equal on all ranks (digests: name -> sha256 on this rank)."""
    per_rank = gather_over_ranks(dt_rank / max(steps, 1) * 1e3, world)
    all_dig = gather_over_ranks(digests, world)
    rep = {"ms_per_step_per_rank": per_rank,
           "bitwise_equal_across_ranks": {k: all(d.get(k) == all_dig[0].get(k) for d in all_dig) for k in all_dig[0]}}
    if event_ms is not None:
        ev = gather_over_ranks(event_ms, world)
        rep["collective_ms_per_step_per_rank"] = ev
    bad = [k for k, v in rep["bitwise_equal_across_ranks"].items() if not v]
    if bad:
        print("bench.py: rank %d: replicated results differ between ranks after the collective: %s" % (rank, bad), file=sys.stderr, flush=True)
        sys.exit(4)
    return rep


def tv_workload(ctx, g, coll, w, mean, iv, dev, rank, world, U, frames, R, steps, warmup, check=True, cpu=True, overlap=False, force=False, traffic=None,
                cpu_utterances=64):
    """BASELINE.json configs[3]: N / F of this rank's U utterances computed once (untimed, like TotalVariability loads them),
    then `steps` EM iterations timed, each the tool's full sequence: restore + substractM, estimateTETt, estimateAandC,
    updateTestimate (sharded), minDivergence.  Returns the JSON fields of the workload."""
    from lia_ral_amd import dist as gd
    import torch.distributed as dist
    invvar = torch.from_numpy(iv.ravel().copy()).to(dev)
    means = torch.from_numpy(mean.ravel().copy()).to(dev)
    N = torch.empty((U, C), dtype=torch.float64, device=dev)
    F_raw = torch.empty((U, C * D), dtype=torch.float64, device=dev)
    t_stats = time.perf_counter()
    CH = 1250
    for u0 in range(0, U, CH):
        n = min(CH, U - u0)
        x = synth_frames(w, mean, iv, n * frames, dev, seed=9000 + 131 * rank + u0)
        screen_once(ctx, x, "T-matrix workload frames")
        g.tv_stats(x, np.arange(n + 1, dtype=np.int64) * frames, N[u0:u0 + n], F_raw[u0:u0 + n])
        del x
    torch.cuda.synchronize()
    t_stats = time.perf_counter() - t_stats
    F = torch.empty_like(F_raw)                               # the centred working copy, rebuilt by every iteration's recentre
    gen = torch.Generator(device=dev); gen.manual_seed(5)     # the same initial T on every rank
    Tm = 0.01 * torch.randn((R, C * D), dtype=torch.float64, device=dev, generator=gen)
    ops = GpuTvOps(ctx, N, F, Tm, invvar, means, R, F_raw=F_raw, world=world)
    n_total = U * world

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        gd.tv_em_iteration(ops, n_total, C, D, rank, world, coll, overlap=overlap, force_collectives=force)
    coll.take_bytes()
    phases = {"sync": torch.cuda.synchronize}
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        gd.tv_em_iteration(ops, n_total, C, D, rank, world, coll, phases, overlap=overlap, force_collectives=force)
    barrier()
    dt_rank = time.perf_counter() - t0
    dt = max_over_ranks(dt_rank, world, dev)
    nbytes = coll.take_bytes() / max(steps, 1)
    ph = {k: v / steps * 1e3 for k, v in phases.items() if k != "sync"}
    multi = None
    if world > 1:      # T and the UBM means are replicated: the all-gather of T and the all-reduce of R / r / meanW must leave the same bits everywhere
        multi = multi_rank_report(world, rank, dt_rank, steps, {"T_after_iteration": tensor_digest(ops.T), "ubm_means_after_min_divergence": tensor_digest(ops.means)},
                                  {k: ph.get(k) for k in ("reduce_scatter", "allgather", "min_divergence", "estep")})
    estep_tf = TV_FLOP_PER_UTT * U / (ph["estep"] * 1e-3) / 1e12
    finite = bool(torch.isfinite(ops.T).all().item())
    tr = tr_note = None
    if traffic and traffic[0] and traffic[0].get("tv_em"):
        tr = traffic[0]["tv_em"]["estep_hbm_bytes_per_utterance"] * U; tr_note = traffic[0]["tv_em"].get("source")
    elif traffic:
        tr_note = traffic[1] or "no tv_em entry in profiles/traffic.json"
    res = {
        "metric": "utterances/s (TotalVariability: one T-matrix EM iteration, 2048-g UBM, rank %d)" % R,
        "value": n_total * steps / dt, "unit": "utterances/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "TotalVariability T-matrix EM: 2048-g UBM, rank %d, %d utterances x %d frames per GPU (50 k over 8 GPUs), "
                               "N/F resident in HBM, 1 EM iteration per step (restore + substractM, estimateTETt, estimateAandC, "
                               "updateTestimate, minDivergence)" % (R, U, frames),
                   "gaussians": C, "dim": D, "rank": R, "utterances_per_gpu": U,
                   "partitioning": "utterances sharded per rank; reduce-scatter of A_packed / Cmx by Gaussian blocks, sharded "
                                   "updateTestimate, all-gather of T, all-reduce of R / r / meanW"},
        "phases_ms": ph, "collective_bytes_per_step_per_rank": nbytes, "collectives": coll.name,
        "overlap": bool(overlap and (world > 1 or force) and getattr(coll, "supports_overlap", False)),
        "statistics_once_s": t_stats, "finite": finite,
        "roofline": {"bound": "mfma", "kernel": "E-step (k_dgemm: L, aux, A, Cmx + chol_fused)", "achieved": estep_tf, "peak": PEAK_F64_TFLOPS,
                     "unit": "TFLOP/s", "frac": estep_tf / PEAK_F64_TFLOPS, "traffic": tr, "traffic_source": tr_note,
                     "traffic_unit": "HBM bytes of the E-step of one iteration over this rank's %d utterances (all kernels)" % U,
                     "algorithmic_flop_per_utterance": TV_FLOP_PER_UTT, "kernel_ms": ph["estep"],
                     "algorithmic_bytes": float(U) * (C * D + C) * 8.0},      # the N / F rows read once
    }
    if multi:
        res["multi_rank"] = multi
    if rank == 0 and check:
        res["parity"], cpub = tv_parity_and_cpu_baseline(ops, R, want_cpu=cpu)
        if cpub:
            res["cpu_baseline_1thread_estep"] = cpub
        if cpu:
            res["parity"]["whole_iteration"], res["cpu_baseline"] = tv_iteration_cpu_baseline(ops, R, cpu_utterances, min(32, physical_cores()))
    return res


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run with N ranks on this node (the
    same command line the driver uses) and hand its exit status back."""
    import socket
    import subprocess
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=10_000_000, help="frames per GPU")
    ap.add_argument("--workload", choices=["em", "tv"], default="em", help="em: TrainWorld EM pass (headline); tv: TotalVariability T-matrix EM iteration (configs[3])")
    ap.add_argument("--tv-utterances", type=int, default=6250, help="utterances per GPU of the T-matrix EM workload")
    ap.add_argument("--tv-rank", type=int, default=400)
    ap.add_argument("--tv-cpu-utterances", type=int, default=64, help="utterances of the whole-iteration CPU baseline of the T-matrix workload")
    ap.add_argument("--iv-utterances", type=int, default=10_000, help="utterances per GPU of the IvExtractor block (configs[2]: 10 000)")
    ap.add_argument("--score-vectors", type=int, default=100_000, help="enrolment = test i-vectors of the scoring block (configs[4]: 100 000)")
    ap.add_argument("--collectives", choices=["gmmiv", "torch"], default="gmmiv")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-host-layer", action="store_true", help="skip the host_layer block (the C++ host layer timed on the same workloads)")
    ap.add_argument("--mean-spread", type=float, default=2.0,
                    help="std of the synthetic UBM means (SURVEY 8(d): 2.0; smaller = overlapping Gaussians)")
    ap.add_argument("--wg-waves", type=int, default=0, help="A/B knob: waves per workgroup of the MFMA kernels (4 or 8)")
    ap.add_argument("--overlap", type=int, default=0,
                    help="T-matrix EM: 1 = the reduce-scatter of A starts inside the E-step (under the Cmx GEMM), the all-gather of T is joined "
                         "inside minDivergence (gmmiv_*_begin / gmmiv_comm_join; bitwise the serial results)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="one rank only: build the communicator through RCCL anyway (GMMIV_COMM_FORCE_RCCL=1) and walk the sharded "
                         "exchange of the T-matrix workload with it -- every RCCL call of the step executes on this one GPU")
    ap.add_argument("--share-gpu", action="store_true",
                    help="correctness mode: let the ranks share the visible GPU(s) (rank r on device r %% count) over the C ABI's shm transport")
    args = ap.parse_args()

    ndev = torch.cuda.device_count()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the ranks ourselves (the reference tools fan their worker threads out the same way)
        if ndev < args.gpus and not args.share_gpu:
            print("bench.py: --gpus %d but %d GPU(s) visible: one rank per GPU is required (pass --share-gpu to run the ranks on "
                  "shared devices for a correctness check only)" % (args.gpus, ndev), file=sys.stderr, flush=True)
            sys.exit(2)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started %d rank(s): refusing to report a number for the wrong job size"
              % (args.gpus, world), file=sys.stderr, flush=True)
        sys.exit(2)
    if ndev < 1 or (ndev < world and not args.share_gpu):
        print("bench.py: %d rank(s) but %d GPU(s) visible (one rank per GPU; --share-gpu for a correctness-only run)" % (world, ndev),
              file=sys.stderr, flush=True)
        sys.exit(2)
    shared = world > ndev
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if shared:       # RCCL refuses two ranks on one device: the launcher group runs on gloo, the data path on gmmiv's shm transport
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI

    from conftest import make_gmm
    from lia_ral_amd import capi

    w, mean, iv = make_gmm(C, D, seed=0, spread=args.mean_spread)
    T = args.frames
    # ONE stream for everything: torch's kernels (synthetic data, re-layouts of the sharded M-step) and libgmmiv's kernels and
    # collectives are enqueued on it in program order -- no cross-stream hazards, and the HIP events of the library bracket
    # exactly its own kernels
    side = torch.cuda.Stream(dev)
    torch.cuda.set_stream(side)
    ctx = capi.Context(local, side.cuda_stream)
    ctx.set_option("timing", 1)
    if args.wg_waves:
        ctx.set_option("wg_waves", args.wg_waves)
    g = ctx.gmm(w, mean, iv)
    force = bool(args.force_collectives and world == 1)
    if force:
        os.environ["GMMIV_COMM_FORCE_RCCL"] = "1"
    coll, coll_note = make_collectives(ctx, dev, world, rank, args.collectives, "shm" if shared else None)
    if coll.world != world:
        print("bench.py: the communicator spans %d rank(s), the job %d" % (coll.world, world), file=sys.stderr, flush=True)
        sys.exit(2)
    comm_info = {"world": coll.world, "backend": getattr(coll, "backend", coll.name), "launcher_group": (dist.get_backend() if world > 1 else None),
                 "physical_gpus": min(ndev, world), "gpu_sharing": shared}
    if hasattr(coll, "comm"):       # what RCCL itself reports for the product's communicator: version code and ncclCommCount
        comm_info.update(coll.comm.info())
    if shared:
        comm_info["note"] = "ranks SHARE the GPU(s): a correctness run of the multi-rank orchestration, not a scaling measurement"
    check = not args.no_cpu_baseline
    if args.workload == "tv":
        res = tv_workload(ctx, g, coll, w, mean, iv, dev, rank, world, args.tv_utterances, 3000, args.tv_rank, args.steps, args.warmup,
                          check=check, cpu=(world == 1), overlap=bool(args.overlap), force=force, traffic=load_traffic(),
                          cpu_utterances=args.tv_cpu_utterances)
        if rank == 0:
            res["comm"] = comm_info
            if coll_note:
                res["collectives_note"] = coll_note
            print(json.dumps(res), flush=True)
        g.close()
        ctx.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    x = synth_frames(w, mean, iv, T, dev, seed=1234 + rank)
    screening = screen_once(ctx, x, "headline frames")      # a check of the data only: the timed calls keep the default "assume_finite" 0
    screening["assume_finite"] = int(ctx.set_option("assume_finite", 0))
    traffic = load_traffic()
    nacc = g.em_acc_len()
    acc = torch.zeros(nacc, dtype=torch.float64, device=dev)
    mean_d = torch.from_numpy(mean).to(dev)
    cov_d = torch.from_numpy(1.0 / iv).to(dev)
    w_d = torch.empty(C, dtype=torch.float64, device=dev)
    nm_d = torch.empty_like(mean_d)
    nc_d = torch.empty_like(cov_d)
    cov_signal = cov_d.mean(0).contiguous()       # stands in for the global covariance of computeMeanCov

    kern_ms = {}
    ar_events = []

    def step(record=False):
        nonlocal mean_d, cov_d, nm_d, nc_d
        acc.zero_()
        g.em_accumulate(x, acc=acc)                         # K1 (lse) + K2 (statistics) + reduce
        if record:
            for name in KERNEL_FLOP:
                if ctx.kernel_launches(name) > 0:
                    kern_ms.setdefault(name, []).append((ctx.kernel_ms(name), ctx.kernel_launches(name)))
        if world > 1:
            if record:                                      # HIP events on the stream the collective is enqueued on (the context's = torch's current)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); coll.allreduce(acc); e1.record()
                ar_events.append((e0, e1))
            else:
                coll.allreduce(acc)                         # EM sufficient statistics, 1.98 MB fp64 (gmmiv_allreduce_f64)
        # M-step (MixtureStat::getEM) + variance flooring + re-pack of the device model
        capi._chk(capi.lib.gmmiv_em_get(ctx._h, C, D, capi._ptr(acc), capi._ptr(mean_d), capi._ptr(cov_d),
                                        capi._ptr(w_d), capi._ptr(nm_d), capi._ptr(nc_d)))
        ctx.variance_control(nc_d, 1e-3, 10.0, cov_signal, C, D, count=False)
        g.set_cov(w_d, nm_d, nc_d)
        mean_d, nm_d = nm_d, mean_d
        cov_d, nc_d = nc_d, cov_d

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    barrier()
    dt_rank = time.perf_counter() - t0
    dt = max_over_ranks(dt_rank, world, dev)
    multi = None
    if world > 1:      # acc still holds the LAST step's all-reduced accumulator; the M-step outputs are replicated too
        ar_ms = float(np.mean([a.elapsed_time(b) for a, b in ar_events])) if ar_events else None
        multi = multi_rank_report(world, rank, dt_rank, args.steps, {"em_accumulator_after_allreduce": tensor_digest(acc), "weights_after_m_step": tensor_digest(w_d),
                                                                     "means_after_m_step": tensor_digest(mean_d)}, {"allreduce": ar_ms})
    parity = em_parity(ctx, g, w, mean, iv, x, acc, T, world) if rank == 0 and check else None   # untimed checker leg

    pairs_per_step = float(T) * C * world
    value = pairs_per_step * args.steps / dt / 1e9
    out = {
        "metric": "Gframe-Gaussian evals/s (UBM EM pass: LLK + full-posterior statistics)",
        "value": value, "unit": "Gframe-Gaussian/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "TrainWorld EM: 2048-Gaussian diag UBM, 60-dim float32 frames, %d frames per GPU "
                               "resident in HBM, 1 EM iteration per step" % T,
                   "gaussians": C, "dim": D, "frames_per_gpu": T, "partitioning": "frames sharded per rank, "
                   "one RCCL all-reduce of %d doubles per step" % nacc},
        "collectives": coll.name, "comm": comm_info,
    }
    if parity:
        out["parity"] = parity
    if multi:
        out["multi_rank"] = multi
    if coll_note:
        out["collectives_note"] = coll_note
    # the same E-step on a heavily overlapping mixture (means ~ N(0, 0.3^2)): hundreds of Gaussians carry
    # posterior mass per frame, none of the data-dependent skips of K1/K2 can fire -- the dense floor.
    dense = None
    if not args.no_secondary:
        wd, md, ivd = make_gmm(C, D, seed=3, spread=0.3)
        Td = min(T, 2_000_000)
        xd = synth_frames(wd, md, ivd, Td, dev, seed=4321 + rank)
        screen_once(ctx, xd, "dense-mixture frames")
        g.set(wd, md, ivd)
        accd = torch.zeros(nacc, dtype=torch.float64, device=dev)
        g.em_accumulate(xd, acc=accd)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            accd.zero_()
            g.em_accumulate(xd, acc=accd)
        torch.cuda.synchronize()
        dd = (time.perf_counter() - t1) / 2
        sname = "k_stats_z" if ctx.kernel_launches("k_stats_z") > 0 else "k_stats_mfma"
        dense_k = (ctx.kernel_ms("k_llk_mfma"), ctx.kernel_ms(sname))
        ctx.set_option("prune_log2", 100)      # opt-in pruning on the SURVEY-spec data (one dominant Gaussian per frame)
        g.set(w, mean, iv)
        xs = x[:Td]
        g.em_accumulate(xs, acc=accd)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            accd.zero_()
            g.em_accumulate(xs, acc=accd)
        torch.cuda.synchronize()
        dp = (time.perf_counter() - t1) / 2
        pruned = {"value": Td * C / dp / 1e9, "unit": "Gframe-Gaussian/s per GPU", "frames": Td, "prune_log2": 100,
                  "k_stats_ms": ctx.kernel_ms(sname),
                  "note": "opt-in: groups of 4 frames x 16 Gaussians with all posteriors < 2^-100 skip exp + statistics MFMAs"}
        ctx.set_option("prune_log2", 0)
        dense = {"pruned_posteriors": pruned, "value": Td * C / dd / 1e9, "unit": "Gframe-Gaussian/s per GPU", "frames": Td, "mean_spread": 0.3,
                 "k_llk_ms": dense_k[0], "k_stats_ms": dense_k[1],
                 "k_stats_tflops": KERNEL_FLOP[sname] * Td * C / (dense_k[1] * 1e-3) / 1e12}
        del xd, accd
    secondary = None
    computetest = None
    tv_em = None
    hostl = None
    scoring = None
    if not args.no_secondary:
        g.set(w, mean, iv)     # back to the seed model for the i-vector block
        # configs[2] at its stated size (10 000 utterances x 3000 frames per GPU; --iv-utterances for a smaller dry run)
        secondary = ivector_secondary(ctx, g, w, mean, iv, dev, rank, world, U=args.iv_utterances, check=check, traffic=traffic)
        computetest = computetest_secondary(ctx, g, w, mean, iv, dev, rank, world, check=check)
        if world == 1 and not args.no_host_layer:
            hostl = host_layer(ctx, g, w, mean, iv, x, dev, T, value, secondary, computetest, check=check)
        for blk in (secondary, computetest):     # tensors kept for the host-layer parity, not part of the line
            for k in [k for k in blk if k.startswith("_")]:
                del blk[k]
        torch.cuda.empty_cache()
        # configs[3]: one T-matrix EM iteration on utterance-sharded statistics -- 6250 utterances per GPU (50 k over 8 GPUs), at
        # world 1 too: the per-GPU share is what one MI355X does in the 8-GPU job
        x_keep = x
        x = xs = None              # release the 2.4 GB frame block of the EM workload (the CPU baseline below makes its own frames)
        del x_keep
        torch.cuda.empty_cache()
        tv_em = tv_workload(ctx, g, coll, w, mean, iv, dev, rank, world, args.tv_utterances, 3000, args.tv_rank, 3, 1, check=check, cpu=(world == 1),
                            overlap=bool(args.overlap), traffic=traffic, cpu_utterances=args.tv_cpu_utterances)
        torch.cuda.empty_cache()
        if world == 1:             # configs[4] is a single-GPU workload (model blocks shard without any exchange: lia_ral_amd.dist.score_model_block)
            scoring = scoring_block(ctx, dev, M=args.score_vectors, S=args.score_vectors, check=check, traffic=traffic)
        if world == 1 and check and rank == 0:
            secondary["cpu_baseline"] = ivector_cpu_baseline(w, mean, iv)
            scoring["cpu_baseline"] = scoring_cpu_baseline()
    if rank == 0:
        if secondary:
            out["secondary"] = secondary
        if computetest:
            out["computetest"] = computetest
        if hostl:
            out["host_layer"] = hostl
        if tv_em:
            out["tv_em"] = tv_em
        if scoring:
            out["scoring"] = scoring
        out["screening"] = screening
        out["library"] = traffic[2]
        if dense:
            out["dense_data"] = dense
        # per kernel: total ms per step, launches per step (frame chunks), algorithmic TFLOP/s
        kernels = {}
        for name, recs in kern_ms.items():
            ms = float(np.mean([r[0] for r in recs])); nl = int(recs[-1][1])
            kernels[name] = {"ms_per_step": ms, "launches_per_step": nl, "ms_per_launch": ms / nl,
                             "tflops": KERNEL_FLOP[name] * T * C / (ms * 1e-3) / 1e12,
                             "gpairs_per_s": T * C / (ms * 1e-3) / 1e9}
        out["kernels"] = kernels
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])     # the dominant kernel of the step
            kd = kernels[dom]
            # HBM bytes per launch come from PMC counters, which only a rocprofv3 run can read: the figure of the committed PMC passes
            # of the SAME command on the SAME library build (load_traffic), scaled to the frames of a launch of this run -- else null
            tj, traffic_note, _ = traffic
            tr = traffic_src = None
            if tj and tj.get(dom + "_hbm_bytes_per_launch") and tj.get("frames_per_launch"):
                tr = tj[dom + "_hbm_bytes_per_launch"] / tj["frames_per_launch"] * (T / kd["launches_per_step"])
                traffic_src = tj.get("source")
            else:
                traffic_src = traffic_note or "no entry for %s in profiles/traffic.json" % dom
            traffic_val = tr
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": kd["tflops"], "peak": PEAK_F64_TFLOPS,
                               "unit": "TFLOP/s", "frac": kd["tflops"] / PEAK_F64_TFLOPS, "traffic": traffic_val,
                               "traffic_source": traffic_src, "kernel_ms": kd["ms_per_launch"], "launches_per_step": kd["launches_per_step"],
                               "algorithmic_flop_per_launch": KERNEL_FLOP[dom] * T * C / kd["launches_per_step"]}
        # the whole EM step against SURVEY 8(d)'s per-pair figure (240 logit + 242 statistics flop; this design
        # executes exactly that: every logit once, every statistic once), over all ranks
        step_tf = (FLOP_PER_PAIR_LLK + FLOP_PER_PAIR_ACC) * pairs_per_step / (dt / args.steps) / 1e12
        out["step_roofline"] = {"bound": "mfma", "algorithmic_flop_per_pair": FLOP_PER_PAIR_LLK + FLOP_PER_PAIR_ACC,
                                "achieved": step_tf, "peak": PEAK_F64_TFLOPS * world, "unit": "TFLOP/s",
                                "frac": step_tf / (PEAK_F64_TFLOPS * world)}
        if world == 1 and check:
            out["cpu_baseline"] = cpu_baseline(w, mean, iv, seed=99)
        out["summary"] = summarize(out)     # LAST key: the five BASELINE configs in a few hundred bytes (a truncated tail of the line still shows them)
        print(json.dumps(out), flush=True)
    g.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def summarize(out):
    """One compact entry per BASELINE.json config from the blocks of the line: value, unit, fraction of the fp64 peak, the CPU
    baseline (value / threads) and the parity error of the block's checker leg."""
    def cpu(b):
        c = (b or {}).get("cpu_baseline")
        return [c["value"], c["unit"], c["cores"]] if c else None

    def entry(b, err_key, frac=None, cpu_of=None):
        if not b:
            return None
        par = b.get("parity") or {}
        e = par.get(err_key)
        return {"value": b.get("value"), "unit": b.get("unit"), "frac": frac if frac is not None else (b.get("roofline") or {}).get("frac"),
                "cpu": cpu(cpu_of or b), "parity_err": e, "ok": par.get("ok")}
    ct = out.get("computetest")
    s = {"configs[1] TrainWorld EM": entry(out, "max_rel_err", frac=(out.get("step_roofline") or {}).get("frac")),
         "configs[2] IvExtractor": entry(out.get("secondary"), "max_rel_err_vs_oracle"),
         "configs[3] TotalVariability": entry(out.get("tv_em"), "max_rel_err"),
         "configs[4] IvTest scoring": entry(out.get("scoring"), "max_rel_err_vs_oracle"),
         "configs[0] ComputeTest": entry(ct, "max_abs_err_llk")}
    if s["configs[0] ComputeTest"] and (ct or {}).get("config0"):
        s["configs[0] ComputeTest"]["literal_config_llr_abs_err"] = ct["config0"]["abs_err"]
    if s["configs[1] TrainWorld EM"] and out.get("roofline"):
        s["configs[1] TrainWorld EM"]["k1_frac"] = out["roofline"]["frac"]
    tv = out.get("tv_em") or {}
    if s["configs[3] TotalVariability"] and (tv.get("parity") or {}).get("whole_iteration"):
        s["configs[3] TotalVariability"]["parity_err_T_whole_iteration"] = tv["parity"]["whole_iteration"]["max_rel_err_T"]
    return s


if __name__ == "__main__":
    main()
