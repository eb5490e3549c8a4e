#!/usr/bin/env python3
"""bench.py -- headline benchmark: one UBM EM pass (TrainWorld hot path) per step.

Workload (BASELINE.json configs[1]): 2048-Gaussian diagonal UBM, 60-dim float32 frames,
10 M synthetic frames resident in HBM per GPU.  One step = the E-step of one EM iteration:
log-likelihood pass + full-posterior sufficient statistics over every frame (HIP, fp64 MFMA),
the RCCL all-reduce of the 1.98 MB statistics when N > 1, the M-step and the model re-pack.
Metric: Gframe-Gaussian evaluations/s, whole job (all ranks).  Scaling is weak (10 M frames/GPU).

Launch: python bench.py --gpus 1   |   python -m torch.distributed.run --nproc-per-node N bench.py --gpus N
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
               "k_em_fused": FLOP_PER_PAIR_LLK + FLOP_PER_PAIR_ACC}
PEAK_F64_TFLOPS = 78.6        # MI355X fp64 matrix = vector peak (AMD datasheet; measured ceiling in DESIGN.md)


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


def cpu_baseline(w, mean, iv, seed):
    """Oracle port (-O3 -ffast-math, pthreads with the reference's frame partitioning) on the host cores."""
    from conftest import make_frames
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    frames = min(30000 * cores, 2_000_000)
    x = make_frames(w, mean, iv, frames, seed=seed).astype(np.float64)
    g = orc.Gmm(w, mean, iv)
    orc.em_accumulate(g, x[:2000], fast=True, threads=cores)       # warm-up / page-in
    t = time.time()
    orc.em_accumulate(g, x, fast=True, threads=cores)
    dt = time.time() - t
    return {"value": frames * C / dt / 1e9, "unit": "Gframe-Gaussian/s", "cores": cores, "kind": "port",
            "sample": "%d frames x %d Gaussians, one EM statistics pass, %d pthreads (oracle/oracle_mt.c, "
                      "gcc -O3 -ffast-math like the reference), %.1f s" % (frames, C, cores, dt)}


def ivector_secondary(ctx, g, w, mean, iv, dev, rank, world, U=512, frames=3000, R=400):
    """BASELINE.json configs[2] on a bounded slice: IvExtractor end-to-end (Baum-Welch N/F statistics,
    substractM, L = I + sum N TETt, SPD inverse, w = L^-1 T Sigma^-1 F) for U utterances x 3000 frames
    per GPU; TETt is precomputed once (T is fixed during extraction, IvExtractor.cpp:136)."""
    T = U * frames
    x = synth_frames(w, mean, iv, T, dev, seed=777 + rank)
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

    def run():
        t0 = time.perf_counter()
        g.tv_stats(x, ub, N, F)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ctx.tv_subtract_m(N, F, means, C, D)
        ctx.tv_estimate_w(N, F, Tm, invvar, tett, C, D, out=W)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        times["stats_ms"] = (t1 - t0) * 1e3
        times["solve_ms"] = (t2 - t1) * 1e3
        return t2 - t0

    run()                      # warm-up (workspace allocation)
    dt = min(run(), run())
    # the same statistics through the single-pass cooperative kernel (opt-in, em_fused.hip)
    fused = {}
    try:
        ctx.set_option("em_fused", 1)
        N2 = torch.empty_like(N); F2 = torch.empty_like(F)
        g.tv_stats(x, ub, N2, F2); torch.cuda.synchronize()
        t0 = time.perf_counter(); g.tv_stats(x, ub, N2, F2); torch.cuda.synchronize()
        fused["stats_ms"] = (time.perf_counter() - t0) * 1e3
        ctx.tv_subtract_m(N2, F2, means, C, D)
        fused["max_rel_diff_F"] = float(((F2 - F).abs().max() / F.abs().max()).item())
        fused["i-vectors/s"] = U * world / (fused["stats_ms"] * 1e-3 + times["solve_ms"] * 1e-3)
    finally:
        ctx.set_option("em_fused", 0)
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return {"metric": "i-vectors/s (IvExtractor end-to-end, 2048-g UBM, rank 400, 3000-frame utterances)",
            "value": U * world / dt, "unit": "i-vectors/s", "utterances_per_gpu": U, "stats_ms": times["stats_ms"],
            "solve_ms": times["solve_ms"], "finite": bool(torch.isfinite(W).all().item()), "fused_stats": fused}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=10_000_000, help="frames per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--mean-spread", type=float, default=2.0,
                    help="std of the synthetic UBM means (SURVEY 8(d): 2.0; smaller = overlapping Gaussians)")
    ap.add_argument("--em-fused", type=int, default=-1, help="A/B knob: 1 = single-pass cooperative EM kernel, 0 = two-kernel path")
    ap.add_argument("--wg-waves", type=int, default=0, help="A/B knob: waves per workgroup of the MFMA kernels (4 or 8)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI

    from conftest import make_gmm
    from lia_ral_amd import capi

    w, mean, iv = make_gmm(C, D, seed=0, spread=args.mean_spread)
    T = args.frames
    x = synth_frames(w, mean, iv, T, dev, seed=1234 + rank)
    ctx = capi.Context(local, torch.cuda.current_stream().cuda_stream)
    ctx.set_option("timing", 1)
    if args.wg_waves:
        ctx.set_option("wg_waves", args.wg_waves)
    if args.em_fused >= 0:
        ctx.set_option("em_fused", args.em_fused)
    g = ctx.gmm(w, mean, iv)
    nacc = g.em_acc_len()
    acc = torch.zeros(nacc, dtype=torch.float64, device=dev)
    mean_d = torch.from_numpy(mean).to(dev)
    cov_d = torch.from_numpy(1.0 / iv).to(dev)
    w_d = torch.empty(C, dtype=torch.float64, device=dev)
    nm_d = torch.empty_like(mean_d)
    nc_d = torch.empty_like(cov_d)
    cov_signal = cov_d.mean(0).contiguous()       # stands in for the global covariance of computeMeanCov

    kern_ms = {}

    def step(record=False):
        nonlocal mean_d, cov_d, nm_d, nc_d
        acc.zero_()
        g.em_accumulate(x, acc=acc)                         # K1 (lse) + K2 (statistics) + reduce
        if record:
            for name in KERNEL_FLOP:
                if ctx.kernel_launches(name) > 0:
                    kern_ms.setdefault(name, []).append((ctx.kernel_ms(name), ctx.kernel_launches(name)))
        if world > 1:
            dist.all_reduce(acc)                            # EM sufficient statistics, 1.98 MB fp64
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
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

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
    }
    # the same E-step on a heavily overlapping mixture (means ~ N(0, 0.3^2)): hundreds of Gaussians carry
    # posterior mass per frame, none of the data-dependent skips of K1/K2 can fire -- the dense floor.
    dense = None
    if not args.no_secondary:
        wd, md, ivd = make_gmm(C, D, seed=3, spread=0.3)
        Td = min(T, 2_000_000)
        xd = synth_frames(wd, md, ivd, Td, dev, seed=4321 + rank)
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
    if not args.no_secondary:
        g.set(w, mean, iv)     # back to the seed model for the i-vector slice
        secondary = ivector_secondary(ctx, g, w, mean, iv, dev, rank, world)
    if rank == 0:
        if secondary:
            out["secondary"] = secondary
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
            traffic = None
            tf = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tf):
                try:
                    traffic = json.load(open(tf)).get(dom + "_hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": kd["tflops"], "peak": PEAK_F64_TFLOPS,
                               "unit": "TFLOP/s", "frac": kd["tflops"] / PEAK_F64_TFLOPS, "traffic": traffic,
                               "kernel_ms": kd["ms_per_launch"], "launches_per_step": kd["launches_per_step"],
                               "algorithmic_flop_per_launch": KERNEL_FLOP[dom] * T * C / kd["launches_per_step"]}
        # the whole EM step against SURVEY 8(d)'s per-pair figure (240 logit + 242 statistics flop; this design
        # executes exactly that: every logit once, every statistic once), over all ranks
        step_tf = (FLOP_PER_PAIR_LLK + FLOP_PER_PAIR_ACC) * pairs_per_step / (dt / args.steps) / 1e12
        out["step_roofline"] = {"bound": "mfma", "algorithmic_flop_per_pair": FLOP_PER_PAIR_LLK + FLOP_PER_PAIR_ACC,
                                "achieved": step_tf, "peak": PEAK_F64_TFLOPS * world, "unit": "TFLOP/s",
                                "frac": step_tf / (PEAK_F64_TFLOPS * world)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, mean, iv, seed=99)
        print(json.dumps(out), flush=True)
    g.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
