#!/usr/bin/env python3
"""BASELINE.json configs[2] (IvExtractor: 10 k utterances x 3 k frames) and configs[4] (IvTest: 100 k x 100 k trials at dim
400) run ONCE at full size on one MI355X, with oracle spot checks; prints one JSON object (kept under profiles/rNN/).

Where the bytes live (config 5): the 100 000 x 100 000 fp64 score matrix = 80 GB stays in HBM (288 GB part); what leaves the
device is a checksum, the per-model maxima and the sampled trials that are compared with the oracle -- shipping the matrix
over PCIe would take 80 GB / 63 GB/s = 1.3 s against 0.16-0.3 s of compute.

The oracle (oracle/) is used here as the CHECKER of sampled outputs only; nothing that is timed touches it.
usage: python tools/run_configs.py [ivextract] [scoring]   (default: both);  env SMALL=1 shrinks both for a dry run."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from conftest import make_gmm
from lia_ral_amd import capi

SMALL = bool(int(os.environ.get("SMALL", "0")))
dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
C, D, R = 2048, 60, 400
P = R * (R + 1) // 2
out = {}


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize()
    return time.perf_counter() - t


def ivextract():
    from bench import synth_frames
    from oracle import oracle as orc          # checker of the sampled utterances
    U, frames = (256, 3000) if SMALL else (10_000, 3000)
    w, mean, iv = make_gmm(C, D, seed=0)
    g = ctx.gmm(w, mean, iv)
    T = U * frames
    x = synth_frames(w, mean, iv, T, dev, seed=4242)            # 7.2 GB float32, resident
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    Tm = 0.01 * torch.randn((R, C * D), dtype=torch.float64, device=dev, generator=gen)
    invvar = torch.from_numpy(iv.ravel().copy()).to(dev)
    means = torch.from_numpy(mean.ravel().copy()).to(dev)
    tett = torch.empty((C, P), dtype=torch.float64, device=dev)
    N = torch.empty((U, C), dtype=torch.float64, device=dev)
    F = torch.empty((U, C * D), dtype=torch.float64, device=dev)  # 9.8 GB
    W = torch.empty((U, R), dtype=torch.float64, device=dev)
    ub = np.arange(U + 1, dtype=np.int64) * frames
    t_tett = timed(lambda: ctx.tv_tett(Tm, invvar, C, D, out=tett))

    def run():
        ts = timed(lambda: g.tv_stats(x, ub, N, F))
        tw = timed(lambda: (ctx.tv_subtract_m(N, F, means, C, D), ctx.tv_estimate_w(N, F, Tm, invvar, tett, C, D, out=W)))
        return ts, tw
    run()                                                        # warm-up: workspace allocation
    ts, tw = run()
    # parity: four utterances end to end (statistics -> centring -> i-vector) against the oracle
    rows = [0, 1, U // 2, U - 1]
    og = orc.Gmm(w, mean, iv)
    Tm_h = Tm.cpu().numpy()
    te_o = orc.tv_tett(Tm_h, iv.ravel(), C, D)
    errs = []
    for u in rows:
        xu = x[u * frames:(u + 1) * frames].cpu().numpy().astype(np.float64)
        No, Fo = orc.tv_stats(og, xu, np.zeros(frames, np.int64), 1)
        Fo = orc.tv_subtract_m(No, Fo, mean.ravel())
        Wo = orc.tv_estimate_w(No, Fo, Tm_h, iv.ravel(), te_o)
        errs.append(relerr(W[u].cpu().numpy(), Wo[0]))
    flop_stats = float(T) * C * (240 + 2 * (1 + D))              # 362 flop per pair
    flop_solve = U * 448.5e6
    out["config3_ivextractor"] = {
        "workload": "IvExtractor: 2048-g UBM, rank 400, %d utterances x %d frames, 1 x MI355X, everything resident in HBM" % (U, frames),
        "frames": T, "tett_ms": t_tett * 1e3, "stats_ms": ts * 1e3, "solve_ms": tw * 1e3,
        "ivectors_per_s": U / (ts + tw), "stats_gpairs_per_s": T * C / ts / 1e9,
        "stats_tflops": flop_stats / ts / 1e12, "solve_tflops": flop_solve / tw / 1e12,
        "end_to_end_frac_of_fp64_peak": (flop_stats + flop_solve) / (ts + tw) / 78.6e12,
        "hbm_resident_gb": {"features_f32": T * D * 4 / 1e9, "F": U * C * D * 8 / 1e9, "N": U * C * 8 / 1e9, "tett_packed": C * P * 8 / 1e9},
        "parity": {"utterances_checked": rows, "max_rel_err_vs_oracle": max(errs), "tolerance": 1e-6, "ok": max(errs) < 1e-6},
        "finite": bool(torch.isfinite(W).all().item())}
    g.close()
    del x, N, F, W, tett, Tm
    torch.cuda.empty_cache()


def scoring():
    from oracle import oracle as orc          # checker of the sampled trials
    M = S = 4096 + 37 if SMALL else 100_000
    gen = torch.Generator(device=dev); gen.manual_seed(11)
    models = torch.randn((R, M), dtype=torch.float64, device=dev, generator=gen)
    segs = torch.randn((R, S), dtype=torch.float64, device=dev, generator=gen)
    models /= models.norm(dim=0, keepdim=True); segs /= segs.norm(dim=0, keepdim=True)
    scores = torch.empty((M, S), dtype=torch.float64, device=dev)       # 80 GB at 100 k x 100 k: stays on the device
    Q = torch.randn((R, R), dtype=torch.float64, device=dev, generator=gen)
    Mah = (Q @ Q.T / R + torch.eye(R, dtype=torch.float64, device=dev)).contiguous()
    Gm = ((Q + Q.T) / R).contiguous(); Hm = ((Q @ Q.T) / (R * R)).contiguous()
    rf = 200
    Fp = torch.randn((R, rf), dtype=torch.float64, device=dev, generator=gen) / np.sqrt(R)
    FTJF = (Fp.T @ Fp + 0.1 * torch.eye(rf, dtype=torch.float64, device=dev)).contiguous()
    nsess = np.sort(np.random.default_rng(3).integers(1, 4, M)).astype(np.int64)
    mp = torch.randn((rf, M), dtype=torch.float64, device=dev, generator=gen) * torch.from_numpy(nsess).to(dev)
    sp = torch.randn((rf, S), dtype=torch.float64, device=dev, generator=gen)
    rng = np.random.default_rng(7)
    rows = np.unique(np.concatenate([[0, 127, 128, M - 129, M - 1], rng.integers(0, M, 11)]))
    cols = np.unique(np.concatenate([[0, 63, 128, S - 130, S - 1], rng.integers(0, S, 11)]))
    ri = torch.from_numpy(rows).to(dev); ci = torch.from_numpy(cols).to(dev)

    def sub(t2, idx):
        return np.ascontiguousarray(t2[:, idx].cpu().numpy())
    rules = {
        "cosine": (lambda: ctx.score_cosine(models, segs, out=scores), lambda: orc.score_cosine(sub(models, ri), sub(segs, ci)), R),
        "mahalanobis": (lambda: ctx.score_mahalanobis(models, segs, Mah, out=scores),
                        lambda: orc.score_mahalanobis(sub(models, ri), sub(segs, ci), Mah.cpu().numpy()), R),
        "twocov": (lambda: ctx.score_twocov(models, segs, Gm, Hm, out=scores),
                   lambda: orc.score_twocov(sub(models, ri), sub(segs, ci), Gm.cpu().numpy(), Hm.cpu().numpy()), R),
        "plda(rankF=200)": (lambda: ctx.score_plda(mp, nsess, sp, FTJF, out=scores),
                            lambda: orc.score_plda(sub(mp, ri), nsess[rows], sub(sp, ci), FTJF.cpu().numpy()), rf),
    }
    res = {}
    for name, (run, ref, k) in rules.items():
        run(); dt = min(timed(run), timed(run))
        got = scores[ri][:, ci].cpu().numpy()
        err = relerr(got, ref())
        res[name] = {"ms": dt * 1e3, "Gtrials_per_s": M * S / dt / 1e9, "tflops": 2.0 * k * M * S / dt / 1e12,
                     "frac_of_fp64_peak": 2.0 * k * M * S / dt / 78.6e12, "score_write_TBps": M * S * 8 / dt / 1e12,
                     "checksum": float(scores.sum().item()), "max_rel_err_vs_oracle": err, "ok": err < 1e-9,
                     "finite": bool(torch.isfinite(scores).all().item())}
    out["config5_ivtest_scoring"] = {
        "workload": "IvTest scoring: %d enrol x %d test %d-dim i-vectors, 1 x MI355X" % (M, S, R),
        "scores_resident_gb": M * S * 8 / 1e9, "scores_location": "HBM (never copied to the host; checksum + sampled trials leave the device)",
        "trials_checked": int(len(rows) * len(cols)), "tolerance": 1e-9, "rules": res}


if __name__ == "__main__":
    which = sys.argv[1:] or ["ivextract", "scoring"]
    if "ivextract" in which:
        ivextract()
    if "scoring" in which:
        scoring()
    print(json.dumps(out))
