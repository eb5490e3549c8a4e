#!/usr/bin/env python3
"""TrainWorld through the C++ host layer (liagpu::trainModelStream, host/liatools_gpu.cpp) timed per EM iteration: T frames
of BASELINE configs[1]'s model (2048 Gaussians, 60 dimensions) in 3000-frame segments, baggedFrameProbability 1.0 and 0.4.
The per-iteration time is the difference between a run of `hi` iterations and one of `lo` (the upload of the features, the
context and the first iteration's workspace allocations cancel).  LIAGPU_TRACE=1 prints the stages of every iteration.

  python tools/host_world_time.py [T] [p ...]        default T = 1 000 000, p = 1.0 0.4
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_frames, make_gmm
from lia_ral_amd import host_capi as h

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
ps = [float(a) for a in sys.argv[2:]] or [1.0, 0.4]
C, D, SEG = 2048, 60, 3000
w, mean, iv = make_gmm(C, D, seed=0)
x = make_frames(w, mean, iv, min(T, 200_000), seed=1)
x = np.ascontiguousarray(np.tile(x, ((T + len(x) - 1) // len(x), 1))[:T])
begin = np.arange(0, T, SEG); length = np.minimum(SEG, T - begin)
lo, hi = 2, 6


def run(p, its):
    t = time.perf_counter()
    r = h.train_world(x, begin, length, w, mean, 1.0 / iv, its, bagged_p=p, init_floor=0.0, final_floor=0.0, init_ceil=10.0, final_ceil=10.0)
    return (time.perf_counter() - t) * 1e3, r


for p in ps:
    run(p, 1)
    a, _ = run(p, lo)
    b, r = run(p, hi)
    per = (b - a) / (hi - lo)
    print(json.dumps({"frames": T, "segments": len(begin), "bagged_p": p, "ms_%d_iterations" % lo: a, "ms_%d_iterations" % hi: b,
                      "ms_iterations_timed_in_the_library": r["it_ms"].tolist(), "ms_per_iteration": per, "Gpairs_per_s_all_frames": T * C / per / 1e6,
                      "Gpairs_per_s_selected_frames": T * p * C / per / 1e6, "mean_llk_last": float(r["llk"][-1])}), flush=True)
