#!/usr/bin/env python3
"""The C++ host layer (liagpu::TVAcc, device-resident) timed on the T-matrix EM of BASELINE configs[3]'s shape next to the
Python / torch path of tools/bench_tv.py: U utterances, C = 2048, D = 60, R = 400, one rank.  The statistics cross PCIe ONCE
(setStats); every iteration after that runs on device buffers."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_gmm
from lia_ral_amd import host_capi as h

U = int(os.environ.get("TV_U", "1024"))
C, D, R = 2048, 60, 400
rng = np.random.default_rng(0)
w, mean, iv = make_gmm(C, D, seed=0)
N = rng.gamma(0.6, 2.5, (U, C))
F = rng.normal(size=(U, C * D)) * np.sqrt(np.repeat(N, D, axis=1) + 0.05) + np.repeat(N, D, axis=1) * mean.ravel()
Tm = rng.normal(0, 0.01, (R, C * D))
t = time.perf_counter()
Tn, means, times = h.tv_train_dist(N, F, (w, mean, 1.0 / iv), Tm, 3)
wall = time.perf_counter() - t
it = times[1:].mean(0)          # the first iteration allocates the workspace
print(json.dumps({"utterances": U, "iterations": 3, "wall_s_including_upload_and_download": wall,
                  "per_iteration_ms": {"tett": it[0], "estep": it[1], "mstep": it[2], "min_divergence": it[3]},
                  "estep_us_per_utterance": it[1] * 1e3 / U, "first_iteration_ms": times[0].tolist(), "finite": bool(np.isfinite(Tn).all())}))
