#!/usr/bin/env python3
"""k_llk_mfma<WZ> on the same 768 000 frames and model, eight times, each launch right behind
   em : the EM statistics kernel k_stats_z<SQ=1> of the previous call   (python tools/k1_clock_probe.py em)
   tv : the N / F statistics kernel k_stats_z<SQ=0> of the previous call (python tools/k1_clock_probe.py tv)
Run under `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` (tools/profile_r05.sh): GRBM_GUI_ACTIVE / duration of k_llk_mfma is
the shader clock the kernel ran at in each order -- round 4 measured it 4.4 % slower behind k_stats_z<SQ=0> and guessed at the
clock / power state without a counter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
mode = sys.argv[1]
C, D, T, U = 2048, 60, 768000, 256
w, mean, iv = make_gmm(C, D, seed=0)
dev = torch.device("cuda", 0)
x = bench.synth_frames(w, mean, iv, T, dev, seed=777)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1); ctx.set_option("assume_finite", 1)
g = ctx.gmm(w, mean, iv)
acc = torch.zeros(g.em_acc_len(), dtype=torch.float64, device=dev)
N = torch.empty((U, C), dtype=torch.float64, device=dev); F = torch.empty((U, C * D), dtype=torch.float64, device=dev)
ub = np.arange(U + 1, dtype=np.int64) * 3000
ms = []
for rep in range(9):
    if mode == "em":
        acc.zero_(); g.em_accumulate(x, acc=acc)
    else:
        g.tv_stats(x, ub, N, F)
    torch.cuda.synchronize()
    ms.append(ctx.kernel_ms("k_llk_mfma"))
print("mode %s: k_llk_mfma<WZ> ms per launch (HIP events, first call dropped): %s  mean %.4f" % (mode, ["%.3f" % m for m in ms[1:]], float(np.mean(ms[1:]))))
