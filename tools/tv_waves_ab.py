import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
C, D, U, frames = 2048, 60, 1024, 3000
w, mean, iv = make_gmm(C, D, seed=0)
dev = torch.device("cuda", 0)
x = bench.synth_frames(w, mean, iv, U * frames, dev, seed=1)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1)
g = ctx.gmm(w, mean, iv)
ub = np.arange(U + 1, dtype=np.int64) * frames
N = torch.empty((U, C), dtype=torch.float64, device=dev); F = torch.empty((U, C * D), dtype=torch.float64, device=dev)
ref = None
for rep in range(2):
    for zw, tv4 in ((8, 1), (8, 0), (16, 0), (4, 0)):
        ctx.set_option("z_waves", zw); ctx.set_option("z_tv4", tv4)
        g.tv_stats(x, ub, N, F); g.tv_stats(x, ub, N, F); torch.cuda.synchronize()
        a = F.clone()
        if ref is None: ref = a
        ms = ctx.kernel_ms("k_stats_z")
        print("z_waves %2d tv4 %d: k_stats_z %.2f ms (%.1f Gpair/s, %.2f TB/s)  max rel diff %.1e" % (zw, tv4, ms, U * frames * C / ms / 1e6, U * frames * C * 8 / ms / 1e9, float(((a - ref).abs().max() / ref.abs().max()).item())))
