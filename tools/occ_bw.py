import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch, ctypes as ct
from conftest import make_gmm
from lia_ral_amd import capi
from lia_ral_amd.capi import lib, _ptr, _chk
from bench import synth_frames
dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
C, D, T = 2048, 60, 200000
w, mean, iv = make_gmm(C, D, seed=0, spread=2.0)
x = synth_frames(w, mean, iv, T, dev, seed=5)
g = ctx.gmm(w, mean, iv)
out = torch.empty((T, C), dtype=torch.float64, device=dev)
def run(): _chk(lib.gmmiv_occ(ctx._h, g._h, _ptr(x), capi.F32, ct.c_int64(T), ct.c_int64(D), _ptr(out)))
run(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): run()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
print("gmmiv_occ T=%d: %.2f ms  %.1f Gpair/s  row sums ok: %s" % (T, ms, T * C / ms / 1e6, bool(torch.allclose(out.sum(1), torch.ones(T, dtype=torch.float64, device=dev)))))
