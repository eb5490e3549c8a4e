#!/usr/bin/env python3
"""ComputeTest kernels on a 2048-Gaussian world model: plain LLK (k_llk_mfma), top-C determination, top-C use; device-resident."""
import ctypes as ct, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
from lia_ral_amd.capi import lib, _ptr, _chk
from bench import synth_frames
dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("timing", 1)
ctx.set_option("topc_z", int(os.environ.get("TOPC_Z", "1")))
ctx.set_option("topc_fused", int(os.environ.get("TOPC_FUSED", "1")))
ctx.set_option("topc_rank_direct", int(os.environ.get("TOPC_RANK_DIRECT", "0")))   # 1 = survivors of every frame re-evaluated in the direct form (round 2)
ctx.set_option("topc_use_lanes", int(os.environ.get("TOPC_USE_LANES", "4")))   # lanes per candidate in the client pass: 4 (default) or 1
ctx.set_option("topc_rank2", int(os.environ.get("TOPC_RANK2", "1")))   # 0 = one frame per wave for every frame
C, D, T, ctop = 2048, 60, int(os.environ.get("T", "1000000")), 10
w, mean, iv = make_gmm(C, D, seed=0, spread=float(os.environ.get("SPREAD", "2.0")))
x = synth_frames(w, mean, iv, T, dev, seed=5)
g = ctx.gmm(w, mean, iv)
out = torch.empty(T, dtype=torch.float64, device=dev)
idx = torch.empty((T, ctop), dtype=torch.int32, device=dev); lk = torch.empty((T, ctop), dtype=torch.float64, device=dev)
nlk = torch.empty(T, dtype=torch.float64, device=dev); nllk = torch.empty_like(nlk); nw = torch.empty_like(nlk)
def tm(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
a = tm(lambda: g.llk(x, out=out))
b = tm(lambda: _chk(lib.gmmiv_llk_determine_top(ctx._h, g._h, _ptr(x), capi.F32, ct.c_int64(T), ct.c_int64(D), ctop, capi.TOP_COMPLETE, ct.c_double(-200.0), ct.c_double(200.0), _ptr(idx), _ptr(lk), _ptr(nlk), _ptr(nllk), _ptr(nw), _ptr(out))))
kk = {k: round(ctx.kernel_ms(k), 3) for k in ("k_llk_mfma", "k_topc_rank", "k_topc_from_z", "k_topc_determine")}
kk["fallbacks"] = ctx.set_option("topc_fallbacks", 0)
c = tm(lambda: _chk(lib.gmmiv_llk_use_top(ctx._h, g._h, _ptr(x), capi.F32, ct.c_int64(T), ct.c_int64(D), ctop, _ptr(idx), _ptr(nllk), capi.TOP_COMPLETE, ct.c_double(-200.0), ct.c_double(200.0), _ptr(out))))
print(kk)
print("T=%d: llk %.2f ms (%.1f Gpair/s) | determine_top %.2f ms (%.1f Gpair/s) | use_top %.3f ms (%.1f Mframe/s)" % (T, a, T * C / a / 1e6, b, T * C / b / 1e6, c, T / c / 1e3))
