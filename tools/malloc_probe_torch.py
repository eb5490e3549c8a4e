#!/usr/bin/env python3
"""hipMalloc latency by size INSIDE a PyTorch process (the bench's situation), fresh process per size: where does the slow path start?"""
import subprocess, sys
code = r'''
import ctypes as ct, time, sys, torch
torch.zeros(1, device="cuda")
hip = ct.CDLL("libamdhip64.so")
g = float(sys.argv[1]); p = ct.c_void_p()
t = time.perf_counter(); rc = hip.hipMalloc(ct.byref(p), ct.c_size_t(int(g * 2**30))); dt = time.perf_counter() - t
t2 = time.perf_counter(); q = ct.c_void_p(); hip.hipMalloc(ct.byref(q), ct.c_size_t(int(g * 2**30))); dt2 = time.perf_counter() - t2
print("%5.1f GiB: first hipMalloc %9.2f ms (rc %d), a second block of the same size %9.2f ms" % (g, dt * 1e3, rc, dt2 * 1e3))
'''
for g in (8, 12, 14, 16, 18, 20, 24, 28, 32, 48, 64):
    subprocess.run([sys.executable, "-c", code, str(g)])
