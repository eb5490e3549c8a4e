#!/usr/bin/env python3
"""FETCH_SIZE calibration (MI355X_MICROARCH.md, HBM section): stream a KNOWN byte count with the same
access pattern as the statistics kernel's frame staging (one coalesced dword per lane) and with
16-byte loads, under `rocprofv3 --pmc FETCH_SIZE`, and compare the counter with the byte count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lia_ral_amd import capi

T, D = 10_000_000, 60
x = torch.randn((T, D), dtype=torch.float32, device="cuda")          # 2.4 GB > 256 MiB Infinity Cache
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
acc = torch.zeros(2 * D + 1, dtype=torch.float64, device="cuda")
for _ in range(3):
    ctx.frame_moments(x, acc)            # k_frame_moments<float>: dword loads, 2.4e9 bytes per launch
y = x.clone()                            # torch copy kernel: 16-B loads, 2.4e9 bytes read + written
torch.cuda.synchronize()
print("bytes per k_frame_moments launch:", T * D * 4)
ctx.close()
