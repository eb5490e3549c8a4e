#!/usr/bin/env python3
"""The i-vector secondary of bench.py on its own (IvExtractor end to end on 512 utterances x 3000 frames): the command the rocprofv3
summary profiles/r04/iv_secondary_kernel_stats.txt was taken from.  Prints the secondary's JSON block."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_gmm
from lia_ral_amd import capi
import bench
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev); torch.cuda.set_stream(side)
ctx = capi.Context(0, side.cuda_stream); ctx.set_option("timing", 1)
w, mean, iv = make_gmm(bench.C, bench.D, seed=0)
g = ctx.gmm(w, mean, iv)
s = bench.ivector_secondary(ctx, g, w, mean, iv, dev, 0, 1, check=("--check" in sys.argv))
for k in [k for k in s if k.startswith("_")]:
    del s[k]
print(json.dumps(s))
