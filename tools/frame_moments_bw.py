import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lia_ral_amd import capi
dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream); ctx.set_option("timing", 1)
T, D = 10_000_000, 60
x = torch.randn((T, D), dtype=torch.float32, device=dev)
acc = torch.zeros(2 * D + 1, dtype=torch.float64, device=dev)
for _ in range(3):
    acc.zero_(); ctx.frame_moments(x, acc=acc); torch.cuda.synchronize()
    ms = ctx.kernel_ms("k_frame_moments")
    print("k_frame_moments %.3f ms  %.2f TB/s" % (ms, T * D * 4 / ms / 1e9))
ref = x.double().sum(0)
print("rel err", float(((acc[:D] - ref).abs().max() / ref.abs().max()).item()))
