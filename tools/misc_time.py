import sys, os, time
R0 = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import numpy as np, torch
from conftest import make_gmm
from lia_ral_amd import capi
dev = torch.device("cuda", 0)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
C, D, R, U = 2048, 60, 400, 1024
SV = C * D
g = torch.Generator(device=dev); g.manual_seed(0)
def tm(f, reps=2):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
T = 0.01 * torch.randn((R, SV), dtype=torch.float64, device=dev, generator=g)
w, mean, iv = make_gmm(C, D, seed=0)
N = torch.rand((U, C), dtype=torch.float64, device=dev, generator=g) * 3
F = torch.randn((U, SV), dtype=torch.float64, device=dev, generator=g)
means = torch.from_numpy(mean.ravel().copy()).to(dev); invvar = torch.from_numpy(iv.ravel().copy()).to(dev)
print("orthonormalize_t %.1f ms" % tm(lambda: ctx.tv_orthonormalize_t(T.clone()), 1))
print("subtract_m %.2f ms" % tm(lambda: ctx.tv_subtract_m(N, F, means, C, D)))
Wv = torch.randn((U, R), dtype=torch.float64, device=dev, generator=g)
print("subtract_m_plus_tw %.2f ms" % tm(lambda: ctx.tv_subtract_m_plus_tw(N, F, means, T, Wv, C, D)))
print("norm_statistics %.2f ms" % tm(lambda: ctx.tv_norm_statistics(N, F, means, invvar, C, D)))
Z = torch.randn((U, SV), dtype=torch.float64, device=dev, generator=g); Dm = torch.rand(SV, dtype=torch.float64, device=dev, generator=g)
print("jfa_estimate_z_and_d %.2f ms" % tm(lambda: ctx.jfa_estimate_z_and_d(N, F, invvar, Dm.clone(), C, D, out=Z)))
print("jfa_subtract(M+VY+DZ) %.2f ms" % tm(lambda: ctx.jfa_subtract(N, F, C, D, means=means, T=T, W=Wv, Dm=Dm, Z=Z)))
# approximate extractors (IvExtractor modes ubmWeight / eigenDecomposition)
wgt = torch.rand(C, dtype=torch.float64, device=dev, generator=g); wgt /= wgt.sum()
Wm = torch.empty((R, R), dtype=torch.float64, device=dev)
print("weighted_cov %.2f ms" % tm(lambda: ctx.tv_weighted_cov(T, wgt, C, D, out=Wm)))
Wm2 = (Wm + Wm.T) / 2 + torch.eye(R, dtype=torch.float64, device=dev)
Wout = torch.zeros((U, R), dtype=torch.float64, device=dev)
print("estimate_w_ubm_weight %.2f ms" % tm(lambda: ctx.tv_estimate_w_ubm_weight(N, F, T, Wm2, C, D, out=Wout)))
Q = torch.linalg.qr(torch.randn((R, R), dtype=torch.float64, device=dev, generator=g))[0].contiguous()
Dq = torch.empty((C, R), dtype=torch.float64, device=dev)
print("approximate_tctc %.2f ms" % tm(lambda: ctx.tv_approximate_tctc(T, Q, C, D, out=Dq)))
print("estimate_w_eigen %.2f ms" % tm(lambda: ctx.tv_estimate_w_eigen(N, F, T, Dq, Q, C, D, out=Wout)))
tett = torch.empty((C, R * (R + 1) // 2), dtype=torch.float64, device=dev)
print("tett %.2f ms" % tm(lambda: ctx.tv_tett(T, invvar, C, D, out=tett)))
print("estimate_w (exact) %.2f ms" % tm(lambda: ctx.tv_estimate_w(N, F, T, invvar, tett, C, D, out=Wout)))
