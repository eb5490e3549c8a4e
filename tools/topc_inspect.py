"""Inspect the raw outputs of k_llk_mfma<TC> (candidate lists, thresholds) against numpy logits: every top-C Gaussian must be in its
frame list, every record must carry the logit of its own Gaussian (this is how the VALU -> MFMA operand hazard was found)."""
import sys, os, ctypes as ct
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from conftest import make_gmm, make_frames
from lia_ral_amd import capi
C, D, T, ctop = 128, 60, 1000, 10
w, mean, iv = make_gmm(C, D, seed=C)
x = make_frames(w, mean, iv, T, seed=T + 1)
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
g = ctx.gmm(w, mean, iv)
raw = (ct.c_char * 96).from_address(g._h.value)
ints = np.frombuffer(raw, np.int32, 5, 8); ptrs = np.frombuffer(raw, np.uint64, 8, 32)
Cc, Dd, KS, nct, Cp64 = [int(v) for v in ints]
Pt = int(ptrs[5])
print("C D KS nct", Cc, Dd, KS, nct)
lib = capi.lib
f = getattr(lib, "_Z13gmmk_llk_topcP12ihipStream_tiiPKvlliPKdiiiPdPiS5_S5_S6_")
CAP = 256
Tp = (T + 255) // 256 * 256
xd = torch.from_numpy(x).cuda()
cand = torch.zeros((Tp, CAP, 2), dtype=torch.float64, device="cuda")
cnt = torch.zeros(Tp, dtype=torch.int32, device="cuda"); theta = torch.zeros(Tp, dtype=torch.float64, device="cuda")
slow = torch.zeros(Tp, dtype=torch.float64, device="cuda"); efin = torch.zeros(Tp, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
rc = f(ct.c_void_p(st), KS, 0, ct.c_void_p(xd.data_ptr()), ct.c_long(T), ct.c_long(D), D, ct.c_void_p(Pt), nct, 1, ctop,
       ct.c_void_p(cand.data_ptr()), ct.c_void_p(cnt.data_ptr()), ct.c_void_p(theta.data_ptr()), ct.c_void_p(slow.data_ptr()), ct.c_void_p(efin.data_ptr()))
torch.cuda.synchronize()
print("rc", rc)
cand = cand.cpu().numpy(); cnt = cnt.cpu().numpy(); theta = theta.cpu().numpy()
xx = x.astype(np.float64)
lw = np.log(w) - 0.5 * D * np.log(2 * np.pi) + 0.5 * np.log(iv).sum(1)
nbad = 0
for t in range(T):
    z = lw - 0.5 * (((xx[t][None, :] - mean) ** 2) * iv).sum(1)
    n = cnt[t]
    idx = cand[t, :n, 1].copy().view(np.int64)
    zz = cand[t, :n, 0]
    top = np.argsort(-z)[:ctop]
    missing = [c for c in top if c not in idx]
    dup = len(idx) != len(set(idx.tolist()))
    zerr = np.max(np.abs(zz - z[idx])) if n else 0
    if missing or dup or zerr > 1e-6 or theta[t] > z[top[-1]]:
        nbad += 1
        if nbad <= 6:
            print("frame", t, "n", n, "theta", theta[t], "10th", z[top[-1]], "missing", missing, [(c, z[c], c // 32, c % 16) for c in missing], "dup", dup, "zerr", zerr)
            print("   list", sorted(idx.tolist()))
print("bad", nbad)
