"""In-kernel stamps of gemm_nt_bf16_exp_kernel (MCD_GEMM_EXP_ABLATE=12: K loop only + s_memtime stamps of workgroup 0's
first loader wave and first compute wave): where a stage's cycles go."""
import os, sys, ctypes
os.environ["MCD_GEMM_EXP_ABLATE"] = "12"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mammo_clip_dissect_amd as m
from mammo_clip_dissect_amd import core, _lib
N, C, D = 25000, 10000, 512
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
I = torch.randn(N, D, device=dev, generator=g); T = torch.randn(C, D, device=dev, generator=g)
L = _lib.load()
nws = L.mcd_embed_gemm_exp_workspace(N, C, D)
ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
E = torch.empty((N, 10112), dtype=torch.bfloat16, device=dev); rinv = torch.empty(N, device=dev)
for _ in range(3):
    core.check(L.mcd_embed_gemm_exp(I.data_ptr(), D, T.data_ptr(), D, N, C, D, 10.0, 1, E.data_ptr(), 10112, rinv.data_ptr(), ws.data_ptr(), nws, None))
torch.cuda.synchronize()
tm = int(os.environ.get("MCD_GEMM_EXP_TM", "256"))
pitch = 512 + 64
ops = ((N + C) * pitch * 2 + 255) // 256 * 256
st = ws[ops:ops + 4096 * 8].view(torch.int64).cpu().numpy()
ld, cp = st[:2048].reshape(512, 4), st[2048:].reshape(512, 4)
valid = int((cp[:, 1] != 0).sum())
ld, cp = ld[:valid], cp[:valid]
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "gexp_stamps.npy"), np.stack([ld, cp]))
print("valid stages of workgroup 0:", valid, " total ticks:", int(cp[-1, 3] - cp[0, 0]), " per stage:", (cp[-1, 3] - cp[0, 0]) / valid)
def stats(name, a):
    a = a[8:].astype(np.float64)
    print("%-44s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f" % (name, a.mean(), *np.percentile(a, [10, 50, 90])))
print("s_memtime ticks (100 MHz constant clock? -> see period); stage period from compute stamps:")
stats("compute: stage period (barrier-to-barrier)", np.diff(cp[:, 1]))
stats("compute: wait at barrier", cp[:, 1] - cp[:, 0])
stats("compute: k-step 0 reads + MFMAs done", cp[:, 2] - cp[:, 1])
stats("compute: k-step 1 issued", cp[:, 3] - cp[:, 2])
stats("loader: wait for its share of the stage (vmcnt)", ld[:, 1] - ld[:, 0])
stats("loader: wait at barrier", ld[:, 2] - ld[:, 1])
stats("loader: issue next stage", ld[:, 3] - ld[:, 2])
stats("loader: stage period", np.diff(ld[:, 2]))
print("first 40 stages, compute wave: [wait barrier, k0 done, k1 issued, gap to next]")
for g in range(30, 70):
    print(g, int(cp[g, 1] - cp[g, 0]), int(cp[g, 2] - cp[g, 1]), int(cp[g, 3] - cp[g, 2]), int(cp[g + 1, 0] - cp[g, 3]) if g + 1 < valid else -1,
          "| loader [vmcnt wait, barrier wait, issue]", int(ld[g, 1] - ld[g, 0]), int(ld[g, 2] - ld[g, 1]), int(ld[g, 3] - ld[g, 2]))
