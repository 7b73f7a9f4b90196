"""In-kernel stamps of gemm_nt_bf16_exp_w4_kernel (MCD_GEMM_EXP_ABLATE=12: K loop only + s_memtime stamps of workgroup 0's
wave 0): where a stage's cycles go."""
import os, sys
os.environ["MCD_GEMM_EXP_ABLATE"] = "12"
os.environ["MCD_GEMM_EXP_LAYOUT"] = "w4"
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
parts = 2 * ((C + 191) // 192) * ((N + 63) // 64 * 64) * 4
ops = nws - parts
st = ws[ops:ops + 2048 * 8].view(torch.int64).cpu().numpy().reshape(512, 4)
valid = int((st[:, 3] != 0).sum())
st = st[:valid]
print("valid stages of workgroup 0:", valid, " ticks per stage:", (st[-1, 0] - st[0, 0]) / (valid - 1))
def stats(name, a):
    a = a[8:].astype(np.float64)
    print("%-58s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f" % (name, a.mean(), *np.percentile(a, [10, 50, 90])))
stats("stage period (top to top)", np.diff(st[:, 0]))
stats("[0->1] reads k1 issued + 16 MFMAs of k-step 0 issued", st[:, 1] - st[:, 0])
stats("[1->2] vmcnt (next stage landed) + lgkmcnt(0)", st[:, 2] - st[:, 1])
stats("[2->3] barrier", st[:, 3] - st[:, 2])
stats("[3->next 0] 8 DMA + 8 reads + 16 MFMAs of k-step 1 issued", st[1:, 0] - st[:-1, 3])
print("stages 30..70: [0->1, 1->2, 2->3, 3->next]")
for g_ in range(30, min(70, valid - 1)):
    print(g_, int(st[g_, 1] - st[g_, 0]), int(st[g_, 2] - st[g_, 1]), int(st[g_, 3] - st[g_, 2]), int(st[g_ + 1, 0] - st[g_, 3]))
