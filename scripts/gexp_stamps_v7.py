"""Tile period of gemm_nt_bf16_exp_v7_kernel in shader cycles (dev build, MCD_GEMM_EXP_V7=1, MCD_GEMM_EXP_ABLATE=8: s_memtime at every
tile top of workgroup 0's four waves; 9: the same without the E stores).  argv: [ablate]"""
import os, sys
ab = sys.argv[1] if len(sys.argv) > 1 else "8"
os.environ["MCD_GEMM_EXP_ABLATE"] = ab
os.environ["MCD_GEMM_EXP_V7"] = "1"
os.environ.setdefault("MCD_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mammo-clip-dissect_amd", "csrc", "libmcd_hip_dev.so"))
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
for _ in range(20):
    core.check(L.mcd_embed_gemm_exp(I.data_ptr(), D, T.data_ptr(), D, N, C, D, 10.0, 1, E.data_ptr(), 10112, rinv.data_ptr(), ws.data_ptr(), nws, None))
torch.cuda.synchronize()
parts = 2 * ((C + 255) // 256) * ((N + 63) // 64 * 64) * 4
ops = nws - parts
st = ws[ops:ops + 4 * 256 * 8].view(torch.int64).cpu().numpy().reshape(4, 256)
print("v7 ablate", ab, "- cycles (s_memtime) between tile tops, workgroup 0")
for w in range(4):
    d = np.diff(st[w, :31].astype(np.float64))
    print("wave", w, " tiles 2..28: mean %.0f  min %.0f  max %.0f   (per k-step %.0f)" % (d[2:28].mean(), d[2:28].min(), d[2:28].max(), d[2:28].mean() / 16))
