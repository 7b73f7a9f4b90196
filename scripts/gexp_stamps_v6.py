"""In-kernel stamps of gemm_nt_bf16_exp_v6_kernel (MCD_GEMM_EXP_ABLATE=8: product + s_memtime stamps of workgroup 0's four waves;
9: the same without the E stores): where a tile's cycles go around the tile boundary.  argv: [ablate]   (MCD_GEMM_EXP_OVERLAP from the environment: with 1 / 2 the boundary phase
sits between the 'k0 barrier' and 'k1 wait' stamps, and for 2 the 'k1' stamps are the sync point inside the phase)"""
import os, sys
ab = sys.argv[1] if len(sys.argv) > 1 else "8"
os.environ["MCD_GEMM_EXP_ABLATE"] = ab
os.environ.setdefault("MCD_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mammo-clip-dissect_amd", "csrc", "libmcd_hip_dev.so"))   # dev build (make dev)
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
parts = 2 * ((C + 255) // 256) * ((N + 63) // 64 * 64) * 4          # mcd_embed_gemm_exp_workspace: one row per (concept tile, wave row)
ops = nws - parts
st = ws[ops:ops + 4 * 256 * 8].view(torch.int64).cpu().numpy().reshape(4, 16, 16)
names = ["tile top -> k0 wait", "k0 vmcnt+lgkm wait", "k0 barrier", "k0 body -> k1 wait", "k1 wait", "k1 barrier", "k1 body", "k2 wait", "k2 barrier",
         "k2 body", "k3 wait", "k3 barrier", "k3 body + rounds 1-3", "epilogue"]
print("ablate", ab, "- cycles (s_memtime), workgroup 0, mean over tiles 2..13 (each stamp costs ~45)")
for w in range(4):
    d = np.diff(st[w, 2:14, :15].astype(np.float64), axis=1)
    nxt = st[w, 3:15, 0] - st[w, 2:14, 14]
    print("wave", w, " tile period %.0f" % np.mean(st[w, 3:15, 0] - st[w, 2:14, 0]), " epilogue end -> next tile top %.0f" % nxt.mean())
    print("   " + "  ".join("%s %.0f" % (n, v) for n, v in zip(names, d.mean(axis=0))))
