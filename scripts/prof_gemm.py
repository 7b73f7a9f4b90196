"""Runs the large bf16 GEMM (stress shape) a few times for rocprofv3 passes.  argv[1]: bf16|bf16x3|f32"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import core
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
N, C, D = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (50000, 10000, 512)))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
I = core.normalize_rows(torch.randn(N, D, device=dev, generator=g))
T = core.normalize_rows(torch.randn(C, D, device=dev, generator=g))
out = torch.empty(N, C, device=dev)
for _ in range(4):
    core.embed_gemm(I, T, mode=mode, out=out)
torch.cuda.synchronize()
