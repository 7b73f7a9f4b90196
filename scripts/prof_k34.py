"""Runs K3 and K4 at config-2 shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_dissect_amd as m
from mammo_clip_dissect_amd import core
dev = torch.device("cuda:0")
N, C, D, L, UL, K = 10000, 763, 512, 12, 768, 100
U = L * UL
g = torch.Generator(device=dev).manual_seed(0)
At = torch.randn(U, N, device=dev, generator=g)
P = torch.randn(N, C, device=dev, generator=g) * 0.05
S = core.row_softmax(P, 10.0)
p = (0.998 - (torch.arange(0, K) / K * (0.998 - 0.97))).float().to(dev)
for _ in range(3):
    vals, idx = core.col_topk(At, K, neuron_major=True)
    pdge = core.wpmi_score(S, idx, p, 1e-7, True)
    sim = core.logsumexp_sub(pdge, 1.0, seg_offsets=[i * UL for i in range(L + 1)])
torch.cuda.synchronize()
