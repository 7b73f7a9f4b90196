"""Runs K4 (soft-WPMI sums, config-2 shape) a few times for rocprofv3 --pmc passes.  argv[1]: checked|trusted|fast"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_dissect_amd as m
from mammo_clip_dissect_amd import core
mode = sys.argv[1] if len(sys.argv) > 1 else "trusted"
dev = torch.device("cuda:0")
N, C, L, UL, K = 10000, 763, 12, 768, 100
U = L * UL
g = torch.Generator(device=dev).manual_seed(0)
At = torch.randn(U, N, device=dev, generator=g)
P = torch.randn(N, C, device=dev, generator=g) * 0.0442      # unit-vector dot products in 512 dimensions
S = core.row_softmax(P, 10.0)
p = (0.998 - (torch.arange(0, K) / K * (0.998 - 0.97))).float().to(dev)
vals, idx = core.col_topk(At, K, neuron_major=True)
for _ in range(4):
    pdge = core.wpmi_score(S, idx, p, 1e-7, True, s_is_prob=(mode == "trusted"), fast_log=(mode == "fast"))
torch.cuda.synchronize()
