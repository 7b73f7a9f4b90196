"""Runs K4s (wpmi_score_bf16) at one rank's share of configs[4] a few times, for rocprofv3 kernel-trace / --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_dissect_amd as m  # noqa: F401
from mammo_clip_dissect_amd import core
dev = torch.device("cuda:0")
N, C, D, U, K = 25000, 10000, 512, 9216, 100
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = torch.Generator(device=dev).manual_seed(0)
I = torch.randn(N, D, device=dev, generator=g)
T = torch.randn(C, D, device=dev, generator=g)
E, rinv = core.embed_gemm_exp(I, T, 10.0, normalize=True)
idx = torch.stack([torch.randperm(N, device=dev, generator=g)[:K] for _ in range(64)]).int()
idx = idx.repeat(U // 64, 1)
idx = ((idx + torch.arange(U, device=dev).unsqueeze(1) * 131) % N).int().contiguous()   # every neuron its own rows
fold = int(os.environ.get("K4S_FOLD_ROWS", "0"))     # experiment: gather from the first `fold` rows only (smaller L2 working set)
if fold:
    idx = (idx % fold).int().contiguous()
p = (0.998 - (torch.arange(0, K) / K * (0.998 - 0.97))).float().to(dev)
out = core.wpmi_score_bf16(E, rinv, idx, p, 1e-7, True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for i in range(reps):
    out = core.wpmi_score_bf16(E, rinv, idx, p, 1e-7, True)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
print("K4s %s: median %.4f ms  min %.4f   out[0,0] = %r" % (os.environ.get("MCD_LIB_PATH", "product"), ms[len(ms) // 2], ms[0], float(out[0, 0])))
