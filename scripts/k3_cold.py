"""K3 at 9 216 x 10 000: warm (the same matrix again and again) against cold (2 GB written between the calls) reads, and the driver's
row pitch (10 048) against a dense matrix.  Kernel times come from rocprofv3 around this script (scripts/k3_cold.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_dissect_amd  # noqa: F401
from mammo_clip_dissect_amd import core

dev = torch.device("cuda:0")
mode = sys.argv[1]
N, U = 10000, 9216
g = torch.Generator(device=dev).manual_seed(0)
if "pitch" in mode:
    buf = torch.zeros(U, 10048, device=dev)
    At = buf[:, :N]
    At.copy_(torch.randn(U, N, device=dev, generator=g))
else:
    At = torch.randn(U, N, device=dev, generator=g)
flush = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev) if "cold" in mode else None
for _ in range(8):
    if flush is not None:
        flush.fill_(1.0)
    vals, idx = core.col_topk(At, 100, neuron_major=True)
torch.cuda.synchronize()
