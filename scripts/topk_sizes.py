"""K3 (top-100 images per neuron) across probe-set sizes (dev tool): time and read rate per size class."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import core
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for N in (256, 1000, 4096, 6250, 10000, 12000, 16384, 20000, 25000, 30000, 40000, 50000, 65536, 100000):
    U = max(256, min(9216, int(2.0e8 // N)))
    At = torch.randn(U, N, device=dev, generator=g)
    K = min(100, N)
    for _ in range(2): core.col_topk(At, K, neuron_major=True)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): core.col_topk(At, K, neuron_major=True)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print("N=%6d U=%5d  %8.3f ms  %5.2f TB/s" % (N, U, ms, 4.0 * N * U / ms / 1e9), flush=True)
