"""hipBLASLt via libmcd_blaslt.so (best of 32 heuristic candidates) against PyTorch's F.linear (+ TunableOp picks when
enabled) at the four GEMM shapes of a ViT-B block, 250 images.  Dev tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import mammo_clip_dissect_amd as m
from mammo_clip_dissect_amd import core, tuning

if "tunable" in sys.argv:
    print("tunableop:", tuning.enable_gemm_tuning())
dev = torch.device("cuda:0")
M = int(os.environ.get("MCD_ATTN_B", "250")) * 197


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for name, N, K, res in [("qkv", 2304, 768, False), ("proj", 768, 768, True), ("fc1", 3072, 768, False), ("fc2", 768, 3072, True)]:
    h = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev)
    t_torch = timeit((lambda: r + F.linear(h, W, b)) if res else (lambda: F.linear(h, W, b)))
    t_mine = timeit(lambda: core.linear_residual(r, h, W, b, out=out))
    fl = 2.0 * M * N * K
    print("%-5s N=%4d K=%4d  torch %s %.3f ms (%.0f TF)   libmcd_blaslt %.3f ms (%.0f TF)" % (
        name, N, K, "linear+add" if res else "linear    ", t_torch, fl / t_torch / 1e9, t_mine, fl / t_mine / 1e9), flush=True)
