"""K9 against PyTorch's SDPA at the headline shape (250 images x 12 heads x 197 tokens, fp32).  Dev tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import mammo_clip_dissect_amd as m
from mammo_clip_dissect_amd import core

dev = torch.device("cuda:0")
B, T, H = (int(os.environ.get("MCD_ATTN_B", "250")), 197, 12)
qkv = torch.randn(B, T, 3 * H * 64, device=dev)
flops = 4.0 * B * H * T * T * 64


def timeit(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print("%-28s %8.3f ms  %6.1f TFLOP/s" % (name, ms, flops / ms / 1e9), flush=True)


def sdpa():
    q, k, v = qkv.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, H * 64)


if "k9only" in sys.argv:      # for rocprofv3 --pmc passes
    for _ in range(4): core.vit_attention(qkv, H)
    torch.cuda.synchronize()
    sys.exit(0)
timeit("torch SDPA (+ reshape)", sdpa)
timeit("K9 mcd_vit_attention", lambda: core.vit_attention(qkv, H))
print("max |diff|", (sdpa() - core.vit_attention(qkv, H)).abs().max().item())

# K10 LayerNorm at the headline shape
x = torch.randn(250 * 197, 768, device=dev)
w = torch.randn(768, device=dev); b = torch.randn(768, device=dev)
def timeit_b(name, fn, nbytes, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print("%-28s %8.3f ms  %6.2f TB/s" % (name, ms, nbytes / ms / 1e9), flush=True)
timeit_b("torch layer_norm", lambda: F.layer_norm(x, (768,), w, b, 1e-12), 2 * x.numel() * 4)
timeit_b("K10 mcd_layer_norm", lambda: core.layer_norm(x, w, b, 1e-12), 2 * x.numel() * 4)
