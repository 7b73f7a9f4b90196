"""Which SDPA backend is fastest for the ViT-B/16 attention shape in fp32 (dev tool)."""
import torch, torch.nn.functional as F
from torch.nn.attention import sdpa_kernel, SDPBackend
dev = torch.device("cuda:0")
q, k, v = (torch.randn(250, 12, 197, 64, device=dev) for _ in range(3))
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
print("default          %.3f ms" % t(lambda: F.scaled_dot_product_attention(q, k, v)))
for name, b in (("flash", SDPBackend.FLASH_ATTENTION), ("mem_efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
    try:
        with sdpa_kernel(b):
            print("%-16s %.3f ms" % (name, t(lambda: F.scaled_dot_product_attention(q, k, v))))
    except Exception as ex:
        print(name, "unavailable:", str(ex)[:80])
def manual():
    a = torch.matmul(q, k.transpose(-1, -2)) * 0.125
    return torch.matmul(torch.softmax(a, dim=-1), v)
print("manual bmm       %.3f ms" % t(manual))
