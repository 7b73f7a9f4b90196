"""The HIP core at the shape of BASELINE config 5 as ONE rank of 8 sees it (dev tool, not the bench):
200 000 probe images / 8 GPUs = 25 000 images per rank (argv[1] overrides), 10 000 concepts, 12 x 768 neurons,
soft-WPMI top_k = 100, bf16 MFMA similarity GEMM.  Synthetic activations / embeddings; per-stage HIP-event times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_dissect_amd  # noqa: F401
from mammo_clip_dissect_amd.pipeline import Dissector

N = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
C, D, L, UL, K = 10000, 512, 12, 768, 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
dis = Dissector(N, ["l%d" % i for i in range(L)], [UL] * L, C, D, dev, top_k=K, gemm_mode=mode)
dis.At.normal_(generator=g)
dis.E_img.normal_(generator=g)
dis.cursor = N
E_txt = torch.randn(C, D, device=dev, generator=g)
for it in range(3):
    marks = []
    def mark(name):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
    res = dis.finish(E_txt, marks=mark)
    torch.cuda.synchronize()
    ms = {b[0]: a[1].elapsed_time(b[1]) for a, b in zip(marks[:-1], marks[1:])}
print("N=%d C=%d U=%d K=%d gemm=%s" % (N, C, L * UL, K, mode))
tot = 0.0
for k, v in ms.items():
    print("  %-10s %8.3f ms" % (k, v)); tot += v
print("  core total %.3f ms -> %.2f M images/s" % (tot, N / tot / 1e3))
U = L * UL
flops = 2.0 * N * C * D
print("  gemm %.1f TFLOP/s; topk %.2f TB/s; wpmi %.2f TB/s (algorithmic)" % (
    flops / ms["gemm"] / 1e9, (4.0 * N * U + 8.0 * K * U) / ms["topk"] / 1e9,
    (4.0 * C * min(N, UL * K) * L + 4.0 * K * U + 4.0 * U * C) / ms["wpmi"] / 1e9))
assert torch.isfinite(res.sim).all()
