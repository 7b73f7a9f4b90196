"""Per-kernel timing of the dissection core at BASELINE config-2 shape (dev tool, not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_dissect_amd as m
from mammo_clip_dissect_amd import core

dev = torch.device("cuda:0")
N, C, D, L, UL, K = 10000, 763, 512, 12, 768, 100
U = L * UL
g = torch.Generator(device=dev).manual_seed(0)
E_img = torch.randn(N, D, device=dev, generator=g)
E_txt = torch.randn(C, D, device=dev, generator=g)
At = torch.randn(U, N, device=dev, generator=g)      # neuron-major activations
A_im = At[:UL].t().contiguous()                      # one layer, image-major
p = (0.998 - (torch.arange(0, K) / K * (0.998 - 0.97))).float().to(dev)

def timeit(name, fn, n=10, bytes_=None, flops=None):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    extra = ""
    if bytes_: extra += "  %.2f TB/s" % (bytes_ / ms / 1e9)
    if flops: extra += "  %.1f TFLOP/s" % (flops / ms / 1e9)
    print("%-28s %9.3f ms%s" % (name, ms, extra), flush=True)
    return ms

I = core.normalize_rows(E_img); T = core.normalize_rows(E_txt)
timeit("K1a normalize_rows", lambda: core.normalize_rows(E_img), bytes_=8 * N * D)
P = core.embed_gemm(I, T)
timeit("K1 embed_gemm f32", lambda: core.embed_gemm(I, T), flops=2 * N * C * D, bytes_=4 * (N * D + C * D + N * C))
timeit("K1 embed_gemm bf16x3", lambda: core.embed_gemm(I, T, mode="bf16x3"), flops=2 * N * C * D)
timeit("K1 embed_gemm bf16", lambda: core.embed_gemm(I, T, mode="bf16"), flops=2 * N * C * D)
S = core.row_softmax(P, 10.0)
timeit("K2 row_softmax", lambda: core.row_softmax(P, 10.0), bytes_=8 * N * C)
vals, idx = core.col_topk(At, K, neuron_major=True)
timeit("K3 col_topk all layers (nm)", lambda: core.col_topk(At, K, neuron_major=True), bytes_=4 * N * U + 8 * K * U)
timeit("K3 col_topk 1 layer (im)", lambda: core.col_topk(A_im, K), bytes_=4 * N * UL)
timeit("transpose 1 layer", lambda: core.transpose(A_im), bytes_=8 * N * UL)
pdge = core.wpmi_score(S, idx, p, 1e-7, True)
by4 = 4 * C * min(N, UL * K) * L + 8 * K * U + 4 * U * C
timeit("K4 wpmi_score all layers", lambda: core.wpmi_score(S, idx, p, 1e-7, True), bytes_=by4)
timeit("K4 wpmi_score 1 layer", lambda: core.wpmi_score(S, idx[:UL], p, 1e-7, True))
timeit("K4 all layers, S_IS_PROB", lambda: core.wpmi_score(S, idx, p, 1e-7, True, s_is_prob=True), bytes_=by4)
timeit("K4 all layers, FAST_LOG", lambda: core.wpmi_score(S, idx, p, 1e-7, True, fast_log=True), bytes_=by4)
a = core.wpmi_score(S, idx, p, 1e-7, True); b = core.wpmi_score(S, idx, p, 1e-7, True, s_is_prob=True)
print("trusted == checked:", bool(torch.equal(a, b)), flush=True)
segs = [i * UL for i in range(L + 1)]
sim = core.logsumexp_sub(pdge, 1.0, seg_offsets=segs)
timeit("K5 logsumexp_sub 12 seg", lambda: core.logsumexp_sub(pdge, 1.0, seg_offsets=segs), bytes_=8 * U * C)
timeit("K6 row_topk k=10", lambda: core.row_topk(sim, 10), bytes_=4 * U * C)
timeit("K6 row_topk k=1", lambda: core.row_topk(sim, 1), bytes_=4 * U * C)
x = torch.randn(256, 197, 768, device=dev)
dst = torch.empty(U, N, device=dev)
timeit("K0 hook_pool CLS B=256", lambda: core.hook_pool(x, "avg", dst, 0, 0, True))
x4 = torch.randn(256, 176, 14, 14, device=dev)
timeit("K0 hook_pool avg 176x14x14", lambda: core.hook_pool(x4, "avg", dst, 0, 0, True), bytes_=x4.numel() * 4)

# the whole scoring side: eager launches vs one hipGraph replay
from mammo_clip_dissect_amd.pipeline import Dissector
dis = Dissector(N, ["l%d" % i for i in range(L)], [UL] * L, C, D, dev, top_k=K)
dis.At[:, :N] = At
dis.E_img[:] = E_img
dis.cursor = N
timeit("core, eager (12 launches)", lambda: dis.finish(E_txt))
timeit("core, one hipGraph replay", lambda: dis.finish_graphed(E_txt))
