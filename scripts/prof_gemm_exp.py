"""K1s (bf16 MFMA GEMM with the exp epilogue) at one rank's share of the stress configuration, piece by piece (HIP
events), for rocprofv3 --kernel-trace / --pmc passes.  argv: [N] [C] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_dissect_amd  # noqa: F401
from mammo_clip_dissect_amd import core

N = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
D = 512
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
E_img = torch.randn(N, D, device=dev, generator=g)
E_txt = torch.randn(C, D, device=dev, generator=g)


def timed(fn, n=reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


I = core.normalize_rows(E_img)
T = core.normalize_rows(E_txt)
t_norm = timed(lambda: (core.normalize_rows(E_img), core.normalize_rows(E_txt)))
t_exp = timed(lambda: core.embed_gemm_exp(I, T, 10.0))
out = torch.empty(N, C, device=dev)
t_p32 = timed(lambda: core.embed_gemm(I, T, mode="bf16", out=out))
fl = 2.0 * N * C * D
print("N=%d C=%d D=%d" % (N, C, D))
print("  normalize I + T                 %8.3f ms" % t_norm)
print("  embed_gemm_exp (conv+gemm+rsum) %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 2.5 PF)" % (t_exp, fl / t_exp / 1e9, fl / t_exp / 1e9 / 25))
print("  embed_gemm bf16 -> fp32 P       %8.3f ms  %7.1f TFLOP/s" % (t_p32, fl / t_p32 / 1e9))
if os.environ.get("MCD_PROF_LIBRARY", "1") != "0":
    # yardstick: the vendor library's plain bf16 GEMM of the same shape (bf16 in, bf16 out, no exp, no row sums, no
    # conversion) through torch.matmul = hipBLASLt / rocBLAS, both operand orders
    Ib, Tb = I.to(torch.bfloat16), T.to(torch.bfloat16)
    o1 = torch.empty(N, C, device=dev, dtype=torch.bfloat16)
    o2 = torch.empty(C, N, device=dev, dtype=torch.bfloat16)
    t_l1 = timed(lambda: torch.matmul(Ib, Tb.t(), out=o1))
    t_l2 = timed(lambda: torch.matmul(Tb, Ib.t(), out=o2))
    print("  library bf16 GEMM, [N,C] out    %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 2.5 PF)" % (t_l1, fl / t_l1 / 1e9, fl / t_l1 / 1e9 / 25))
    print("  library bf16 GEMM, [C,N] out    %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 2.5 PF)" % (t_l2, fl / t_l2 / 1e9, fl / t_l2 / 1e9 / 25))
    # and at a K where the epilogue no longer weighs: same output, 8 x the reduction depth
    Ik, Tk = torch.randn(N, 4096, device=dev, dtype=torch.bfloat16), torch.randn(C, 4096, device=dev, dtype=torch.bfloat16)
    t_l3 = timed(lambda: torch.matmul(Ik, Tk.t(), out=o1))
    print("  library bf16 GEMM, K = 4096     %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 2.5 PF)" % (t_l3, 8 * fl / t_l3 / 1e9, 8 * fl / t_l3 / 1e9 / 25))
