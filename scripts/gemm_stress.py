"""K1 at the stress shape of BASELINE config 5 (scaled to one launch that fits comfortably): N x 10000 x 512."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import core
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for (N, C, D) in [(50000, 10000, 512), (10000, 763, 512)]:
    I = core.normalize_rows(torch.randn(N, D, device=dev, generator=g))
    T = core.normalize_rows(torch.randn(C, D, device=dev, generator=g))
    out = torch.empty(N, C, device=dev)
    for mode in ("f32", "bf16x3", "bf16"):
        for _ in range(2): core.embed_gemm(I, T, mode=mode, out=out)
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): core.embed_gemm(I, T, mode=mode, out=out)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        print("N=%d C=%d D=%d %-7s %8.3f ms  %7.1f TFLOP/s (2NCD)  out-write %.2f TB/s" % (N, C, D, mode, ms, 2.0 * N * C * D / ms / 1e9, 4.0 * N * C / ms / 1e9), flush=True)
