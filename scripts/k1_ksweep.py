import os, sys
sys.path.insert(0, os.getcwd())
import torch
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import core
dev = torch.device("cuda:0")
N, C = 10000, 763
REPS = int(os.environ.get("K1_REPS", "40"))      # launches per timing (back to back: the clock the chip holds depends on how long it has been busy)
for D in ([int(a) for a in sys.argv[1:]] or [128, 256, 384, 512, 768, 1024, 2048]):      # argv: reduction depths
    g = torch.Generator(device=dev).manual_seed(0)
    I = core.normalize_rows(torch.randn(N, D, device=dev, generator=g)); T = core.normalize_rows(torch.randn(C, D, device=dev, generator=g))
    for _ in range(5): core.embed_gemm(I, T)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS): core.embed_gemm(I, T)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / REPS
    print("D=%5d  %.4f ms  %.1f TFLOP/s  per 32-k tile %.2f us" % (D, ms, 2.0 * N * C * D / ms / 1e9, ms * 1e3 / (D / 32)), flush=True)
