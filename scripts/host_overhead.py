"""What one wrapper call costs the HOST (us per call, GPU-box Python): the pieces -- device context, current stream, allocation, event record --
and whole core.* calls on kernels too short to hide them.  The scoring pass of one run is ~14 such calls: 0.73 ms of wall time for 0.5 ms of kernels
(bench.py `core_ms` against the kernel trace), most of the difference in front of K1, where the stream starts dry."""
import time, torch, ctypes, sys, os
sys.path.insert(0, os.getcwd())
import mammo_clip_dissect_amd
from mammo_clip_dissect_amd import core
dev = torch.device("cuda:0")
x = torch.randn(763, 512, device=dev)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
def ctx():
    with torch.cuda.device(dev): pass
print("with torch.cuda.device(dev)      %.2f us" % t(ctx))
print("torch.cuda.current_device()      %.2f us" % t(lambda: torch.cuda.current_device()))
print("torch.cuda.current_stream()      %.2f us" % t(lambda: torch.cuda.current_stream().cuda_stream))
print("torch.empty_like                 %.2f us" % t(lambda: torch.empty_like(x)))
print("x.data_ptr()                     %.2f us" % t(lambda: x.data_ptr()))
ev = torch.cuda.Event(enable_timing=True)
print("event.record()                   %.2f us" % t(lambda: ev.record()))
print("new Event + record               %.2f us" % t(lambda: torch.cuda.Event(enable_timing=True).record()))
out = torch.empty_like(x)
print("core.normalize_rows (host side)  %.2f us  (kernel ~3 us: host-bound loop)" % t(lambda: core.normalize_rows(x, out=out)))
I = torch.randn(100, 512, device=dev)
print("core.embed_gemm small (host)     %.2f us" % t(lambda: core.embed_gemm(I, x)))
