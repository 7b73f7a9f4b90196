"""K1 (gemm_nt_f32_dma_kernel, fp32 MFMA, parity mode) at 10 000 x 763 x 512: where a launch's time goes, from s_memrealtime stamps of
EVERY workgroup (dev build, MCD_GEMM_K1_STAMPS=1; 100 MHz: 10 ns a tick): entry, first K-tile landed, second K-tile landed, K loop done,
stores issued, stores written.  argv: [D]"""
import ctypes, os, sys
os.environ["MCD_GEMM_K1_STAMPS"] = "1"
os.environ.setdefault("MCD_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mammo-clip-dissect_amd", "csrc", "libmcd_hip_dev.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mammo_clip_dissect_amd  # noqa: F401
from mammo_clip_dissect_amd import core, _lib
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N, C = 10000, 763
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
I = core.normalize_rows(torch.randn(N, D, device=dev, generator=g)); T = core.normalize_rows(torch.randn(C, D, device=dev, generator=g))
out = torch.empty(N, C, device=dev)
L = _lib.load()
L.mcd_dev_k1_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
nwg = ((N + 127) // 128 + 7) // 8 * 8 * ((C + 127) // 128)
for rep in range(12):
    core.embed_gemm(I, T, mode="f32", out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for rep in range(20):
    core.embed_gemm(I, T, mode="f32", out=out)
e.record(); torch.cuda.synchronize()
print("D=%d: %.4f ms per launch (20 back to back, stamps on)" % (D, s.elapsed_time(e) / 20))
buf = np.zeros((nwg, 32), dtype=np.uint64)
assert L.mcd_dev_k1_stamps(buf.ctypes.data, nwg) == 0
st = buf[:, :6].astype(np.int64)
live = st[:, 3] > 0
kt = buf[live, 8:8 + D // 32].astype(np.int64)
st = st[live]; hw = buf[live, 7]
t0 = st[:, 0].min()
us = (st - t0) / 100.0
names = ["entry", "tile 0 landed", "tile 1 landed", "K loop done", "stores issued", "stores written"]
print("%d workgroups; microseconds from the first workgroup's entry: min / median / max" % len(st))
for i, n in enumerate(names):
    print("  %-16s %7.2f %7.2f %7.2f" % (n, us[:, i].min(), np.median(us[:, i]), us[:, i].max()))
d = np.diff(us, axis=1)
for i, n in enumerate(["entry -> tile 0 landed", "tile 0 -> tile 1 landed", "tile 1 landed -> K loop done", "K loop done -> stores issued", "stores issued -> written"]):
    print("  %-30s min %6.2f  median %6.2f  max %6.2f" % (n, d[:, i].min(), np.median(d[:, i]), d[:, i].max()))
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = ((se.astype(np.int64) * 2 + sh) * 16 + cu)
xcd = np.arange(len(buf))[live] % 8
slots = {}
for k, x in zip(key, xcd):
    slots[(x, k)] = slots.get((x, k), 0) + 1
cnt = np.bincount(np.array(list(slots.values())))
print("workgroups per (XCD, CU id):", {i: int(c) for i, c in enumerate(cnt) if c})

# per-K-tile pace: time between consecutive K-tile starts, for the workgroups that share a CU (pairs) and those alone
nper = {}
for k, x in zip(key, xcd):
    nper[(x, k)] = nper.get((x, k), 0) + 1
pair = np.array([nper[(x, k)] == 2 for k, x in zip(key, xcd)])
dk = np.diff(kt, axis=1) / 100.0
print("K-tile period (us) by K-tile index, median over workgroups: pairs | alone")
print("  pairs:", " ".join("%.2f" % v for v in np.median(dk[pair], axis=0)))
print("  alone:", " ".join("%.2f" % v for v in np.median(dk[~pair], axis=0)))
print("  pairs, slower half:", " ".join("%.2f" % v for v in np.percentile(dk[pair], 90, axis=0)))

clk = (buf[live, 31].astype(np.int64) - buf[live, 6].astype(np.int64)) / ((st[:, 3] - st[:, 0]) / 100.0) / 1e3
print("shader clock over the K loop (s_memtime / s_memrealtime), GHz: min %.2f median %.2f max %.2f" % (clk.min(), np.median(clk), clk.max()))
