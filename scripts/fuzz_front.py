"""Randomised differential test of the front of the path against the CPU oracle, all bit-exact: K1a (row L2-normalise), K1
(fp32 MFMA GEMM in MKL's K-block order) on random (N, C, D) incl. unaligned leading dimensions, and K0 (hook pooling:
max exact, avg within 1e-6) on random [B, C, H, W].  argv: [cases] [seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import mammo_clip_dissect_amd  # noqa: F401
from mammo_clip_dissect_amd import core
import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
DS = [1, 2, 3, 4, 5, 8, 31, 32, 33, 64, 100, 127, 128, 129, 255, 256, 383, 384, 385, 388, 511, 512, 513, 520, 640, 766, 767, 768,
      769, 772, 1000, 1023, 1024, 1025, 1152, 1153, 1536, 2048, 2049]
bad = 0
for c in range(cases):
    N, C, D = int(rng.integers(1, 400)), int(rng.integers(1, 300)), int(rng.choice(DS))
    tag = "N=%d C=%d D=%d" % (N, C, D)
    x = (rng.standard_normal((N, D)) * rng.choice([1.0, 30.0, 1e-3])).astype(np.float32)
    y = rng.standard_normal((C, D)).astype(np.float32)
    pad = int(rng.integers(0, 4))                       # extra columns: an unaligned / padded leading dimension
    xg = torch.from_numpy(np.concatenate([x, np.ones((N, pad), np.float32)], axis=1)).to(dev)[:, :D]
    yg = torch.from_numpy(y).to(dev)
    In, Tn = core.normalize_rows(xg), core.normalize_rows(yg)
    Io, To = O.normalize_rows(x), O.normalize_rows(y)
    if not (np.array_equal(In.cpu().numpy(), Io) and np.array_equal(Tn.cpu().numpy(), To)):
        bad += 1
        print("MISMATCH K1a " + tag, flush=True)
        continue
    ref = np.empty((N, C), np.float32)
    O.lib().mcd_o_gemm_nt(O._f(Io), O._f(To), O._i64(N), O._i64(C), O._i64(D), O._f(ref))
    got = core.embed_gemm(In, Tn).cpu().numpy()
    if not np.array_equal(got, ref):
        bad += 1
        print("MISMATCH K1 %s pad=%d: max %.3e" % (tag, pad, np.abs(got - ref).max()), flush=True)
        continue
    # K0
    B, Ch = int(rng.integers(1, 40)), int(rng.choice([1, 3, 24, 40, 64, 128, 176, 304, 512, 77]))
    H, W = int(rng.choice([1, 2, 7, 14, 28, 9])), int(rng.choice([1, 2, 7, 14, 28, 5]))
    t = rng.standard_normal((B, Ch, H, W)).astype(np.float32)
    B = min(B, 40)
    for mode in ("avg", "max"):
        r0 = int(rng.integers(0, 64 - B + 1))
        At = torch.zeros((Ch + 5, 64), dtype=torch.float32, device=dev)
        n = core.hook_pool(torch.from_numpy(t).to(dev), mode, At, r0, 3, True)
        ref0 = O.hook_pool(t, mode)
        blk = At.cpu().numpy()
        want = np.zeros_like(blk)
        want[3:3 + Ch, r0:r0 + B] = ref0.T
        # max is exact; the mean over H*W is an fp32 sum in another order than the oracle's fp64 one (<= 1e-6 on unit-variance data)
        same = np.array_equal(blk, want) if mode == "max" else (
            float(np.abs(blk - want).max()) <= 1e-6 and np.array_equal(blk == 0, want == 0) or float(np.abs(blk - want).max()) <= 1e-6
            and np.array_equal(np.delete(blk, np.s_[3:3 + Ch], 0), np.delete(want, np.s_[3:3 + Ch], 0)))
        if n != Ch or not same:
            bad += 1
            print("MISMATCH K0 %s B=%d Ch=%d HW=%dx%d r0=%d" % (mode, B, Ch, H, W, r0), flush=True)
    if (c + 1) % 50 == 0:
        print("%d cases, %d mismatches" % (c + 1, bad), flush=True)
print("done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
