"""Randomised test of the stress chain's kernels against float64 on random shapes: K1s (E = bf16(exp(a(P-1))), rinv) and K4s
(log-sum over the K gathered rows, products of four arguments per log) -- tolerances as in tests/test_gpu_kernels.py (bf16
operands; v_log_f32).  argv: [cases] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mammo_clip_dissect_amd  # noqa: F401
from mammo_clip_dissect_amd import core

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
bad = 0
for c in range(cases):
    N = int(rng.choice([1, 2, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513, 700]))
    C = int(rng.choice([1, 3, 7, 8, 9, 127, 128, 129, 191, 255, 256, 257, 383, 385, 700, 1000]))
    D = int(rng.choice([8, 31, 64, 70, 128, 512, 520]))
    a = float(rng.choice([2.0, 10.0]))
    K = int(rng.integers(1, min(N, 130) + 1))
    U = int(rng.integers(1, 50))
    soft = bool(rng.integers(0, 2))
    tag = "N=%d C=%d D=%d K=%d U=%d soft=%d a=%g" % (N, C, D, K, U, soft, a)
    g = torch.Generator().manual_seed(seed * 100003 + c)
    I = core.normalize_rows(torch.randn(N, D, generator=g).to(dev))
    T = core.normalize_rows(torch.randn(C, D, generator=g).to(dev))
    E, rinv = core.embed_gemm_exp(I, T, a)
    ref = torch.exp(a * (I.double() @ T.double().t() - 1.0))
    rel = float((E.double() / ref - 1.0).abs().max())
    rs = float((rinv.double() * ref.sum(dim=1) - 1.0).abs().max())
    # (a row sum over a handful of concepts does not average the elements' errors: E's own bound then)
    if rel > a * 8e-3 + 2.0 ** -7 or rs > (a * 4e-3 + 1e-3 if C >= 16 else a * 8e-3 + 2.0 ** -7):
        bad += 1
        print("MISMATCH K1s %s: E rel %.3e rinv rel %.3e" % (tag, rel, rs), flush=True)
        continue
    idx = torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(U)]).int().to(dev)
    p = torch.linspace(0.998, 0.97, K).float().to(dev)
    out = core.wpmi_score_bf16(E, rinv, idx, p if soft else None, 1e-7, soft)
    S = E.double() * rinv.double()[:, None]
    gsel = S[idx.long()]
    w = 1.0 + p.double()[None, :, None] * (gsel - 1.0) + 1e-7 if soft else gsel + 1e-7
    want = torch.log(w).sum(dim=1)
    err = float((out.double() - want).abs().max())
    tol = 2e-5 * K * max(1.0, float(torch.log(w).abs().max())) + 1e-5
    if not err <= tol:
        bad += 1
        print("MISMATCH K4s %s: err %.3e tol %.3e" % (tag, err, tol), flush=True)
    if (c + 1) % 50 == 0:
        print("%d cases, %d mismatches" % (c + 1, bad), flush=True)
print("done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
