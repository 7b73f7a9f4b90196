"""Randomised differential test of the scoring chain K2 -> K3 -> K4 -> K5 -> K6 against the CPU oracle on random shapes:
concept counts around the 32-column groups that switch ATen's summation order, ragged neuron counts, every K, soft / hard
WPMI, padded and unpadded S.  S, the image indices and the top concepts must be exact; soft-WPMI sums may differ by one
rounding of a log in a few entries per ten thousand.  argv: [cases] [seed]"""
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
CS = list(range(1, 40)) + [60, 63, 64, 65, 95, 96, 97, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300, 511, 512, 513,
                           736, 737, 762, 763, 764, 767, 768, 769, 1000, 1023, 1024, 1025, 1500]
KS = [1, 2, 3, 4, 5, 8, 15, 16, 17, 28, 31, 32, 33, 48, 63, 64, 65, 100, 127, 128]
bad = 0
worst = 0.0
for c in range(cases):
    K = int(rng.choice(KS))
    N = int(rng.integers(K, max(K + 1, 700)))
    C = int(rng.choice(CS))
    U = int(rng.integers(1, 70))
    soft = bool(rng.integers(0, 2))
    a = float(rng.choice([10.0, 2.0]))
    P = (rng.standard_normal((N, C)) * 0.05).astype(np.float32)
    A = rng.standard_normal((N, U)).astype(np.float32)
    tag = "N=%d C=%d U=%d K=%d soft=%d a=%g" % (N, C, U, K, soft, a)
    Sg = core.row_softmax(torch.from_numpy(P).to(dev), a, pad_to=(192 if rng.random() < 0.7 else None))
    So = O.row_softmax(P, a)
    if not np.array_equal(Sg.cpu().numpy(), So):
        bad += 1
        print("MISMATCH K2 " + tag, flush=True)
        continue
    vg, ig = core.col_topk(torch.from_numpy(A).to(dev), K)
    vo, io = O.col_topk(A, K)
    if not np.array_equal(ig.cpu().numpy().T, io):
        bad += 1
        print("MISMATCH K3 " + tag, flush=True)
        continue
    p = O.p_in_examples(K) if soft else None
    pg = torch.from_numpy(p).to(dev) if soft else None
    trust = bool(rng.integers(0, 2))
    dg = core.wpmi_score(Sg, ig, pg, 1e-7, soft, s_is_prob=trust)
    do = O.wpmi_score(So, io, p, np.float32(1e-7), 1 if soft else 0)
    d = np.abs(dg.cpu().numpy().astype(np.float64) - do.astype(np.float64))
    tol = 1.3e-4 * max(1.0, float(np.abs(do).max()) / 512.0)
    exact = float((d == 0).mean())
    worst = max(worst, float(d.max()))
    if d.max() > tol or exact < 0.995:
        bad += 1
        print("MISMATCH K4 %s trust=%d: max %.3e exact %.5f" % (tag, trust, d.max(), exact), flush=True)
        continue
    lam = 1.0 if soft else 0.6
    sg = core.logsumexp_sub(dg, np.float32(lam).item())
    so = O.logsumexp_sub(dg.cpu().numpy(), np.float32(lam))
    d5 = np.abs(sg.cpu().numpy().astype(np.float64) - so.astype(np.float64))
    cols_off = int((d5.max(axis=0) > 0).sum())        # one ulp of a column's prob_d moves that whole column by one ulp
    if d5.max() > tol or cols_off > max(1, C // 200):
        bad += 1
        print("MISMATCH K5 %s: max %.3e, %d columns differ" % (tag, d5.max(), cols_off), flush=True)
        continue
    k = min(10, C)
    v6, i6 = core.row_topk(sg, k)
    vo6, io6 = O.row_topk(sg.cpu().numpy(), k)
    if not (np.array_equal(i6.cpu().numpy(), io6) and np.array_equal(v6.cpu().numpy(), vo6)):
        bad += 1
        print("MISMATCH K6 " + tag, flush=True)
    if (c + 1) % 50 == 0:
        print("%d cases, %d mismatches, worst K4 deviation %.2e" % (c + 1, bad, worst), flush=True)
print("done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
