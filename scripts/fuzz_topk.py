"""Randomised differential test of K3 (col_topk) and K6 (row_topk) against a stable reference order (NaN first, value
descending, lower index first): shapes around every size-class boundary, heavy ties, ReLU zeros, NaN / +-inf, constants.
argv: [cases] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mammo_clip_dissect_amd  # noqa: F401
from mammo_clip_dissect_amd import core

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
EDGES = [1, 2, 63, 64, 65, 255, 256, 257, 1024, 1025, 2048, 2049, 4095, 4096, 4097, 6144, 6145, 8192, 8193, 10240, 10241,
         12288, 12289, 16384, 16385, 20480, 20481, 26624, 26625, 32768, 32769, 40000, 65536, 65537, 70001]


def make(rows, n, mode):
    a = rng.standard_normal((rows, n)).astype(np.float32)
    if mode == 1:
        a = np.round(a, 1) + 0.0                       # heavy ties (no -0.0)
    elif mode == 2:
        a = np.maximum(a, 0.0)                         # ReLU: half the entries tie at +0
    elif mode == 3:
        a[:] = rng.standard_normal((rows, 1)).astype(np.float32)    # constant rows
    elif mode == 4:
        a = np.round(a, 0) + 0.0
        a[rng.random(a.shape) < 0.01] = np.inf
        a[rng.random(a.shape) < 0.01] = -np.inf
    if mode in (0, 1, 4) and rng.random() < 0.5 and n > 3:
        for r in range(rows):
            a[r, rng.integers(0, n, size=rng.integers(1, 4))] = np.nan
    return a


def ref_order(row, k):
    nan = np.isnan(row)
    key = np.where(nan, np.inf, row)
    order = np.lexsort((np.arange(row.size), -key, ~nan))   # NaN first, value descending, index ascending
    return order[:k]


bad = 0
for c in range(cases):
    n = int(rng.choice(EDGES)) + int(rng.integers(-1, 2)) if rng.random() < 0.7 else int(rng.integers(1, 40000))
    n = max(1, n)
    rows = int(rng.integers(1, 6))
    mode = int(rng.integers(0, 5))
    a = make(rows, n, mode)
    which = int(rng.integers(0, 3))
    if which == 2:
        k = int(min(n, rng.choice([1, 3, 10, 16])))
        v, i = core.row_topk(torch.from_numpy(a).to(dev), k)
        got_i, got_v, name = i.cpu().numpy(), v.cpu().numpy(), "row_topk"
    else:
        k = int(min(n, rng.choice([1, 5, 28, 100, 128, 129, 300, 1024, 2500]))) if n > 1 else 1
        t = torch.from_numpy(a).to(dev)
        if which == 0:
            v, i = core.col_topk(t, k, neuron_major=True)
        else:
            v, i = core.col_topk(t.t().contiguous(), k)
        got_i, got_v, name = i.cpu().numpy(), v.cpu().numpy(), "col_topk nm" if which == 0 else "col_topk im"
    for r in range(rows):
        o = ref_order(a[r], k)
        ok = np.array_equal(got_i[r], o) and np.array_equal(np.isnan(got_v[r]), np.isnan(a[r][o])) \
            and np.array_equal(got_v[r][~np.isnan(got_v[r])], a[r][o][~np.isnan(a[r][o])])
        if not ok:
            bad += 1
            print("MISMATCH %s n=%d k=%d mode=%d row=%d: got %s want %s" % (name, n, k, mode, r, got_i[r][:8], o[:8]), flush=True)
            break
    if (c + 1) % 50 == 0:
        print("%d cases, %d mismatches" % (c + 1, bad), flush=True)
print("done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
