"""A CPU stand-in for mammo_clip_dissect_amd.core built on the ORACLE, for tests only.

The product's Dissector takes its compute backend as a parameter (`ops`); the default -- and the only one
the package ships -- is the HIP library.  The multi-rank host logic (sharding, index offsets, the three
all-gathers, the candidate merge, the neuron split) has no arithmetic of its own, so the tests run it on CPU
under gloo with this oracle-backed backend and require bit-identical results for 1 and 2 ranks."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle as O  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy()


def normalize_rows(x, out=None):
    y = torch.from_numpy(O.normalize_rows(_np(x)))
    if out is not None:
        out.copy_(y)
        return out
    return y


def embed_gemm(I, T, mode="f32", out=None):
    P = np.empty((I.shape[0], T.shape[0]), np.float32)
    O.lib().mcd_o_gemm_nt(O._f(np.ascontiguousarray(_np(I))), O._f(np.ascontiguousarray(_np(T))),
                          O._i64(I.shape[0]), O._i64(T.shape[0]), O._i64(I.shape[1]), O._f(P))   # k-ordered: deterministic
    return torch.from_numpy(P)


def row_softmax(P, a, pad_to=192):
    S = O.row_softmax(_np(P), float(a))
    C = S.shape[1]
    ld = (C + pad_to - 1) // pad_to * pad_to
    buf = torch.zeros(S.shape[0], ld)
    buf[:, :C] = torch.from_numpy(S)
    return buf[:, :C]


def col_topk(A, K, neuron_major=False, want_vals=True):
    a = _np(A).T if neuron_major else _np(A)
    v, i = O.col_topk(np.ascontiguousarray(a), int(K))
    return (torch.from_numpy(v.T.copy()) if want_vals else None), torch.from_numpy(i.T.astype(np.int32).copy())


def wpmi_score(S, idx, p, min_prob, soft, split=-1, out=None, fast_log=False, s_is_prob=False):
    r = O.wpmi_score(np.ascontiguousarray(_np(S)), _np(idx).T.astype(np.int64).copy(), _np(p) if soft else None,
                     np.float32(min_prob), 1 if soft else 0, split)
    r = torch.from_numpy(r)
    if out is not None:
        out.copy_(r)
        return out
    return r


def logsumexp_sub(pdge, lam, seg_offsets=None, split=-1, out=None):
    x = np.ascontiguousarray(_np(pdge))
    segs = seg_offsets or [0, x.shape[0]]
    r = np.concatenate([O.logsumexp_sub(x[a:b], float(lam), split) for a, b in zip(segs[:-1], segs[1:])])
    return torch.from_numpy(r)


def row_topk(sim, k):
    v, i = O.row_topk(np.ascontiguousarray(_np(sim)), int(k))
    return torch.from_numpy(v), torch.from_numpy(i.astype(np.int32))


def hook_pool(x, mode, dst, row0, col0, neuron_major):
    r = torch.from_numpy(O.hook_pool(_np(x), mode))
    B, W = r.shape
    if neuron_major:
        dst[col0:col0 + W, row0:row0 + B] = r.t()
    else:
        dst[row0:row0 + B, col0:col0 + W] = r
    return W
