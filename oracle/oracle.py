"""CPU oracle for the Mammo-CLIP-Dissect dissection core (numpy + oracle/mcd_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  Each function cites the reference
lines it restates (paths relative to /root/reference).  The reference has no tests or
golden vectors of its own; the pin is tests/golden/*.npz, generated from the
reference's similarity.py by tests/golden/make_golden.py and checked against this
module by tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32 = np.float32
_i64 = ctypes.c_int64
_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int64)


def build():
    """Compile libmcd_oracle.so (gcc) if missing or stale."""
    so = os.path.join(_HERE, "libmcd_oracle.so")
    src = os.path.join(_HERE, "mcd_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libmcd_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.mcd_o_num_threads.restype = ctypes.c_int
        L.mcd_o_sum_split.restype = ctypes.c_int
        L.mcd_o_col_topk.restype = ctypes.c_int
        L.mcd_o_row_topk.restype = ctypes.c_int
        _LIB = L
    return _LIB


def num_threads():
    return int(lib().mcd_o_num_threads())


def set_num_threads(n):
    lib().mcd_o_set_num_threads(ctypes.c_int(int(n)))


def _f(a):
    return a.ctypes.data_as(_fp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _c(a, dtype=f32):
    return np.ascontiguousarray(a, dtype=dtype)


def sum_split(C):
    """Column below which torch.sum(dim=0) uses the cascade order (SumKernel.cpp)."""
    return int(lib().mcd_o_sum_split(ctypes.c_int(int(C))))


# ----------------------------------------------------------------------------------------------
# utils.py:570-594  get_similarity_from_activations: normalise rows, P = I @ T.T
# ----------------------------------------------------------------------------------------------
def normalize_rows(x):
    x = _c(x).copy()
    lib().mcd_o_normalize_rows(_f(x), _i64(x.shape[0]), _i64(x.shape[1]))
    return x


def embed_gemm(E_img, E_txt, blas=True):
    """clip_feats = image_features @ text_features.T on row-normalised inputs (utils.py:577-594).

    blas=False is the restatement in C of what torch's CPU matmul (MKL sgemm) computes on this image: fma chains over
    MKL's K-blocks, added in order (mcd_oracle.c: mcd_o_gemm_nt) -- bit-identical to the reference's P;
    blas=True uses numpy's own BLAS (another accumulation order, <= 2.5e-7 away)."""
    I = normalize_rows(E_img)
    T = normalize_rows(E_txt)
    if blas:
        return np.matmul(I, T.T).astype(f32, copy=False)
    P = np.empty((I.shape[0], T.shape[0]), f32)
    lib().mcd_o_gemm_nt(_f(I), _f(T), _i64(I.shape[0]), _i64(T.shape[0]), _i64(I.shape[1]), _f(P))
    return P


# ----------------------------------------------------------------------------------------------
# similarity.py:54  softmax(a*clip_feats, dim=1)
# ----------------------------------------------------------------------------------------------
def row_softmax(P, a):
    P = _c(P)
    S = np.empty_like(P)
    lib().mcd_o_row_softmax(_f(P), _i64(P.shape[0]), _i64(P.shape[1]), ctypes.c_float(a), _f(S))
    return S


# ----------------------------------------------------------------------------------------------
# similarity.py:55  torch.topk(target_feats, dim=0, k) -> (values [K,U], indices [K,U])
# ----------------------------------------------------------------------------------------------
def col_topk(A, K):
    A = _c(A)
    N, U = A.shape
    vals = np.empty((K, U), f32)
    idx = np.empty((K, U), np.int64)
    rc = lib().mcd_o_col_topk(_f(A), _i64(N), _i64(U), _i64(U), _i64(K), _f(vals), _i(idx))
    if rc != 0:
        raise RuntimeError("selected index k out of range")  # torch.topk's message
    return vals, idx


def row_topk(sim, k):
    """torch.topk(sim, k, dim=1) / torch.max(sim, dim=1) for k=1 (describe_*_neurons.py)."""
    sim = _c(sim)
    U, C = sim.shape
    vals = np.empty((U, k), f32)
    idx = np.empty((U, k), np.int64)
    rc = lib().mcd_o_row_topk(_f(sim), _i64(U), _i64(C), _i64(C), _i64(k), _f(vals), _i(idx))
    if rc != 0:
        raise RuntimeError("selected index k out of range")
    return vals, idx


# ----------------------------------------------------------------------------------------------
# similarity.py:58  p_in_examples = p_start-(arange(0,top_k)/top_k*(p_start-p_end))
# ----------------------------------------------------------------------------------------------
def p_in_examples(top_k, p_start=0.998, p_end=0.97):
    j = np.arange(top_k, dtype=np.int64).astype(f32) / f32(top_k)
    return (f32(p_start) - j * f32(p_start - p_end)).astype(f32)


def wpmi_score(S, idx, p, min_prob, soft, split=-1):
    """pdge[u,c] = sum_j log(term(S[idx[j,u], c]))  (similarity.py:59-65 / :84-88)."""
    S = _c(S)
    idx = _c(idx, np.int64)
    K, U = idx.shape
    C = S.shape[1]
    p = _c(p) if p is not None else np.zeros(K, f32)
    out = np.empty((U, C), f32)
    lib().mcd_o_wpmi_score(_f(S), _i64(S.shape[1]), _i(idx), _f(p), _i64(C), _i64(U), _i64(K),
                           ctypes.c_float(min_prob), ctypes.c_int(int(soft)), ctypes.c_int(split), _f(out))
    return out


def logsumexp_sub(pdge, lam, split=-1):
    """pdge - lam*(logsumexp(pdge, 0) - log(U))  (similarity.py:70-72)."""
    pdge = _c(pdge)
    U, C = pdge.shape
    out = np.empty_like(pdge)
    lib().mcd_o_logsumexp_sub(_f(pdge), _i64(U), _i64(C), ctypes.c_float(lam), ctypes.c_int(split), _f(out))
    return out


# ----------------------------------------------------------------------------------------------
# The five similarity functions, same signatures as concept_vit/similarity.py (minus `device`)
# ----------------------------------------------------------------------------------------------
def soft_wpmi(clip_feats, target_feats, top_k=100, a=10, lam=1, min_prob=1e-7, p_start=0.998, p_end=0.97,
              return_parts=False):
    """similarity.py:49-73."""
    S = row_softmax(clip_feats, float(a))
    _, idx = col_topk(target_feats, top_k)
    p = p_in_examples(top_k, p_start, p_end)
    pdge = wpmi_score(S, idx, p, f32(min_prob), soft=1)
    out = logsumexp_sub(pdge, float(lam))
    if return_parts:
        return out, dict(S=S, idx=idx, p=p, pdge=pdge)
    return out


def wpmi(clip_feats, target_feats, top_k=28, a=2, lam=0.6, min_prob=1e-7):
    """similarity.py:75-97."""
    S = row_softmax(clip_feats, float(a))
    _, idx = col_topk(target_feats, top_k)
    pdge = wpmi_score(S, idx, None, f32(min_prob), soft=0)
    return logsumexp_sub(pdge, float(f32(lam)))


def cos_similarity(clip_feats, target_feats):
    """similarity.py:33-47: columns L2-normalised over the image axis, target.T @ clip."""
    c = _c(clip_feats)
    t = _c(target_feats)
    c = c / np.sqrt((c * c).sum(0, keepdims=True, dtype=f32))
    t = t / np.sqrt((t * t).sum(0, keepdims=True, dtype=f32))
    return np.matmul(t.T, c).astype(f32)


def cos_similarity_cubed(clip_feats, target_feats, min_norm=1e-3):
    """similarity.py:7-31: centre columns, cube, normalise (norm clipped at min_norm), target.T @ clip."""
    c = _c(clip_feats)
    t = _c(target_feats)
    c = c - c.mean(0, keepdims=True, dtype=f32)
    t = t - t.mean(0, keepdims=True, dtype=f32)
    c = c * c * c
    t = t * t * t
    c = c / np.clip(np.sqrt((c * c).sum(0, keepdims=True, dtype=f32)), f32(min_norm), None)
    t = t / np.clip(np.sqrt((t * t).sum(0, keepdims=True, dtype=f32)), f32(min_norm), None)
    return np.matmul(t.T, c).astype(f32)


def sum0(x, split=-1):
    """torch.sum(x, dim=0) of a [R, C] float32 matrix in ATen's CPU order (cascade / row_sum by column)."""
    x = _c(x)
    R, C = x.shape
    out = np.empty((C,), f32)
    lib().mcd_o_sum0(_f(x), _i64(R), _i64(C), ctypes.c_int(split), _f(out))
    return out


def rank_reorder_perms(U, top_n, n_rep=5):
    """The permutations the reference draws (similarity.py:119): for every neuron, in order, five
    torch.randperm(top_n) calls on torch's global CPU generator.  Under torch.manual_seed(s) this reproduces the
    reference's stream call for call.  Returns int64 [U, n_rep, top_n]."""
    import torch
    out = np.empty((U, n_rep, top_n), np.int64)
    for u in range(U):
        for k in range(n_rep):
            out[u, k] = torch.randperm(top_n).numpy()
    return out


def rank_reorder(clip_feats, target_feats, p=3, top_fraction=0.05, scale_p=0.5, perms=None):
    """similarity.py:99-132.  For every neuron u: the top_n most activating images (descending activations t),
    G = clip_feats[those images]; per concept c the ascending RANK of G[j,c] among the top_n rows picks a value of
    the ascending-sorted activations, and
        err[u,c] = mean_j |t_j - sorted_t[rank_jc]|^p / baseline_u / mean_j(G[j,c])^scale_p,   result = -err
    baseline_u = mean over 5 random permutations of |sorted_t - sorted_t[perm]|^p  (torch.randperm, :119).
    mean(G) < 0 gives NaN (sqrt of a negative), as in the reference.  perms: [U, 5, top_n] or None to draw them
    from torch's global generator in the reference's order."""
    P = _c(clip_feats)
    A = _c(target_feats)
    N, C = P.shape
    U = A.shape[1]
    top_n = int(N * top_fraction)
    vals, idx = col_topk(A, top_n)                 # [top_n, U] descending
    if perms is None:
        perms = rank_reorder_perms(U, top_n)
    out = np.empty((U, C), f32)
    for u in range(U):
        G = P[idx[:, u]]                            # [top_n, C]
        avg = (sum0(G) / f32(top_n)).astype(f32)
        rank = np.argsort(np.argsort(G, axis=0, kind="stable"), axis=0, kind="stable")
        t = vals[:, u:u + 1]                        # [top_n, 1] descending
        st = t[::-1]                                # ascending
        base = st - np.concatenate([st[perms[u, k]] for k in range(perms.shape[1])], axis=1)
        ab = np.abs(base).astype(f32)
        base = (_pow(ab, p).sum(dtype=f32) / f32(ab.size)).astype(f32)
        reorg = st[:, 0][rank]                      # [top_n, C]
        d = np.abs(t - reorg).astype(f32)
        err = (sum0(_pow(d, p)) / f32(top_n)).astype(f32) / base
        with np.errstate(invalid="ignore"):
            out[u] = -(err / _pow(avg, scale_p)).astype(f32)
    return out


def _pow(x, p):
    """ATen pow_tensor_scalar: exponents 2, 3, 0.5 are x*x, (x*x)*x, sqrt(x); anything else is powf."""
    x = np.asarray(x, f32)
    if p == 2:
        return x * x
    if p == 3:
        return (x * x) * x
    if p == 0.5:
        return np.sqrt(x)
    return np.power(x, f32(p)).astype(f32)


def hook_pool(x, mode="avg"):
    """utils.py:27-52 get_activation() on a 4-D/3-D/2-D hook output -> [B, U]."""
    x = np.asarray(x, f32)
    if x.ndim == 4:
        B, Cc, H, W = x.shape
        out = np.empty((B, Cc), f32)
        xc = _c(x)
        lib().mcd_o_hook_pool(_f(xc), _i64(B), _i64(Cc), _i64(H * W), ctypes.c_int(0 if mode == "avg" else 1),
                              _f(out), _i64(0), _i64(0), _i64(Cc))
        return out
    if x.ndim == 3:
        return x[:, 0].copy()
    return x.copy()


def dissect_layer(E_img, E_txt, A, similarity_fn="soft_wpmi", top_k=100, k_desc=10, k_img=5, blas=True):
    """One layer of describe_broad_neurons.py:83-116 on in-memory tensors.

    utils.py:566-612 get_similarity_from_activations (normalise, P, similarity_fn) followed by
    topk(similarities, k=10, dim=1) and topk(target_feats, k=5, dim=0) (describe_broad_neurons.py:101-102).
    Returns dict(sim, vals [U,k_desc], ids [U,k_desc], top_ids [k_img,U]).
    """
    P = embed_gemm(E_img, E_txt, blas=blas)
    if similarity_fn == "soft_wpmi":
        sim = soft_wpmi(P, A, top_k=top_k)
    elif similarity_fn == "wpmi":
        sim = wpmi(P, A, top_k=top_k)
    elif similarity_fn == "cos_similarity":
        sim = cos_similarity(P, A)
    elif similarity_fn == "cos_similarity_cubed":
        sim = cos_similarity_cubed(P, A)
    else:
        raise ValueError(similarity_fn)
    vals, ids = row_topk(sim, k_desc)
    _, top_ids = col_topk(A, k_img)
    return dict(P=P, sim=sim, vals=vals, ids=ids, top_ids=top_ids)
