"""Tensor-level wrappers over the C ABI (include/mcd_hip.h).

Every function takes torch tensors that already live in HBM, passes their device pointers and
strides to libmcd_hip.so on torch's current stream, and returns torch tensors.  torch is used for
memory and streams only; all arithmetic happens in the HIP kernels.  Nothing here runs on the CPU:
a CPU tensor is a TypeError, a missing library an ImportError.
"""
import ctypes
import functools

import torch

from . import _lib
from ._lib import McdError, check  # noqa: F401

POOL_MODES = {"avg": 0, "max": 1, "cls": 2, "none": 3}
GEMM_MODES = {"f32": 0, "bf16x3": 1, "bf16": 2}


def _stream():
    """torch's current stream of the CURRENT device; every wrapper runs under _on_device, which makes the tensors'
    device current first."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_device(fn):
    """Run `fn` with the device of its tensor arguments current: the library launches on torch's current stream and
    keeps its one-time kernel attributes per HIP device, so a call on tensors of cuda:1 from a process whose current
    device is cuda:0 must switch first.  Tensors on different devices are an error (no silent peer access)."""
    @functools.wraps(fn)
    def wrapper(*args, **kw):
        dev = None
        for a in list(args) + list(kw.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise ValueError("%s: tensors on different devices (%s and %s)" % (fn.__name__, dev, a.device))
        if dev is None:
            return fn(*args, **kw)          # CPU or no tensors: the wrapper's own checks raise
        with torch.cuda.device(dev):
            return fn(*args, **kw)
    return wrapper


def _need_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise TypeError("mammo-clip-dissect_amd runs on the GPU only: expected a CUDA/HIP tensor, got %s"
                            % (t.device if isinstance(t, torch.Tensor) else type(t)))


def _f32_rows(t, name):
    """2-D fp32 tensor with unit inner stride (a row-major matrix with leading dimension stride(0))."""
    _need_gpu(t)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if t.dim() != 2:
        raise ValueError("%s must be 2-D, got shape %s" % (name, tuple(t.shape)))
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _f32_out(t, name):
    """An OUTPUT matrix: the kernel must write into the caller's memory, so a layout the ABI cannot address is an
    error, never a silent copy."""
    _need_gpu(t)
    if t.dtype != torch.float32 or t.dim() != 2:
        raise TypeError("%s must be a 2-D float32 tensor" % name)
    if (t.shape[1] > 1 and t.stride(1) != 1) or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        raise ValueError("%s must have unit inner stride and a leading dimension >= its width (strides %s)"
                         % (name, tuple(t.stride())))
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def sum_split(C):
    """ATen torch.sum(dim=0) CPU rule: columns below use the cascade order, the rest row_sum."""
    return (C // 32) * 32 if C >= 8 else (C // 4) * 4


def pad_cols(C, mult=64):
    return (C + mult - 1) // mult * mult


# ---- K1a / K1 ----------------------------------------------------------------------------------
@_on_device
def normalize_rows(x, out=None):
    """y = x / ||x||_2 per row (utils.py:577-578).  out may be x itself (in place)."""
    x = _f32_rows(x, "x")
    if out is None:
        out = torch.empty_like(x, memory_format=torch.contiguous_format)
    out = _f32_out(out, "out")
    L = _lib.load()
    check(L.mcd_normalize_rows(x.data_ptr(), _ld(x), x.shape[0], x.shape[1], out.data_ptr(), _ld(out), _stream()))
    return out


@_on_device
def center_cube_normalize_rows(x, min_norm=1e-3, out=None):
    """Rows centred, cubed and scaled to unit norm (norm clipped at min_norm): similarity.py:15-22 with the
    image axis contiguous."""
    x = _f32_rows(x, "x")
    if out is None:
        out = torch.empty_like(x, memory_format=torch.contiguous_format)
    out = _f32_out(out, "out")
    L = _lib.load()
    check(L.mcd_center_cube_normalize_rows(x.data_ptr(), _ld(x), x.shape[0], x.shape[1], float(min_norm), out.data_ptr(),
                                           _ld(out), _stream()))
    return out


@_on_device
def embed_gemm(I, T, mode="f32", out=None, use_workspace=True):
    """P = I @ T.T for I [N,D], T [C,D] (utils.py:594).  mode: "f32" (the parity mode: exact fp32 fma chains
    over the K-blocks MKL's sgemm uses, so P equals torch's CPU matmul to the bit),
    "bf16x3" (split bf16, fp32-class accuracy) or "bf16" (single pass, stress configuration only)."""
    I = _f32_rows(I, "I")
    T = _f32_rows(T, "T")
    if I.shape[1] != T.shape[1]:
        raise RuntimeError("mat1 and mat2 shapes cannot be multiplied (%dx%d and %dx%d)"
                           % (I.shape[0], I.shape[1], T.shape[1], T.shape[0]))
    N, D = I.shape
    C = T.shape[0]
    if out is None:
        out = torch.empty((N, C), dtype=torch.float32, device=I.device)
    out = _f32_out(out, "out")
    L = _lib.load()
    nws = L.mcd_embed_gemm_workspace(N, C, D, GEMM_MODES[mode]) if use_workspace else 0
    ws = torch.empty((nws,), dtype=torch.uint8, device=I.device) if nws else None
    check(L.mcd_embed_gemm(I.data_ptr(), _ld(I), T.data_ptr(), _ld(T), N, C, D, GEMM_MODES[mode], out.data_ptr(),
                           _ld(out), ws.data_ptr() if ws is not None else None, nws, _stream()))
    return out


# ---- K1s / K4s: the stress chain (bf16, no parity claim) ----------------------------------------------------
@_on_device
def embed_gemm_exp(I, T, a, normalize=False):
    """K1 + K2 of the stress configuration in one kernel that never writes fp32 P (utils.py:594 + similarity.py:54):
    returns (E, rinv) with E = bf16(exp(a (I @ T.T - 1))) as a [N, C] view of a buffer whose rows are padded with zeros
    to a multiple of 128 concepts, and rinv[n] = 1 / rowsum, so that softmax(a P) = E * rinv[:, None].  I and T must
    be row-normalised, or raw embeddings with normalize=True (utils.py:577-578 folded into the bf16 conversion)."""
    I = _f32_rows(I, "I")
    T = _f32_rows(T, "T")
    if I.shape[1] != T.shape[1]:
        raise RuntimeError("mat1 and mat2 shapes cannot be multiplied (%dx%d and %dx%d)"
                           % (I.shape[0], I.shape[1], T.shape[1], T.shape[0]))
    N, D = I.shape
    C = T.shape[0]
    ldE = pad_cols(C, 128)
    E = torch.empty((N, ldE), dtype=torch.bfloat16, device=I.device)
    rinv = torch.empty((N,), dtype=torch.float32, device=I.device)
    L = _lib.load()
    nws = L.mcd_embed_gemm_exp_workspace(N, C, D)
    ws = torch.empty((max(nws, 16),), dtype=torch.uint8, device=I.device)
    check(L.mcd_embed_gemm_exp(I.data_ptr(), _ld(I), T.data_ptr(), _ld(T), N, C, D, float(a), 1 if normalize else 0, E.data_ptr(), ldE,
                               rinv.data_ptr(), ws.data_ptr(), nws, _stream()))
    return E[:, :C], rinv


@_on_device
def wpmi_score_bf16(E, rinv, idx, p, min_prob, soft, out=None):
    """K4 on the stress chain's representation: pdge[u,c] = sum_j log(term(E[idx[u,j], c] * rinv[idx[u,j]])).
    E: the [N, C] bf16 view embed_gemm_exp returns (rows padded to a multiple of 128); idx [U,K] int32."""
    _need_gpu(E, rinv, idx)
    if E.dtype != torch.bfloat16 or E.dim() != 2 or (E.shape[1] > 1 and E.stride(1) != 1) or E.stride(0) % 128 != 0:
        raise TypeError("E must be a bfloat16 [N, C] view with unit inner stride and rows padded to a multiple of 128")
    if rinv.dtype != torch.float32 or rinv.numel() != E.shape[0] or not rinv.is_contiguous():
        raise TypeError("rinv must be N contiguous float32 values")
    if idx.dtype != torch.int32 or idx.dim() != 2 or (idx.shape[1] > 1 and idx.stride(1) != 1):
        raise TypeError("idx must be a [U,K] int32 tensor with unit inner stride")
    N, C = E.shape
    U, K = idx.shape
    if out is None:
        out = torch.empty((U, C), dtype=torch.float32, device=E.device)
    out = _f32_out(out, "out")
    if soft:
        _need_gpu(p)
        if p.dtype != torch.float32 or p.numel() != K:
            raise TypeError("p must be K float32 values")
        p = p.contiguous()
    L = _lib.load()
    nws = int(L.mcd_wpmi_score_bf16_workspace(U, K))
    ws = torch.empty(max(nws, 8) // 8, dtype=torch.int64, device=E.device)
    check(L.mcd_wpmi_score_bf16(E.data_ptr(), E.stride(0), N, C, rinv.data_ptr(), idx.data_ptr(),
                                idx.stride(0) if U > 1 else K, U, K, p.data_ptr() if soft else None, float(min_prob),
                                1 if soft else 0, out.data_ptr(), _ld(out), ws.data_ptr(), nws, _stream()))
    return out


# ---- K2 ------------------------------------------------------------------------------------------
@_on_device
def row_softmax(P, a, pad_to=192):
    """S = softmax(a*P, dim=1) (similarity.py:54) into a buffer whose leading dimension is padded to a
    multiple of `pad_to` floats (padding columns are 0).  Returns the [N, C] view of that buffer.
    192 = lcm(64, 96): whole 96-concept slices for K4's XCD-sliced kernel at any C (763 -> 768, 10 000 -> 10 176)."""
    P = _f32_rows(P, "clip_feats")
    N, C = P.shape
    ldS = pad_cols(C, pad_to) if pad_to else C
    S = torch.empty((N, ldS), dtype=torch.float32, device=P.device)
    L = _lib.load()
    check(L.mcd_row_softmax(P.data_ptr(), _ld(P), N, C, float(a), S.data_ptr(), ldS, _stream()))
    return S[:, :C]


# ---- K3 ------------------------------------------------------------------------------------------
@_on_device
def col_topk(A, K, neuron_major=False, want_vals=True):
    """Per neuron, the K most activating images, sorted descending (similarity.py:55).

    A is image-major [N,U] (what the reference passes) or, with neuron_major=True, [U,N].
    Returns (vals [U,K] float32 or None, idx [U,K] int32) -- neuron-major, i.e. torch.topk(A, K, dim=0)
    transposed.  K > N raises RuntimeError like torch.topk.
    """
    A = _f32_rows(A, "target_feats")
    if neuron_major:
        U, N = A.shape
        sn, su = 1, _ld(A)
    else:
        N, U = A.shape
        sn, su = _ld(A), 1
    L = _lib.load()
    K = int(K)
    idx = torch.empty((U, K), dtype=torch.int32, device=A.device)
    vals = torch.empty((U, K), dtype=torch.float32, device=A.device) if want_vals else None
    ws_bytes = L.mcd_col_topk_workspace(N, U, sn, su, K) if (N > 0 and K >= 1) else 0
    ws = torch.empty((max(ws_bytes, 4) // 4,), dtype=torch.float32, device=A.device)
    rc = L.mcd_col_topk(A.data_ptr(), N, U, sn, su, K, vals.data_ptr() if want_vals else None, idx.data_ptr(),
                        K, ws.data_ptr(), ws_bytes, _stream())
    if rc == _lib.MCD_E_RANGE:
        raise RuntimeError("selected index k out of range")
    check(rc)
    return vals, idx


@_on_device
def transpose(A, out=None):
    """image-major [N,U] -> neuron-major [U,N]."""
    A = _f32_rows(A, "A")
    N, U = A.shape
    if out is None:
        out = torch.empty((U, N), dtype=torch.float32, device=A.device)
    out = _f32_out(out, "out")
    L = _lib.load()
    check(L.mcd_transpose(A.data_ptr(), _ld(A), N, U, out.data_ptr(), _ld(out), _stream()))
    return out


# ---- K4 / K5 -------------------------------------------------------------------------------------
@_on_device
def wpmi_score(S, idx, p, min_prob, soft, split=-1, out=None, fast_log=False, s_is_prob=False):
    """pdge[u,c] = sum_j log(term(S[idx[u,j], c])) (similarity.py:59-65, 84-88); idx is [U,K] int32.
    fast_log=True trades the accurate (near correctly rounded) log for the v_log_f32 based one (<= ~1.5 ulp).
    s_is_prob=True promises S in [0,1] (it came from row_softmax) and p in [0,1]: the log-table range check is skipped."""
    S = _f32_rows(S, "S")
    _need_gpu(idx)
    if idx.dtype != torch.int32 or idx.dim() != 2 or (idx.shape[1] > 1 and idx.stride(1) != 1):
        raise TypeError("idx must be a [U,K] int32 tensor with unit inner stride")
    N, C = S.shape
    U, K = idx.shape
    if out is None:
        out = torch.empty((U, C), dtype=torch.float32, device=S.device)
    out = _f32_out(out, "out")
    if soft:
        _need_gpu(p)
        if p.dtype != torch.float32 or p.numel() != K:
            raise TypeError("p must be K float32 values")
        p = p.contiguous()
    L = _lib.load()
    check(L.mcd_wpmi_score(S.data_ptr(), _ld(S), N, C, idx.data_ptr(), idx.stride(0) if U > 1 else K, U, K,
                           p.data_ptr() if soft else None, float(min_prob), (1 if soft else 0) | (2 if fast_log else 0) | (4 if s_is_prob else 0), int(split),
                           out.data_ptr(), _ld(out), _stream()))
    return out


@_on_device
def logsumexp_sub(pdge, lam, seg_offsets=None, split=-1, out=None):
    """out = pdge - lam*(logsumexp(pdge, 0) - log U), per row segment (one segment per layer)."""
    pdge = _f32_rows(pdge, "pdge")
    U, C = pdge.shape
    if seg_offsets is None:
        seg_offsets = [0, U]
    if out is None:
        out = torch.empty((U, C), dtype=torch.float32, device=pdge.device)
    out = _f32_out(out, "out")
    L = _lib.load()
    for s0 in range(0, len(seg_offsets) - 1, 64):  # the ABI takes at most 64 segments per call
        seg = list(seg_offsets[s0:s0 + 65])
        arr = (ctypes.c_int64 * len(seg))(*seg)
        ws_bytes = L.mcd_logsumexp_sub_workspace(seg[-1] - seg[0], C, len(seg) - 1)
        ws = torch.empty((max(ws_bytes, 4) // 4,), dtype=torch.float32, device=pdge.device)
        check(L.mcd_logsumexp_sub(pdge.data_ptr(), _ld(pdge), C, arr, len(seg) - 1, float(lam), int(split),
                                  out.data_ptr(), _ld(out), ws.data_ptr(), ws_bytes, _stream()))
    return out


# ---- K6 ------------------------------------------------------------------------------------------
@_on_device
def row_topk(sim, k):
    """torch.topk(sim, k, dim=1) (k=1: torch.max(sim, dim=1)); returns (vals [U,k], idx [U,k] int32)."""
    sim = _f32_rows(sim, "similarities")
    U, C = sim.shape
    k = int(k)
    vals = torch.empty((U, k), dtype=torch.float32, device=sim.device)
    idx = torch.empty((U, k), dtype=torch.int32, device=sim.device)
    L = _lib.load()
    rc = L.mcd_row_topk(sim.data_ptr(), _ld(sim), U, C, k, vals.data_ptr(), idx.data_ptr(), _stream())
    if rc == _lib.MCD_E_RANGE:
        raise RuntimeError("selected index k out of range")
    check(rc)
    return vals, idx


# ---- K8 ------------------------------------------------------------------------------------------
@_on_device
def rank_reorder(P, tvals, tidx, perms, p=3, scale_p=0.5, out=None):
    """rank_reorder scores of one layer (similarity.py:107-132).  tvals/tidx: [U, top_n] from col_topk (descending
    activations, image indices); perms: int32 [U, n_perm, top_n] baseline permutations.  Returns [U, C]."""
    P = _f32_rows(P, "P")
    N, C = P.shape
    if _ld(P) % 4 != 0 or P.data_ptr() % 16 != 0:     # 16-byte gathers need rows padded to whole quads
        Pp = torch.zeros((N, pad_cols(C, 4)), dtype=torch.float32, device=P.device)
        Pp[:, :C] = P
        P = Pp[:, :C]
    tvals = _f32_rows(tvals, "tvals")
    _need_gpu(tidx, perms)
    U, top_n = tvals.shape
    if tidx.dtype != torch.int32 or tuple(tidx.shape) != (U, top_n) or perms.dtype != torch.int32 or perms.dim() != 3 \
            or perms.shape[0] != U or perms.shape[2] != top_n:
        raise ValueError("tidx must be int32 [U, top_n] and perms int32 [U, n_perm, top_n]")
    tidx = tidx.contiguous()
    tvals = tvals.contiguous()
    perms = perms.contiguous()
    if out is None:
        out = torch.empty((U, C), dtype=torch.float32, device=P.device)
    out = _f32_out(out, "out")
    ws = torch.empty((max(U, 1),), dtype=torch.float32, device=P.device)
    L = _lib.load()
    check(L.mcd_rank_reorder(P.data_ptr(), _ld(P), N, C, tvals.data_ptr(), tidx.data_ptr(), top_n, U, top_n,
                             perms.data_ptr(), perms.shape[1], float(p), float(scale_p), ws.data_ptr(), out.data_ptr(),
                             _ld(out), _stream()))
    return out


# ---- K9 ------------------------------------------------------------------------------------------
VIT_ATTENTION_MAX_T = 256


@_on_device
def vit_attention(qkv, heads, out=None):
    """softmax(q k^T / 8) v per head for the ViT tower: qkv [B, T, 3*heads*64] (the fused projection's output,
    q | k | v along the last axis, heads inside each) -> [B, T, heads*64].  fp32, head dimension 64, T <= 256."""
    _need_gpu(qkv)
    if qkv.dtype != torch.float32 or qkv.dim() != 3 or not qkv.is_contiguous():
        raise TypeError("qkv must be a contiguous float32 [B, T, 3*heads*64] tensor")
    B, T, W = qkv.shape
    if W != 3 * heads * 64:
        raise ValueError("qkv last dimension %d is not 3 * %d heads * 64" % (W, heads))
    if out is None:
        out = torch.empty((B, T, heads * 64), dtype=torch.float32, device=qkv.device)
    elif out.dtype != torch.float32 or tuple(out.shape) != (B, T, heads * 64) or not out.is_contiguous():
        raise TypeError("out must be a contiguous float32 [B, T, heads*64] tensor")
    L = _lib.load()
    check(L.mcd_vit_attention(qkv.data_ptr(), B, T, heads, out.data_ptr(), _stream()))
    return out


# ---- K10 -----------------------------------------------------------------------------------------
@_on_device
def layer_norm(x, weight, bias, eps):
    """LayerNorm over the last dimension of a contiguous fp32 tensor (torch.nn.functional.layer_norm semantics)."""
    _need_gpu(x, weight, bias)
    D = x.shape[-1]
    if x.dtype != torch.float32 or not x.is_contiguous() or weight.shape != (D,) or bias.shape != (D,):
        raise TypeError("layer_norm: contiguous float32 x and [D] weight / bias")
    y = torch.empty_like(x)
    L = _lib.load()
    check(L.mcd_layer_norm(x.data_ptr(), x.numel() // D, D, weight.data_ptr(), bias.data_ptr(), float(eps), y.data_ptr(),
                           _stream()))
    return y


# ---- K11 -----------------------------------------------------------------------------------------
@_on_device
def patchify(x, patch):
    """[B, Cin, H, W] -> [B, 1 + (H/patch)(W/patch), Cin*patch*patch]: row 0 of every image zero (class-token slot),
    then the patches in (c, dy, dx) order -- the operand of the patch embedding written as a GEMM."""
    _need_gpu(x)
    if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous():
        raise TypeError("patchify: contiguous float32 [B, Cin, H, W]")
    B, Cin, H, W = x.shape
    if H % patch or W % patch:
        raise ValueError("patchify: %dx%d is not a multiple of the %d-pixel patch" % (H, W, patch))
    out = torch.empty((B, 1 + (H // patch) * (W // patch), Cin * patch * patch), dtype=torch.float32, device=x.device)
    L = _lib.load()
    check(L.mcd_patchify(x.data_ptr(), B, Cin, H, W, patch, out.data_ptr(), _stream()))
    return out


# ---- encoder-side linear + bias + residual on hipBLASLt (libmcd_blaslt.so) ------------------------
_blaslt_ws = {}
# bench.py sets this to a list to time the library GEMMs inside the forwards: every call then appends
# (start, end, M, N, K) with two HIP events recorded on the launch stream around the hipBLASLt call.
LINEAR_EVENTS = None


def linear_residual_available():
    return _lib.load_blaslt() is not None


def encoder_gemm_picks():
    """[(M, N, K, has_res, pick)]: the hipBLASLt algorithm (index into the heuristic's list) this process keeps for every
    encoder GEMM shape it has run through linear_residual."""
    L = _lib.load_blaslt()
    if L is None:
        return []
    n = L.mcd_linear_residual_get_picks(None, 0)
    buf = (ctypes.c_int64 * (5 * max(n, 1)))()
    n = min(n, L.mcd_linear_residual_get_picks(buf, n))
    return [tuple(int(buf[5 * i + j]) for j in range(5)) for i in range(n)]


def set_encoder_gemm_picks(picks):
    """Force the algorithm for the given shapes (what another process reported with encoder_gemm_picks()): same
    algorithm => same summation order => the same image encodes to the same bits on every rank."""
    L = _lib.load_blaslt()
    if L is None:
        return
    for M, N, K, has_res, pick in picks:
        L.mcd_linear_residual_set_pick(int(M), int(N), int(K), int(has_res), int(pick))


@_on_device
def linear_residual(res, h, weight, bias=None, out=None):
    """out = res + h @ weight.T + bias in ONE hipBLASLt GEMM (bias epilogue + beta*C), instead of nn.Linear followed
    by an elementwise add over the whole residual stream.  res, h: [..., N] / [..., K] contiguous fp32 with the same
    leading shape; weight [N, K]; out defaults to a new tensor (pass out=res for in place).  res may be None."""
    L = _lib.load_blaslt()
    if L is None:
        raise ImportError("libmcd_blaslt.so is not available (make -C mammo-clip-dissect_amd/csrc)")
    _need_gpu(h, weight, bias, res, out)
    K = h.shape[-1]
    N = weight.shape[0]
    if weight.shape[1] != K or (res is not None and (res.shape[-1] != N or res.shape[:-1] != h.shape[:-1])):
        raise ValueError("linear_residual: shapes h %s, weight %s, res %s do not match"
                         % (tuple(h.shape), tuple(weight.shape), None if res is None else tuple(res.shape)))
    for t in (h, weight, bias, res, out):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise TypeError("linear_residual: contiguous float32 tensors only")
    M = h.numel() // K
    if out is None:
        out = torch.empty(h.shape[:-1] + (N,), dtype=torch.float32, device=h.device)
    elif tuple(out.shape) != tuple(h.shape[:-1]) + (N,):
        raise ValueError("linear_residual: out has shape %s" % (tuple(out.shape),))
    ws = _blaslt_ws.get(h.device)
    if ws is None:
        ws = _blaslt_ws[h.device] = torch.empty((L.mcd_linear_residual_workspace(),), dtype=torch.uint8, device=h.device)
    ev = LINEAR_EVENTS
    if ev is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = L.mcd_linear_residual(h.data_ptr(), K, weight.data_ptr(), K, bias.data_ptr() if bias is not None else None,
                               res.data_ptr() if res is not None else None, N, out.data_ptr(), N, M, N, K,
                               ws.data_ptr(), ws.numel(), _stream())
    if ev is not None:
        e1.record()
        ev.append((e0, e1, M, N, K))
    if rc != 0:
        raise _lib.McdError(rc, L.mcd_blaslt_last_error().decode("utf-8", "replace"))
    return out


# ---- K0 ------------------------------------------------------------------------------------------
@_on_device
def hook_pool(x, mode, dst, row0, col0, neuron_major):
    """Pool a hooked tensor (utils.py:27-52) and write it into the activation matrix `dst`.

    mode: "avg"/"max" for 4-D [B,Cout,H,W]; 3-D [B,T,F] takes token 0; 2-D [B,F] is copied.
    dst: [U_total, N] when neuron_major else [N, U_total]; rows row0.. / columns col0.. are written.
    Returns the number of neurons written.
    """
    _need_gpu(x, dst)
    if x.dtype != torch.float32:
        x = x.float()
    x = x.contiguous()
    if x.dim() == 4:
        B, Cout, HW = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
        m = POOL_MODES[mode]
    elif x.dim() == 3:
        B, HW, Cout = x.shape
        m = POOL_MODES["cls"]
    elif x.dim() == 2:
        B, Cout = x.shape
        HW = 1
        m = POOL_MODES["none"]
    else:
        raise ValueError("unsupported hook output shape %s" % (tuple(x.shape),))
    if dst.dtype != torch.float32 or dst.dim() != 2 or dst.stride(1) != 1:
        raise TypeError("dst must be a 2-D float32 matrix with unit inner stride")
    need = (col0 + Cout, row0 + B) if neuron_major else (row0 + B, col0 + Cout)
    if row0 < 0 or col0 < 0 or dst.shape[0] < need[0] or dst.shape[1] < need[1]:
        raise IndexError("hook_pool: block [%d:%d, %d:%d] outside the activation matrix %s"
                         % (row0, row0 + B, col0, col0 + Cout, tuple(dst.shape)))
    if neuron_major:
        sn, su = 1, dst.stride(0)
    else:
        sn, su = dst.stride(0), 1
    L = _lib.load()
    check(L.mcd_hook_pool(x.data_ptr(), B, Cout, HW, m, dst.data_ptr(), int(row0), int(col0), sn, su, _stream()))
    return Cout
