"""Drop-in for the reference's concept_vit/similarity.py, computed by the gfx950 HIP kernels.

Same function names, argument names, defaults and return contract as the reference
(/root/reference/concept_vit/similarity.py): each function takes
    clip_feats   [N images, C concepts]  the CLIP image-text matrix P
    target_feats [N images, U neurons]   pooled activations of one layer
and returns a fresh float32 tensor [U, C] on `device`; inputs are never modified.

Differences, all deliberate:
  * GPU only.  `device` must name a CUDA/HIP device; "cpu" raises (there is no CPU path here --
    the CPU restatement lives in oracle/ and is test infrastructure).
  * torch.topk ties: the reference's order is unspecified; here ties go to the lower image index.
  * rank_reorder draws its baseline permutations with torch.randperm on torch's global CPU generator
    (similarity.py:119).  The mirror draws the same permutations in the same order (5 per neuron, neuron by
    neuron) from the same generator, so under torch.manual_seed(s) both produce the same scores; only the
    permutation indices are made on the host, all arithmetic is in K8.
"""
import torch

from .. import core

__all__ = ["soft_wpmi", "wpmi", "rank_reorder", "cos_similarity", "cos_similarity_cubed"]


def _dev(device):
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError("mammo-clip-dissect_amd: similarity functions run on the GPU only (device=%r)" % (device,))
    return d


def _to(t, d):
    if not isinstance(t, torch.Tensor):
        raise TypeError("expected a torch.Tensor")
    return t.to(device=d, dtype=torch.float32)


def _check_pair(clip_feats, target_feats):
    if clip_feats.dim() != 2 or target_feats.dim() != 2:
        raise ValueError("clip_feats and target_feats must be 2-D")
    if clip_feats.shape[0] != target_feats.shape[0]:
        raise RuntimeError("clip_feats has %d images, target_feats %d" % (clip_feats.shape[0], target_feats.shape[0]))


def _score(clip_feats, target_feats, top_k, a, lam, device, min_prob, p):
    d = _dev(device)
    with torch.no_grad():
        P = _to(clip_feats, d)
        A = _to(target_feats, d)
        _check_pair(P, A)
        if A.shape[1] == 0:   # the reference ends in torch.cat([]) for a layer without neurons (similarity.py:67)
            raise RuntimeError("torch.cat(): expected a non-empty list of Tensors")
        S = core.row_softmax(P, a)                          # similarity.py:54 / :80
        _, inds = core.col_topk(A, top_k, want_vals=False)  # similarity.py:55 / :82  ([U,K] here)
        S_full = S  # [N,C] view of the padded buffer
        # S is our own softmax output (values in [0,1], or NaN rows out of NaN/inf similarities, which stay NaN through
        # K4's arithmetic); p is known on the host: promise the kernel its log arguments are in range
        p_ok = p is None or bool(((p >= 0) & (p <= 1)).all())
        pdge = core.wpmi_score(S_full, inds, p.to(d) if p is not None else None, min_prob, soft=p is not None,
                               s_is_prob=p_ok)
        # similarity.py:70-72: lam*prob_d is a float32 multiply by the Python scalar lam
        return core.logsumexp_sub(pdge, float(torch.tensor(lam, dtype=torch.float32)))


def soft_wpmi(clip_feats, target_feats, top_k=100, a=10, lam=1, device='cuda',
              min_prob=1e-7, p_start=0.998, p_end=0.97):
    """reference similarity.py:49-73."""
    # similarity.py:58, evaluated with the same torch CPU ops so the K coefficients are bit-identical
    p_in_examples = p_start - (torch.arange(start=0, end=top_k) / top_k * (p_start - p_end))
    return _score(clip_feats, target_feats, top_k, a, lam, device, min_prob, p_in_examples.float())


def wpmi(clip_feats, target_feats, top_k=28, a=2, lam=0.6, device='cuda', min_prob=1e-7):
    """reference similarity.py:75-97."""
    return _score(clip_feats, target_feats, top_k, a, lam, device, min_prob, None)


def rank_reorder(clip_feats, target_feats, device="cuda", p=3, top_fraction=0.05, scale_p=0.5):
    """reference similarity.py:99-132.
    top fraction: percentage of mostly highly activating target images to use for eval. Between 0 and 1"""
    d = _dev(device)
    with torch.no_grad():
        P = _to(clip_feats, d)
        A = _to(target_feats, d)
        _check_pair(P, A)
        top_n = int(A.shape[0] * top_fraction)                       # similarity.py:106
        if top_n < 1:
            raise RuntimeError("rank_reorder: top_fraction*N = %d images (the reference divides by zero here)" % top_n)
        vals, inds = core.col_topk(A, top_n)                         # similarity.py:107 ([U, top_n] here)
        U = A.shape[1]
        # similarity.py:119: `for _ in range(5)` torch.randperm(len(sorted_target)) per neuron, global CPU generator
        perms = torch.stack([torch.randperm(top_n) for _ in range(5 * U)]).view(U, 5, top_n).to(torch.int32)
        return core.rank_reorder(P, vals, inds, perms.to(d), p=p, scale_p=scale_p)


def _cos(clip_feats, target_feats, device, prep):
    """target.T @ clip on column-normalised matrices (reference similarity.py:25-31 / :37-47), as one NT GEMM on the
    MFMA kernel: both operands are transposed so the contraction axis (images) is contiguous."""
    d = _dev(device)
    with torch.no_grad():
        P = _to(clip_feats, d)
        A = _to(target_feats, d)
        _check_pair(P, A)
        Pt = prep(core.transpose(P))        # [C, N]
        At = prep(core.transpose(A))        # [U, N]
        return core.embed_gemm(At, Pt)      # [U, C]


def cos_similarity(clip_feats, target_feats, device='cuda'):
    """reference similarity.py:33-47: x / ||x||_2 over the image axis, then target.T @ clip."""
    return _cos(clip_feats, target_feats, device, lambda t: core.normalize_rows(t, out=t))


def cos_similarity_cubed(clip_feats, target_feats, device='cuda', batch_size=10000, min_norm=1e-3):
    """reference similarity.py:7-31: subtract the mean over images, cube, normalise (norm clipped at min_norm),
    then target.T @ clip.  batch_size only blocked the reference's matmul; it has no effect here."""
    return _cos(clip_feats, target_feats, device,
                lambda t: core.center_cube_normalize_rows(t, min_norm=min_norm, out=t))
