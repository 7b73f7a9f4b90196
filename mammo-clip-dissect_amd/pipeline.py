"""Fused all-layer dissection and its image-sharded multi-GPU form.

What the reference does per layer, with a disk round trip in between (describe_broad_neurons.py:83-116,
utils.py:566-612), this module does once per run with everything resident in HBM:

  hooks   -> K0 pools every hooked layer straight into ONE neuron-major activation matrix At[sum U, N]
             (no list-append + torch.cat, utils.py:143; no second encoder pass when target == dissector)
  finish  -> K1a/K1 P = I_hat T_hat^T and K2 S = softmax(a P) ONCE (the reference recomputes them for
             every layer, utils.py:570-594), K3 top-K images for all neurons of all layers in one launch,
             K4 soft-WPMI sums, K5 per-layer logsumexp normalisation, K6 top-10 concepts.

Multi-GPU: the probe images are sharded over the ranks (one process per GPU); the only
exchanges are all-gathers (RCCL over xGMI; nothing is reduced, so results are bit-identical for any
number of ranks):
  1. S shard [N/G, C]                          -> all-gather -> S [N, C] on every rank
  2. local top-K (value, global image index)   -> all-gather -> merged to the global top-K per neuron
  3. neurons are split over the ranks for K4   -> all-gather of prob_d_given_e [sum U / G, C]
K5/K6 are replicated.  Shards may be UNEVEN (the reference walks any N, utils.py:174-181): every rank tells the
others how many images it holds, shorter shards are padded for the fixed-size all-gather and the padding is dropped
on arrival, so a probe set of any size can be split over any number of ranks (shard_bounds()).
The compute backend (`ops`) and the row all-gather (`gather`) are injectable so the host logic above can be exercised
on CPU under gloo by the tests; the defaults -- and the only ones the package ships -- are the HIP library and
torch.distributed's all_gather_into_tensor on device tensors (backend "nccl" = RCCL over xGMI).
"""
import os

import torch
import torch.distributed as dist

from . import core as _hip_ops


# Optional callable(name) invoked between the stages of every Dissector.finish ("start", "gemm", "softmax", ...): bench.py
# records HIP events there when the dissection runs inside the drop-in driver, which has no argument for it.
STAGE_MARK = None


def _round_up(x, m):
    return (x + m - 1) // m * m


def shard_align():
    """Images a shard boundary must be a multiple of (environment MCD_SHARD_ALIGN; 1 = any boundary).

    Encoder-inclusive bit-identity across rank counts needs it set to the encoder batch size: the towers' fp32 GEMMs are hipBLASLt
    stream-K kernels -- none of the ~500 fp32 solutions the library offers for these shapes is anything else
    (profiles/r04_blaslt_algos.txt) -- whose split of the K loop depends on the number of rows, so an image encodes to the same bits
    only inside the same batch: same size, same position.  With boundaries on batch multiples every batch of the global image
    order is encoded whole by exactly one rank, whichever it is, and the single shorter batch at the end of the probe set
    exists once at any rank count."""
    try:
        return max(1, int(os.environ.get("MCD_SHARD_ALIGN", "1")))
    except ValueError:
        return 1


def shard_bounds(n_total, world, rank, align=None):
    """Contiguous, balanced split of n_total probe images over `world` ranks: rank r holds images
    [lo, hi); the first n_total % world ranks hold one image more.  Contiguity keeps the global image order, which
    is what makes the sharded result identical to the 1-rank one (ties in the top-K go to the lower global index).
    align > 1 (default: shard_align()): the UNITS of `align` images are split that way instead (the last unit may be short), so
    every boundary is a multiple of align."""
    n_total, world, rank = int(n_total), int(world), int(rank)
    align = shard_align() if align is None else max(1, int(align))
    units = (n_total + align - 1) // align
    q, r = divmod(units, world)
    lo = rank * q + min(rank, r)
    hi = lo + q + (1 if rank < r else 0)
    return min(lo * align, n_total), min(hi * align, n_total)


def rccl_all_gather_rows(t, group=None):
    """[r, c] on every rank (same shape everywhere) -> [world * r, c], rank-major: one all_gather_into_tensor on
    device memory.  This is the package's only transport; tests inject a host-staged one to rehearse several ranks
    on one GPU."""
    world = dist.get_world_size(group)
    t = t.contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


def sync_encoder_gemm_picks(group=None):
    """Multi-rank runs: make every rank run the encoder GEMMs with rank 0's hipBLASLt algorithm picks (made by timing,
    so they can differ from process to process; different algorithms sum in different orders).  Call it after the warm-up
    pass that made rank 0 plan its shapes.  With equal batch shapes on every rank the same image then encodes to the same
    bits whichever rank holds it.  No-op for one rank or without libmcd_blaslt.so."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    obj = [_hip_ops.encoder_gemm_picks() if dist.get_rank(group) == 0 else None]
    dist.broadcast_object_list(obj, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if dist.get_rank(group) != 0:
        _hip_ops.set_encoder_gemm_picks(obj[0])


class DissectResult:
    """Per-neuron outputs for all layers (rows follow layer order, then unit order)."""

    def __init__(self, layer_names, layer_widths, sim, vals, ids, top_ids, top_vals, n_images):
        self.layer_names, self.layer_widths = layer_names, layer_widths
        self.sim = sim            # [sum U, C] float32 similarities (soft-WPMI)
        self.vals = vals          # [sum U, k_desc] float32: torch.topk(sim, k, dim=1).values
        self.ids = ids            # [sum U, k_desc] int32 concept indices
        self.top_ids = top_ids    # [sum U, k_img] int32: most activating images (global indices)
        self.top_vals = top_vals  # [sum U, k_img] float32
        self.n_images = n_images

    def layer_slices(self):
        o = 0
        for name, w in zip(self.layer_names, self.layer_widths):
            yield name, slice(o, o + w)
            o += w


class Dissector:
    def __init__(self, n_images, layer_names, layer_widths, n_concepts, embed_dim, device, top_k=100,
                 similarity_fn="soft_wpmi", a=None, lam=None, min_prob=1e-7, p_start=0.998, p_end=0.97,
                 pool_mode="avg", group=None, ops=None, gemm_mode="f32", gather=None):
        """n_images: images of THIS rank's shard (ranks may hold different numbers; see shard_bounds()).
        gemm_mode: "f32" (exact fp32 MFMA chain: the parity mode); "bf16x3" (split bf16, fp32-class P); "bf16" (the
        stress chain: bf16 MFMA GEMM with the softmax numerator fused into its epilogue, bf16 similarity matrix,
        v_log_f32 log in K4 -- no parity claim); "bf16_p" (bf16 MFMA GEMM that writes fp32 P, then the fp32 chain).
        gather: callable(tensor[r, c]) -> tensor[world * r, c]; default rccl_all_gather_rows.
        Collective: with more than one rank the constructor exchanges the shard sizes (every rank must build its
        Dissector at the same point)."""
        self.ops = ops if ops is not None else _hip_ops
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.n_local = int(n_images)
        self._gather = gather if gather is not None else (lambda t: rccl_all_gather_rows(t, self.group))
        if self.world > 1:
            mine = torch.tensor([[self.n_local]], dtype=torch.int64, device=self.device)
            self.counts = [int(v) for v in self._gather(mine).view(-1).tolist()]
        else:
            self.counts = [self.n_local]
        self.n_total = sum(self.counts)
        self.n_max = max(self.counts)
        self.row0 = sum(self.counts[:self.rank])         # global index of this shard's first image
        self.layer_names = list(layer_names)
        self.layer_widths = [int(w) for w in layer_widths]
        self.offsets = [0]
        for w in self.layer_widths:
            self.offsets.append(self.offsets[-1] + w)
        self.U = self.offsets[-1]
        self.C, self.D = int(n_concepts), int(embed_dim)
        self.pool_mode = pool_mode
        self.gemm_mode = gemm_mode
        self.set_scoring(similarity_fn, top_k, a=a, lam=lam, min_prob=min_prob, p_start=p_start, p_end=p_end)
        self.ldA = _round_up(max(self.n_local, 1), 64)
        self.At = torch.zeros((self.U, self.ldA), dtype=torch.float32, device=self.device)  # neuron-major
        self.E_img = torch.zeros((self.n_local, self.D), dtype=torch.float32, device=self.device)
        self.cursor = 0

    def set_scoring(self, similarity_fn="soft_wpmi", top_k=None, a=None, lam=None, min_prob=1e-7, p_start=0.998,
                    p_end=0.97):
        """Which similarity function finish() computes and with what parameters (defaults: the reference's,
        similarity.py:49 / :75).  Only the scoring side depends on it, so the drivers can extract first and choose
        afterwards."""
        if similarity_fn not in ("soft_wpmi", "wpmi"):
            raise NotImplementedError("fused pipeline supports soft_wpmi and wpmi (got %r)" % (similarity_fn,))
        self.similarity_fn = similarity_fn
        soft = similarity_fn == "soft_wpmi"
        self.top_k = int(top_k if top_k is not None else (100 if soft else 28))
        self.a = float(a if a is not None else (10 if soft else 2))                 # similarity.py:49 / :75
        self.lam = lam if lam is not None else (1 if soft else 0.6)
        self.min_prob = float(min_prob)
        # similarity.py:58, same torch CPU ops as the reference so the coefficients are bit-identical
        self.p = None
        self.p_ok = True
        if soft:
            self.p = (p_start - (torch.arange(start=0, end=self.top_k) / self.top_k * (p_start - p_end))).float().to(
                self.device)
            self.p_ok = 0.0 <= min(p_start, p_end) and max(p_start, p_end) <= 1.0

    # ---- extraction side -------------------------------------------------------------------------
    def reset(self):
        self.cursor = 0

    def hook(self, layer_index):
        """forward hook for target layer `layer_index`: reference get_activation(outputs, mode), utils.py:27-52,
        writing into the activation matrix instead of appending to a list."""
        col0 = self.offsets[layer_index]
        width = self.layer_widths[layer_index]

        def _hook(module, inputs, output):
            if type(output) is tuple:
                output = output[0]
            n = self.ops.hook_pool(output.detach(), self.pool_mode, self.At, self.cursor, col0, True)
            if n != width:
                raise RuntimeError("layer %s produced %d neurons, expected %d" % (self.layer_names[layer_index], n, width))
        return _hook

    def add_image_features(self, feats):
        """dissector image embeddings of the current batch (utils.py:329-331), rows cursor..cursor+B."""
        B = feats.shape[0]
        self.E_img[self.cursor:self.cursor + B].copy_(feats)

    def advance(self, batch):
        self.cursor += int(batch)
        if self.cursor > self.n_local:
            raise RuntimeError("more images than the dissector was sized for")

    # ---- collectives ---------------------------------------------------------------------------
    def _all_gather_rows(self, t):
        """[r, c] on every rank -> [world*r, c], rank-major."""
        if self.world == 1:
            return t
        return self._gather(t.contiguous())

    def _all_gather_ragged(self, t, lens):
        """Rank r contributes the first lens[r] rows of its [>= lens[r], c] tensor -> [sum(lens), c], rank-major.
        Shorter contributions are zero-padded to max(lens) rows for the fixed-size all-gather and the padding is
        cut away on arrival (nothing is reduced, so the padding never meets the data)."""
        if self.world == 1:
            return t[:lens[0]]
        m = max(lens)
        mine = lens[self.rank]
        if t.shape[0] != m:
            pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[:mine] = t[:mine]
            t = pad
        full = self._gather(t.contiguous())
        if all(n == m for n in lens):
            return full
        return torch.cat([full[r * m:r * m + lens[r]] for r in range(self.world)], dim=0)

    # ---- scoring side ----------------------------------------------------------------------------
    def finish(self, E_txt, k_desc=10, k_img=5, marks=None):
        """Score every neuron of every layer.  E_txt: [C, D] text embeddings (replicated on every rank).
        marks: optional callable(name) invoked between stages (the bench records HIP events there)."""
        ops = self.ops
        mark = marks if marks is not None else (STAGE_MARK if STAGE_MARK is not None else (lambda name: None))
        if self.cursor != self.n_local:
            raise RuntimeError("dissector holds %d of %d images" % (self.cursor, self.n_local))
        G, N_l, K = self.world, self.n_local, self.top_k
        if K > self.n_total or k_img > self.n_total:
            raise RuntimeError("selected index k out of range")
        with torch.no_grad():
            # utils.py:577-594 on this rank's images
            mark("start")
            fused_exp = self.gemm_mode == "bf16"
            T = E_txt.to(self.device, torch.float32)
            if not fused_exp:
                T = ops.normalize_rows(T)
            rinv = None
            if fused_exp:
                # the stress chain (configs[4]): K1 + K2 as ONE bf16-MFMA kernel that writes E = bf16(exp(a (P - 1))) and the
                # reciprocal row sums -- fp32 P is never written, S = E * rinv is never materialised (no parity claim)
                if N_l > 0:           # K1a folded in: the rows are normalised while they are converted to bf16
                    mark("gemm:begin")    # the whole call: conversion + GEMM kernel + row-sum finish
                    S, rinv = ops.embed_gemm_exp(self.E_img, T, self.a, normalize=True)   # [N_l, C] bf16 view, rows padded to 128
                    mark("gemm:end")
                    ldS = S.stride(0)
                else:
                    ldS = _round_up(self.C, 128)
                    S = torch.zeros((0, ldS), dtype=torch.bfloat16, device=self.device)[:, :self.C]
                    rinv = torch.zeros((0,), dtype=torch.float32, device=self.device)
                mark("gemm")
            elif N_l > 0:
                I = ops.normalize_rows(self.E_img)
                mode = {"bf16_p": "bf16"}.get(self.gemm_mode, self.gemm_mode)
                mark("gemm:begin")        # K1 alone: one kernel launch (the stage above it also holds K1a x 2 and host gaps)
                P = ops.embed_gemm(I, T, mode=mode) if mode != "f32" else ops.embed_gemm(I, T)
                mark("gemm:end")
                mark("gemm")
                mark("softmax:begin")     # K2 alone
                S = ops.row_softmax(P, self.a)                   # [N_l, C] view, leading dim padded
                mark("softmax:end")
                ldS = S.stride(0)
            else:                                                # a rank without images (N < G) only takes part
                mark("gemm")
                ldS = _round_up(self.C, 192)
                S = torch.zeros((0, ldS), dtype=torch.float32, device=self.device)[:, :self.C]
            mark("softmax")
            if G > 1:
                full = torch.as_strided(S, (N_l, ldS), (ldS, 1)) if N_l > 0 else torch.zeros(
                    (0, ldS), dtype=S.dtype, device=self.device)
                S = self._all_gather_ragged(full, self.counts)[:, :self.C]
                if fused_exp:
                    rinv = self._all_gather_ragged(rinv.view(-1, 1), self.counts).view(-1)
                mark("gather_S")
            # similarity.py:55 for all layers at once (local shard), then the cross-shard merge
            Kl = min(K, N_l)
            if Kl > 0:
                mark("topk:begin")        # K3 alone (the local selection; the cross-shard merge below is a second, small launch)
                vals, idx = ops.col_topk(self.At[:, :N_l], Kl, neuron_major=True)
                mark("topk:end")
            else:
                vals = torch.zeros((self.U, 0), dtype=torch.float32, device=self.device)
                idx = torch.zeros((self.U, 0), dtype=torch.int32, device=self.device)
            if G > 1:
                # one message per rank: [U, 2*Km] = (values | global indices), the ranks' first min(K, n_r) columns valid
                kls = [min(K, n) for n in self.counts]
                Km = max(kls)
                packed = torch.zeros((self.U, 2 * Km), dtype=torch.float32, device=self.device)
                packed[:, :Kl] = vals
                # (int32 indices carried as float32 BITS: only ever copied -- an all-gather moves bytes.  A reducing collective
                # (all-reduce, reduce-scatter) would do arithmetic on these bit patterns and destroy them: never swap one in.)
                packed[:, Km:Km + Kl] = (idx + self.row0).view(torch.float32)
                allp = self._all_gather_rows(packed).view(G, self.U, 2 * Km)
                cand_v = torch.cat([allp[r, :, :kls[r]] for r in range(G)], dim=1).contiguous()
                cand_i = torch.cat([allp[r, :, Km:Km + kls[r]] for r in range(G)], dim=1).contiguous().view(torch.int32)
                vals, pos = ops.col_topk(cand_v, K, neuron_major=True)  # ties -> lower position = lower image index
                idx = torch.gather(cand_i, 1, pos.long())
            mark("topk")
            # similarity.py:59-65: neurons split over the ranks
            per = (self.U + G - 1) // G
            u0, u1 = min(self.rank * per, self.U), min((self.rank + 1) * per, self.U)
            pdge_l = torch.empty((per, self.C), dtype=torch.float32, device=self.device)
            if u1 - u0 < per:
                pdge_l[u1 - u0:].zero_()         # rows of the all-gather message that no neuron of this rank fills
            mark("wpmi:begin")         # K4 alone (the stage above it also holds host gaps and the index bookkeeping)
            if u1 > u0 and fused_exp:
                ops.wpmi_score_bf16(S, rinv, idx[u0:u1].contiguous(), self.p, self.min_prob, self.p is not None,
                                    out=pdge_l[:u1 - u0])
            elif u1 > u0:
                ops.wpmi_score(S, idx[u0:u1].contiguous(), self.p, self.min_prob, self.p is not None, out=pdge_l[:u1 - u0],
                               s_is_prob=self.p_ok)   # S is this pipeline's own softmax output (NaN rows stay NaN)
            mark("wpmi:end")
            mark("wpmi")
            pdge = self._all_gather_rows(pdge_l)[:self.U] if G > 1 else pdge_l
            # similarity.py:70-72 per layer; lam*prob_d is a float32 multiply by the Python scalar
            lam32 = float(torch.tensor(self.lam, dtype=torch.float32))
            mark("logsumexp:begin")    # K5 alone
            sim = ops.logsumexp_sub(pdge, lam32, seg_offsets=self.offsets)
            mark("logsumexp:end")
            mark("logsumexp")
            # describe_broad_neurons.py:101-102
            mark("row_topk:begin")     # K6 alone
            v, ids = ops.row_topk(sim, min(k_desc, self.C))
            mark("row_topk:end")
            mark("row_topk")
        return DissectResult(self.layer_names, self.layer_widths, sim, v, ids, idx[:, :k_img].contiguous(),
                             vals[:, :k_img].contiguous(), self.n_total)


    def finish_graphed(self, E_txt, k_desc=10, k_img=5):
        """finish() as ONE hipGraph launch (single rank): the ~12 kernels of the scoring side are captured once --
        nothing inside the C ABI allocates or synchronises, and torch's allocations during capture come from the
        graph's private pool -- and replayed on later calls with the new text embeddings copied into the captured
        input.  Same bits as finish().  The returned tensors are the graph's output buffers: they are overwritten by
        the next replay."""
        if self.world > 1:
            return self.finish(E_txt, k_desc=k_desc, k_img=k_img)
        key = (int(k_desc), int(k_img))
        if getattr(self, "_graph_key", None) != key:
            self._E_static = torch.empty((self.C, self.D), dtype=torch.float32, device=self.device)
            self._E_static.copy_(E_txt)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):                      # warm-up outside the capture (first-call attributes)
                self.finish(self._E_static, k_desc=k_desc, k_img=k_img)
            torch.cuda.current_stream(self.device).wait_stream(side)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._graph_out = self.finish(self._E_static, k_desc=k_desc, k_img=k_img)
            self._graph_key = key
        self._E_static.copy_(E_txt)
        self._graph.replay()
        return self._graph_out


# ---- CSV cell formatting -----------------------------------------------------------------------------
# pandas writes an ndarray cell as str(ndarray), i.e. numpy's array2string: ~50 us of Python per cell, 0.46 s for
# the 2 x 9216 cells of a ViT-B run (13 % of a whole dissection step).  The two helpers below produce the SAME
# characters for the two cell shapes the drivers emit -- 1-D float32 (similarities) and 1-D int64 (image ids) --
# by calling numpy's own per-element formatter (dragon4, via np.format_float_positional with array2string's
# settings) and re-doing only the padding and the 75-column wrapping of numpy/_core/arrayprint.py.  Any row
# outside the plain regime (non-finite, zeros, values that make numpy switch to exponent notation, long rows)
# is handed to str(row) itself, so the output is numpy's by construction; tests/test_host_logic_cpu.py compares
# the two on tens of thousands of random rows.
_F32_1E8, _F32_1EM4, _F32_1E3 = 9.9e7, 1.001e-4, 999.0   # conservative: anything near numpy's thresholds falls back


_host_lib = None


def _load_host_lib():
    """csrc/libmcd_host.so (plain C, built by the same Makefile): numpy's float32 row formatting, natively.
    Optional: without it the Python formatter below produces the same characters, 30x slower."""
    global _host_lib
    if _host_lib is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmcd_host.so")
        try:
            L = ctypes.CDLL(path)
            for fn in (L.mcd_fmt_f32_rows, L.mcd_fmt_i64_rows):
                fn.restype = None
                fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
            L.mcd_csv_og_rows.restype = ctypes.c_int64
            L.mcd_csv_og_rows.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
            _host_lib = L
        except OSError:
            _host_lib = False
    return _host_lib


def format_f32_rows(a, native=True):
    """[str(row) for row in a] for a 2-D float32 array, fast."""
    import numpy as np
    a = np.asarray(a)
    if a.dtype != np.float32 or a.ndim != 2 or a.shape[1] == 0 or a.shape[1] > 64:
        return [str(r) for r in a]
    L = _load_host_lib() if native else False
    if L:
        a = np.ascontiguousarray(a)
        rows, n = a.shape
        cap = 8 + n * 24 + 2 * (n // 3 + 2)
        buf = np.empty((rows, cap), np.uint8)
        lens = np.empty((rows,), np.int32)
        L.mcd_fmt_f32_rows(a.ctypes.data, rows, n, buf.ctypes.data, cap, lens.ctypes.data)
        raw = buf.tobytes()
        out = []
        for r in range(rows):
            k = lens[r]
            out.append(raw[r * cap:r * cap + k].decode("ascii") if k >= 0 else format_f32_rows(a[r:r + 1], native=False)[0])
        return out
    fmt = np.format_float_positional
    absa = np.abs(a.astype(np.float64))
    with np.errstate(all="ignore"):
        ok = np.isfinite(a).all(1) & (absa.min(1) >= _F32_1EM4) & (absa.max(1) < _F32_1E8) \
            & (absa.max(1) / absa.min(1) <= _F32_1E3)
    out = []
    for r in range(a.shape[0]):
        row = a[r]
        if not ok[r]:
            out.append(str(row))
            continue
        parts = [fmt(x, precision=8, unique=True, fractional=True, trim=".", min_digits=0).split(".") for x in row]
        pl = max(len(p[0]) for p in parts)
        pr = max(len(p[1]) for p in parts)
        words = [p[0].rjust(pl) + "." + p[1].ljust(pr) for p in parts]
        s, line = "", " "
        for i, w in enumerate(words):            # numpy _extendLine: wrap when the word would pass column 74
            if len(line) + len(w) > 74 and len(line) > 1:
                s += line.rstrip() + "\n"
                line = " "
            line += w
            if i + 1 < len(words):
                line += " "
        s += line
        out.append("[" + s[1:] + "]")
    return out


def format_i64_rows(a, native=True):
    """[str(row) for row in a] for a 2-D int64 array, fast."""
    import numpy as np
    a = np.asarray(a)
    if a.dtype != np.int64 or a.ndim != 2 or a.shape[1] == 0:
        return [str(r) for r in a]
    L = _load_host_lib() if (native and a.shape[1] <= 64) else False
    if L:
        a = np.ascontiguousarray(a)
        rows, n = a.shape
        cap = 80
        buf = np.empty((rows, cap), np.uint8)
        lens = np.empty((rows,), np.int32)
        L.mcd_fmt_i64_rows(a.ctypes.data, rows, n, buf.ctypes.data, cap, lens.ctypes.data)
        raw = buf.tobytes()
        return [raw[r * cap:r * cap + lens[r]].decode("ascii") if lens[r] >= 0 else format_i64_rows(a[r:r + 1], native=False)[0]
                for r in range(rows)]
    out = []
    for row in a.tolist():
        strs = [str(v) for v in row]
        w = max(len(t) for t in strs)
        if (w + 1) * len(strs) + 1 > 74:
            out.append(str(np.asarray(row, dtype=np.int64)))
            continue
        out.append("[" + " ".join(t.rjust(w) for t in strs) + "]")
    return out


class _Cell(str):
    """A pre-formatted cell: pandas' CSV writer calls str() on object cells, which returns the text unchanged."""
    __slots__ = ()


def results_to_dataframe(result, words, variant="og", fast_format=True):
    """The drivers' `outputs` dict -> pandas DataFrame, column for column as the reference builds it.

    variant 'og'  : describe_og_neurons.py:77-122 / describe_broad_neurons.py:79-122 -- description = list of
                    k strings, similarity = float32[k], images = int64[k_img]
    variant 'clip': describe_clip_neurons.py:49-84 -- description = one string, similarity = float32 scalar
    numpy/pandas do the formatting (hard part 5): with fast_format the array cells are
    pre-rendered by format_f32_rows / format_i64_rows, which reproduce str(ndarray) character for character.
    """
    import pandas as pd
    vals = result.vals.cpu().numpy()
    ids = result.ids.cpu().numpy()
    top_ids = result.top_ids.cpu().numpy().astype("int64")   # torch.topk indices are int64 in the reference
    outputs = {"layer": [], "unit": [], "description": [], "similarity": [], "images": []}
    img_cells = [_Cell(t) for t in format_i64_rows(top_ids)] if fast_format else None
    sim_cells = [_Cell(t) for t in format_f32_rows(vals)] if (fast_format and variant != "clip") else None
    for name, sl in result.layer_slices():
        n = sl.stop - sl.start
        outputs["unit"].extend([i for i in range(n)])
        outputs["layer"].extend([name] * n)
        if variant == "clip":
            outputs["description"].extend([words[int(i)] for i in ids[sl, 0]])
            outputs["similarity"].extend(vals[sl, 0])
        else:
            outputs["description"].extend([[words[int(i)] for i in row] for row in ids[sl]])
            outputs["similarity"].extend(sim_cells[sl] if fast_format else vals[sl])
        outputs["images"].extend(img_cells[sl] if fast_format else top_ids[sl])
    return pd.DataFrame(outputs)


def write_descriptions_csv(result, words, path_or_buf, variant="og"):
    """DataFrame(outputs).to_csv(path, index=False) of the drivers (describe_og_neurons.py:122-123,
    describe_broad_neurons.py:122-172), written directly.  For the og/broad variant every cell is already text (layer
    name, the list of k concept strings, the pre-rendered similarity and image arrays) or an int (unit), so the rows go
    straight to Python's csv.writer with pandas' dialect (QUOTE_MINIMAL, '"', os.linesep) -- the same writer pandas
    drives, minus the DataFrame in between; the bytes are identical (tests/test_host_logic_cpu.py compares them with
    DataFrame.to_csv and with the reference-made golden CSV).  The clip variant (a float32 column that pandas formats
    itself) goes through pandas."""
    import csv
    import os
    if variant == "clip":
        results_to_dataframe(result, words, variant).to_csv(path_or_buf, index=False)
        return
    import ctypes
    import numpy as np
    vals = np.ascontiguousarray(result.vals.cpu().numpy(), dtype=np.float32)
    ids = np.ascontiguousarray(result.ids.cpu().numpy(), dtype=np.int32)
    top_ids = np.ascontiguousarray(result.top_ids.cpu().numpy().astype("int64"))
    wl = [repr(w) for w in words]          # str(list_of_str) == "[" + ", ".join(map(repr, list)) + "]"
    own = isinstance(path_or_buf, (str, bytes, os.PathLike))
    # ---- native path: whole rows assembled in csrc/mcd_host.c (same dialect), one write ----
    L = _load_host_lib()
    if L and os.linesep == "\n" and vals.ndim == 2 and vals.shape[1] <= 64 and top_ids.shape[1] <= 64:
        enc = [w.encode("utf-8") for w in wl]
        arr = (ctypes.c_char_p * len(enc))(*enc)
        lens = np.array([len(b) for b in enc], np.int32)
        k, k_img = vals.shape[1], top_ids.shape[1]
        per_row = 2 * (k * (int(lens.max()) + 2) + 4) + 3 * 4096 + 256
        chunks = [b"layer,unit,description,similarity,images\n"]
        ok = True
        for name, sl in result.layer_slices():
            n = sl.stop - sl.start
            if n == 0:
                continue
            nb = name.encode("utf-8")
            cap = n * (2 * len(nb) + 64 + 2 * (k * (int(lens.max()) + 2) + 4) + 48 * k + 24 * k_img) + per_row
            buf = ctypes.create_string_buffer(cap)
            got = L.mcd_csv_og_rows(nb, len(nb), n, ids[sl].ctypes.data, k, arr, lens.ctypes.data, len(enc),
                                    vals[sl].ctypes.data, top_ids[sl].ctypes.data, k_img, buf, cap)
            if got < 0:
                ok = False
                break
            chunks.append(buf.raw[:got])
        if ok:
            data = b"".join(chunks)
            if own:
                with open(path_or_buf, "wb") as f:
                    f.write(data)
            else:
                path_or_buf.write(data.decode("utf-8"))
            return
    # ---- csv-module path (rows that need numpy's own formatting, non-Linux line ends, no libmcd_host.so) ----
    sim_cells = format_f32_rows(vals)
    img_cells = format_i64_rows(top_ids)
    f = open(path_or_buf, "w", newline="", encoding="utf-8") if own else path_or_buf
    try:
        w = csv.writer(f, lineterminator=os.linesep, quoting=csv.QUOTE_MINIMAL, quotechar='"')
        w.writerow(["layer", "unit", "description", "similarity", "images"])
        for name, sl in result.layer_slices():
            block = ids[sl].tolist()
            w.writerows((name, u, "[" + ", ".join([wl[i] for i in row]) + "]", sim_cells[sl.start + u],
                         img_cells[sl.start + u]) for u, row in enumerate(block))
    finally:
        if own:
            f.close()
