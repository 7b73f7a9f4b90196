/*
 * mcd_hip.h -- C ABI of libmcd_hip.so: the MI355X (gfx950) dissection core of
 * Mammo-CLIP-Dissect, hand-written HIP.
 *
 * The reference is pure Python: its "operator interface" for this path is the set of torch
 * calls inside concept_vit/similarity.py and around it in concept_vit/utils.py and the
 * describe_*_neurons.py drivers.  Each entry point below replaces one of those call sites
 * (file:line relative to the reference root).  Plain pointers and sizes only: every pointer
 * is a DEVICE pointer (HBM) unless stated, the caller owns every buffer, nothing is
 * allocated or synchronised inside, every call is asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = the default stream) and is hipGraph-capturable.
 *
 * Return value: 0 on success; negative on error (MCD_E_*), with a human readable message in
 * mcd_last_error() (thread local).  No call ever falls back to a CPU path.
 *
 * Layout vocabulary:
 *   N images, C concepts, D embedding width, U neurons (of one layer or of all layers
 *   concatenated), K = top_k activating images per neuron.
 *   "image-major"  [N, U]: element (n,u) at base[n*ld + u]   (what torch.cat of hook outputs gives)
 *   "neuron-major" [U, N]: element (n,u) at base[u*ld + n]   (what the fused pipeline keeps in HBM)
 */
#ifndef MCD_HIP_H
#define MCD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mcd_stream_t; /* hipStream_t */

enum {
    MCD_OK = 0,
    MCD_E_ARG = -1,     /* bad shape / stride / alignment / NULL pointer */
    MCD_E_RANGE = -2,   /* k out of range (torch: "selected index k out of range") */
    MCD_E_WORKSPACE = -3,
    MCD_E_LAUNCH = -4,  /* hipGetLastError() after a launch */
    MCD_E_UNSUPPORTED = -5
};

/* GEMM arithmetic modes for mcd_embed_gemm */
enum {
    MCD_GEMM_F32 = 0,     /* v_mfma_f32_32x32x2_f32: exact fp32 fma chains over MKL's K-blocks (parity mode:
                             bit-identical to torch's CPU matmul for D <= 768 and D = 1024) */
    MCD_GEMM_BF16X3 = 1,  /* split-bf16 hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate */
    MCD_GEMM_BF16 = 2     /* single-pass bf16 MFMA (stress config only; no parity claim) */
};

/* bits of the `soft` argument of mcd_wpmi_score */
enum { MCD_WPMI_SOFT = 1, MCD_WPMI_FAST_LOG = 2, MCD_WPMI_S_IS_PROB = 4 };

/* hook pooling modes for mcd_hook_pool */
enum { MCD_POOL_AVG = 0, MCD_POOL_MAX = 1, MCD_POOL_CLS = 2, MCD_POOL_NONE = 3 };

const char* mcd_last_error(void);
int mcd_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * K1a  rows of x scaled to unit L2 norm:  y[r,:] = x[r,:] / sqrt(sum_k x[r,k]^2)
 * replaces  image_features /= image_features.norm(dim=-1, keepdim=True)   concept_vit/utils.py:577
 *           text_features  /= text_features.norm(dim=-1, keepdim=True)    concept_vit/utils.py:578
 *           (same lines: og_utils.py:485-486, CLIP_og_utils.py:158-159)
 * y may alias x (the reference normalises in place).
 * ------------------------------------------------------------------------------------------- */
int mcd_normalize_rows(const float* x, int64_t ldx, int64_t n, int64_t d, float* y, int64_t ldy,
                       mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K7   per row r (length n):  d = x - mean(x);  c = d^3;  y = c / max(||c||_2, min_norm)
 * replaces  x - mean(dim=0); x**3; x / clip(norm(dim=0), min_norm)         concept_vit/similarity.py:15-22
 *           (cos_similarity_cubed; rows here are the reference's columns: the matrices are passed
 *           neuron-major / concept-major so that the image axis is contiguous).  y may alias x.
 * ------------------------------------------------------------------------------------------- */
int mcd_center_cube_normalize_rows(const float* x, int64_t ldx, int64_t rows, int64_t n, float min_norm,
                                   float* y, int64_t ldy, mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K1   P[n,c] = sum_k I[n,k] * T[c,k]      (I: [N,D] ld ldi, T: [C,D] ld ldt, P: [N,C] ld ldp)
 * replaces  clip_feats = image_features @ text_features.T                 concept_vit/utils.py:594
 *           (og_utils.py:501, CLIP_og_utils.py:160)
 * ws: optional scratch of mcd_embed_gemm_workspace() bytes (0 for MCD_GEMM_F32 and for small problems).  With
 * it, the bf16 modes convert the operands to bf16 once and run the 256x256-tile kernel (the stress shape);
 * without it (NULL / too small) they run the 128x128 kernel that converts while staging -- same results per mode.
 * ------------------------------------------------------------------------------------------- */
size_t mcd_embed_gemm_workspace(int64_t N, int64_t C, int64_t D, int mode);
int mcd_embed_gemm(const float* I, int64_t ldi, const float* T, int64_t ldt, int64_t N, int64_t C, int64_t D,
                   int mode, float* P, int64_t ldp, void* ws, size_t ws_bytes, mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2   S[n,c] = softmax_c(a * P[n,c]);  columns C..lds-1 of S are written as 0 (padding).
 * replaces  clip_feats = torch.nn.functional.softmax(a*clip_feats, dim=1)  concept_vit/similarity.py:54, :80
 * ------------------------------------------------------------------------------------------- */
int mcd_row_softmax(const float* P, int64_t ldp, int64_t N, int64_t C, float a, float* S, int64_t lds,
                    mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3   per neuron u: the K largest activations over the N images, sorted descending; ties go to
 *      the lower image index; NaN ranks above +inf (torch.topk's rule).
 * replaces  inds = torch.topk(target_feats, dim=0, k=top_k)[1]             concept_vit/similarity.py:55, :82
 *           _, top_ids = torch.topk(target_feats, k=5, dim=0)              describe_clip_neurons.py:66
 *                                                                          describe_og_neurons.py:100
 *                                                                          describe_broad_neurons.py:102
 * A element (n,u) at A[n*stride_n + u*stride_u]; exactly one of the strides must be 1
 * (image-major: stride_u == 1; neuron-major: stride_n == 1).
 * Outputs are NEURON-major: vals[u*ldo + j], idx[u*ldo + j], j < K (either may be NULL).
 * Image-major input is transposed through `ws` (mcd_col_topk_workspace bytes; 0 for neuron-major).
 * Returns MCD_E_RANGE when K > N (torch raises "selected index k out of range").
 * ------------------------------------------------------------------------------------------- */
size_t mcd_col_topk_workspace(int64_t N, int64_t U, int64_t stride_n, int64_t stride_u, int K);
int mcd_col_topk(const float* A, int64_t N, int64_t U, int64_t stride_n, int64_t stride_u, int K, float* vals,
                 int32_t* idx, int64_t ldo, void* ws, size_t ws_bytes, mcd_stream_t stream);

/* image-major [N,U] -> neuron-major [U,N] (dst ld ldd >= N).  Used by K3 and by the activation cache. */
int mcd_transpose(const float* src, int64_t lds, int64_t N, int64_t U, float* dst, int64_t ldd,
                  mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4   pdge[u,c] = sum_{j<K} log(w_j),  g = S[idx[u,j], c]
 *        soft != 0:  w_j = (1 + p[j]*(g - 1)) + min_prob      concept_vit/similarity.py:59-65
 *        soft == 0:  w_j = g + min_prob                        concept_vit/similarity.py:84-88
 *      every fp32 operation rounded on its own (no contraction); the sum over j follows
 *      torch.sum(dim=0) on CPU: columns c < split in ATen's cascade order, columns c >= split in
 *      its row_sum order (4 interleaved partials).  split < 0 selects ATen's rule for C
 *      ((C/32)*32 for C >= 8, (C/4)*4 below).
 * replaces the Python loop `for orig_id in tqdm(range(target_feats.shape[1]))` with its
 *      gather / log / sum(dim=0) / cat.
 * idx is neuron-major int32 [U, K] (ld ldidx), every entry in [0, N); p is [K] (ignored for hard WPMI).
 * `soft` is a bit set: bit 0 = soft-WPMI terms; bit 1 (MCD_WPMI_FAST_LOG) = use the v_log_f32 based log
 * (<= ~1.5 ulp) instead of the default accurate log (near correctly rounded, like the reference's MKL vsLn);
 * bit 2 (MCD_WPMI_S_IS_PROB) = the caller promises that S holds probabilities in [0,1] (a softmax output) and p
 * lies in [0,1], which lets the kernel skip the range check in front of its log table.  NaN entries of S (softmax
 * rows of NaN/inf similarities) are allowed: they propagate as NaN through the arithmetic.  Finite values outside [0,1]
 * break the promise and yield unspecified results.
 * ------------------------------------------------------------------------------------------- */
int mcd_wpmi_score(const float* S, int64_t ldS, int64_t N, int64_t C, const int32_t* idx, int64_t ldidx, int64_t U,
                   int K, const float* p, float min_prob, int soft, int split, float* pdge, int64_t ldo,
                   mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K1s + K4s  the STRESS chain (BASELINE configs[4]: 10 000 concepts, bf16 MFMA similarity; no parity claim).
 * K1 and K2 in one kernel that never writes fp32 P:
 *      E[n,c]  = bf16( exp(a * (P[n,c] - 1)) ),  P = I_hat @ T_hat^T on the bf16 MFMA (fp32 accumulate)
 *      rinv[n] = 1 / sum_c exp(a * (P[n,c] - 1))
 *   so that softmax(a*P)[n,c] = E[n,c] * rinv[n].  I and T must be row-normalised (|P| <= 1: no row maximum is needed),
 *   or raw with MCD_GEMM_EXP_NORMALIZE in `flags`;
 *   E is [N, ldE] bf16, ldE a multiple of 16 (anything else: MCD_E_UNSUPPORTED; K4s wants a multiple of 128), columns
 *   C..ldE-1 written as 0; rinv[n] = 1 / (the sum of row n's STORED bf16 values), so E * rinv sums to 1 over a row; ws of
 *   mcd_embed_gemm_exp_workspace() bytes holds the bf16 operands and the per-tile partial row sums.
 * replaces  clip_feats = image_features @ text_features.T                 concept_vit/utils.py:594
 *           clip_feats = torch.nn.functional.softmax(a*clip_feats, dim=1)  concept_vit/similarity.py:54, :80
 * K4 on that representation:
 *      pdge[u,c] = sum_j log(w_j),  g = E[idx[u,j], c] * rinv[idx[u,j]],  w_j as in mcd_wpmi_score;
 *   v_log_f32-based log, sums kept in the log2 domain.  ldE % 128 == 0.
 * replaces  the Python loop of soft_wpmi / wpmi                            concept_vit/similarity.py:59-65, :84-88
 * ------------------------------------------------------------------------------------------- */
enum { MCD_GEMM_EXP_NORMALIZE = 1 };  /* flags: I and T are RAW embeddings; rows are L2-normalised while they are converted
                                         to bf16 (utils.py:577-578 folded in; D <= 2048) */
size_t mcd_embed_gemm_exp_workspace(int64_t N, int64_t C, int64_t D);
int mcd_embed_gemm_exp(const float* I, int64_t ldi, const float* T, int64_t ldt, int64_t N, int64_t C, int64_t D,
                       float a, int flags, uint16_t* E, int64_t ldE, float* rinv, void* ws, size_t ws_bytes,
                       mcd_stream_t stream);
/* Measurement hook (bench.py; not on the data path): after mcd_embed_gemm_exp_time_kernel(reps), reps > 0, mcd_embed_gemm_exp
 * launches its GEMM kernel `reps` times back to back (same arguments, same output) between a pair of HIP events on its stream --
 * the kernel alone, not the bf16 conversion in front of it nor the row-sum finish behind it; mcd_embed_gemm_exp_kernel_ms() waits
 * for the pair of the calling thread's current device and returns the elapsed milliseconds PER LAUNCH of the last timed call
 * (< 0: none recorded).  One launch between two events reads 10-25 us long (marker packets, dispatch gaps): use reps >= 8.
 * reps = 0 (the default) turns it off: the entry point then neither creates events nor synchronises, and stays capturable in a
 * hipGraph.  (One kernel serves every shape since round 5; the quotient uses the number of launches actually issued.) */
int mcd_embed_gemm_exp_time_kernel(int reps);
float mcd_embed_gemm_exp_kernel_ms(void);
size_t mcd_wpmi_score_bf16_workspace(int64_t U, int K);   /* {row, p_j * rinv[row]} per (neuron, j): 8 U K bytes */
int mcd_wpmi_score_bf16(const uint16_t* E, int64_t ldE, int64_t N, int64_t C, const float* rinv, const int32_t* idx,
                        int64_t ldidx, int64_t U, int K, const float* p, float min_prob, int soft, float* pdge,
                        int64_t ldo, void* ws, size_t ws_bytes, mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K5   per segment s (= one layer, rows seg[s]..seg[s+1]-1 of pdge), per column c:
 *        prob_d = logsumexp_u(pdge[u,c]) - log(U_s);   out[u,c] = pdge[u,c] - lam*prob_d
 * replaces  prob_d = torch.logsumexp(prob_d_given_e, dim=0, keepdim=True) - torch.log(U*ones([1]))
 *           mutual_info = prob_d_given_e - lam*prob_d                     concept_vit/similarity.py:70-72, :92-96
 * seg_offsets is a HOST array of n_seg+1 row offsets (n_seg <= 64); out may alias pdge.
 * ws: device scratch of mcd_logsumexp_sub_workspace(total rows, C, n_seg) bytes (column maxima and the
 * 16-row partial sums that ATen's summation order chains).
 * ------------------------------------------------------------------------------------------- */
size_t mcd_logsumexp_sub_workspace(int64_t U_total, int64_t C, int n_seg);
int mcd_logsumexp_sub(const float* pdge, int64_t ld, int64_t C, const int64_t* seg_offsets, int n_seg, float lam,
                      int split, float* out, int64_t ldo, void* ws, size_t ws_bytes, mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K6   per row u of sim [U,C]: the k largest entries, sorted descending, ties to the lower
 *      concept index.  k = 1 is torch.max(dim=1).  k <= 16.
 * replaces  vals, ids = torch.max(similarities, dim=1)                    describe_clip_neurons.py:64
 *           vals, ids = torch.topk(similarities, k=10, dim=1)             describe_og_neurons.py:99
 *                                                                         describe_broad_neurons.py:101
 * vals/idx are [U,k] contiguous.
 * ------------------------------------------------------------------------------------------- */
int mcd_row_topk(const float* sim, int64_t ld, int64_t U, int64_t C, int k, float* vals, int32_t* idx,
                 mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K0   forward-hook pooling written straight into the activation matrix.
 *      x is the hooked tensor: [B, Cout, HW] for AVG/MAX (mean / amax over HW),
 *      [B, T, F] for CLS (token 0: x[b,0,:], Cout = F, HW = T), [B, F] for NONE (HW = 1).
 *      Result (b, ch) goes to dst[(row0+b)*stride_n + (col0+ch)*stride_u].
 * replaces  get_activation(outputs, mode) hook bodies                     concept_vit/utils.py:27-52
 *           (og_utils.py:31-56, CLIP_og_utils.py:13-36) and the later torch.cat (utils.py:143).
 * ------------------------------------------------------------------------------------------- */
int mcd_hook_pool(const float* x, int64_t B, int64_t Cout, int64_t HW, int mode, float* dst, int64_t row0,
                  int64_t col0, int64_t stride_n, int64_t stride_u, mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K8   rank_reorder scoring of one layer.
 *      tvals/tidx are neuron-major [U, top_n] (ld ldt): the top_n largest activations of every neuron in
 *      descending order and their image indices (mcd_col_topk's output); perms is int32 [U, n_perm, top_n],
 *      the random permutations of the baseline (the reference draws them with torch.randperm on the global CPU
 *      generator, 5 per neuron in neuron order -- the host mirror does the same so a seeded run reproduces the
 *      reference); baseline_ws is caller-owned scratch of U floats.
 *      out[u,c] = -( mean_j |t_j - st[rank_jc]|^p / baseline_u ) / mean_j(P[idx_j, c])^scale_p
 * replaces  the per-neuron Python loop of rank_reorder                    concept_vit/similarity.py:107-132
 * ------------------------------------------------------------------------------------------- */
int mcd_rank_reorder(const float* P, int64_t ldP, int64_t N, int64_t C, const float* tvals, const int32_t* tidx,
                     int64_t ldt, int64_t U, int top_n, const int32_t* perms, int n_perm, float p, float scale_p,
                     float* baseline_ws, float* out, int64_t ldo, mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K9   fp32 multi-head self-attention of the ViT image tower (head dimension 64, no mask, T <= 256 tokens).
 *      qkv is the fused projection's output [B, T, 3, H, 64] (q, k, v of a token adjacent), out is [B, T, H*64]:
 *      out[b, t, h, :] = softmax_j(q[b,t,h].k[b,j,h] / 8) v[b,j,h].  One workgroup per (image, head), K and V in
 *      LDS, flash-style online softmax, v_mfma_f32_32x32x2_f32 for both products.  Both pointers 16-byte aligned.
 *      Encoder-side op (the forwards that the extraction loop drives, concept_vit/utils.py:117-148): fp32-accurate
 *      (<= 2e-6 from torch's SDPA), no bit-exactness claim -- the reference's own encoders run on whatever
 *      backend torch picks.
 * replaces  the attention inside ViTModel(...)                             model/modules/image_encoder.py:37
 *           nn.MultiheadAttention(x, x, x, need_weights=False)            concept_vit/clip/model.py:171-183
 * ------------------------------------------------------------------------------------------- */
int mcd_vit_attention(const float* qkv, int64_t B, int64_t T, int64_t H, float* out, mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K10  fp32 LayerNorm over the last dimension (biased variance, like torch): rows x D contiguous, D a multiple of 4
 *      up to 2048, pointers 16-byte aligned.  One wave per row, the row register-resident, two-pass statistics.
 *      Encoder-side op, fp32-accurate (<= 1e-6 relative from torch's), no bit-exactness claim.
 * replaces  nn.LayerNorm inside the ViT blocks                              model/modules/image_encoder.py:37
 *           (ViTLayer.layernorm_before / layernorm_after), ln_1 / ln_2      concept_vit/clip/model.py:172-176
 * ------------------------------------------------------------------------------------------- */
int mcd_layer_norm(const float* x, int64_t rows, int64_t D, const float* gamma, const float* beta, float eps, float* y,
                   mcd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K11  patch extraction for the ViT patch embedding: x [B, Cin, H, W] -> out [B, 1 + (H/P)(W/P), Cin*P*P], row 0 of
 *      every image zero (class-token slot), row 1 + patch in (c, dy, dx) order, so that the Conv2d(Cin, dim, P, P) is
 *      the GEMM  out . weight.view(dim, Cin*P*P)^T  (P, W multiples of 4; pointers 16-byte aligned).  A permutation
 *      of the pixels: exact.
 * replaces  the patch-embedding convolution of the image tower              model/modules/image_encoder.py:37
 *           (ViTPatchEmbeddings.projection), conv1                          concept_vit/clip/model.py:206-223
 * ------------------------------------------------------------------------------------------- */
int mcd_patchify(const float* x, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t P, float* out, mcd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MCD_HIP_H */
