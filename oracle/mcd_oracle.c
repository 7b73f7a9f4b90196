/*
 * mcd_oracle.c -- CPU restatement of the Mammo-CLIP-Dissect dissection core.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (mammo-clip-dissect_amd/) never links, imports or falls back to it.
 *
 * Every function restates, in plain fp32 C with the reference's operation order,
 * one step of the reference's hot path (paths relative to /root/reference):
 *
 *   mcd_o_normalize_rows   concept_vit/utils.py:577-578   x /= x.norm(dim=-1, keepdim=True)
 *   mcd_o_gemm_nt          concept_vit/utils.py:594       clip_feats = image_features @ text_features.T
 *   mcd_o_row_softmax      concept_vit/similarity.py:54   softmax(a*clip_feats, dim=1)
 *   mcd_o_col_topk         concept_vit/similarity.py:55   topk(target_feats, dim=0, k)
 *   mcd_o_wpmi_score       concept_vit/similarity.py:59-65 (soft_wpmi) / :84-88 (wpmi)
 *   mcd_o_logsumexp_sub    concept_vit/similarity.py:70-72 / :92-96
 *   mcd_o_row_topk         concept_vit/describe_clip_neurons.py:64 (max) /
 *                          describe_broad_neurons.py:101 (topk k=10, dim=1)
 *   mcd_o_hook_pool        concept_vit/utils.py:27-52     get_activation(): mean/amax over H,W
 *
 * Pinning: the reference holds no tests or golden vectors, so
 * the pin is tests/golden/ (npz files), generated in the build container by importing
 * /root/reference/concept_vit/similarity.py (tests/golden/make_golden.py).
 * tests/test_oracle_golden.py checks this file against every one of them.
 *
 * Summation order.  torch.sum(x, dim=0) on a contiguous [R, C] fp32 tensor runs ATen's
 * cascade_sum (aten/src/ATen/native/cpu/SumKernel.cpp, torch 2.10 CPU): columns below
 * `split` use multi_row_sum (4 accumulator levels, 16-row chunks), the remaining
 * columns use row_sum (4 row-interleaved partial sums, each cascaded).  On the torch
 * build in this image the sum kernel uses 8-float vectors, so
 *     split = (C/32)*32  when C >= 8,  (C/4)*4 otherwise
 * (verified bit-exactly against torch.sum for C in {1,3,5,7,8,20,33,40,100,763}).
 * Both orders are restated below so the oracle reproduces torch's bits for the
 * sums.  softmax's exp is the restated SLEEF expf_u10 (bit-exact); torch.log / torch.exp (unary ops) are MKL's
 * high-accuracy vsLn / vsExp, practically correctly rounded: the K4 log here is the double-precision log rounded
 * to fp32 (99.99 % identical with torch.log); K5's exp/log stay libm's (<= 1 ulp).
 *
 * Build:  make -C oracle      (gcc -O2 -fopenmp -ffp-contract=off, see Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int mcd_o_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void mcd_o_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ATen's split between the multi_row_sum columns and the row_sum columns. */
int mcd_o_sum_split(int C) { return C >= 8 ? (C / 32) * 32 : (C / 4) * 4; }

/* ------------------------------------------------------------------------------------------
 * cascade_sum building blocks.  `term(i)` yields the i-th addend of one output column.
 * ---------------------------------------------------------------------------------------- */
static int ceil_log2_i64(int64_t x) {
    if (x <= 2) return 1; /* ATen utils::CeilLog2: x<=2 -> 1 */
    int l = 0;
    uint64_t v = (uint64_t)(x - 1);
    while (v) { v >>= 1; ++l; }
    return l;
}

typedef float (*term_fn)(const void* ctx, int64_t i);

/* multi_row_sum for ONE column: addends term(first + i*stride), i in [0,size). */
static float cascade_1col(term_fn term, const void* ctx, int64_t first, int64_t stride, int64_t size) {
    const int num_levels = 4;
    int lp = ceil_log2_i64(size) / num_levels;
    if (lp < 4) lp = 4;
    const int64_t step = (int64_t)1 << lp;
    const int64_t mask = step - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t i = 0;
    for (; i + step <= size;) {
        for (int64_t j = 0; j < step; ++j, ++i) acc[0] += term(ctx, first + i * stride);
        for (int j = 1; j < num_levels; ++j) {
            acc[j] += acc[j - 1];
            acc[j - 1] = 0.f;
            const int64_t m = mask << (j * lp);
            if ((i & m) != 0) break;
        }
    }
    for (; i < size; ++i) acc[0] += term(ctx, first + i * stride);
    for (int j = 1; j < num_levels; ++j) acc[0] += acc[j];
    return acc[0];
}

/* row_sum for ONE column: 4 interleaved partial sums, leftovers into partial 0. */
static float rowsum_1col(term_fn term, const void* ctx, int64_t size) {
    const int64_t q = size / 4;
    float part[4];
    for (int k = 0; k < 4; ++k) part[k] = cascade_1col(term, ctx, k, 4, q);
    for (int64_t i = q * 4; i < size; ++i) part[0] += term(ctx, i);
    for (int k = 1; k < 4; ++k) part[0] += part[k];
    return part[0];
}

static float torch_sum0_col(term_fn term, const void* ctx, int64_t size, int col, int split) {
    return col < split ? cascade_1col(term, ctx, 0, 1, size) : rowsum_1col(term, ctx, size);
}

/* torch.sum(x, dim=0) of a row-major [R, C] matrix in ATen's order (used by rank_reorder's column means,
 * similarity.py:110, and error means, :129). */
typedef struct { const float* x; int64_t ld, c; } col_ctx;
static float col_term(const void* v, int64_t i) {
    const col_ctx* k = (const col_ctx*)v;
    return k->x[i * k->ld + k->c];
}
void mcd_o_sum0(const float* x, int64_t R, int64_t C, int split, float* out /* [C] */) {
    if (split < 0) split = mcd_o_sum_split((int)C);
    for (int64_t c = 0; c < C; ++c) {
        col_ctx k = {x, C, c};
        out[c] = torch_sum0_col(col_term, &k, R, (int)c, split);
    }
}

/* ------------------------------------------------------------------------------------------
 * utils.py:577-578   image_features /= image_features.norm(dim=-1, keepdim=True)
 * ---------------------------------------------------------------------------------------- */
/* ATen's norm kernel for p = 2 over a contiguous last dim (aten/src/ATen/native/cpu/ReduceOpsKernel.cpp, the
 * "reduce_lastdim" fast path): 8 lane accumulators acc[j] = fma(x[8i+j], x[8i+j], acc[j]), the lanes added
 * sequentially from lane 0, then the d % 8 tail one element at a time -- as the torch 2.10 build of this image
 * compiled it (a loop unrolled by four): the first 4*(tail/4) elements as a rounded product plus a rounded add, the
 * last tail % 4 ones fused --
 * sqrt, and a true division per element.  Verified bit-exactly against that torch for
 * d in {1, 3, 5, 6, 7, 8, 12, 13, 15, 20, 36, 100, 512, 515, 768, 1027}. */
float mcd_o_row_norm(const float* row, int64_t d) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int64_t full = d - d % 8;
    for (int64_t k = 0; k < full; k += 8)
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(row[k + j], row[k + j], acc[j]);
    float ss = acc[0];
    for (int j = 1; j < 8; ++j) ss = ss + acc[j];
    for (int64_t k = full; k < d; ++k) {
        if (k - full < (d - full) / 4 * 4) {
            const float sq = row[k] * row[k];
            ss = ss + sq;
        } else {
            ss = fmaf(row[k], row[k], ss);
        }
    }
    return sqrtf(ss);
}

void mcd_o_normalize_rows(float* x, int64_t n, int64_t d) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        float* row = x + r * d;
        const float nrm = mcd_o_row_norm(row, d);
        for (int64_t k = 0; k < d; ++k) row[k] = row[k] / nrm;
    }
}

/* utils.py:594   clip_feats = image_features @ text_features.T
 *
 * torch's CPU matmul is MKL sgemm.  On the torch 2.10 build of this image (the one that made the golden vectors) its
 * result is, bit for bit -- verified on [256..10000] x [5..763] outputs, 1 to 8 threads --
 *     P[n,c] = chain(block 0) + chain(block 1) + ...      (added in block order)
 * where chain(block) = fma(a_k, b_k, fma(a_{k-1}, b_{k-1}, ... 0)) over the block's k in order, and the K axis is cut as
 *     K <= 384:        one block
 *     384 < K <= 768:  two blocks, the first of roundup4(ceil(K/2)) elements   (512 -> 256 + 256, 700 -> 352 + 348)
 *     K  > 768:        blocks of 384 (verified for K = 1024: 384 + 384 + 256; other K > 768 unverified)
 * mcd_o_gemm_kblocks() returns the cut; the HIP kernel (K1, MCD_GEMM_F32) follows the same rule. */
int mcd_o_gemm_kblocks(int64_t K, int64_t* sizes /* >= K/384 + 2 entries */) {
    int n = 0;
    if (K <= 384) {
        sizes[n++] = K;
    } else if (K <= 768) {
        const int64_t k1 = ((K + 1) / 2 + 3) / 4 * 4;
        sizes[n++] = k1;
        sizes[n++] = K - k1;
    } else {
        for (int64_t r = K; r > 0; r -= 384) sizes[n++] = r < 384 ? r : 384;
    }
    return n;
}

void mcd_o_gemm_nt(const float* I, const float* T, int64_t N, int64_t C, int64_t D, float* P) {
    int64_t sizes[64];
    if (D / 384 + 2 > 64) return;
    const int nb = mcd_o_gemm_kblocks(D, sizes);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        const float* a = I + n * D;
        for (int64_t c = 0; c < C; ++c) {
            const float* b = T + c * D;
            float tot = 0.f;
            int64_t k = 0;
            for (int blk = 0; blk < nb; ++blk) {
                float s = 0.f;
                for (const int64_t ke = k + sizes[blk]; k < ke; ++k) s = fmaf(a[k], b[k], s);
                tot = (blk == 0) ? s : tot + s;
            }
            P[n * C + c] = tot;
        }
    }
}

/* similarity.py:54   clip_feats = softmax(a*clip_feats, dim=1)
 *
 * Bit-exact restatement of ATen's CPU kernel for a contiguous last dim (vec_softmax_lastdim, torch 2.10) as it
 * runs on the AVX-512 host the golden vectors were made on -- verified with 0 mismatches in 1.5 M entries:
 *   x = a*P (rounded);  m = max x;  e = Sleef_expf16_u10(x - m);
 *   sum = vec::reduce_all over 16-float vectors: lane l accumulates e[l], e[16+l], ... in order (a partial
 *         last vector only touches its first lanes), then halves are added: 16 -> 8 -> 4 -> 2 -> 1;
 *         rows shorter than one vector are summed left to right;
 *   S = e * (1/sum).
 * Sleef_expf_u10 (public SLEEF algorithm, FMA build): q = rint(d*log2e); s = fma(q,-L2U,d); s = fma(q,-L2L,s);
 * degree-6 Horner polynomial in s; u = 1 + fma(s*s, poly, s); result = u * 2^(q>>1) * 2^(q-(q>>1)).
 * On a host without AVX-512 torch uses 8-float vectors and the last bits of the sums differ.            */
static inline float sleef_expf_u10(float d) {
    const float qf = nearbyintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    const int q = (int)qf;
    float s = fmaf(qf, -0.693145751953125f, d);
    s = fmaf(qf, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    union { int32_t i; float f; } p0, p1;
    p0.i = ((q >> 1) + 0x7f) << 23;
    p1.i = ((q - (q >> 1)) + 0x7f) << 23;
    u = (u * p0.f) * p1.f;
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = INFINITY;
    return u;
}

#define MCD_O_VEC 16
static float aten_vec_sum(const float* e, int64_t n) {
    if (n < MCD_O_VEC) {
        float acc = e[0];
        for (int64_t i = 1; i < n; ++i) acc += e[i];
        return acc;
    }
    float acc[MCD_O_VEC];
    for (int l = 0; l < MCD_O_VEC; ++l) acc[l] = e[l];
    int64_t d = MCD_O_VEC;
    for (; d < n - (n % MCD_O_VEC); d += MCD_O_VEC)
        for (int l = 0; l < MCD_O_VEC; ++l) acc[l] += e[d + l];
    for (int l = 0; d + l < n; ++l) acc[l] += e[d + l];
    for (int w = MCD_O_VEC / 2; w >= 1; w /= 2)
        for (int l = 0; l < w; ++l) acc[l] += acc[l + w];
    return acc[0];
}

void mcd_o_row_softmax(const float* P, int64_t N, int64_t C, float a, float* S) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        const float* p = P + n * C;
        float* s = S + n * C;
        float m = -INFINITY;
        for (int64_t c = 0; c < C; ++c) {
            const float x = a * p[c];
            s[c] = x;
            if (x > m) m = x;
        }
        for (int64_t c = 0; c < C; ++c) s[c] = sleef_expf_u10(s[c] - m);
        const float r = 1.0f / aten_vec_sum(s, C);
        for (int64_t c = 0; c < C; ++c) s[c] = s[c] * r;
    }
}

/* ------------------------------------------------------------------------------------------
 * similarity.py:55   inds = topk(target_feats, dim=0, k)[1]   -> [K, U], sorted descending.
 * Ties: torch leaves the order unspecified; the build defines lowest image index first.
 * NaN: ordered above +inf like torch.topk.  A is [N, U] row-major with leading dim lda.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float v; int64_t i; } vi_t;

static int vi_before(float av, int64_t ai, float bv, int64_t bi) {
    /* 1 if (av,ai) ranks before (bv,bi): larger value first, NaN largest, then lower index */
    const int an = isnan(av), bn = isnan(bv);
    if (an != bn) return an;
    if (!an && av != bv) return av > bv;
    return ai < bi;
}

static int vi_cmp(const void* a, const void* b) {
    const vi_t* x = (const vi_t*)a;
    const vi_t* y = (const vi_t*)b;
    if (vi_before(x->v, x->i, y->v, y->i)) return -1;
    if (vi_before(y->v, y->i, x->v, x->i)) return 1;
    return 0;
}

/* strided top-k of one vector of n elements (element j at base[j*stride]) */
static void topk_strided(const float* base, int64_t n, int64_t stride, int64_t K, vi_t* heap /* K */) {
    /* min-heap on rank: heap[0] is the WORST of the current best K */
    int64_t hs = 0;
    for (int64_t j = 0; j < n; ++j) {
        const float v = base[j * stride];
        if (hs < K) {
            int64_t c = hs++;
            heap[c].v = v; heap[c].i = j;
            while (c > 0) {
                const int64_t p = (c - 1) / 2;
                if (vi_before(heap[p].v, heap[p].i, heap[c].v, heap[c].i)) {
                    vi_t t = heap[p]; heap[p] = heap[c]; heap[c] = t; c = p;
                } else break;
            }
        } else if (vi_before(v, j, heap[0].v, heap[0].i)) {
            heap[0].v = v; heap[0].i = j;
            int64_t c = 0;
            for (;;) {
                int64_t l = 2 * c + 1, r = l + 1, w = c;
                if (l < hs && vi_before(heap[w].v, heap[w].i, heap[l].v, heap[l].i)) w = l;
                if (r < hs && vi_before(heap[w].v, heap[w].i, heap[r].v, heap[r].i)) w = r;
                if (w == c) break;
                vi_t t = heap[w]; heap[w] = heap[c]; heap[c] = t; c = w;
            }
        }
    }
    qsort(heap, (size_t)hs, sizeof(vi_t), vi_cmp);
}

/* returns 0, or -1 when K > N (torch: "selected index k out of range") */
int mcd_o_col_topk(const float* A, int64_t N, int64_t U, int64_t lda, int64_t K, float* vals /* [K,U] */,
                   int64_t* idx /* [K,U] */) {
    if (K > N || K < 0) return -1;
    if (K == 0) return 0;
#pragma omp parallel
    {
        vi_t* heap = (vi_t*)malloc(sizeof(vi_t) * (size_t)K);
#pragma omp for schedule(static)
        for (int64_t u = 0; u < U; ++u) {
            topk_strided(A + u, N, lda, K, heap);
            for (int64_t j = 0; j < K; ++j) {
                if (vals) vals[j * U + u] = heap[j].v;
                idx[j * U + u] = heap[j].i;
            }
        }
        free(heap);
    }
    return 0;
}

/* describe_clip_neurons.py:64 torch.max(sim, dim=1) (k=1) / describe_broad_neurons.py:101 topk(k=10, dim=1) */
int mcd_o_row_topk(const float* sim, int64_t U, int64_t C, int64_t ld, int64_t k, float* vals /* [U,k] */,
                   int64_t* idx /* [U,k] */) {
    if (k > C || k < 0) return -1;
    if (k == 0) return 0;
#pragma omp parallel
    {
        vi_t* heap = (vi_t*)malloc(sizeof(vi_t) * (size_t)k);
#pragma omp for schedule(static)
        for (int64_t u = 0; u < U; ++u) {
            topk_strided(sim + u * ld, C, 1, k, heap);
            for (int64_t j = 0; j < k; ++j) {
                vals[u * k + j] = heap[j].v;
                idx[u * k + j] = heap[j].i;
            }
        }
        free(heap);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * similarity.py:59-65 (soft_wpmi), :84-88 (wpmi)
 *   g      = S[inds[j,u], c]
 *   soft:  t = log((1 + p[j]*(g - 1)) + min_prob)      each op rounded to fp32 on its own
 *   hard:  t = log(g + min_prob)
 *   pdge[u,c] = torch.sum(t, dim=0)[c]   (cascade order, see header)
 * idx is [K,U] like torch.topk's output.  split < 0 -> ATen's rule for C.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const float* S;
    int64_t ldS;
    const int64_t* idx;
    int64_t U, u, c;
    const float* p;
    float min_prob;
    int soft;
} wpmi_ctx;

static float wpmi_term(const void* vctx, int64_t j) {
    const wpmi_ctx* x = (const wpmi_ctx*)vctx;
    const float g = x->S[x->idx[j * x->U + x->u] * x->ldS + x->c];
    float w;
    if (x->soft) {
        const float d = g - 1.0f;
        const float y = x->p[j] * d;
        const float z = 1.0f + y;
        w = z + x->min_prob;
    } else {
        w = g + x->min_prob;
    }
    /* torch.log on CPU is MKL vsLn (high accuracy): 99.99 % of its results equal the correctly rounded log, which
     * the double-precision log rounded to fp32 reproduces; glibc's logf (< 1 ulp) agrees with it less often */
    return (float)log((double)w);
}

void mcd_o_wpmi_score(const float* S, int64_t ldS, const int64_t* idx, const float* p, int64_t C, int64_t U,
                      int64_t K, float min_prob, int soft, int split, float* pdge /* [U, C] */) {
    if (split < 0) split = mcd_o_sum_split((int)C);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t u = 0; u < U; ++u) {
        wpmi_ctx ctx = {S, ldS, idx, U, u, 0, p, min_prob, soft};
        for (int64_t c = 0; c < C; ++c) {
            ctx.c = c;
            pdge[u * C + c] = torch_sum0_col(wpmi_term, &ctx, K, (int)c, split);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * similarity.py:70-72
 *   prob_d = logsumexp(pdge, dim=0, keepdim=True) - log(U * ones([1]))
 *   out    = pdge - lam * prob_d
 * torch.logsumexp: m = amax(x,0) (inf -> 0); log(sum(exp(x - m), 0)) + m, sum in cascade order.
 * exp / log: torch's are MKL vsExp / vsLn (closed source, practically correctly rounded).  The checker and the kernel
 * (k_wpmi.hip: k5_exp / k5_log) take the SAME form -- the double-precision function rounded to float once -- so that they
 * cannot differ from each other by construction (round 5; glibc's expf / logf before: against live torch.logsumexp both
 * forms are off by one ulp equally often, 16 / 19 of 191 182 random columns).
 * ---------------------------------------------------------------------------------------- */
static inline float k5_exp(float x) { return (float)exp((double)x); }
static inline float k5_log(float x) { return (float)log((double)x); }

typedef struct {
    const float* x;
    int64_t ld, c;
    float m;
} lse_ctx;

static float lse_term(const void* vctx, int64_t u) {
    const lse_ctx* x = (const lse_ctx*)vctx;
    return k5_exp(x->x[u * x->ld + x->c] - x->m);
}

void mcd_o_logsumexp_sub(const float* pdge, int64_t U, int64_t C, float lam, int split, float* out) {
    if (split < 0) split = mcd_o_sum_split((int)C);
    const float logU = k5_log((float)U);
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < C; ++c) {
        float m = -INFINITY;
        for (int64_t u = 0; u < U; ++u) {
            const float v = pdge[u * C + c];
            if (v > m || isnan(v)) m = v;
        }
        if (isinf(m)) m = 0.f;
        lse_ctx ctx = {pdge, C, c, m};
        const float s = torch_sum0_col(lse_term, &ctx, U, (int)c, split);
        const float lse = k5_log(s) + m;
        const float prob_d = lse - logU;
        const float scaled = lam * prob_d;
        for (int64_t u = 0; u < U; ++u) out[u * C + c] = pdge[u * C + c] - scaled;
    }
}

/* ------------------------------------------------------------------------------------------
 * utils.py:27-52 get_activation(): 4-D hook output [B,Cout,H,W] -> mean(dim=[2,3]) (mode 0)
 * or amax(dim=[2,3]) (mode 1).  Written into dst[(row0+b)*ld + col0 + ch].
 * The mean is accumulated in double and rounded once: it is the checker for a kernel whose
 * tolerance is stated in the test (torch's own mean order is device dependent).
 * ---------------------------------------------------------------------------------------- */
void mcd_o_hook_pool(const float* x, int64_t B, int64_t Cout, int64_t HW, int mode, float* dst, int64_t row0,
                     int64_t col0, int64_t ld) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t ch = 0; ch < Cout; ++ch) {
            const float* px = x + (b * Cout + ch) * HW;
            float r;
            if (mode == 0) {
                double s = 0.0;
                for (int64_t i = 0; i < HW; ++i) s += (double)px[i];
                r = (float)(s / (double)HW);
            } else {
                r = -INFINITY;
                for (int64_t i = 0; i < HW; ++i)
                    if (px[i] > r || isnan(px[i])) r = px[i];
            }
            dst[(row0 + b) * ld + col0 + ch] = r;
        }
    }
}
