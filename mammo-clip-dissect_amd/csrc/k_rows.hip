// k_rows.hip -- row-wise kernels over the embedding / similarity matrices (HBM-bound).
//   K1a mcd_normalize_rows   concept_vit/utils.py:577-578   (bit-exact restatement of ATen's CPU norm + division)
//   K2  mcd_row_softmax      concept_vit/similarity.py:54   (bit-exact restatement of ATen's CPU kernel)
// Rows live in registers between the passes; reductions are lane shuffles (no LDS).
#include "mcd_common.h"
#include <stdlib.h>

namespace {


// ---- K1a -----------------------------------------------------------------------------------
// Bit-exact with ATen's CPU x / x.norm(dim=-1, keepdim=True) (the reduce_lastdim fast path of the p = 2 norm kernel,
// torch 2.10): eight accumulators acc[j] = fma(x[8i+j], x[8i+j], acc[j]), added lane 0 first; the d % 8 tail one
// element at a time (the first 4*(tail/4) as product + add, the last tail % 4 fused -- that build's unrolled loop);
// sqrt; true division.  The eight chains are the unit of parallelism: 8 lanes own a row, 8 rows per wave.
// A row's 8 lanes keep its elements in registers (NV per lane, all loads issued before the chain starts), so the
// division pass reads nothing; NV = 0 streams rows longer than 8 * 128 floats (two passes).  One wave (8 rows) per
// workgroup: 1250 workgroups at N = 10 000.
constexpr int NORM_ROWS_PER_BLOCK = 8;

template <int NV>
__global__ __launch_bounds__(64) void normalize_rows_kernel(const float* x, int64_t ldx, int64_t n, int64_t d, float* y,
                                                             int64_t ldy) {
    const int j = threadIdx.x & 7;
    int64_t row = (int64_t)blockIdx.x * NORM_ROWS_PER_BLOCK + (threadIdx.x >> 3);
    const bool live = row < n;
    if (!live) row = n - 1;  // keep the shuffles below convergent
    const float* xr = x + row * ldx;
    const int64_t full = d - d % 8;
    float acc = 0.f;
    float v[NV > 0 ? NV : 1];
    if (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = (8 * i + j < d) ? xr[8 * i + j] : 0.f;   // tail elements included
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (8 * i < full) acc = __builtin_fmaf(v[i], v[i], acc);
    } else {
#pragma unroll 8
        for (int64_t k = j; k < full; k += 8) {
            const float t = xr[k];
            acc = __builtin_fmaf(t, t, acc);
        }
    }
    float ss = __shfl(acc, 0, 8);
#pragma unroll
    for (int l = 1; l < 8; ++l) ss = ss + __shfl(acc, l, 8);
    const int64_t unfused_end = full + (d - full) / 4 * 4;
    for (int64_t k = full; k < d; ++k) {
        const float t = xr[k];
        if (k < unfused_end) {
            const float sq = t * t;
            ss = ss + sq;
        } else {
            ss = __builtin_fmaf(t, t, ss);
        }
    }
    const float nrm = sqrtf(ss);
    if (!live) return;
    float* yr = y + row * ldy;
    if (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (8 * i + j < d) yr[8 * i + j] = v[i] / nrm;
    } else {
        for (int64_t k = j; k < d; k += 8) yr[k] = xr[k] / nrm;
    }
}

// ---- K2 ------------------------------------------------------------------------------------
// Bit-exact with ATen's CPU softmax (vec_softmax_lastdim on an AVX-512 host, torch 2.10):
//   x = a*P;  m = max x;  e = Sleef_expf_u10(x - m);  sum over 16-float "vectors" (lane l adds e[l], e[16+l], ...
//   in order; a partial last vector only touches its first lanes; then 16 -> 8 -> 4 -> 2 -> 1);  S = e * (1/sum).
// The 16 accumulation chains are the unit of parallelism, so 16 GPU lanes own a row (4 rows per wave): lane l
// holds elements l, l+16, ... in registers, runs its chain sequentially and the halving tree is 4 shuffles.
__device__ __forceinline__ float sleef_expf_u10(float d) {
    const float qf = __builtin_rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    const int q = (int)qf;
    float s = __builtin_fmaf(qf, -0.693145751953125f, d);
    s = __builtin_fmaf(qf, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
    u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
    u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
    u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
    u = __builtin_fmaf(u, s, 0.5f);
    u = 1.0f + __builtin_fmaf(s * s, u, s);
    u = (u * __int_as_float(((q >> 1) + 0x7f) << 23)) * __int_as_float(((q - (q >> 1)) + 0x7f) << 23);
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = INFINITY;
    return u;
}

__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 16));
    return v;
}

// sum of the 16 lane accumulators in ATen's order (halves added); valid in lane 0 of the group, then broadcast
__device__ __forceinline__ float group16_aten_sum(float acc) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc = acc + __shfl_down(acc, o, 16);
    return __shfl(acc, 0, 16);
}

// JMAX > 0: elements per lane held in registers (C <= 16*JMAX); JMAX == 0: any C, exp recomputed in the last pass
template <int JMAX>
__global__ __launch_bounds__(256) void row_softmax_kernel(const float* __restrict__ P, int64_t ldp, int64_t N,
                                                           int64_t C, float a, float* __restrict__ S, int64_t lds) {
    const int l = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= N) return;
    const float* pr = P + row * ldp;
    float* sr = S + row * lds;
    const int64_t nfull = C / 16;  // complete 16-float vectors
    if (C < 16) {  // ATen sums a row shorter than one vector left to right
        const float x = (l < C) ? a * pr[l] : -INFINITY;
        const float m = group16_max(x);
        const float e = (l < C) ? sleef_expf_u10(x - m) : 0.f;
        float acc = __shfl(e, 0, 16);
        for (int i = 1; i < (int)C; ++i) acc = acc + __shfl(e, i, 16);
        const float r = 1.0f / acc;
        if (l < lds) sr[l] = e * r;
        for (int64_t c = 16 + l; c < lds; c += 16) sr[c] = 0.f;
        return;
    }
    if constexpr (JMAX > 0) {
        float v[JMAX];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int64_t c = l + 16 * j;
            v[j] = (c < C) ? a * pr[c] : -INFINITY;  // x = a*clip_feats, rounded (similarity.py:54)
            m = fmaxf(m, v[j]);
        }
        m = group16_max(m);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int64_t c = l + 16 * j;
            if (c < C) {
                v[j] = sleef_expf_u10(v[j] - m);
                acc = (j == 0) ? v[j] : acc + v[j];
            } else {
                v[j] = 0.f;
            }
        }
        const float r = 1.0f / group16_aten_sum(acc);
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int64_t c = l + 16 * j;
            if (c < lds) sr[c] = v[j] * r;  // padding columns C..lds-1 get exactly 0
        }
    } else {
        float m = -INFINITY;
        for (int64_t c = l; c < C; c += 16) m = fmaxf(m, a * pr[c]);
        m = group16_max(m);
        float acc = sleef_expf_u10(a * pr[l] - m);
        for (int64_t c = l + 16; c < C; c += 16) acc = acc + sleef_expf_u10(a * pr[c] - m);
        const float r = 1.0f / group16_aten_sum(acc);
        for (int64_t c = l; c < lds; c += 16) sr[c] = (c < C) ? sleef_expf_u10(a * pr[c] - m) * r : 0.f;
    }
    (void)nfull;
}

// Rows of 256 to 16 384 concepts (763; the stress configuration's 10 000): one 128- or 256-thread workgroup per row, the
// row lives in LDS.  P is read once and S written once with 16-byte accesses (the 16-lanes-per-row kernel above
// re-reads a long row three times in 64-byte pieces and evaluates exp twice).  The sum keeps ATen's order: 16 chains
// (chain l adds the terms c = l, l+16, ... one after the other) reduced by halves -- 16 lanes walk the chains out
// of LDS while the workgroup's other rows-in-flight on the CU keep the memory pipe busy.
template <int T>
__global__ __launch_bounds__(T) void row_softmax_lds_kernel(const float* __restrict__ P, int64_t ldp, int64_t C, float a,
                                                               float* __restrict__ S, int64_t lds, int vec4) {
    extern __shared__ float s_e[];   // [C rounded up to 4]
    __shared__ float s_red[4];
    __shared__ float s_r;
    const int tid = threadIdx.x;
    const float* pr = P + (int64_t)blockIdx.x * ldp;
    float* sr = S + (int64_t)blockIdx.x * lds;
    const int Ci = (int)C, C4 = Ci & ~3;
    float m = -INFINITY;
    if constexpr (T == 128) {
        // C <= 1 024: a thread's (up to) 8 elements as 8 buffer loads issued together -- in a loop of runtime length hipcc waits for
        // each load before it issues the next (6 memory round trips in a row per thread at 763 concepts, at any alignment);
        // dwords at or past the row's end return 0 and are not used
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pr, 0, Ci * 4, 0x00020000);
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, tid * 4, j * T * 4, 0));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = tid + j * T;
            if (c < Ci) {
                const float x = a * xv[j];                 // x = a*clip_feats, rounded (similarity.py:54)
                s_e[c] = x;
                m = fmaxf(m, x);
            }
        }
    } else if (vec4) {
        for (int c = 4 * tid; c < C4; c += 4 * T) {
            const float4 v = *reinterpret_cast<const float4*>(pr + c);
            const float4 x = make_float4(a * v.x, a * v.y, a * v.z, a * v.w);   // x = a*clip_feats, rounded (similarity.py:54)
            *reinterpret_cast<float4*>(s_e + c) = x;
            m = fmaxf(fmaxf(m, x.x), fmaxf(fmaxf(x.y, x.z), x.w));
        }
        for (int c = C4 + tid; c < Ci; c += T) {
            const float x = a * pr[c];
            s_e[c] = x;
            m = fmaxf(m, x);
        }
    } else {
        for (int c = tid; c < Ci; c += T) {
            const float x = a * pr[c];
            s_e[c] = x;
            m = fmaxf(m, x);
        }
    }
    m = mcd_wave_max(m);
    if ((tid & 63) == 0) s_red[tid >> 6] = m;
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < T / 64; ++w) m = fmaxf(m, s_red[w]);
    for (int c = tid; c < Ci; c += T) s_e[c] = sleef_expf_u10(s_e[c] - m);   // (the barrier above ordered the fills)
    __syncthreads();
    if (tid < 16) {
        float acc = s_e[tid];
#pragma unroll 8
        for (int c = tid + 16; c < Ci; c += 16) acc = acc + s_e[c];
        const float r = 1.0f / group16_aten_sum(acc);
        if (tid == 0) s_r = r;
    }
    __syncthreads();
    const float r = s_r;
    const int L = (int)lds;
    if (vec4) {
        for (int c = 4 * tid; c < L; c += 4 * T) {   // lds % 4 == 0
            float4 o;
            o.x = (c + 0 < Ci) ? s_e[c + 0] * r : 0.f;
            o.y = (c + 1 < Ci) ? s_e[c + 1] * r : 0.f;
            o.z = (c + 2 < Ci) ? s_e[c + 2] * r : 0.f;
            o.w = (c + 3 < Ci) ? s_e[c + 3] * r : 0.f;
            *reinterpret_cast<float4*>(sr + c) = o;
        }
    } else {
        for (int c = tid; c < L; c += T) sr[c] = (c < Ci) ? s_e[c] * r : 0.f;   // padding columns C..lds-1 get exactly 0
    }
}

// ---- K7: per-row centre / cube / normalise, the pre-processing of cos_similarity_cubed --------------
// One workgroup per row (a neuron's activations, or a concept's similarities, over the N images):
//   d = x - mean(x);  c = d*d*d;  y = c / max(||c||_2, min_norm)       (reference similarity.py:15-22)
// three passes over a row that stays in L2 (40 KB at N = 10 000); block reductions through LDS.
__device__ __forceinline__ float block256_sum(float v, float* s_red) {
    v = mcd_wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float t = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    return t;
}

// x and y may be the same buffer (the mirror calls it in place): no __restrict__; every thread reads its own elements
// of a pass before it writes them, and the passes are separated by the barriers of block256_sum().
__global__ __launch_bounds__(256) void center_cube_normalize_kernel(const float* x, int64_t ldx, int64_t n,
                                                                     float min_norm, float* y, int64_t ldy) {
    __shared__ float s_red[4];
    const float* xr = x + (int64_t)blockIdx.x * ldx;
    float* yr = y + (int64_t)blockIdx.x * ldy;
    float s = 0.f;
    for (int64_t k = threadIdx.x; k < n; k += 256) s += xr[k];
    const float mean = block256_sum(s, s_red) / (float)n;
    float ss = 0.f;
    for (int64_t k = threadIdx.x; k < n; k += 256) {
        const float d = xr[k] - mean;
        const float c = (d * d) * d;
        ss += c * c;
    }
    const float nrm = fmaxf(sqrtf(block256_sum(ss, s_red)), min_norm);
    for (int64_t k = threadIdx.x; k < n; k += 256) {
        const float d = xr[k] - mean;
        yr[k] = ((d * d) * d) / nrm;
    }
}

}  // namespace

extern "C" int mcd_normalize_rows(const float* x, int64_t ldx, int64_t n, int64_t d, float* y, int64_t ldy,
                                  mcd_stream_t stream) {
    MCD_REQUIRE(x && y, MCD_E_ARG, "mcd_normalize_rows: NULL pointer");
    MCD_REQUIRE(n >= 0 && d > 0 && ldx >= d && ldy >= d, MCD_E_ARG, "mcd_normalize_rows: bad shape n=%lld d=%lld",
                (long long)n, (long long)d);
    if (n == 0) return MCD_OK;
    const unsigned grid = (unsigned)mcd_cdiv(n, NORM_ROWS_PER_BLOCK);
#define MCD_NORM_LAUNCH(NV) \
    hipLaunchKernelGGL(normalize_rows_kernel<NV>, dim3(grid), dim3(64), 0, (hipStream_t)stream, x, ldx, n, d, y, ldy)
    const int64_t per_lane = mcd_cdiv(d, 8);
    if (per_lane <= 16) MCD_NORM_LAUNCH(16);
    else if (per_lane <= 64) MCD_NORM_LAUNCH(64);
    else if (per_lane <= 128) MCD_NORM_LAUNCH(128);
    else MCD_NORM_LAUNCH(0);
#undef MCD_NORM_LAUNCH
    MCD_LAUNCH_CHECK("normalize_rows_kernel");
    return MCD_OK;
}

extern "C" int mcd_row_softmax(const float* P, int64_t ldp, int64_t N, int64_t C, float a, float* S, int64_t lds,
                               mcd_stream_t stream) {
    MCD_REQUIRE(P && S, MCD_E_ARG, "mcd_row_softmax: NULL pointer");
    MCD_REQUIRE(N >= 0 && C > 0 && ldp >= C && lds >= C, MCD_E_ARG, "mcd_row_softmax: bad shape N=%lld C=%lld",
                (long long)N, (long long)C);
    if (N == 0) return MCD_OK;
    const dim3 grid((unsigned)mcd_cdiv(N, 16)), block(256);
    hipStream_t st = (hipStream_t)stream;
    // short rows: 16 lanes per row, the row in registers; from 256 concepts on: one workgroup per row, the row in LDS
    // (763 concepts: 0.021 ms against 0.028 for the register kernel; 10 000: 0.57 against 1.24 for the streaming one)
    const int vec4 = (ldp % 4 == 0) && (lds % 4 == 0) && (((uintptr_t)P) % 16 == 0) && (((uintptr_t)S) % 16 == 0);
    const size_t sh = (size_t)((C + 3) / 4 * 4) * sizeof(float);
    if (lds <= 16 * 16)
        hipLaunchKernelGGL(row_softmax_kernel<16>, grid, block, 0, st, P, ldp, N, C, a, S, lds);
    else if (C <= 1024 && N <= 0x7fffffffLL)
        hipLaunchKernelGGL(row_softmax_lds_kernel<128>, dim3((unsigned)N), dim3(128), sh, st, P, ldp, C, a, S, lds, vec4);
    else if (C <= 16384 && N <= 0x7fffffffLL)
        hipLaunchKernelGGL(row_softmax_lds_kernel<256>, dim3((unsigned)N), dim3(256), sh, st, P, ldp, C, a, S, lds, vec4);
    else
        hipLaunchKernelGGL(row_softmax_kernel<0>, grid, block, 0, st, P, ldp, N, C, a, S, lds);
    MCD_LAUNCH_CHECK("row_softmax_kernel");
    return MCD_OK;
}

extern "C" int mcd_center_cube_normalize_rows(const float* x, int64_t ldx, int64_t rows, int64_t n, float min_norm,
                                              float* y, int64_t ldy, mcd_stream_t stream) {
    MCD_REQUIRE(x && y, MCD_E_ARG, "mcd_center_cube_normalize_rows: NULL pointer");
    MCD_REQUIRE(rows >= 0 && n > 0 && ldx >= n && ldy >= n, MCD_E_ARG, "mcd_center_cube_normalize_rows: bad shape");
    if (rows == 0) return MCD_OK;
    hipLaunchKernelGGL(center_cube_normalize_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, ldx, n,
                       min_norm, y, ldy);
    MCD_LAUNCH_CHECK("center_cube_normalize_kernel");
    return MCD_OK;
}
