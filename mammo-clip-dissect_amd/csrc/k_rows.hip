// k_rows.hip -- row-wise kernels over the embedding / similarity matrices (HBM-bound).
//   K1a mcd_normalize_rows   concept_vit/utils.py:577-578
//   K2  mcd_row_softmax      concept_vit/similarity.py:54
// One 64-lane wavefront owns one row: coalesced dword reads (lane l reads columns l, l+64, ...),
// the row lives in registers between the passes, reductions are wave butterflies (no LDS).
#include "mcd_common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // 4 waves of 64

// ---- K1a -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* x, int64_t ldx, int64_t n,
                                                              int64_t d, float* y, int64_t ldy) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* xr = x + row * ldx;
    float* yr = y + row * ldy;
    float ss = 0.f;
    for (int64_t k = lane; k < d; k += 64) {
        const float v = xr[k];
        ss += v * v;
    }
    ss = mcd_wave_sum(ss);
    const float nrm = sqrtf(ss);
    for (int64_t k = lane; k < d; k += 64) yr[k] = xr[k] / nrm;
}

// ---- K2 ------------------------------------------------------------------------------------
// ITEMS > 0: the row (C <= 64*ITEMS) is held in registers: one read of P, one write of S.
// ITEMS == 0: any C, three reads of P (the re-reads come from L2).
template <int ITEMS>
__global__ __launch_bounds__(256) void row_softmax_kernel(const float* __restrict__ P, int64_t ldp, int64_t N,
                                                           int64_t C, float a, float* __restrict__ S, int64_t lds) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* pr = P + row * ldp;
    float* sr = S + row * lds;
    if constexpr (ITEMS > 0) {
        float v[ITEMS];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int64_t c = lane + 64 * i;
            v[i] = (c < C) ? a * pr[c] : -INFINITY;  // x = a*clip_feats, rounded (similarity.py:54)
            m = fmaxf(m, v[i]);
        }
        m = mcd_wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int64_t c = lane + 64 * i;
            v[i] = (c < C) ? expf(v[i] - m) : 0.f;
            s += v[i];
        }
        s = mcd_wave_sum(s);
        const float r = 1.0f / s;  // ATen's CPU softmax multiplies by the reciprocal of the row sum
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int64_t c = lane + 64 * i;
            if (c < lds) sr[c] = v[i] * r;  // padding columns C..lds-1 get exactly 0
        }
    } else {
        float m = -INFINITY;
        for (int64_t c = lane; c < C; c += 64) m = fmaxf(m, a * pr[c]);
        m = mcd_wave_max(m);
        float s = 0.f;
        for (int64_t c = lane; c < C; c += 64) s += expf(a * pr[c] - m);
        s = mcd_wave_sum(s);
        const float r = 1.0f / s;
        for (int64_t c = lane; c < lds; c += 64) sr[c] = (c < C) ? expf(a * pr[c] - m) * r : 0.f;
    }
}

}  // namespace

extern "C" int mcd_normalize_rows(const float* x, int64_t ldx, int64_t n, int64_t d, float* y, int64_t ldy,
                                  mcd_stream_t stream) {
    MCD_REQUIRE(x && y, MCD_E_ARG, "mcd_normalize_rows: NULL pointer");
    MCD_REQUIRE(n >= 0 && d > 0 && ldx >= d && ldy >= d, MCD_E_ARG, "mcd_normalize_rows: bad shape n=%lld d=%lld",
                (long long)n, (long long)d);
    if (n == 0) return MCD_OK;
    const unsigned grid = (unsigned)mcd_cdiv(n, ROWS_PER_BLOCK);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, n, d, y, ldy);
    MCD_LAUNCH_CHECK("normalize_rows_kernel");
    return MCD_OK;
}

extern "C" int mcd_row_softmax(const float* P, int64_t ldp, int64_t N, int64_t C, float a, float* S, int64_t lds,
                               mcd_stream_t stream) {
    MCD_REQUIRE(P && S, MCD_E_ARG, "mcd_row_softmax: NULL pointer");
    MCD_REQUIRE(N >= 0 && C > 0 && ldp >= C && lds >= C, MCD_E_ARG, "mcd_row_softmax: bad shape N=%lld C=%lld",
                (long long)N, (long long)C);
    if (N == 0) return MCD_OK;
    const dim3 grid((unsigned)mcd_cdiv(N, ROWS_PER_BLOCK)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (lds <= 64 * 4)
        hipLaunchKernelGGL(row_softmax_kernel<4>, grid, block, 0, st, P, ldp, N, C, a, S, lds);
    else if (lds <= 64 * 12)
        hipLaunchKernelGGL(row_softmax_kernel<12>, grid, block, 0, st, P, ldp, N, C, a, S, lds);
    else if (lds <= 64 * 32)
        hipLaunchKernelGGL(row_softmax_kernel<32>, grid, block, 0, st, P, ldp, N, C, a, S, lds);
    else
        hipLaunchKernelGGL(row_softmax_kernel<0>, grid, block, 0, st, P, ldp, N, C, a, S, lds);
    MCD_LAUNCH_CHECK("row_softmax_kernel");
    return MCD_OK;
}
