// k_gemm.hip -- K1: P = I_hat @ T_hat^T, the one dense contraction of the path (MFMA).
//   replaces  clip_feats = image_features @ text_features.T   concept_vit/utils.py:594
//
// Both operands are K-contiguous ("NT" GEMM: I is [N,D], T is [C,D]), which is exactly the
// fragment order the MFMA A and B operands want.
//
// mode MCD_GEMM_F32 (parity mode): v_mfma_f32_32x32x2_f32 -- exact fp32, bit-for-bit a k-ordered
//   fmaf chain (64 FLOP/clk/SIMD, 157 TFLOP/s peak).  128x128 output tile per 256-thread workgroup,
//   4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator registers), BK = 32, operands
//   staged through a double-buffered LDS image with 33-float rows so the per-lane fragment reads
//   (row = lane&31, k = kk + lane>>5) hit 32 distinct banks; the next K-tile is prefetched into registers
//   under the MFMAs.
#include "mcd_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128, BN = 128, BK = 32, LDK = BK + 1;

// global -> registers: thread t fetches 4 quads of the 128 x 32 tile: f = t + 256*it -> row f/8, k-quad (f%8)*4
template <bool ALIGNED>
__device__ __forceinline__ void fetch_tile(const float* __restrict__ G, int64_t ldg, int64_t rows, int64_t row0,
                                           int64_t Kd, int64_t k0, float4 (&v)[4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int f = threadIdx.x + 256 * it;
        const int r = f >> 3, kq = (f & 7) * 4;
        const int64_t gr = row0 + r, gk = k0 + kq;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < rows) {
            const float* src = G + gr * ldg + gk;
            if (ALIGNED && gk + 3 < Kd) {
                t = *reinterpret_cast<const float4*>(src);
            } else {
                if (gk + 0 < Kd) t.x = src[0];
                if (gk + 1 < Kd) t.y = src[1];
                if (gk + 2 < Kd) t.z = src[2];
                if (gk + 3 < Kd) t.w = src[3];
            }
        }
        v[it] = t;
    }
}

// registers -> LDS tile (33-float rows)
__device__ __forceinline__ void store_tile(const float4 (&v)[4], float* __restrict__ L) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int f = threadIdx.x + 256 * it;
        float* d = L + (f >> 3) * LDK + (f & 7) * 4;
        d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
    }
}

// Software pipeline: the loads of K-tile t+1 are issued before the 64 MFMAs of tile t and land in registers
// while the matrix pipe works; they are written to the OTHER LDS buffer after the MFMAs, one barrier per tile.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ B, int64_t ldb, int64_t M,
                                                           int64_t Nc, int64_t Kd, float* __restrict__ Cc,
                                                           int64_t ldc) {
    __shared__ float As[2][BM * LDK];
    __shared__ float Bs[2][BN * LDK];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t row0 = (int64_t)blockIdx.y * BM, col0 = (int64_t)blockIdx.x * BN;
    const int fr = lane & 31, fk = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    float4 ra[4], rb[4];
    fetch_tile<ALIGNED>(A, lda, M, row0, Kd, 0, ra);
    fetch_tile<ALIGNED>(B, ldb, Nc, col0, Kd, 0, rb);
    store_tile(ra, As[0]);
    store_tile(rb, Bs[0]);
    __syncthreads();
    const int64_t nt = (Kd + BK - 1) / BK;
    for (int64_t t = 0; t < nt; ++t) {
        const int cur = (int)(t & 1);
        if (t + 1 < nt) {
            fetch_tile<ALIGNED>(A, lda, M, row0, Kd, (t + 1) * BK, ra);
            fetch_tile<ALIGNED>(B, ldb, Nc, col0, Kd, (t + 1) * BK, rb);
        }
        const float* as = As[cur];
        const float* bs = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = as[(wr * 64 + mi * 32 + fr) * LDK + kk + fk];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = bs[(wc * 64 + ni * 32 + fr) * LDK + kk + fk];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (t + 1 < nt) {
            store_tile(ra, As[cur ^ 1]);
            store_tile(rb, Bs[cur ^ 1]);
        }
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gr = row0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                const int64_t gc = col0 + wc * 64 + ni * 32 + fr;
                if (gr < M && gc < Nc) Cc[gr * ldc + gc] = acc[mi][ni][r];
            }
}

}  // namespace

extern "C" int mcd_embed_gemm(const float* I, int64_t ldi, const float* T, int64_t ldt, int64_t N, int64_t C,
                              int64_t D, int mode, float* P, int64_t ldp, mcd_stream_t stream) {
    MCD_REQUIRE(I && T && P, MCD_E_ARG, "mcd_embed_gemm: NULL pointer");
    MCD_REQUIRE(N >= 0 && C > 0 && D > 0 && ldi >= D && ldt >= D && ldp >= C, MCD_E_ARG,
                "mcd_embed_gemm: bad shape N=%lld C=%lld D=%lld", (long long)N, (long long)C, (long long)D);
    MCD_REQUIRE(mode == MCD_GEMM_F32, MCD_E_UNSUPPORTED, "mcd_embed_gemm: mode %d not built yet (only MCD_GEMM_F32)", mode);
    if (N == 0) return MCD_OK;
    const dim3 grid((unsigned)mcd_cdiv(C, BN), (unsigned)mcd_cdiv(N, BM));
    MCD_REQUIRE(grid.y <= 65535u, MCD_E_UNSUPPORTED, "mcd_embed_gemm: N too large for one launch");
    const bool aligned = (ldi % 4 == 0) && (ldt % 4 == 0) && (((uintptr_t)I) % 16 == 0) && (((uintptr_t)T) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (aligned)
        hipLaunchKernelGGL(gemm_nt_f32_kernel<true>, grid, dim3(256), 0, st, I, ldi, T, ldt, N, C, D, P, ldp);
    else
        hipLaunchKernelGGL(gemm_nt_f32_kernel<false>, grid, dim3(256), 0, st, I, ldi, T, ldt, N, C, D, P, ldp);
    MCD_LAUNCH_CHECK("gemm_nt_f32_kernel");
    return MCD_OK;
}
