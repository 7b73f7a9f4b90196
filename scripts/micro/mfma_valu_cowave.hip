// Microbenchmark (round 5): do the vector instructions of one wave execute under ANOTHER wave's MFMAs on the same SIMD when the MFMA
// wave's stream is what K1s' K loop issues -- v_mfma_f32_16x16x32_bf16 on INDEPENDENT accumulators (round 2's mfma_valu_coexec.hip ran
// one dependent chain per wave and found the pair's time to be the SUM) -- and the vector wave runs the epilogue's mix (8 v_exp_f32 +
// 4 v_cvt_pk_bf16_f32 per group)?  512-thread workgroups, one per CU: waves 0-3 MFMA only, waves 4-7 vector only; also the same with
// the roles given to the YOUNGER / OLDER half swapped, and with s_setprio 1 on the MFMA waves.
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_cowave mfma_valu_cowave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE /*1 mfma waves, 2 vector waves, 3 both*/, int SWAP /*1: waves 4-7 do the MFMAs*/, int PRIO>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b, int groups_per_iter) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool mf = SWAP ? wave >= 4 : wave < 4;
    float s = 0.f;
    if (mf) {
        if (!(MODE & 1)) return;
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        f32x4 d[16];
        for (int i = 0; i < 16; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 av, bv;
        for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(a + j + (threadIdx.x & 7)); bv[j] = (__bf16)(b + (threadIdx.x & 63) * 0.01f); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 64; ++u) d[u & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, d[u & 15], 0, 0, 0);   // one k-step
        }
        for (int i = 0; i < 16; ++i) s += d[i][0];
    } else {
        if (!(MODE & 2)) return;
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = -(a + j) * 0.1f - (threadIdx.x & 63) * 0.001f;
        unsigned acc = 0;
        for (int it = 0; it < iters; ++it) {
            for (int g = 0; g < groups_per_iter; ++g) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_exp2f(x[j]);
                unsigned p0, p1, p2, p3;
                asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %6, %7\n\tv_cvt_pk_bf16_f32 %2, %8, %9\n\tv_cvt_pk_bf16_f32 %3, %10, %11"
                             : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(e[4]), "v"(e[5]), "v"(e[6]), "v"(e[7]));
                acc ^= p0 ^ p1 ^ p2 ^ p3;
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = x[j] * 0.999f;
            }
        }
        s = (float)acc;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int SWAP, int PRIO>
float run(float* out, int iters, int gpi) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, SWAP, PRIO><<<256, 512>>>(out, 10, 1.f, 1.f, gpi);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, SWAP, PRIO><<<256, 512>>>(out, iters, 1.f, 1.f, gpi);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 2000;
    for (int gpi : {2, 4, 8}) {   // epilogue groups (8 exp + 4 cvt + 8 mul) per 64-MFMA k-step: K1s needs 32 groups per 16 k-steps = 2
        const float m = run<1, 0, 0>(out, iters, gpi), v = run<2, 0, 0>(out, iters, gpi), b = run<3, 0, 0>(out, iters, gpi);
        const float bs = run<3, 1, 0>(out, iters, gpi), bp = run<3, 0, 1>(out, iters, gpi), bsp = run<3, 1, 1>(out, iters, gpi);
        printf("%d groups per k-step: MFMA waves alone %.3f ms (%.1f cycles per MFMA at 2.4 GHz), vector waves alone %.3f ms; both %.3f "
               "(MFMA = older half), %.3f (MFMA = younger half), %.3f / %.3f with s_setprio 1 on the MFMA waves; sum %.3f, max %.3f\n",
               gpi, m, m * 2.4e6 / (iters * 64.0), v, b, bs, bp, bsp, m + v, m > v ? m : v);
    }
    return 0;
}
