// Microbenchmark: issue rate of fp32 MFMAs on gfx950, as a function of waves per SIMD and independent accumulator
// chains per wave.  Dev tool (hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// MFMA + independent VALU work in the same wave: do they overlap?
template <int NVALU>
__global__ __launch_bounds__(1024) void kv(float* out, int iters, float a, float b) {
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = a + j + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NVALU; ++v) x[v & 7] = x[v & 7] * b + a;   // -ffp-contract=off: v_mul + v_add, independent of c
    }
    float s = c[0];
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NVALU>
void runv(int waves_per_simd, float* out) {
    const int iters = 2000;
    dim3 grid(256), block(64 * 4 * waves_per_simd);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kv<NVALU><<<grid, block>>>(out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kv<NVALU><<<grid, block>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)256 * 4 * waves_per_simd * iters * 16 * 4096.0;
    printf("32x32x2 + %3d x (v_mul, v_add) per 16 MFMAs, waves/SIMD=%d  %.3f ms  %.1f TFLOP/s (MFMA only)\n", NVALU, waves_per_simd,
           ms, flops / ms / 1e9);
}

template <int CHAINS, bool SMALL>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a, float b) {
    f32x16 c[CHAINS];
    f32x4 d[CHAINS];
    for (int i = 0; i < CHAINS; ++i) {
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
        for (int r = 0; r < 4; ++r) d[i][r] = 0.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (SMALL) d[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d[i], 0, 0, 0);
                else c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < CHAINS; ++i) s += SMALL ? d[i][0] : c[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS, bool SMALL>
void run(int waves_per_simd, float* out) {
    const int iters = 2000;
    dim3 grid(256), block(64 * 4 * waves_per_simd);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS, SMALL><<<grid, block>>>(out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CHAINS, SMALL><<<grid, block>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)256 * 4 * waves_per_simd * iters * 16 * CHAINS * (SMALL ? 2048.0 : 4096.0);
    printf("%s chains=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", SMALL ? "16x16x4" : "32x32x2", CHAINS, waves_per_simd, ms,
           flops / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    run<1, false>(1, out); run<2, false>(1, out); run<4, false>(1, out);
    run<1, false>(2, out); run<2, false>(2, out); run<1, false>(4, out); run<2, false>(4, out);
    run<1, true>(1, out); run<2, true>(1, out); run<4, true>(1, out);
    run<1, true>(2, out); run<4, true>(2, out); run<1, true>(4, out); run<4, true>(4, out);
    runv<0>(2, out); runv<32>(2, out); runv<64>(2, out); runv<128>(2, out);
    runv<0>(4, out); runv<64>(4, out); runv<128>(4, out);
    return 0;
}
