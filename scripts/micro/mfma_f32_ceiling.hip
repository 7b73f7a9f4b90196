// What a bare v_mfma_f32_32x32x2_f32 loop sustains on this part with constant and with random operands (the clock the chip holds
// depends on the data), 1 and 2 waves per SIMD, 4 independent accumulator chains per wave -- the ceiling K1 (fp32 MFMA, parity mode)
// is priced against beside the nominal 157.3 TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f32_ceiling scripts/micro/mfma_f32_ceiling.hip && /tmp/mfma_f32_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ __launch_bounds__(512) void k(const float* __restrict__ in, float* out, int iters) {
    f32x16 c[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    float a[4], b[4];
    for (int j = 0; j < 4; ++j) {
        a[j] = in[(threadIdx.x * 8 + j) & 4095];
        b[j] = in[(threadIdx.x * 8 + 4 + j) & 4095];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 3], b[i], c[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += c[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4);
    hipMalloc(&out, 256 * 512 * 4);
    std::vector<float> h(4096);
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (auto& x : h) x = mode ? (float)rand() / RAND_MAX * 0.1f - 0.05f : 0.03125f;
        hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
        for (int wps = 1; wps <= 2; ++wps) {
            const int iters = 4000;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            k<<<256, 256 * wps>>>(in, out, 50);
            hipDeviceSynchronize();
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                k<<<256, 256 * wps>>>(in, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double flops = 256.0 * 4 * wps * iters * 64 * 4096.0;
                printf("%-8s operands, %d wave(s) per SIMD: %.3f ms  %.1f TFLOP/s  (%.2f of 157.3)\n", mode ? "random" : "constant", wps, ms,
                       flops / ms / 1e9, flops / ms / 1e9 / 157.3);
            }
        }
    }
    return 0;
}
