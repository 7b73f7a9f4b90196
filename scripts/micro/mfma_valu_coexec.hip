// Microbenchmark: do VALU instructions of one wave execute while ANOTHER wave of the same SIMD runs bf16 MFMAs?
// Two waves per SIMD (512-thread workgroups, one per CU): waves 0-3 issue MFMAs only, waves 4-7 VALU only (v_mul + v_add chains,
// or the K1s epilogue mix: v_fma + v_exp + v_add + v_cvt_pk).  Times: MFMA waves alone, VALU waves alone, both together.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_valu_coexec mfma_valu_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE /*1 = mfma waves, 2 = valu waves, 3 = both*/, int KIND /*0 bf16 32x32x16, 1 f32 32x32x2, 2 bf16 16x16x32*/>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (!(MODE & 1)) return;
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        __attribute__((ext_vector_type(4))) float d = {0.f, 0.f, 0.f, 0.f};
        bf16x8 av, bv;
        for (int j = 0; j < 8; ++j) { av[j] = (__bf16)(a + j); bv[j] = (__bf16)(b + threadIdx.x); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (KIND == 0) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
                else if (KIND == 1) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
                else { d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, d, 0, 0, 0); d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, d, 0, 0, 0); }
            }
        }
        s = c[0] + d[0];
    } else {
        if (!(MODE & 2)) return;
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = a + j + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < 128; ++v) x[v & 7] = x[v & 7] * b + a;   // 128 x (v_mul, v_add)
        }
        for (int j = 0; j < 8; ++j) s += x[j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int KIND>
float run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, KIND><<<256, 512>>>(out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, KIND><<<256, 512>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 4000;
    const char* names[3] = {"bf16 32x32x16", "f32 32x32x2", "bf16 16x16x32 (x2)"};
    float m[3], v, bt[3];
    m[0] = run<1, 0>(out, iters); m[1] = run<1, 1>(out, iters); m[2] = run<1, 2>(out, iters);
    v = run<2, 0>(out, iters);
    bt[0] = run<3, 0>(out, iters); bt[1] = run<3, 1>(out, iters); bt[2] = run<3, 2>(out, iters);
    printf("per iteration: 16 MFMAs in one wave of a SIMD, 256 VALU instructions (128 x v_mul + v_add) in the other\n");
    printf("VALU waves alone: %.3f ms\n", v);
    for (int i = 0; i < 3; ++i)
        printf("%-20s MFMA waves alone %.3f ms; both %.3f ms; sum %.3f ms  -> overlap %.0f %% of the shorter\n", names[i], m[i], bt[i],
               m[i] + v, 100.0 * (m[i] + v - bt[i]) / (m[i] < v ? m[i] : v));
    return 0;
}
