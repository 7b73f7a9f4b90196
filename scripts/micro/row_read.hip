// What does K3's access pattern cost by itself, warm (the 369 MB matrix read again and again: part of it stays in the 256 MB
// Infinity Cache) and cold (2 GB written between the reads)?  One workgroup per 10 000-float row, every thread's 16-byte loads
// issued together, a maximum as the only arithmetic -- against a plain grid-stride stream over the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/row_read scripts/micro/row_read.hip && /tmp/row_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int THREADS, int QUADS>
__global__ __launch_bounds__(THREADS) void row_max(const float* __restrict__ A, int ld, int N, float* __restrict__ out) {
    const float* row = A + (size_t)blockIdx.x * ld;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, N * 4, 0x00020000);
    f32x4 v[QUADS];
#pragma unroll
    for (int q = 0; q < QUADS; ++q) v[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16, q * THREADS * 16, 0));
    float m = -1e30f;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) m = fmaxf(fmaxf(fmaxf(m, v[q][0]), fmaxf(v[q][1], v[q][2])), v[q][3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)] = m;
}

__global__ __launch_bounds__(256) void stream_max(const f32x4* __restrict__ A, size_t n4, float* __restrict__ out) {
    float m = -1e30f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = A[i];
        m = fmaxf(fmaxf(fmaxf(m, v[0]), fmaxf(v[1], v[2])), v[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = m;
}

__global__ void fill(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

int main() {
    const int U = 9216, N = 10000, ld = 10000;
    float *A, *out, *flush;
    const size_t nflush = (size_t)512 << 20;
    hipMalloc(&A, (size_t)U * ld * 4);
    hipMalloc(&out, (size_t)U * 16 * 4);
    hipMalloc(&flush, nflush * 4);
    fill<<<4096, 256>>>(A, (size_t)U * ld, 1.0f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double bytes = (double)U * N * 4;
    for (int cold = 0; cold < 2; ++cold)
        for (int k = 0; k < 4; ++k) {
            float best = 1e9f, sum = 0.f;
            for (int rep = 0; rep < 8; ++rep) {
                if (cold) fill<<<4096, 256>>>(flush, nflush, (float)rep);
                hipEventRecord(e0);
                if (k == 0) row_max<256, 10><<<U, 256>>>(A, ld, N, out);
                if (k == 1) row_max<512, 5><<<U, 512>>>(A, ld, N, out);
                if (k == 2) stream_max<<<256 * 8, 256>>>((const f32x4*)A, (size_t)U * ld / 4, out);
                if (k == 3) stream_max<<<256 * 32, 256>>>((const f32x4*)A, (size_t)U * ld / 4, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
            }
            const char* names[4] = {"row per workgroup, 256 x 10 quads", "row per workgroup, 512 x 5 quads", "grid-stride stream, 8 WG/CU", "grid-stride stream, 32 WG/CU"};
            printf("%-5s %-36s best %.1f us (%.2f TB/s)  mean %.1f us (%.2f TB/s)\n", cold ? "cold" : "warm", names[k], best * 1e3, bytes / best / 1e9,
                   sum / 6 * 1e3, bytes / (sum / 6) / 1e9);
        }
    return 0;
}
