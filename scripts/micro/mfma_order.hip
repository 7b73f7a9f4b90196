// Does the ORDER in which a k-step's 64 MFMAs walk the wave tile's 8 x 8 block grid change what the chip sustains?  MI355X / gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_order scripts/micro/mfma_order.hip && /tmp/mfma_order
//
// Background (profiles/r05_mfma_ceiling.txt): the bare v_mfma_f32_16x16x32_bf16 loop sustains 1.8 PFLOP/s on random operands and 2.3 on
// zeros at the same 16.4 cycles per MFMA -- the clock follows the power the data's toggling draws.  K1s holds 8 A fragments (concept
// blocks) and 8 B fragments (image blocks) per k-step and issues the 64 block products in some order; between two consecutive MFMAs
// either one operand changes or both do.  This loop has K1s' register picture (64 accumulator blocks in AGPRs, 8 + 8 fragments in
// VGPRs, one wave per SIMD, 256 CUs) and nothing else, and walks the grid
//   0 row-major      A held for 8 MFMAs, B changes every MFMA (wraps 7 -> 0 at the row's end: both change there)
//   1 serpentine     as 0 but every other row backwards: exactly ONE operand changes at every step
//   2 diagonal       (i, (i + d) % 8): both operands change at every step
//   3 column-major   B held for 8, A changes every MFMA
//   4 2 x 2 quads    Z-order: neither operand is held longer than two MFMAs, one changes per step three times out of four
//   5 same block     one A, one B, one accumulator... (64 different accumulators, SAME a[0], b[0]): the floor of operand toggling
// on random operands and on zeros.  ms = HIP events around one launch; cycles = s_memtime of the loop (median wave).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA16A(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))

__device__ __forceinline__ bf16x8 rnd_frag(unsigned seed, int zero) {
    bf16x8 v;
    unsigned s = seed * 2654435761u + 12345u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s = s * 1664525u + 1013904223u;
        // unit-norm 512-d embeddings: components ~ N(0, 1/512); a uniform of the same scale toggles the same mantissa bits
        v[j] = zero ? (__bf16)0.f : (__bf16)((((int)(s >> 9) & 0xffff) * (2.0f / 65536.0f) - 1.0f) * 0.0765f);
    }
    return v;
}

template <int ORDER>
__device__ __forceinline__ constexpr int blk_i(int u) {
    return ORDER == 0 ? u >> 3 : ORDER == 1 ? u >> 3 : ORDER == 2 ? (u & 7) : ORDER == 3 ? (u & 7)
         : ORDER == 4 ? (((u >> 4) & 3) * 2 + ((u >> 1) & 1)) : 0;
}
template <int ORDER>
__device__ __forceinline__ constexpr int blk_j(int u) {
    return ORDER == 0 ? (u & 7) : ORDER == 1 ? (((u >> 3) & 1) ? 7 - (u & 7) : (u & 7)) : ORDER == 2 ? (((u & 7) + (u >> 3)) & 7) : ORDER == 3 ? u >> 3
         : ORDER == 4 ? (((u >> 2) & 3) * 2 + (u & 1)) : 0;
}

template <int ORDER, int ZERO>
__global__ __launch_bounds__(256, 1) void order_kernel(float* out, unsigned long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x4 d[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = rnd_frag(threadIdx.x * 11u + i + blockIdx.x, ZERO);
        b[i] = rnd_frag(threadIdx.x * 13u + i * 5u + 3u, ZERO);
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            constexpr int dummy = 0; (void)dummy;
            const int i = blk_i<ORDER>(u), j = blk_j<ORDER>(u);
            MFMA16A(d[ORDER == 5 ? u : i * 8 + j], a[i], b[j]);
        }
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) s += d[i][0] + d[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

static float* g_out;
static unsigned long long* g_cyc;

template <int ORDER, int ZERO>
static void run(const char* name) {
    const int iters = 4000;     // x 64 MFMAs x 16.4 cycles = 4.2 M cycles ~ 2.3 ms
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((order_kernel<ORDER, ZERO>), dim3(256), dim3(256), 0, 0, g_out, g_cyc, iters);   // warm, and lets the clock settle
    hipLaunchKernelGGL((order_kernel<ORDER, ZERO>), dim3(256), dim3(256), 0, 0, g_out, g_cyc, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((order_kernel<ORDER, ZERO>), dim3(256), dim3(256), 0, 0, g_out, g_cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 4;
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), g_cyc, 1024 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cy = (double)h[512] / (iters * 64.0);
    const double fl = 2.0 * 16 * 16 * 32 * 64.0 * iters * 1024;
    printf("  %-13s %s: %.3f ms = %4.0f TFLOP/s, %.2f cycles per MFMA, %.2f GHz\n", name, ZERO ? "zeros " : "random", ms, fl / ms * 1e-9, cy,
           (double)h[512] / (ms * 1e6) );
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    hipMalloc(&g_out, 256 * 256 * sizeof(float));
    hipMalloc(&g_cyc, 1024 * sizeof(unsigned long long));
    for (int round = 0; round < 3; ++round) {
        printf("round %d\n", round);
        run<0, 0>("row-major"); run<1, 0>("serpentine"); run<2, 0>("diagonal"); run<3, 0>("column-major"); run<4, 0>("2x2 quads"); run<5, 0>("same operands");
        run<0, 1>("row-major"); run<2, 1>("diagonal");
    }
    return 0;
}
