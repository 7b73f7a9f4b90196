// Does a wave's 1-KB store overlap with its own vector work?  K1s' epilogue issues one buffer_store_dwordx4 (4 rows x 256 B) per
// group of ~34 vector instructions (8 v_exp, 8 moves, 4 packed adds, 4 conversions ...) and pays ~30 us per launch for the stores
// wherever they sit (profiles/r04_gexp_v4.txt, r04_gexp_v5.txt).  One workgroup of 4 waves per CU; per iteration a group of
// NV x (v_exp + v_fma) and (STORE) one store; cycles per iteration by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o store_overlap scripts/micro/store_overlap.hip && ./store_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NV, int STORE, int SKEW>
__global__ __launch_bounds__(256, 1) void k(unsigned short* out, long long pitch_b, int iters, unsigned long long* cyc, float seed) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* base = (char*)out + (long long)(blockIdx.x * 4 + wave) * 8 * 64 * pitch_b;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    const unsigned lane_off = (unsigned)((lane >> 4) * pitch_b + (lane & 15) * 16);
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = seed + i * 0.01f + lane * 1e-3f;
    u32x4 v = {(unsigned)lane, 2u, 3u, 4u};
    if (SKEW) for (int i = 0; i < wave; ++i) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {                       // 16 groups per iteration = 64 rows of a band
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                float t;
                asm volatile("v_exp_f32 %0, %1" : "=v"(t) : "v"(x[i & 7]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i & 7]) : "v"(t), "v"(1e-3f));
            }
            if (STORE) {
                v.x = __float_as_uint(x[0]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane_off + (unsigned)(((it & 7) * 64 + g * 4) * pitch_b), 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 1234.5f) out[0] = 1;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NV, int STORE, int SKEW>
void run(unsigned short* out, unsigned long long* cyc, long long pitch_b, int grid) {
    const int iters = 100;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NV, STORE, SKEW>), dim3(grid), dim3(256), 0, 0, out, pitch_b, iters, cyc, 0.5f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    (void)hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%3d CUs  %2d x (v_exp + v_fma) per group, store %d, skew %d: %7.1f cycles per group (median wave)\n", grid, NV, STORE, SKEW,
           (double)h[grid * 2] / (iters * 16.0));
}

int main() {
    const long long pitch_b = 20224;
    unsigned short* out; unsigned long long* cyc;
    if (hipMalloc(&out, (size_t)256 * 4 * 8 * 64 * pitch_b) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&cyc, 1024 * 8);
    for (int grid : {32, 256}) {
        run<0, 1, 0>(out, cyc, pitch_b, grid);
        run<8, 0, 0>(out, cyc, pitch_b, grid);  run<8, 1, 0>(out, cyc, pitch_b, grid);  run<8, 1, 1>(out, cyc, pitch_b, grid);
        run<16, 0, 0>(out, cyc, pitch_b, grid); run<16, 1, 0>(out, cyc, pitch_b, grid); run<16, 1, 1>(out, cyc, pitch_b, grid);
        run<32, 0, 0>(out, cyc, pitch_b, grid); run<32, 1, 0>(out, cyc, pitch_b, grid); run<32, 1, 1>(out, cyc, pitch_b, grid);
    }
    return 0;
}
