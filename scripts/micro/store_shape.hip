// What does a 1-KB store instruction cost by SHAPE?  K1s' epilogue writes 128 KB of E per tile and CU; round 4 measured ~45 TA cycles
// per buffer_store_dwordx4 of 4 rows x 256 B and no gain from moving the stores into the K loop (profiles/r04_gexp_v5.txt).
//   hipcc --offload-arch=gfx950 -O3 -o store_shape scripts/micro/store_shape.hip && ./store_shape
// One workgroup of 4 waves per CU (one wave per SIMD), each wave stores its own 16-byte pieces from registers, NST instructions back
// to back per iteration, rows of a [rows x 20224 B] buffer (K1s' pitch at 10 000 concepts).  Shapes: R rows x (1024 / R) bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int ROWS>
__global__ __launch_bounds__(256, 1) void store_kernel(unsigned short* out, long long pitch_b, int iters, unsigned long long* cyc, int aux_dummy) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER = 64 / ROWS;                     // lanes per row
    const int r = lane / PER, c = lane % PER;
    // this wave's window: rows [(block * 4 + wave) * 64 ...), a fresh 64-row band per iteration (mod 8 bands: 2 MB per CU)
    char* base = (char*)out + (long long)(blockIdx.x * 4 + wave) * 8 * 64 * pitch_b;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned band = (unsigned)((it & 7) * 64) * (unsigned)pitch_b;
#pragma unroll
        for (int k = 0; k < 64 / ROWS; ++k) {          // 64 rows x 1 KB... each instruction: ROWS rows x (1024 / ROWS) bytes
#pragma unroll
            for (int q = 0; q < 1; ++q) {
                const unsigned off = band + (unsigned)((k * ROWS + r) * pitch_b) + (unsigned)(c * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
            }
        }
        v.x += 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int ROWS>
void run(unsigned short* out, unsigned long long* cyc, long long pitch_b, int grid) {
    const int iters = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(store_kernel<ROWS>, dim3(grid), dim3(256), 0, 0, out, pitch_b, iters, cyc, 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(store_kernel<ROWS>, dim3(grid), dim3(256), 0, 0, out, pitch_b, iters, cyc, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + grid * 4);
    const double ninstr = (double)iters * (64 / ROWS);                 // store instructions per wave
    const double bytes = (double)grid * 4 * ninstr * 1024;
    printf("%3d CUs  %2d rows x %4d B per instruction: %7.3f ms  %6.2f TB/s  %6.1f cycles per store instruction and wave (median), %5.1f B/clk/CU\n",
           grid, ROWS, 1024 / ROWS, ms, bytes / ms / 1e9, (double)h[grid * 2] / ninstr, 4.0 * 1024 / ((double)h[grid * 2] / ninstr));
}

int main() {
    const long long pitch_b = 20224;
    unsigned short* out; unsigned long long* cyc;
    const size_t bytes = (size_t)256 * 4 * 8 * 64 * pitch_b;           // 10.6 GB?  (256 CUs x 4 waves x 8 bands x 64 rows x 20 KB) = 10.6 GB
    if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&cyc, 1024 * 8);
    for (int grid : {32, 256}) {
        run<1>(out, cyc, pitch_b, grid); run<2>(out, cyc, pitch_b, grid); run<4>(out, cyc, pitch_b, grid); run<8>(out, cyc, pitch_b, grid);
        run<16>(out, cyc, pitch_b, grid); run<32>(out, cyc, pitch_b, grid);
    }
    return 0;
}
