// Microbenchmark: how many bytes per clock one CU takes in from L2, (a) by LDS-DMA (global_load_lds_dwordx4, 1 KB per
// wave instruction) and (b) by plain global_load_dwordx4 into registers, as a function of the number of issuing waves.
// One workgroup per CU; every wave streams over a small window (its share of `win_kb` per CU: L2-resident) `iters` times.
// Dev tool: hipcc --offload-arch=gfx950 -O3 -o ldsdma_rate ldsdma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT>
__global__ __launch_bounds__(1024) void k_dma(const char* __restrict__ src, size_t win_bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* base = src + (size_t)blockIdx.x * win_bytes;          // this CU's window
    const size_t per_wave = win_bytes / nw;                          // a multiple of 1 KB
    const char* p = base + (size_t)wave * per_wave + lane * 16;
    char* lds = smem + wave * (INFLIGHT * 1024);
    const int pieces = (int)(per_wave / 1024);
    for (int it = 0; it < iters; ++it) {
        for (int q = 0; q < pieces; q += INFLIGHT) {
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)((q + j) % pieces) * 1024),
                                                 (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (sink && lane == 0) sink[blockIdx.x * nw + wave] = *(unsigned*)(lds);
}

template <int INFLIGHT>
__global__ __launch_bounds__(1024) void k_reg(const char* __restrict__ src, size_t win_bytes, int iters, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* base = src + (size_t)blockIdx.x * win_bytes;
    const size_t per_wave = win_bytes / nw;
    const char* p = base + (size_t)wave * per_wave + lane * 16;
    const int pieces = (int)(per_wave / 1024);
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        for (int q = 0; q < pieces; q += INFLIGHT) {
            u32x4 v[INFLIGHT];
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) v[j] = *(const u32x4*)(p + (size_t)((q + j) % pieces) * 1024);
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) acc ^= v[j];
        }
    }
    if (sink && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x * nw + wave] = acc.x;
}

// GEMM-like staging: one DMA instruction fetches 64 / LPR rows x (16 * LPR) bytes (row pitch 1 KB + 128 B), and -- with
// `shared` -- the 32 CUs of an XCD (blockIdx % 8 equal) walk the SAME window, as the workgroups of an XCD share a tile.
template <int INFLIGHT, int LPR, bool SWZ = false>
__global__ __launch_bounds__(1024) void k_rows(const char* __restrict__ src, size_t win_bytes, int iters, unsigned* sink, int shared) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t pitch = 1152;
    const size_t rows = win_bytes / pitch;                           // rows of this CU's window
    const char* base = src + (size_t)(shared ? (blockIdx.x & 7) : blockIdx.x) * win_bytes;
    const int rpi = 64 / LPR;                                        // rows per instruction
    const size_t groups = rows / rpi / nw;                           // row groups per wave
    char* lds = smem + wave * (INFLIGHT * 1024);
    const int kch = (int)(1024 / (16 * LPR));                        // K chunks of 16*LPR bytes per 1 KB row
    for (int it = 0; it < iters; ++it) {
        for (size_t g = 0; g < groups; ++g) {
            const size_t r = (g * nw + wave) * rpi + lane / LPR;
            for (int k = 0; k < kch; k += INFLIGHT) {
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(base + r * pitch + (size_t)((k + j) % kch) * (16 * LPR) +
                                                                        (SWZ ? ((lane % LPR) ^ ((r >> 2) & (LPR - 1))) : (lane % LPR)) * 16),
                        (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    if (sink && lane == 0) sink[blockIdx.x * nw + wave] = *(unsigned*)(lds);
}

// DMA beside consumers: 4 loader waves (the GEMM's staging pattern, L2-resident window shared by the XCD) + 8 consumer waves
// that loop over ds_read_b128 (MODE & 1) and / or v_mfma_f32_32x32x16_bf16 (MODE & 2) with no synchronisation between the
// two groups.  Every wave runs a fixed number of iterations; the DMA rate is bytes / kernel time when the loaders are the
// longer-running group (the host prints both groups' work).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(768) void k_mix(const char* __restrict__ src, size_t win_bytes, int dma_iters, int cons_iters, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 8) {                                   // ---- loaders: 8 DMAs (16 rows x 64 B each) then wait, like one stage
        const int lw = wave - 8;
        const size_t pitch = 1152;
        const char* base = src + (size_t)(blockIdx.x & 7) * win_bytes;
        char* lds = smem + lw * 8192;
        const char* p[8];                              // 8 row groups of 16 rows: 128 of the window's 256 rows per loader pair
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = base + (size_t)(((lw & 1) * 128 + 16 * j + (lane >> 2)) & 255) * pitch + (lane & 3) * 16;
        for (int it = 0; it < dma_iters; ++it) {
            const int k0 = (it & 15) * 64;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p[j] + k0),
                                                 (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        return;
    }
    // ---- consumers
    const char* rd = smem + 32768 + wave * 8192 + lane * 16;
    f32x16_t acc[2];
    for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
    bf16x8_t a = {}, b = {};
    for (int it = 0; it < cons_iters; ++it) {
        if (MODE & 1) {
            bf16x8_t f[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) f[j] = *reinterpret_cast<const bf16x8_t*>(rd + j * 1024);
            if (MODE & 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[j % 4], f[4 + (j & 1)], acc[j & 1], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) asm volatile("" ::"v"(f[j]));
            }
        } else if (MODE & 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 1], 0, 0, 0);
        }
    }
    if (acc[0][0] + acc[1][0] == 12345.f) sink[threadIdx.x] = acc[0][0];
}

template <int MODE>
void run_mix(const char* what, const char* src, float* sink) {
    hipFuncSetAttribute((const void*)k_mix<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t win = 288 * 1024;
    for (int dma_on = 0; dma_on <= 1; ++dma_on) {
        const int dma_iters = dma_on ? 2000 : 0, cons_iters = 4000;     // consumer iteration = half a GEMM stage (6 reads, 8 MFMAs)
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_mix<MODE>, dim3(256), dim3(768), 160 * 1024 - 1024, 0, src, win, 10, 10, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mix<MODE>, dim3(256), dim3(768), 160 * 1024 - 1024, 0, src, win, dma_iters, cons_iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.0e9;
        printf("mix %-22s DMA %s: %7.3f ms = %8.0f cycles; per consumer iteration %6.1f cycles%s\n", what, dma_on ? "on " : "off", ms, cyc,
               cyc / cons_iters, dma_on ? "" : "  (consumers alone)");
        if (dma_on) printf("    if the loaders finished last: %.1f B/clk/CU of DMA (4 x 8 KB per loader iteration = %.0f cycles each)\n",
                           4.0 * 8192 * dma_iters / cyc, cyc / dma_iters);
    }
}

// Both groups run until a common deadline (s_memtime) and report their iteration counts: the DMA rate and the MFMA rate
// that COEXIST on a CU.  LD: 0 = global_load_lds with 64-bit per-lane addresses, 1 = raw_buffer_load_lds (SGPR resource +
// one 32-bit VGPR offset); PRIO: s_setprio for the loaders.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
template <int MODE, int LD, int PRIO>
__global__ __launch_bounds__(768) void k_box(const char* __restrict__ src, size_t win_bytes, unsigned long long ticks, unsigned* counts) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t_end = __builtin_amdgcn_s_memtime() + ticks;
    unsigned n = 0;
    if (wave >= 8) {
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        const int lw = wave - 8;
        const size_t pitch = 1152;
        const char* base = src + (size_t)(blockIdx.x & 7) * win_bytes;
        char* lds = smem + lw * 8192;
        const char* p[8];
        unsigned off[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            off[j] = (unsigned)((((lw & 1) * 128 + 16 * j + (lane >> 2)) & 255) * pitch + (lane & 3) * 16);
            p[j] = base + off[j];
        }
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)win_bytes, 0x00020000);
        while ((n & 7) != 0 || __builtin_amdgcn_s_memtime() < t_end) {
            const int k0 = (n & 15) * 64;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (LD == 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p[j] + k0),
                                                     (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + j * 1024), 16, off[j], k0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ++n;
        }
    } else {
        const char* rd = smem + 32768 + wave * 8192 + lane * 16;
        f32x16_t acc[2];
        for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
        bf16x8_t a = {}, b = {};
        while ((n & 7) != 0 || __builtin_amdgcn_s_memtime() < t_end) {
            if (MODE & 1) {
                bf16x8_t f[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) f[j] = *reinterpret_cast<const bf16x8_t*>(rd + j * 1024);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[j % 4], f[4 + (j & 1)], acc[j & 1], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 1], 0, 0, 0);
            }
            ++n;
        }
        if (acc[0][0] + acc[1][0] == 12345.f) counts[0] = 1;
    }
    if (lane == 0) counts[1 + blockIdx.x * 12 + wave] = n;
}

template <int MODE, int LD, int PRIO>
void run_box(const char* what, const char* src, unsigned* counts) {
    hipFuncSetAttribute((const void*)k_box<MODE, LD, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned long long ticks = 4000000;          // s_memtime ticks = shader cycles: ~2 ms
    hipMemset(counts, 0, (1 + 256 * 12) * 4);
    hipLaunchKernelGGL((k_box<MODE, LD, PRIO>), dim3(256), dim3(768), 160 * 1024 - 1024, 0, src, (size_t)288 * 1024, ticks, counts);
    hipDeviceSynchronize();
    static unsigned h[1 + 256 * 12];
    hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost);
    double ld = 0, cs = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 12; ++w) (w >= 8 ? ld : cs) += h[1 + b * 12 + w];
    const double cyc = (double)ticks;
    // per CU: loader iterations x 8 KB per wave; consumer iterations x 8 MFMAs x 32 cycles, 2 waves per SIMD
    printf("box %-34s: DMA %5.1f B/clk/CU beside MFMA pipes %4.1f %% busy\n", what, ld / 256 * 8192 / cyc,
           100.0 * (cs / 256 / 4) * 8 * 32 / cyc);
}

template <typename K>
void run_rows(const char* name, K kern, int waves, int inflight, int lpr, size_t win_kb, int shared, const char* src, unsigned* sink) {
    const int iters = 200;
    const size_t win = win_kb * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t shmem = (size_t)waves * inflight * 1024;
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), shmem, 0, src, win, 4, sink, shared);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), shmem, 0, src, win, iters, sink, shared);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const size_t rows = win / 1152, rpi = 64 / lpr, groups = rows / rpi / waves, kch = 1024 / (16 * lpr);
    const size_t per_pass = ((kch + inflight - 1) / inflight) * inflight;
    const double bytes = (double)256 * waves * groups * per_pass * 1024.0 * iters;
    printf("%-9s waves/CU=%2d in flight/wave=%2d %3d B per row and instr, window/CU=%4zu KB %s: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU\n",
           name, waves, inflight, 16 * lpr, win_kb, shared ? "shared by the XCD" : "private          ", ms, bytes / ms / 1e9,
           bytes / ms / 1e6 / 256 / 2.0);
}

template <typename K>
void run(const char* name, K kern, int waves, int inflight, size_t win_kb, const char* src, unsigned* sink, size_t shmem) {
    const int iters = 400;
    const size_t win = win_kb * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), shmem, 0, src, win, 4, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), shmem, 0, src, win, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const size_t per_wave = win / waves;
    const size_t pieces = per_wave / 1024;
    const size_t issued = ((pieces + inflight - 1) / inflight) * inflight;     // pieces issued per pass per wave
    const double bytes = (double)256 * waves * issued * 1024.0 * iters;
    printf("%-4s waves/CU=%2d in flight/wave=%2d window/CU=%4zu KB: %7.3f ms  %6.2f TB/s  %5.1f GB/s per CU  (%.1f B/clk/CU at 2.0 GHz)\n",
           name, waves, inflight, win_kb, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.0);
}

int main() {
    const size_t total = (size_t)256 * 4096 * 1024;    // up to 4 MB per CU
    char* src; unsigned* sink;
    hipMalloc((void**)&src, total);
    hipMemset(src, 1, total);
    hipMalloc((void**)&sink, (1 + 256 * 16) * 4);
#define DMA(W, F, KB) do { hipFuncSetAttribute((const void*)k_dma<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                           run("dma", k_dma<F>, W, F, KB, src, sink, (size_t)W * F * 1024); } while (0)
#define REG(W, F, KB) run("reg", k_reg<F>, W, F, KB, src, sink, 0)
    for (size_t kb : {(size_t)96, (size_t)768}) {       // 24 MB (L2-resident across the chip) and 192 MB (Infinity Cache)
        printf("---- window per CU %zu KB (chip: %zu MB)\n", kb, kb * 256 / 1024);
        DMA(1, 8, kb); DMA(2, 8, kb); DMA(4, 8, kb); DMA(4, 16, kb); DMA(8, 8, kb); DMA(12, 8, kb); DMA(16, 8, kb);
        REG(1, 8, kb); REG(2, 8, kb); REG(4, 8, kb); REG(4, 16, kb); REG(8, 8, kb); REG(12, 8, kb); REG(16, 8, kb);
    }
    printf("---- GEMM-like staging pattern (rows of 1152 B, L2-resident windows)\n");
#define ROWS(W, F, LPR, KB, SH) do { hipFuncSetAttribute((const void*)k_rows<F, LPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                                    run_rows("dma-rows", k_rows<F, LPR>, W, F, LPR, KB, SH, src, sink); } while (0)
    // the 16-byte chunks of a row piece fetched in XOR-permuted order (what a swizzled LDS image asks of the source address)
#define ROWS_SWZ(W, F, LPR, KB, SH) do { hipFuncSetAttribute((const void*)k_rows<F, LPR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                                        run_rows("dma-swz ", k_rows<F, LPR, true>, W, F, LPR, KB, SH, src, sink); } while (0)
    ROWS_SWZ(4, 8, 4, 288, 1);
    ROWS_SWZ(4, 8, 8, 288, 1);
    ROWS_SWZ(8, 8, 4, 288, 1);
    for (int sh : {0, 1}) {
        ROWS(4, 8, 4, 288, sh);      // 64 B per row (a 32-element bf16 K-tile): what gemm_nt_bf16_exp_kernel stages
        ROWS(4, 8, 8, 288, sh);      // 128 B per row (64 elements): whole lines
        ROWS(4, 4, 16, 288, sh);     // 256 B per row
        ROWS(4, 1, 64, 288, sh);     // whole 1 KB rows
        ROWS(8, 8, 4, 288, sh);
        ROWS(8, 8, 8, 288, sh);
    }
    printf("---- LDS-DMA beside ds_read_b128 / MFMA consumers (4 loader + 8 consumer waves per CU, no synchronisation)\n");
    run_mix<0>("idle consumers", src, (float*)sink);
    run_mix<1>("ds_read_b128 only", src, (float*)sink);
    run_mix<2>("MFMA only", src, (float*)sink);
    run_mix<3>("ds_read_b128 + MFMA", src, (float*)sink);
    printf("---- time-boxed: DMA rate and MFMA utilisation that coexist on a CU (4 loader + 8 MFMA waves)\n");
    run_box<2, 0, 0>("global_load_lds, prio 0", src, sink);
    run_box<2, 0, 3>("global_load_lds, loader prio 3", src, sink);
    run_box<2, 1, 0>("raw_buffer_load_lds, prio 0", src, sink);
    run_box<2, 1, 3>("raw_buffer_load_lds, loader prio 3", src, sink);
    run_box<3, 0, 0>("global_load_lds + ds_reads, prio 0", src, sink);
    run_box<3, 1, 3>("raw_buffer + ds_reads, prio 3", src, sink);
    return 0;
}
