// Microbenchmarks behind the round-3 K1s layouts (VERDICT r2 #2a); MI355X / gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/mfma_fill scripts/micro/mfma_fill.hip && ./scripts/micro/mfma_fill
//
// A. SAME-WAVE fillers: one wave's own VALU instructions placed between its own bf16 MFMAs (the case the guide's
//    constants table prices: "single-issue instructions HIDDEN per v_mfma_f32_32x32x16_bf16 gap"), at one and at two
//    waves per SIMD; round 2's mfma_valu_coexec.hip measured the OTHER case (MFMA stream and VALU stream in different waves).
// B. CROSS-WAVE with static priority: the MFMA wave at s_setprio 1 / the VALU wave at s_setprio 1 / neither.
// C. Self-issued LDS-DMA inside a one-wave-per-SIMD MFMA stream: one 1-KiB `buffer_load ... lds` piece per NPER MFMAs
//    (+ the fragment reads of a 128 x 128 wave tile), counted vmcnt: what a 4-wave 512-register GEMM pays for having no
//    loader waves.
// D. MFMA shape on RANDOM operands (DVFS): 32x32x16 against 16x16x32, operands in registers, one wave per SIMD.
// Every stream is inline asm (volatile, in program order), so the emitted interleave is the written one.  Cycles are
// s_memtime deltas of the timed loop (median over the launch's waves); ms is the HIP-event time of the launch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA32(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
// accumulators in AGPRs: the "v" class is v0-v255 only, and a 128 x 128 wave tile alone is 256 registers
#define MFMA32A(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define MFMA16A(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define MFMA16(c, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define V_FMA(x, m, k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(k))
#define V_EXP(t, x) asm volatile("v_exp_f32 %0, %1" : "=v"(t) : "v"(x))
#define V_ADD(s, t) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(t))
#define V_CVT(d, a, b) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))

__device__ __forceinline__ bf16x8 rnd_frag(unsigned seed) {
    bf16x8 v;
    unsigned s = seed * 2654435761u + 12345u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s = s * 1664525u + 1013904223u;
        v[j] = (__bf16)(((int)(s >> 9) & 0xffff) * (2.0f / 65536.0f) - 1.0f);   // uniform [-1, 1)
    }
    return v;
}

// KIND: 0 v_fma_f32, 1 v_exp_f32, 2 the K1s epilogue mix per element (v_exp + v_add, a v_cvt_pk per two elements)
template <int KIND, int F>
__device__ __forceinline__ void fillers(float (&x)[8], float (&t)[4], float& s0, float& s1, unsigned& d, float m, float k) {
#pragma unroll
    for (int f = 0; f < F; ++f) {
        if (KIND == 0) V_FMA(x[f & 7], m, k);
        else if (KIND == 1) V_EXP(t[f & 3], x[f & 7]);
        else {
            V_EXP(t[(2 * f) & 3], x[f & 7]);
            V_ADD(s0, t[(2 * f) & 3]);
            V_EXP(t[(2 * f + 1) & 3], x[(f + 1) & 7]);
            V_ADD(s1, t[(2 * f + 1) & 3]);
            V_CVT(d, t[(2 * f) & 3], t[(2 * f + 1) & 3]);
        }
    }
}

// ---- A / B -----------------------------------------------------------------------------------------------------------------
// ROLE: 0 every wave runs MFMA + F fillers per gap (same-wave); 1 waves 0-3 MFMA only, waves 4-7 fillers only, 16 * F per
// iteration (cross-wave; WPS must be 2); 2 / 3: the filler waves / the MFMA waves of ROLE 1 alone.  PRIO: 0 none, 1 MFMA waves at s_setprio 1, 2 filler waves at s_setprio 1.
template <int WPS, int KIND, int F, int ROLE, int PRIO>
__global__ __launch_bounds__(256 * WPS) void fill_kernel(float* out, unsigned long long* cyc, int iters, float m, float k) {
    const int wave = threadIdx.x >> 6;
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    const bf16x8 a = rnd_frag(threadIdx.x + 7u * blockIdx.x), b = rnd_frag(threadIdx.x * 3u + 1u);
    float x[8], t[4] = {0.f, 0.f, 0.f, 0.f}, s0 = 0.f, s1 = 0.f;
    unsigned d = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = KIND == 0 ? 1.0f + j : -0.125f * (j + 1 + (threadIdx.x & 3));
    if (ROLE == 2 && wave < 4) return;      // filler waves alone
    if (ROLE == 3 && wave >= 4) return;     // MFMA waves alone
    const bool do_mfma = ROLE == 0 || wave < 4, do_fill = ROLE == 0 || wave >= 4;
    if (PRIO == 1 && do_mfma && ROLE == 1) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 2 && do_fill && ROLE == 1) __builtin_amdgcn_s_setprio(1);
    if (ROLE < 2) __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (ROLE == 0) {
                MFMA32(c[u & 3], a, b);
                fillers<KIND, F>(x, t, s0, s1, d, m, k);
            } else if (wave < 4) {
                MFMA32(c[u & 3], a, b);
            } else {
                fillers<KIND, F>(x, t, s0, s1, d, m, k);
            }
        }
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = s0 + s1 + __uint_as_float(d & 0x3f800000u);
#pragma unroll
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][7];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    s += t[0] + t[1] + t[2] + t[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

// ---- C ---------------------------------------------------------------------------------------------------------------------
// One wave per SIMD.  Per iteration: 32 MFMAs (one 32-deep stage of a 128 x 128 wave tile: 2 k-steps x 16), READS x 8
// ds_read_b128 per k-step (the tile's 4 + 4 fragments), and NDMA 1-KiB LDS-DMA pieces spread evenly over the 32 gaps
// (a 256 x 256 x 32 stage is 32 pieces over 4 waves = 8 per wave).  The DMA source is a per-workgroup 64 KB window that
// stays in L2; the ring is 4 x 32 KB of LDS; vmcnt(2 * NDMA) before each iteration = two stages in flight.
template <int NDMA, int READS, int FORM /*0 raw_buffer_load lds, 1 global_load_lds, 2 raw_buffer_load lds in K1s' row-major staging pattern: a piece = 16 rows x 64 B at a pitch of 1152 B (16 half lines) instead of 1 KiB contiguous (8 lines)*/>
__global__ __launch_bounds__(256, 1) void dma_kernel(float* out, unsigned long long* cyc, const unsigned short* src, int iters) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    f32x16 c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { fa[i] = rnd_frag(threadIdx.x + i); fb[i] = rnd_frag(threadIdx.x * 5u + i); }
    // FORM 3: every workgroup walks a shared 2 MB region (L2-resident, never in the 32 KB L1): the K1s case
    const unsigned short* base = (FORM == 3 || FORM >= 5) ? src : src + (size_t)(blockIdx.x & 255) * 32768;    // else: 64 KB window per workgroup
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (FORM == 3 || FORM >= 5) ? (2 << 20) : 65536, 0x00020000);
    const unsigned voff = FORM == 2 ? (unsigned)((lane >> 2) * 1152 + (lane & 3) * 16) : (unsigned)(lane * 16);
    // conflict-free fragment read addresses: 64-byte rows, chunk permuted by (row / 4) % 4 (the K1s stage image)
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned roff = (unsigned)(((wave & 1) * 128 + fr) * 64 + ((fh ^ ((fr >> 2) & 3)) * 16));
    for (int i = threadIdx.x; i < 131072 / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    __syncthreads();
    int stage = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
        char* st = smem + (stage & 3) * 32768;
        char* rd = smem + ((stage + 2) & 3) * 32768;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (READS && (u & 15) == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[i] = *reinterpret_cast<const bf16x8*>(rd + roff + i * 2048 + (u >> 4) * 32);
                    fb[i] = *reinterpret_cast<const bf16x8*>(rd + 16384 + roff + i * 2048 + (u >> 4) * 32);
                }
            }
            MFMA32A(c[u & 15], fa[(u >> 2) & 3], fb[u & 3]);
            if (FORM >= 5 && u == 16) {   // K1s' sync point: stage landed, this stage in registers, workgroup barrier
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (NDMA > 0 && (FORM == 6 ? (u >= 16 && u < 16 + NDMA) : (u % (32 / NDMA)) == (32 / NDMA) - 1)) {
                const int piece = (FORM == 6 ? (u - 16) : (u / (32 / NDMA))) * 4 + wave;     // 4 waves x NDMA pieces = the stage
                if (FORM == 0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(st + piece * 1024), 16, voff,
                                                             (piece * 1024) & 0xffff, 0, 0);
                else if (FORM == 3 || FORM >= 5)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(st + piece * 1024), 16, voff,
                                                             (int)(((blockIdx.x * 37u + (unsigned)stage * 32u + (unsigned)piece) * 1024u) & ((2u << 20) - 1u)), 0, 0);
                else if (FORM == 2)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(st + piece * 1024), 16, voff,
                                                             ((piece & 1) * 18432 + (piece >> 1) * 64) & 0xffff, 0, 0);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)base + ((piece * 1024) & 0xffff) + voff),
                                                     (__attribute__((address_space(3))) void*)(st + piece * 1024), 16, 0, 0);
            }
        }
        ++stage;
    }
    asm volatile("s_waitcnt vmcnt(0)\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][9];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// ---- D ---------------------------------------------------------------------------------------------------------------------
template <int SHAPE /*0: 32x32x16, 1: 16x16x32*/, int ZERO>
__global__ __launch_bounds__(256, 1) void shape_kernel(float* out, unsigned long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x16 c[4];
    f32x4 d[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = rnd_frag(threadIdx.x * 11u + i + blockIdx.x);
        b[i] = rnd_frag(threadIdx.x * 13u + i * 5u + 3u);
        if (ZERO)
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)0.f; b[i][j] = (__bf16)0.f; }
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
#pragma unroll
            for (int u = 0; u < 16; ++u) MFMA32(c[u & 3], a[u & 3], b[(u >> 2) & 3]);            // 16 x 32 cycles
        } else {
#pragma unroll
            for (int u = 0; u < 32; ++u) MFMA16(d[u & 15], a[u & 3], b[(u >> 2) & 3]);           // 32 x 16 cycles, same flops
        }
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][5];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += d[i][0] + d[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// ---- host ------------------------------------------------------------------------------------------------------------------
static float* g_out;
static unsigned long long* g_cyc;
static unsigned short* g_src;
struct Res { float ms; double cyc; };

template <typename L>
static Res timed(L launch, int waves) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(4000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    Res r;
    hipEventElapsedTime(&r.ms, e0, e1);
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), g_cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    r.cyc = (double)h[waves / 2];
    hipEventDestroy(e0); hipEventDestroy(e1);
    return r;
}

template <int WPS, int KIND, int F, int ROLE, int PRIO>
static Res run_fill() {
    return timed([](int it) { hipLaunchKernelGGL((fill_kernel<WPS, KIND, F, ROLE, PRIO>), dim3(256), dim3(256 * WPS), 0, 0, g_out, g_cyc, it, 1.0001f, 0.5f); },
                 256 * 4 * WPS);
}
template <int NDMA, int READS, int FORM>
static Res run_dma() {
    hipFuncSetAttribute((const void*)dma_kernel<NDMA, READS, FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    return timed([](int it) { hipLaunchKernelGGL((dma_kernel<NDMA, READS, FORM>), dim3(256), dim3(256), 131072, 0, g_out, g_cyc, g_src, it / 2); }, 1024);
}
template <int SHAPE, int ZERO>
static Res run_shape() {
    return timed([](int it) { hipLaunchKernelGGL((shape_kernel<SHAPE, ZERO>), dim3(256), dim3(256), 0, 0, g_out, g_cyc, it * 4); }, 1024);
}

#define ROW_A(WPS, KIND, F)                                                                                                    \
    do {                                                                                                                       \
        Res r = run_fill<WPS, KIND, F, 0, 0>();                                                                                \
        printf("  %d wave(s)/SIMD  %-8s %d per gap: %7.1f cycles per MFMA (wave), %.3f ms\n", WPS, kn[KIND], F,                \
               r.cyc / (4000.0 * 16), r.ms);                                                                                   \
    } while (0)

int main(int argc, char** argv) {
    const bool only_d = argc > 1 && argv[1][0] == 'D';      // `mfma_fill D`: section D alone (the bf16 MFMA ceiling on random data)
    hipMalloc(&g_out, 256 * 512 * sizeof(float));
    hipMalloc(&g_cyc, 4096 * sizeof(unsigned long long));
    hipMalloc(&g_src, 256 * 65536);
    hipMemset(g_src, 0x3f, 256 * 65536);
    const char* kn[3] = {"v_fma", "v_exp", "epi-mix"};
    if (!only_d) {
    printf("A. same-wave fillers between a wave's own v_mfma_f32_32x32x16_bf16 (32 cycles each when bare); epi-mix = 2 x (v_exp, v_add) + 1 v_cvt_pk\n");
    ROW_A(1, 0, 0); ROW_A(1, 0, 2); ROW_A(1, 0, 4); ROW_A(1, 0, 5); ROW_A(1, 0, 6); ROW_A(1, 0, 8); ROW_A(1, 0, 12);
    ROW_A(1, 1, 1); ROW_A(1, 1, 2); ROW_A(1, 1, 3); ROW_A(1, 1, 4);
    ROW_A(1, 2, 1); ROW_A(1, 2, 2); ROW_A(1, 2, 3);
    ROW_A(2, 0, 0); ROW_A(2, 0, 2); ROW_A(2, 0, 4); ROW_A(2, 0, 6); ROW_A(2, 0, 8);
    ROW_A(2, 1, 1); ROW_A(2, 1, 2); ROW_A(2, 2, 1); ROW_A(2, 2, 2);
    printf("   (2 waves/SIMD: the pair shares the matrix pipe, so 64 cycles per MFMA per wave = the pipe's 32)\n");
    printf("B. cross-wave: waves 0-3 MFMA only, waves 4-7 fillers only (16 x F per 16 MFMAs); cycles of the LONGER wave per 16-MFMA iteration\n");
    Res mf = run_fill<2, 0, 0, 3, 0>();
    {
        Res r0 = run_fill<2, 0, 4, 1, 0>(), r1 = run_fill<2, 0, 4, 1, 1>(), r2 = run_fill<2, 0, 4, 1, 2>();
        Res r3 = run_fill<2, 0, 8, 1, 0>(), r4 = run_fill<2, 0, 8, 1, 1>(), r5 = run_fill<2, 2, 2, 1, 0>(), r6 = run_fill<2, 2, 2, 1, 1>();
        printf("  MFMA waves alone (16 MFMAs / iteration): %.3f ms\n", mf.ms);
        printf("  v_fma 4 per gap : prio none %.3f ms, MFMA waves prio 1 %.3f, filler waves prio 1 %.3f\n", r0.ms, r1.ms, r2.ms);
        printf("  v_fma 8 per gap : prio none %.3f ms, MFMA waves prio 1 %.3f\n", r3.ms, r4.ms);
        printf("  epi-mix 2 per gap: prio none %.3f ms, MFMA waves prio 1 %.3f\n", r5.ms, r6.ms);
        Res f4 = run_fill<2, 0, 4, 2, 0>(), f8 = run_fill<2, 0, 8, 2, 0>(), fm = run_fill<2, 2, 2, 2, 0>();
        printf("  filler waves alone: v_fma 4 per gap %.3f ms, 8 per gap %.3f ms, epi-mix 2 per gap %.3f ms (no overlap = MFMA alone + these)\n",
               f4.ms, f8.ms, fm.ms);
    }
    printf("C. one wave per SIMD, 32 MFMAs (1024 cycles bare) per iteration = one 256 x 256 x 32 stage, self-issued LDS-DMA pieces\n");
#define ROW_C(NDMA, READS, FORM)                                                                                               \
    do {                                                                                                                       \
        Res r = run_dma<NDMA, READS, FORM>();                                                                                  \
        printf("  %d DMA pieces (%s) + %d ds_read_b128 per 32 MFMAs: %7.1f cycles per stage, %.3f ms, %.1f B/clk per CU staged\n", NDMA, \
               FORM == 1 ? "global_load_lds" : FORM == 2 ? "buffer_load lds, 16 half lines" : FORM == 3 ? "buffer_load lds, L1-missing source in L2" : FORM == 5 ? "L2 source + s_barrier per stage" : FORM == 6 ? "L2 source + s_barrier, pieces bunched behind it" : "buffer_load lds", READS * 16, r.cyc / 2000.0, r.ms, NDMA * 4096.0 / (r.cyc / 2000.0)); \
    } while (0)
    ROW_C(0, 0, 0); ROW_C(0, 1, 0); ROW_C(4, 1, 0); ROW_C(8, 0, 0); ROW_C(8, 1, 0); ROW_C(16, 1, 0); ROW_C(8, 1, 1); ROW_C(8, 1, 2); ROW_C(8, 0, 2); ROW_C(8, 0, 3); ROW_C(8, 1, 3); ROW_C(4, 1, 3); ROW_C(16, 1, 3); ROW_C(0, 1, 5); ROW_C(8, 1, 5); ROW_C(8, 0, 5); ROW_C(8, 1, 6);
    }   // !only_d
    printf("D. MFMA shape, operands in registers, one wave per SIMD, 256 CUs, same flops per iteration\n");
    {
        Res a = run_shape<0, 0>(), b = run_shape<1, 0>(), az = run_shape<0, 1>(), bz = run_shape<1, 1>();
        const double fl_ = 2.0 * 32 * 32 * 16 * 16 * 16000.0 * 1024;
        printf("  32x32x16 random: %.3f ms = %.0f TFLOP/s (%.1f cycles per MFMA); zeros: %.3f ms = %.0f TFLOP/s\n", a.ms, fl_ / a.ms * 1e-9,
               a.cyc / (16000.0 * 16), az.ms, fl_ / az.ms * 1e-9);
        printf("  16x16x32 random: %.3f ms = %.0f TFLOP/s (%.1f cycles per MFMA); zeros: %.3f ms = %.0f TFLOP/s\n", b.ms, fl_ / b.ms * 1e-9,
               b.cyc / (16000.0 * 32), bz.ms, fl_ / bz.ms * 1e-9);
    }
    return 0;
}
