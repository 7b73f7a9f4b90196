// K4s' gather pattern without its arithmetic: how fast can a CU pull random 256-byte row pieces out of L2 / the Infinity
// Cache, (R) into registers with global_load_dwordx4 as K4s does, (D) through a per-wave LDS ring with `buffer_load ... lds`
// + ds_read_b128 (all of it inline asm / builtins with hand-counted s_waitcnt, because hipcc puts vmcnt(0) in front of every
// LDS read that may alias an LDS-DMA).  A wave instruction = 4 rows x 256 B (16 lanes x 16 B each), rows random in a table of
// 25 000 rows x 20 224 B, 8 slices of 256 B live at a time (slice = blockIdx % 8 = the XCD), as in wpmi_bf16_kernel.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/gather_path.hip -o scripts/micro/gather_path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 25000, PITCH = 20224;

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// row offset of (job, quad t, row rr) for this lane's 16-lane group
__device__ __forceinline__ uint32_t row_off(uint32_t job, int t, int rr, int grp) {
    return (hash32(job * 4099u + (uint32_t)(t * 4 + rr) * 16u + (uint32_t)grp) % ROWS) * PITCH;
}

// (R) registers: batches of 16 rows in flight per wave
__global__ __launch_bounds__(256) void gather_regs(const char* __restrict__ E, int quads, unsigned* out, int slice_round) {
    extern __shared__ char pad[];
    const int lane = threadIdx.x & 63, grp = lane >> 4;
    const uint32_t job = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t col = (uint32_t)(((blockIdx.x & 7) + 8 * slice_round) * 256 + (lane & 15) * 16);
    u32x4 acc = {0, 0, 0, 0};
    for (int t = 0; t < quads; t += 4) {
        u32x4 g[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = *reinterpret_cast<const u32x4*>(E + row_off(job, t + (r >> 2), r & 3, grp) + col);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc ^= g[r];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
    if (threadIdx.x == 9999) pad[0] = 1;
}

// (D) LDS ring of NQ quads per wave
template <int NQ>
__global__ __launch_bounds__(256) void gather_ring(const char* __restrict__ E, int quads, unsigned* out, int slice_round) {
    extern __shared__ __attribute__((aligned(1024))) char ring_all[];
    const int lane = threadIdx.x & 63, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* ring = ring_all + wave * (NQ * 4096);
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring + (uint32_t)lane * 16u;
    const uint32_t job = blockIdx.x * 4 + wave;
    const uint32_t col = (uint32_t)(((blockIdx.x & 7) + 8 * slice_round) * 256 + (lane & 15) * 16);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)E, 0, ROWS * PITCH, 0x00020000);
    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + slot * 4096 + rr * 1024), 16,
                                                     row_off(job, t, rr, grp) + col, 0, 0, 0);
    };
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NQ; ++s) issue(s, s);
    // quads is a multiple of NQ; the last NQ quads re-issue quad 0's rows (dummy refills keep the counts constant)
    for (int t0 = 0; t0 < quads; t0 += NQ) {
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
            const int t = t0 + s;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NQ - 1)) : "memory");
            u32x4 g0, g1, g2, g3;
            const uint32_t a = ring_lds + (uint32_t)s * 4096u;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3)
                         : "v"(a)
                         : "memory");
            issue(t + NQ < quads ? t + NQ : 0, s);
            acc ^= g0 ^ g1 ^ g2 ^ g3;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

template <typename L>
static float timed(L launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    char* E;
    unsigned* out;
    hipMalloc(&E, (size_t)ROWS * PITCH);
    hipMemset(E, 0x3c, (size_t)ROWS * PITCH);
    hipMalloc(&out, 64);
    const int quads = 24, groups = 576;        // K4s: 25 quads per (4 neurons, slice) job, 576 groups of 16 neurons per slice
    const int rounds = 10;                     // 80 slices
    const double bytes = (double)rounds * 8 * groups * 4 * quads * 4 * 1024.0;
    printf("gather pattern of K4s, no arithmetic: %d rounds x 8 slices x %d workgroups x 4 waves x %d quads x 4 KB = %.2f GB per pass\n", rounds, groups,
           quads, bytes / 1e9);
    auto report = [&](const char* name, float ms) {
        printf("  %-58s %7.3f ms  %6.2f TB/s  %5.1f B/clk per CU at 2.1 GHz\n", name, ms, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.1e9 / 1e3);
    };
    for (int lds_kb : {40, 32, 20}) {
        hipFuncSetAttribute((const void*)gather_regs, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
        float ms = timed([&]() { for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(gather_regs, dim3(8 * groups), dim3(256), lds_kb * 1024, 0, E, quads, out, r); });
        char nm[96];
        snprintf(nm, sizeof nm, "registers, 16 rows in flight per wave, %d workgroups per CU", 160 / lds_kb);
        report(nm, ms);
    }
#define RING(NQ)                                                                                                                 \
    do {                                                                                                                         \
        hipFuncSetAttribute((const void*)gather_ring<NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * NQ * 4096);             \
        float ms = timed([&]() { for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(gather_ring<NQ>, dim3(8 * groups), dim3(256), 4 * NQ * 4096, 0, E, quads, out, r); }); \
        char nm[96];                                                                                                             \
        snprintf(nm, sizeof nm, "LDS ring of %d quads per wave (%d KB), %d workgroups per CU", NQ, 4 * NQ, 160 / (16 * NQ));          \
        report(nm, ms);                                                                                                          \
    } while (0)
    RING(1);
    RING(2);
    RING(3);
    RING(4);
    return 0;
}
