// Issue cost (cycles per wave-instruction, one wave per SIMD, independent operands) of the vector instructions of K1s' epilogue.
//   hipcc --offload-arch=gfx950 -O3 -o valu_cost scripts/micro/valu_cost.hip && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f, t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %0\n v_accvgpr_write_b32 a2, %0\n v_accvgpr_write_b32 a3, %0\n"
                 "v_accvgpr_write_b32 a4, %0\n v_accvgpr_write_b32 a5, %0\n v_accvgpr_write_b32 a6, %0\n v_accvgpr_write_b32 a7, %0" ::"v"(a0) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { REP16(asm volatile("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n v_accvgpr_read_b32 %4, a4\n v_accvgpr_read_b32 %5, a5\n v_accvgpr_read_b32 %6, a6\n v_accvgpr_read_b32 %7, a7" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3), "=v"(t4), "=v"(t5), "=v"(t6), "=v"(t7));) }
        if (KIND == 1) { REP16(asm volatile("v_exp_f32 %0, %8\n v_exp_f32 %1, %9\n v_exp_f32 %2, %10\n v_exp_f32 %3, %11\n v_exp_f32 %4, %12\n v_exp_f32 %5, %13\n v_exp_f32 %6, %14\n v_exp_f32 %7, %15" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3), "=v"(t4), "=v"(t5), "=v"(t6), "=v"(t7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));) }
        if (KIND == 2) { REP16(asm volatile("v_add_f32 %0, %8, %9\n v_add_f32 %1, %9, %10\n v_add_f32 %2, %10, %11\n v_add_f32 %3, %11, %12\n v_add_f32 %4, %12, %13\n v_add_f32 %5, %13, %14\n v_add_f32 %6, %14, %15\n v_add_f32 %7, %15, %8" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3), "=v"(t4), "=v"(t5), "=v"(t6), "=v"(t7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));) }
        if (KIND == 3) { REP16(asm volatile("v_pk_add_f32 %0, %2, %3\n v_pk_add_f32 %1, %3, %4\n v_pk_add_f32 %0, %4, %5\n v_pk_add_f32 %1, %5, %2\n v_pk_add_f32 %0, %2, %4\n v_pk_add_f32 %1, %3, %5\n v_pk_add_f32 %0, %2, %3\n v_pk_add_f32 %1, %4, %5" : "=&v"(*(double*)&t0), "=&v"(*(double*)&t2) : "v"(*(double*)&a0), "v"(*(double*)&a2), "v"(*(double*)&a4), "v"(*(double*)&a6));) }
        if (KIND == 4) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %8, %9\n v_cvt_pk_bf16_f32 %1, %9, %10\n v_cvt_pk_bf16_f32 %2, %10, %11\n v_cvt_pk_bf16_f32 %3, %11, %12\n v_cvt_pk_bf16_f32 %4, %12, %13\n v_cvt_pk_bf16_f32 %5, %13, %14\n v_cvt_pk_bf16_f32 %6, %14, %15\n v_cvt_pk_bf16_f32 %7, %15, %8" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3), "=v"(t4), "=v"(t5), "=v"(t6), "=v"(t7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));) }
        if (KIND == 5) {   // the epilogue's group as written: 8 reads, 8 exp, 4 pk adds, 4 cvt
            REP16(asm volatile("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n v_accvgpr_read_b32 %4, a4\n v_accvgpr_read_b32 %5, a5\n v_accvgpr_read_b32 %6, a6\n v_accvgpr_read_b32 %7, a7\n"
                               "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                               "v_add_f32 %8, %0, %1\n v_add_f32 %9, %2, %3\n v_add_f32 %10, %4, %5\n v_add_f32 %11, %6, %7\n"
                               "v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %6, %6, %7"
                               : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4));) }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7 + a1 + a2 + a3 + a4;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}
template <int KIND> void run(const char* name, int per_iter, float* out, unsigned long long* cyc) {
    const int iters = 200;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), cyc, 8192, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s %6.2f cycles per instruction\n", name, (double)h[512] / ((double)iters * per_iter));
}
int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8192);
    run<0>("v_accvgpr_read_b32", 128, out, cyc);
    run<1>("v_exp_f32", 128, out, cyc);
    run<2>("v_add_f32", 128, out, cyc);
    run<3>("v_pk_add_f32", 128, out, cyc);
    run<4>("v_cvt_pk_bf16_f32", 128, out, cyc);
    run<5>("group: 8 read + 8 exp + 4 add + 4 cvt (per instr)", 16 * 24, out, cyc);
    return 0;
}
