// Translation unit for tests/test_k1s_isa_cpu.py: the round-4 K1s kernel (csrc/k_gexp_v4.inc) alone, compiled to assembly for the
// register / scratch / accumulator-file audit (the kernel names a0..a255 in asm statements; the compiler must stay out of them).
#include "../mammo-clip-dissect_amd/csrc/mcd_common.h"
#include <type_traits>
namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct TileWalkR {
    int xcd, slot, nslot, nrow_x, n_seq;
    __device__ __forceinline__ TileWalkR(int tiles_m, int tiles_n) {
        xcd = blockIdx.x & 7; slot = blockIdx.x >> 3; nslot = gridDim.x >> 3;
        nrow_x = (tiles_m - xcd + 7) / 8; n_seq = tiles_n * nrow_x;
    }
    __device__ __forceinline__ bool next(int& i, int& tm, int& tn) const {
        ++i; const int seq = slot + i * nslot; if (seq >= n_seq) return false;
        tn = seq / nrow_x; tm = xcd + 8 * (seq - tn * nrow_x); return true;
    }
    __device__ __forceinline__ int count() const { return slot < n_seq ? (n_seq - slot + nslot - 1) / nslot : 0; }
};
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const bf16x2 h = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
    unsigned u; __builtin_memcpy(&u, &h, 4); return u;
}
#include "../mammo-clip-dissect_amd/csrc/k_gexp_v4.inc"

#ifndef AB
#define AB 0
#endif
#ifndef SY
#define SY 0
#endif
template __global__ void gemm_nt_bf16_exp_v4_kernel<AB, SY>(const unsigned short*, const unsigned short*, int64_t, int64_t, int64_t, unsigned short*, int64_t, float*, int64_t, float, int, int, int);
}
