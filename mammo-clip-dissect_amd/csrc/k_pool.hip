// k_pool.hip -- K0: forward-hook pooling written straight into the activation matrix (HBM-bound).
//   replaces get_activation() hook bodies, concept_vit/utils.py:27-52 (og_utils.py:31-56,
//   CLIP_og_utils.py:13-36): 4-D -> mean/amax over H,W; 3-D -> token 0; 2-D -> as is,
//   and the list-append + torch.cat that follows (utils.py:143).
// One wave per (image, channel) plane: lanes stride over the H*W plane (coalesced), butterfly
// reduction, one store into dst[(row0+b)*stride_n + (col0+ch)*stride_u].
#include "mcd_common.h"

namespace {

template <bool VEC4>
__global__ __launch_bounds__(256) void pool_plane_kernel(const float* __restrict__ x, int64_t planes, int64_t Cout,
                                                          int64_t HW, int mode, float* __restrict__ dst, int64_t row0,
                                                          int64_t col0, int64_t stride_n, int64_t stride_u) {
    const int lane = threadIdx.x & 63;
    const int64_t plane = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const float* src = x + plane * HW;
    float s = 0.f, m = -INFINITY;
    bool has_nan = false;   // output.amax(dim=[2,3]) propagates NaN (utils.py:44-52); fmaxf alone would drop it
    if constexpr (VEC4) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        const int64_t n4 = HW >> 2;
        for (int64_t i = lane; i < n4; i += 64) {
            const float4 v = s4[i];
            s += (v.x + v.y) + (v.z + v.w);
            m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
            has_nan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
        }
    } else {
        for (int64_t i = lane; i < HW; i += 64) {
            const float v = src[i];
            s += v;
            m = fmaxf(m, v);
            has_nan |= (v != v);
        }
    }
    s = mcd_wave_sum(s);
    m = mcd_wave_max(m);
    if (__ballot(has_nan)) m = __uint_as_float(0x7fc00000u);
    if (lane == 0) {
        const int64_t b = plane / Cout, ch = plane - b * Cout;
        dst[(row0 + b) * stride_n + (col0 + ch) * stride_u] = (mode == MCD_POOL_AVG) ? s / (float)HW : m;
    }
}

// CLS / NONE: element (b, f) = x[b*T*F + f]
__global__ __launch_bounds__(256) void pool_copy_kernel(const float* __restrict__ x, int64_t B, int64_t F,
                                                         int64_t bstride, float* __restrict__ dst, int64_t row0,
                                                         int64_t col0, int64_t stride_n, int64_t stride_u) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * F) return;
    const int64_t b = i / F, f = i - b * F;
    dst[(row0 + b) * stride_n + (col0 + f) * stride_u] = x[b * bstride + f];
}

}  // namespace

extern "C" int mcd_hook_pool(const float* x, int64_t B, int64_t Cout, int64_t HW, int mode, float* dst, int64_t row0,
                             int64_t col0, int64_t stride_n, int64_t stride_u, mcd_stream_t stream) {
    MCD_REQUIRE(x && dst, MCD_E_ARG, "mcd_hook_pool: NULL pointer");
    MCD_REQUIRE(B >= 0 && Cout > 0 && HW > 0 && row0 >= 0 && col0 >= 0, MCD_E_ARG, "mcd_hook_pool: bad shape");
    MCD_REQUIRE(mode >= MCD_POOL_AVG && mode <= MCD_POOL_NONE, MCD_E_ARG, "mcd_hook_pool: bad mode %d", mode);
    if (B == 0) return MCD_OK;
    hipStream_t st = (hipStream_t)stream;
    if (mode == MCD_POOL_AVG || mode == MCD_POOL_MAX) {
        const int64_t planes = B * Cout;
        const unsigned grid = (unsigned)mcd_cdiv(planes, 4);
        const bool vec4 = (HW % 4 == 0) && (((uintptr_t)x) % 16 == 0);
        if (vec4)
            hipLaunchKernelGGL(pool_plane_kernel<true>, dim3(grid), dim3(256), 0, st, x, planes, Cout, HW, mode, dst,
                               row0, col0, stride_n, stride_u);
        else
            hipLaunchKernelGGL(pool_plane_kernel<false>, dim3(grid), dim3(256), 0, st, x, planes, Cout, HW, mode, dst,
                               row0, col0, stride_n, stride_u);
        MCD_LAUNCH_CHECK("pool_plane_kernel");
    } else {
        const int64_t bstride = (mode == MCD_POOL_CLS) ? HW * Cout : Cout;  // [B,T,F] token 0, or [B,F]
        const unsigned grid = (unsigned)mcd_cdiv(B * Cout, 256);
        hipLaunchKernelGGL(pool_copy_kernel, dim3(grid), dim3(256), 0, st, x, B, Cout, bstride, dst, row0, col0,
                           stride_n, stride_u);
        MCD_LAUNCH_CHECK("pool_copy_kernel");
    }
    return MCD_OK;
}
