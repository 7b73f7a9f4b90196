// k_ln.hip -- K10: fp32 LayerNorm over the last dimension for the ViT towers (encoder-side, HBM-bound).
//   replaces  nn.LayerNorm inside the encoder blocks (ViTLayer.layernorm_before/after, model/modules/image_encoder.py:37;
//             ln_1 / ln_2, concept_vit/clip/model.py:172-176) in the forwards that concept_vit/utils.py:117-148 drives.
// ATen's kernel runs at 4.0 TB/s on the [49 250, 768] activations of the headline bench (75 us, 960 calls per step).
// Here one wave owns a row: the row lives in registers (NV float4 per lane), mean and the centred sum of squares are
// two shuffle reductions over it (two-pass in registers: no E[x^2] - E[x]^2 cancellation), one read and one write of
// HBM per element.  y = (x - mean) / sqrt(var + eps) * gamma + beta, biased variance, like torch.
#include "mcd_common.h"

namespace {

template <int NV>
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* __restrict__ x, int64_t rows, int D,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + row * D);
    const int nq = D >> 2;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + 64 * i;
        v[i] = q < nq ? xr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = mcd_wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < nq) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            ss += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(mcd_wave_sum(ss) / (float)D + eps);
    float4* yr = reinterpret_cast<float4*>(y + row * D);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + 64 * i;
        if (q < nq) {
            const float4 g = g4[q], b = b4[q];
            yr[q] = make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                                (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
        }
    }
}

}  // namespace

extern "C" int mcd_layer_norm(const float* x, int64_t rows, int64_t D, const float* gamma, const float* beta, float eps,
                              float* y, mcd_stream_t stream) {
    MCD_REQUIRE(x && y && gamma && beta, MCD_E_ARG, "mcd_layer_norm: NULL pointer");
    MCD_REQUIRE(rows >= 0 && D >= 4 && D % 4 == 0 && D <= 2048, MCD_E_UNSUPPORTED,
                "mcd_layer_norm: D=%lld must be a multiple of 4 in [4, 2048]", (long long)D);
    MCD_REQUIRE(((uintptr_t)x) % 16 == 0 && ((uintptr_t)y) % 16 == 0 && ((uintptr_t)gamma) % 16 == 0 && ((uintptr_t)beta) % 16 == 0,
                MCD_E_ARG, "mcd_layer_norm: pointers must be 16-byte aligned");
    if (rows == 0) return MCD_OK;
    const unsigned grid = (unsigned)mcd_cdiv(rows, 4);
    const int nv = (int)mcd_cdiv(D / 4, 64);
#define MCD_LN(NV) \
    hipLaunchKernelGGL(layer_norm_kernel<NV>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, rows, (int)D, gamma, beta, eps, y)
    if (nv <= 1) MCD_LN(1);
    else if (nv == 2) MCD_LN(2);
    else if (nv == 3) MCD_LN(3);
    else if (nv == 4) MCD_LN(4);
    else if (nv <= 6) MCD_LN(6);
    else MCD_LN(8);
#undef MCD_LN
    MCD_LAUNCH_CHECK("layer_norm_kernel");
    return MCD_OK;
}
