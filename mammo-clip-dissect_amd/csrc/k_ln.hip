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

// ---- K11: patch extraction for the ViT patch embedding ---------------------------------------------------------
//   replaces  the Conv2d(3, dim, P, stride P) of the patch embedding (ViTPatchEmbeddings / conv1, the same reference
//             sites as K9) by its GEMM form: rows of P*P*Cin pixels per patch, times the [dim, Cin*P*P] weight.
// out is [B, 1 + nP, Cin*P*P]: row 0 of every image is zero (the class-token slot: the GEMM that follows adds the
// residual operand there), row 1 + (py*nW + px) is patch (py, px) in (c, dy, dx) order -- the order of the conv
// weight viewed as [dim, Cin*P*P].  One thread per float4 of the output: coalesced writes, 16-byte reads (P % 4 == 0).
namespace {

__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, int B, int Cin, int H, int W, int P,
                                                        float* __restrict__ out) {
    const int nW = W / P, nP = (H / P) * nW, F = Cin * P * P, F4 = F >> 2, P4 = P >> 2;
    const int64_t total = (int64_t)B * (1 + nP) * F4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int f4 = (int)(i % F4);
        const int64_t row = i / F4;
        const int r = (int)(row % (1 + nP));
        const int64_t b = row / (1 + nP);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r > 0) {
            const int p = r - 1, py = p / nW, px = p % nW;
            const int dx4 = f4 % P4, dy = (f4 / P4) % P, c = f4 / (P4 * P);
            v = *reinterpret_cast<const float4*>(x + ((b * Cin + c) * H + (py * P + dy)) * (int64_t)W + px * P + dx4 * 4);
        }
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

}  // namespace

extern "C" int mcd_patchify(const float* x, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t P, float* out,
                            mcd_stream_t stream) {
    MCD_REQUIRE(x && out, MCD_E_ARG, "mcd_patchify: NULL pointer");
    MCD_REQUIRE(B >= 0 && Cin > 0 && P >= 4 && P % 4 == 0 && H > 0 && W > 0 && H % P == 0 && W % P == 0 && W % 4 == 0,
                MCD_E_UNSUPPORTED, "mcd_patchify: bad shape B=%lld Cin=%lld H=%lld W=%lld P=%lld", (long long)B,
                (long long)Cin, (long long)H, (long long)W, (long long)P);
    MCD_REQUIRE(B < (1 << 30) && Cin * H * W < (1LL << 31), MCD_E_UNSUPPORTED, "mcd_patchify: image too large");
    MCD_REQUIRE(((uintptr_t)x) % 16 == 0 && ((uintptr_t)out) % 16 == 0, MCD_E_ARG, "mcd_patchify: pointers must be 16-byte aligned");
    if (B == 0) return MCD_OK;
    const int64_t total = B * (1 + (H / P) * (W / P)) * (Cin * P * P / 4);
    const unsigned grid = (unsigned)(mcd_cdiv(total, 256) < (1 << 20) ? mcd_cdiv(total, 256) : (1 << 20));
    hipLaunchKernelGGL(patchify_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (int)B, (int)Cin, (int)H, (int)W,
                       (int)P, out);
    MCD_LAUNCH_CHECK("patchify_kernel");
    return MCD_OK;
}
