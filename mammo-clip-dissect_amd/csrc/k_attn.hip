// k_attn.hip -- K9: fp32 self-attention of the ViT image tower, one launch per encoder block.
//   replaces, inside the encoder forwards that the extraction loop drives (concept_vit/utils.py:117-148):
//     model/modules/image_encoder.py:37  ViTModel(...)  -> softmax(q k^T / sqrt(64)) v  per head
//     concept_vit/clip/model.py:171-183  nn.MultiheadAttention(x, x, x, need_weights=False) without a mask
// 11 % of the GPU time of the headline bench was PyTorch's generic fp32 SDPA kernel (0.68 ms per call at 250 images
// x 12 heads x 197 tokens, 43 TFLOP/s); the matrix pipe can do the 29.8 GFLOP in 0.19 ms.
//
// One workgroup per (image, head), one wave per 32 queries (7 waves at T = 197), flash style over 32-key tiles.
// The K and V tiles ([32, 64] fp32 = 8 KB each) go global -> LDS by global_load_lds_dwordx4 (no staging registers)
// into a ring of four tiles, in pairs: the pair after the one being consumed is in flight.  A dedicated LOADER wave
// (the last one) issues the sixteen 1 KB pieces of a tile and waits for them; everybody meets at one s_barrier per
// pair.  The compute waves never issue a memory instruction inside the loop: the compiler puts an s_waitcnt
// vmcnt(0) in front of LDS reads that follow a DMA in the same wave (it must assume they alias), which would drain
// the ring every tile.  64 KB of LDS and 128 registers: two workgroups per CU.
// A DMA piece lands contiguously, so rows cannot be padded; instead the 16-byte chunk c of tile row r is kept at
// chunk position c ^ (r & 15), which makes both fragment reads (32 keys x one chunk for K, one key x 32 floats for
// V) bank-conflict free.  Keys past T are clamped to the last row (their scores are masked, so p = 0).
// Everything in fp32 on v_mfma_f32_32x32x2_f32:
//   S^T[key, q]  = K_tile . Q^T     A = K rows from LDS, B = Q (32 registers per lane, pre-scaled by log2(e)/8)
//   the MFMA C layout gives a lane ONE query (q = lane & 31) and 16 keys, so the running max / sum of the online
//   softmax are per-lane scalars plus one exchange between the two lane halves; p = 2^(s - m) on v_exp_f32;
//   O^T[d, q]   += V^T . P^T        B = P straight from the S^T accumulator registers: register r of lane half h is
//   key (r&3) + 8(r>>2) + 4h, and a 32x32x2 step may pair ANY two keys as long as the A operand (V from LDS) uses
//   the same two -- no transpose of P through LDS.
// On this part VALU instructions and fp32 MFMAs do not overlap, not even from different waves of a SIMD
// (scripts/micro/mfma_rate.hip: 16 MFMAs + 2N VALU instructions take 1024 + 8N cycles at any occupancy), so every
// non-MFMA instruction in the tile loop costs matrix time: ~150 per tile now (exp2, max, sum, rescale, swizzled
// addresses) against 4096 MFMA cycles.
// qkv is the [B, T, 3, H, 64] output of the fused qkv projection, out is [B, T, H*64] (what the output projection
// reads): no permute / contiguous copies around the call.
#include "mcd_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int AT_D = 64;           // head dimension
constexpr int AT_MAX_T = 256;      // 8 compute waves + the loader
constexpr int AT_HALF = 32 * 256;  // bytes of a K (or V) tile
constexpr int AT_STAGE = 2 * AT_HALF;
constexpr int AT_NSTAGE = 4;       // ring of tiles: two pairs

__global__ __launch_bounds__(576, 4) void vit_attention_kernel(const float* __restrict__ qkv, int T, int H,
                                                                float* __restrict__ out) {
    __shared__ __attribute__((aligned(1024))) char at_lds[AT_NSTAGE * AT_STAGE];   // [stage][K | V][32 rows x 256 B]
    const int ntile = (T + 31) >> 5;
    const int h = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int64_t row_stride = (int64_t)3 * H * AT_D;                 // floats between two tokens of qkv
    const float* base = qkv + b * T * row_stride + (int64_t)h * AT_D;  // q of token 0; k at +H*64, v at +2*H*64
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ql = lane & 31, half = lane >> 5;
    const int q = wave * 32 + ql;

    if (wave == ntile) {
        // ---- loader wave: DMA of tile kt into its ring slot; piece p (0..7) = tile rows 4p..4p+3 of K and of V ----
        // BUFFER-load DMA (SGPR descriptor of this image's qkv block + one 32-bit VGPR offset per lane): beside waves that
        // keep the matrix pipe busy, global_load_lds with a 64-bit address pair per lane issues 3-4x slower
        // (scripts/micro/ldsdma_rate.hip).  One image's block is T * 3 * H * 64 floats (1.8 MB at ViT-B/16): 32-bit offsets.
        auto stage = [&](int kt) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + b * T * row_stride), 0,
                                                                            (int)(T * row_stride * 4), 0x00020000);
            char* dst = at_lds + (kt % AT_NSTAGE) * AT_STAGE;
            const int v_off = H * AT_D * 4;                               // bytes from a token's k to its v
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int r = 4 * p + (lane >> 4);
                const int c = (lane & 15) ^ (r & 15);
                int key = kt * 32 + r;
                if (key >= T) key = T - 1;
                const unsigned off = (unsigned)(((int64_t)key * row_stride + (int64_t)(h + H) * AT_D + c * 4) * 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, off, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + AT_HALF + p * 1024), 16, off,
                                                         v_off, 0, 0);
            }
        };
        // Tiles travel in pairs (one barrier per 64 keys): pair g is in flight while pair g-1 is consumed.
        stage(0);
        if (ntile > 1) stage(1);
        for (int kt = 0; kt < ntile; kt += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pair kt, kt+1 has landed
            __builtin_amdgcn_s_barrier();                      // ... and everyone is done with the pair before it,
            asm volatile("" ::: "memory");                     // whose two slots the next pair refills
            if (kt + 2 < ntile) stage(kt + 2);
            if (kt + 3 < ntile) stage(kt + 3);
        }
        return;
    }

    // Q fragment: step s = 4j + e of the S^T product uses d = 8j + 4*half + e.  Scores are kept in the log2 domain.
    const float qs = 0.125f * 1.44269504088896340736f;
    float Qr[32];
    const float* qrow = base + (int64_t)(q < T ? q : T - 1) * row_stride + 4 * half;   // rows past T: computed, not stored
    float4 qv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[j] = *reinterpret_cast<const float4*>(qrow + 8 * j);   // eight loads in flight
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 t = qv[j];
        Qr[4 * j + 0] = t.x * qs;
        Qr[4 * j + 1] = t.y * qs;
        Qr[4 * j + 2] = t.z * qs;
        Qr[4 * j + 3] = t.w * qs;
    }

    // per-lane parts of the swizzled fragment addresses
    const int kx = ql & 15;                              // K: chunk (2j + half) of row ql sits at (2j + half) ^ kx
    const int k_off = ql * 256;
    const int v_row = 4 * half * 256 + (ql & 3) * 4;     // V: key k0 + 4*half, float 32i + ql -> chunk 8i + (ql >> 2)
    const int va = (ql >> 2) ^ (half << 2);              //    at ((8i) ^ (key & 8)) | (va ^ (k0 & 3))

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    for (int kt = 0; kt < ntile; ++kt) {
        if ((kt & 1) == 0) __builtin_amdgcn_s_barrier();   // the loader says tiles kt and kt+1 are in LDS
        asm volatile("" ::: "memory");
        const char* kb = at_lds + (kt % AT_NSTAGE) * AT_STAGE;
        const char* vb = kb + AT_HALF;
        // the swizzled offsets are recomputed every tile (a xor and an add each): hoisted out of the loop they cost
        // a dozen registers and the kernel spills at the 128 it may use
        int kx_t = kx, va_t = va;
        asm volatile("" : "+v"(kx_t), "+v"(va_t));
        int v_off[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v_off[e] = v_row + ((va_t ^ e) << 4);

        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 ka = *reinterpret_cast<const float4*>(kb + k_off + (((2 * j + half) ^ kx_t) << 4));
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.x, Qr[4 * j + 0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.y, Qr[4 * j + 1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.z, Qr[4 * j + 2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.w, Qr[4 * j + 3], c, 0, 0, 0);
        }
        const int nk = T - kt * 32;   // valid keys in this tile (>= 1)
        if (nk < 32) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) + 4 * half >= nk) c[r] = -INFINITY;
        }
        float mt = c[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, c[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m, mt);                       // finite: key kt*32 is valid (lane half 0 holds it)
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);  // first tile: 2^-inf = 0
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            c[r] = __builtin_amdgcn_exp2f(c[r] - m_new);
            ls += c[r];
        }
        l = l * alpha + ls;                                     // per lane half; the halves are added at the end
        m = m_new;
        // (rescaling only when some lane's maximum moves by more than 2^8 -- "lazy rescaling" -- measured no gain)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        // step r pairs the keys k0 and k0 + 4 (k0 = (r&3) + 8(r>>2)); row k0 + 4*half, and (key & 8) = (k0 & 8):
        // chunk 8i + .. of that row sits at ((8i) ^ (k0 & 8)) | ..   A clamped key past T has p = 0.
        auto pv_step = [&](int r) {
            const int k0 = (r & 3) + 8 * (r >> 2);
            const char* vr = vb + k0 * 256 + v_off[r & 3];
            const float v0 = *reinterpret_cast<const float*>(vr + ((0 ^ (k0 & 8)) << 4));
            const float v1 = *reinterpret_cast<const float*>(vr + ((8 ^ (k0 & 8)) << 4));
            o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, c[r], o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, c[r], o[1], 0, 0, 0);
        };
        // No branch around the steps, not even for a short last tile (its masked keys have p = 0): a branch per step
        // makes the compiler wait for each MFMA result and copy the accumulators, a branch per tile still costs a
        // copy of all 32 accumulator registers where the two paths meet.
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((r & 3) == 0 && r) asm volatile("" ::: "memory");   // keep the V reads of later groups from piling up
            pv_step(r);
        }
    }

    l = l + __shfl_xor(l, 32);
    if (q < T) {
        const float inv = 1.0f / l;
        float* dst = out + (b * T + q) * (int64_t)H * AT_D + (int64_t)h * AT_D + 4 * half;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // registers 4g..4g+3 = d 32i + 8g + 4*half + 0..3
                const float4 t = make_float4(o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv,
                                             o[i][4 * g + 3] * inv);
                *reinterpret_cast<float4*>(dst + 32 * i + 8 * g) = t;
            }
    }
}

}  // namespace

extern "C" int mcd_vit_attention(const float* qkv, int64_t B, int64_t T, int64_t H, float* out, mcd_stream_t stream) {
    MCD_REQUIRE(qkv && out, MCD_E_ARG, "mcd_vit_attention: NULL pointer");
    MCD_REQUIRE(B >= 0 && T >= 1 && H >= 1, MCD_E_ARG, "mcd_vit_attention: bad shape B=%lld T=%lld H=%lld", (long long)B,
                (long long)T, (long long)H);
    MCD_REQUIRE(T <= AT_MAX_T, MCD_E_UNSUPPORTED, "mcd_vit_attention: T=%lld tokens, at most %d (one wave per 32 queries, 8 waves)",
                (long long)T, AT_MAX_T);
    MCD_REQUIRE(B <= 65535 && H <= 65535, MCD_E_UNSUPPORTED, "mcd_vit_attention: B and H must be <= 65535");
    MCD_REQUIRE(((uintptr_t)qkv) % 16 == 0 && ((uintptr_t)out) % 16 == 0, MCD_E_ARG,
                "mcd_vit_attention: qkv and out must be 16-byte aligned");
    if (B == 0) return MCD_OK;
    const int nwaves = (int)((T + 31) / 32) + 1;   // one per 32 queries + the loader
    hipLaunchKernelGGL(vit_attention_kernel, dim3((unsigned)H, (unsigned)B), dim3(64 * nwaves), 0, (hipStream_t)stream, qkv,
                       (int)T, (int)H, out);
    MCD_LAUNCH_CHECK("vit_attention_kernel");
    return MCD_OK;
}
