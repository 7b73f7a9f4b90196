// k_rank.hip -- K8  mcd_rank_reorder   concept_vit/similarity.py:99-132 (rank_reorder)
//
// Per neuron u the reference takes the top_n most activating images (activations t_0 >= t_1 >= ...), gathers
// their rows of clip_feats, G[j,c] = P[inds[j,u], c], and for every concept c
//     rank_jc  = ascending rank of G[j,c] among G[:,c]             (argsort of argsort, :112-113)
//     err[u,c] = mean_j |t_j - st[rank_jc]|^p / baseline_u / mean_j(G[j,c])^scale_p,   st = t ascending
//     baseline_u = mean over R random permutations pi of |st_j - st[pi_j]|^p            (:119-120)
// and returns -err.  Summed over the SORTED positions r instead of over j (the same terms, another order):
//     sum_j |t_j - st[rank_jc]|^p = sum_r |t[j_r] - t[top_n-1-r]|^p,  j_r = image with the r-th smallest G[.,c]
// so one ascending sort of (G[j,c], j) per (neuron, concept) yields everything.
//
// Mapping: workgroup = (neuron, tile of 4 concepts).  The 4 gathered values of an image are ONE 16-byte load
// (coalescing across images is impossible -- they are random rows -- so the 4-wide load is what limits the
// over-fetch to the 64-byte sector); keys (order-preserving u32 of the value, image slot j) are bitonic-sorted
// in LDS, 4 lists side by side between barriers; NaN keys sort last, like torch.argsort.  The column means follow
// ATen's summation order (they can cancel); the error sums (positive terms) are block reductions.
#include "mcd_common.h"
#include <math.h>

namespace {

constexpr int RK_THREADS = 256;
constexpr int RK_CT = 4;  // concepts per workgroup pass
static_assert(RK_THREADS / 64 == RK_CT, "lane 0 of wave k sums column k");

__device__ __forceinline__ float pow_aten(float x, float p) {
    // ATen pow_tensor_scalar: 2 -> x*x, 3 -> (x*x)*x, 0.5 -> sqrt, 1 -> x; otherwise powf
    if (p == 3.0f) return (x * x) * x;
    if (p == 2.0f) return x * x;
    if (p == 1.0f) return x;
    if (p == 0.5f) return sqrtf(x);
    return powf(x, p);
}

// sum of v over the workgroup, returned to every thread; s_red: RK_THREADS/64 floats per call site, barrier inside
__device__ __forceinline__ float block_sum(float v, float* s_red) {
    v = mcd_wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < RK_THREADS / 64; ++w) t += s_red[w];
    __syncthreads();
    return t;
}

// ATen multi_row_sum state (level_power 4: 16-row chunks), as in k_wpmi.hip
struct Cascade {
    float a0, a1, a2, a3;
    __device__ __forceinline__ void init() { a0 = a1 = a2 = a3 = 0.f; }
    __device__ __forceinline__ void flush(int i) {  // after a complete 16-addend chunk; i = addends consumed so far
        a1 += a0;
        a0 = 0.f;
        if ((i & 0xF0) != 0) return;
        a2 += a1;
        a1 = 0.f;
        if ((i & 0xF00) != 0) return;
        a3 += a2;
        a2 = 0.f;
    }
    __device__ __forceinline__ float total() const { return ((a0 + a1) + a2) + a3; }
};

// torch.sum(dim=0) of the n gathered values of one column, in ATen's order (SumKernel.cpp cascade_sum): columns
// below `split` cascade over the rows, the others add 4 row-interleaved partials.  The column mean is a sum of
// cosine similarities of both signs that can cancel to ~1e-4 of its terms, so only the reference's own order
// reproduces its value there.  One thread per column; the values come back out of the (still unsorted) keys.
__device__ __forceinline__ float column_sum_aten(const unsigned long long* keys, int n, bool row_sum) {
    auto val = [&](int j) { return mcd_key2f((uint32_t)(keys[j] >> 32)); };
    if (!row_sum) {
        Cascade c;
        c.init();
        int i = 0;
        for (; i + 16 <= n; i += 16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c.a0 += val(i + r);
            c.flush(i + 16);
        }
        for (; i < n; ++i) c.a0 += val(i);
        return c.total();
    }
    const int q = n >> 2;
    Cascade part[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) part[k].init();
    int m = 0;
    for (; m + 16 <= q; m += 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) part[k].a0 += val(4 * (m + r) + k);
#pragma unroll
        for (int k = 0; k < 4; ++k) part[k].flush(m + 16);
    }
    for (; m < q; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) part[k].a0 += val(4 * m + k);
    float tot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tot[k] = part[k].total();
    for (int j = 4 * q; j < n; ++j) tot[0] += val(j);
    tot[0] += tot[1];
    tot[0] += tot[2];
    tot[0] += tot[3];
    return tot[0];
}

// baseline_u: one workgroup per neuron
__global__ __launch_bounds__(RK_THREADS) void rank_baseline_kernel(const float* __restrict__ tvals, int64_t ldt,
                                                                    int top_n, const int32_t* __restrict__ perms,
                                                                    int R, float p, float* __restrict__ baseline) {
    __shared__ float s_red[RK_THREADS / 64];
    const int64_t u = blockIdx.x;
    const float* t = tvals + u * ldt;
    const int32_t* pm = perms + u * (int64_t)R * top_n;
    float acc = 0.f;
    for (int k = 0; k < R; ++k)
        for (int j = threadIdx.x; j < top_n; j += RK_THREADS) {
            const float a = t[top_n - 1 - j];                    // st[j]
            const float b = t[top_n - 1 - pm[(int64_t)k * top_n + j]];  // st[perm[j]]
            acc += pow_aten(fabsf(a - b), p);
        }
    const float tot = block_sum(acc, s_red);
    if (threadIdx.x == 0) baseline[u] = tot / (float)((int64_t)R * top_n);
}

template <bool VEC4>
__global__ __launch_bounds__(RK_THREADS) void rank_reorder_kernel(const float* __restrict__ P, int64_t ldP, int64_t C,
                                                                   const float* __restrict__ tvals,
                                                                   const int32_t* __restrict__ tidx, int64_t ldt,
                                                                   int top_n, int npad, int split,
                                                                   const float* __restrict__ baseline, float p,
                                                                   float scale_p, float* __restrict__ out,
                                                                   int64_t ldo) {
    extern __shared__ unsigned long long s_keys[];  // [RK_CT][npad]
    __shared__ float s_red[RK_THREADS / 64];
    __shared__ float s_csum[RK_CT];
    const int64_t u = blockIdx.x;
    const int c0 = blockIdx.y * RK_CT;
    const float* t = tvals + u * ldt;
    const int32_t* ti = tidx + u * ldt;

    // 1. gather, keys
    for (int j = threadIdx.x; j < npad; j += RK_THREADS) {
        float g[RK_CT] = {0.f, 0.f, 0.f, 0.f};
        if (j < top_n) {
            const float* row = P + (int64_t)ti[j] * ldP + c0;
            if constexpr (VEC4) {
                const float4 v = *reinterpret_cast<const float4*>(row);
                g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w;
            } else {
#pragma unroll
                for (int k = 0; k < RK_CT; ++k)
                    if (c0 + k < C) g[k] = row[k];
            }
        }
#pragma unroll
        for (int k = 0; k < RK_CT; ++k) {
            s_keys[k * npad + j] = (j < top_n) ? (((unsigned long long)mcd_f2key(g[k]) << 32) | (uint32_t)j)
                                               : 0xffffffffffffffffull;  // padding sorts last
        }
    }
    __syncthreads();
    // column sums in ATen's order: lane 0 of wave k takes column k (4 SIMDs side by side)
    if ((threadIdx.x & 63) == 0) {
        const int k = threadIdx.x >> 6;
        s_csum[k] = column_sum_aten(s_keys + k * npad, top_n, c0 + k >= split);
    }
    __syncthreads();

    // 2. bitonic sort, ascending, the RK_CT lists side by side
    const int half = npad >> 1;
    for (int k2 = 2; k2 <= npad; k2 <<= 1) {
        for (int s = k2 >> 1; s > 0; s >>= 1) {
            for (int w = threadIdx.x; w < half * RK_CT; w += RK_THREADS) {
                const int list = w / half, e = w - list * half;
                const int i = ((e / s) * (s << 1)) + (e % s);   // lower element of the pair
                const int l = i + s;
                const bool up = ((i & k2) == 0);
                unsigned long long* a = s_keys + list * npad;
                const unsigned long long x = a[i], y = a[l];
                if ((x > y) == up) {
                    a[i] = y;
                    a[l] = x;
                }
            }
            __syncthreads();
        }
    }

    // 3. |t[j_r] - t[top_n-1-r]|^p over the sorted positions
    float esum[RK_CT] = {0.f, 0.f, 0.f, 0.f};
    for (int r = threadIdx.x; r < top_n; r += RK_THREADS) {
        const float st = t[top_n - 1 - r];
#pragma unroll
        for (int k = 0; k < RK_CT; ++k) {
            const int j = (int)(uint32_t)s_keys[k * npad + r];
            esum[k] += pow_aten(fabsf(t[j] - st), p);
        }
    }
    const float base = baseline[u];
#pragma unroll
    for (int k = 0; k < RK_CT; ++k) {
        const float es = block_sum(esum[k], s_red);
        const float cs = s_csum[k];
        if (threadIdx.x == 0 && c0 + k < C) {
            const float err = (es / (float)top_n) / base;          // similarity.py:129
            const float avg = cs / (float)top_n;                   // :110
            out[u * ldo + c0 + k] = -(err / pow_aten(avg, scale_p));  // :130, :133 (avg < 0 -> NaN, as the reference)
        }
    }
}

}  // namespace

extern "C" int mcd_rank_reorder(const float* P, int64_t ldP, int64_t N, int64_t C, const float* tvals,
                                const int32_t* tidx, int64_t ldt, int64_t U, int top_n, const int32_t* perms,
                                int n_perm, float p, float scale_p, float* baseline_ws, float* out, int64_t ldo,
                                mcd_stream_t stream) {
    MCD_REQUIRE(P && tvals && tidx && perms && baseline_ws && out, MCD_E_ARG, "mcd_rank_reorder: NULL pointer");
    MCD_REQUIRE(N > 0 && C > 0 && U >= 0 && ldP >= C && ldo >= C, MCD_E_ARG, "mcd_rank_reorder: bad shape");
    MCD_REQUIRE(top_n >= 1 && top_n <= N && ldt >= top_n, MCD_E_RANGE,
                "selected index k out of range (top_n=%d, N=%lld)", top_n, (long long)N);
    MCD_REQUIRE(top_n <= 4096, MCD_E_UNSUPPORTED, "mcd_rank_reorder: top_n=%d > 4096 not supported", top_n);
    MCD_REQUIRE(n_perm >= 1, MCD_E_ARG, "mcd_rank_reorder: n_perm < 1");
    MCD_REQUIRE(U <= 0x7fffffffLL && mcd_cdiv(C, RK_CT) <= 65535, MCD_E_UNSUPPORTED, "mcd_rank_reorder: grid too large");
    if (U == 0) return MCD_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rank_baseline_kernel, dim3((unsigned)U), dim3(RK_THREADS), 0, st, tvals, ldt, top_n, perms, n_perm,
                       p, baseline_ws);
    MCD_LAUNCH_CHECK("rank_baseline_kernel");
    const int split = (int)(C >= 8 ? (C / 32) * 32 : (C / 4) * 4);  // ATen's cascade / row_sum column split
    int npad = 2;
    while (npad < top_n) npad <<= 1;
    const size_t shmem = (size_t)RK_CT * npad * sizeof(unsigned long long);
    const dim3 grid((unsigned)U, (unsigned)mcd_cdiv(C, RK_CT));
    // whole 16-byte quads: aligned base and pitch, and every tile inside the row (the host pads S/P rows to a
    // multiple of 4 floats, or the last tile takes the scalar kernel below)
    const bool vec4 = (ldP % 4 == 0) && (((uintptr_t)P) % 16 == 0) && (mcd_cdiv(C, RK_CT) * RK_CT <= ldP);
    if (shmem > 48 * 1024) {
        hipError_t e1 = hipFuncSetAttribute((const void*)rank_reorder_kernel<true>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipError_t e2 = hipFuncSetAttribute((const void*)rank_reorder_kernel<false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        MCD_REQUIRE(e1 == hipSuccess && e2 == hipSuccess, MCD_E_LAUNCH, "mcd_rank_reorder: cannot reserve %zu B of LDS",
                    shmem);
    }
    if (vec4)
        hipLaunchKernelGGL(rank_reorder_kernel<true>, grid, dim3(RK_THREADS), shmem, st, P, ldP, C, tvals, tidx, ldt, top_n,
                           npad, split, baseline_ws, p, scale_p, out, ldo);
    else
        hipLaunchKernelGGL(rank_reorder_kernel<false>, grid, dim3(RK_THREADS), shmem, st, P, ldP, C, tvals, tidx, ldt,
                           top_n, npad, split, baseline_ws, p, scale_p, out, ldo);
    MCD_LAUNCH_CHECK("rank_reorder_kernel");
    return MCD_OK;
}
