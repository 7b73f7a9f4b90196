// k_topk.hip -- selection kernels (integer/bit work on fp32 keys; HBM-bound by the one read of A).
//   K3  mcd_col_topk   concept_vit/similarity.py:55  torch.topk(target_feats, dim=0, k)
//   K6  mcd_row_topk   describe_clip_neurons.py:64 (max), describe_broad_neurons.py:101 (topk k=10)
//       mcd_transpose  image-major [N,U] -> neuron-major [U,N]
//
// K3 design (one workgroup per neuron, activations of that neuron contiguous over images):
//   1. the neuron's N activations are read ONCE, coalesced, into registers as order-preserving
//      u32 keys (NaN on top, like torch.topk);
//   2. the K-th largest key is found by bisection on the key bits: per bit one v_cmp per cached
//      key, the wave count lands in an SGPR through ballot + s_bcnt1 (no VALU reduction), waves
//      are combined through LDS with one barrier per bit;
//   3. as soon as the survivors (key >= current lower bound) fit the LDS list (CAP entries) they
//      are compacted as (key, ~index) pairs and bitonic-sorted descending, which also orders ties
//      by the lower image index; the first K pairs are the answer.
//   Heavy ties (more than CAP keys equal to the K-th key, e.g. a dead ReLU channel) take the exact
//   path: all 32 bits are resolved, the winners above the threshold are taken and the remaining
//   slots are filled with the tied entries of lowest image index by an ordered block scan.
#include "mcd_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// transpose: 64x64 tiles through LDS (65-float rows: conflict-free column reads)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int64_t lds_, int64_t N,
                                                         int64_t U, float* __restrict__ dst, int64_t ldd) {
    __shared__ float tile[64][65];
    const int64_t n0 = (int64_t)blockIdx.y * 64, u0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 row groups
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = ty + 4 * i;
        const int64_t n = n0 + r, u = u0 + tx;
        tile[r][tx] = (n < N && u < U) ? src[n * lds_ + u] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = ty + 4 * i;
        const int64_t u = u0 + r, n = n0 + tx;
        if (u < U && n < N) dst[u * ldd + n] = tile[tx][r];
    }
}

// ------------------------------------------------------------------------------------------------
// K3
// ------------------------------------------------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ int block_sum_sgpr(int wave_cnt, int* s_cnt /*[2][NW]*/, int phase) {
    constexpr int NW = THREADS / 64;
    int* buf = s_cnt + (phase & 1) * NW;
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = wave_cnt;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += buf[w];
    return tot;
}

// bitonic sort, descending, of CAP u64 entries in LDS
template <int THREADS, int CAP>
__device__ __forceinline__ void bitonic_desc(unsigned long long* a) {
    for (int k = 2; k <= CAP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < CAP / 2; t += THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j cleared
                const int p = i | j;
                const bool desc = ((i & k) == 0);
                const unsigned long long x = a[i], y = a[p];
                if ((x < y) == desc) {
                    a[i] = y;
                    a[p] = x;
                }
            }
            __syncthreads();
        }
    }
}

template <int THREADS, int ITEMS, int CAP>
__global__ __launch_bounds__(THREADS) void neuron_topk_kernel(const float* __restrict__ At, int64_t ld, int64_t N,
                                                               int K, float* __restrict__ vals,
                                                               int32_t* __restrict__ idx, int64_t ldo) {
    constexpr int NW = THREADS / 64;
    __shared__ unsigned long long s_list[CAP];
    __shared__ int s_cnt[2 * NW];
    __shared__ int s_n;
    __shared__ int s_wtot[NW];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const float* row = At + (int64_t)blockIdx.x * ld;

    // ---- 1. one coalesced read of the neuron's activations -> keys in registers -------------
    uint32_t key[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t n = (int64_t)i * THREADS + tid;
        key[i] = (n < N) ? mcd_f2key(row[n]) : 0u;  // 0 is below every valid key
    }
    if (tid == 0) s_n = 0;

    // ---- 2. bisection for the K-th largest key -----------------------------------------------
    uint32_t T = 0;
    int64_t c = N;  // number of keys >= T
    int phase = 0;
    for (int b = 31; b >= 0 && c > CAP; --b) {
        const uint32_t cand = T | (1u << b);
        int wc = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) wc += __popcll(__ballot(key[i] >= cand));
        const int cnt = block_sum_sgpr<THREADS>(wc, s_cnt, phase++);
        if (cnt >= K) {
            T = cand;
            c = cnt;
        }
    }
    __syncthreads();

    // ---- 3. compact the survivors -------------------------------------------------------------
    if (c <= CAP) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int64_t n = (int64_t)i * THREADS + tid;
            if (n < N && key[i] >= T) {
                const int slot = atomicAdd(&s_n, 1);
                s_list[slot] = ((unsigned long long)key[i] << 32) | (uint32_t)(0xffffffffu - (uint32_t)n);
            }
        }
    } else {
        // exact threshold T, more than CAP keys tie with it: winners above T, then the ties of
        // lowest image index, ranked by an ordered block scan (n = i*THREADS + tid ascending).
        int wc = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) wc += __popcll(__ballot(key[i] > T));
        const int n_gt = block_sum_sgpr<THREADS>(wc, s_cnt, phase++);
        const int need = K - n_gt;
        int base = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {  // fully unrolled: key[] must stay in registers
            if (base < need || i == 0) {   // uniform
                const int64_t n = (int64_t)i * THREADS + tid;
                const bool valid = n < N;
                const bool eq = valid && key[i] == T;
                const unsigned long long m = __ballot(eq);
                const int before = __popcll(m & ((1ull << lane) - 1ull));
                __syncthreads();  // s_wtot reuse
                if (lane == 0) s_wtot[tid >> 6] = __popcll(m);
                __syncthreads();
                int woff = 0, tot = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    if (w < (tid >> 6)) woff += s_wtot[w];
                    tot += s_wtot[w];
                }
                const int rank = base + woff + before;
                if (eq && rank < need) {
                    const int slot = atomicAdd(&s_n, 1);
                    s_list[slot] = ((unsigned long long)key[i] << 32) | (uint32_t)(0xffffffffu - (uint32_t)n);
                }
                base += tot;
            }
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int64_t n = (int64_t)i * THREADS + tid;
            if (n < N && key[i] > T) {
                const int slot = atomicAdd(&s_n, 1);
                s_list[slot] = ((unsigned long long)key[i] << 32) | (uint32_t)(0xffffffffu - (uint32_t)n);
            }
        }
    }
    __syncthreads();
    const int filled = s_n;
    for (int t = filled + tid; t < CAP; t += THREADS) s_list[t] = 0ull;  // pad below every entry
    __syncthreads();

    // ---- 4. sort descending by (key, ~index): value order, ties to the lower image index ------
    bitonic_desc<THREADS, CAP>(s_list);

    for (int j = tid; j < K; j += THREADS) {
        const unsigned long long e = s_list[j];
        if (vals) vals[(int64_t)blockIdx.x * ldo + j] = mcd_key2f((uint32_t)(e >> 32));
        if (idx) idx[(int64_t)blockIdx.x * ldo + j] = (int32_t)(0xffffffffu - (uint32_t)(e & 0xffffffffu));
    }
}

// ------------------------------------------------------------------------------------------------
// K6: one wave per row of sim; each lane keeps a sorted top-KK list of its strided elements, then
// KK rounds of wave arg-max over the list heads.
// ------------------------------------------------------------------------------------------------
template <int KK>
__global__ __launch_bounds__(256) void row_topk_kernel(const float* __restrict__ sim, int64_t ld, int64_t U,
                                                        int64_t C, int k, float* __restrict__ vals,
                                                        int32_t* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= U) return;
    const float* sr = sim + row * ld;
    // entries: (key << 32) | ~col ; larger = better (larger value, then lower column)
    unsigned long long best[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) best[i] = 0ull;
    for (int64_t cidx = lane; cidx < C; cidx += 64) {
        unsigned long long e = ((unsigned long long)mcd_f2key(sr[cidx]) << 32) | (uint32_t)(0xffffffffu - (uint32_t)cidx);
#pragma unroll
        for (int i = 0; i < KK; ++i) {  // insertion into the descending list
            const unsigned long long b = best[i];
            const bool gt = e > b;
            best[i] = gt ? e : b;
            e = gt ? b : e;
        }
    }
    for (int j = 0; j < k; ++j) {
        unsigned long long h = best[0];
        unsigned long long w = h;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(w, o, 64);
            w = other > w ? other : w;
        }
        if (h == w && w != 0ull) {  // the unique owner pops its head
#pragma unroll
            for (int i = 0; i < KK - 1; ++i) best[i] = best[i + 1];
            best[KK - 1] = 0ull;
        }
        if (lane == 0) {
            vals[row * k + j] = mcd_key2f((uint32_t)(w >> 32));
            idx[row * k + j] = (int32_t)(0xffffffffu - (uint32_t)(w & 0xffffffffu));
        }
    }
}

template <int THREADS, int ITEMS, int CAP>
int launch_neuron_topk(const float* At, int64_t ld, int64_t N, int64_t U, int K, float* vals, int32_t* idx,
                       int64_t ldo, hipStream_t st) {
    hipLaunchKernelGGL((neuron_topk_kernel<THREADS, ITEMS, CAP>), dim3((unsigned)U), dim3(THREADS), 0, st, At, ld, N,
                       K, vals, idx, ldo);
    return 0;
}

template <int CAP>
int dispatch_neuron_topk(const float* At, int64_t ld, int64_t N, int64_t U, int K, float* vals, int32_t* idx,
                         int64_t ldo, hipStream_t st) {
    if (N <= 256 * 4) return launch_neuron_topk<256, 4, CAP>(At, ld, N, U, K, vals, idx, ldo, st);
    if (N <= 256 * 8) return launch_neuron_topk<256, 8, CAP>(At, ld, N, U, K, vals, idx, ldo, st);
    if (N <= 256 * 16) return launch_neuron_topk<256, 16, CAP>(At, ld, N, U, K, vals, idx, ldo, st);
    if (N <= 256 * 40) return launch_neuron_topk<256, 40, CAP>(At, ld, N, U, K, vals, idx, ldo, st);
    // 1024-thread workgroups are capped at 128 VGPRs: 32/64 keys per thread spill a little
    if (N <= 1024 * 16) return launch_neuron_topk<1024, 16, CAP>(At, ld, N, U, K, vals, idx, ldo, st);
    if (N <= 1024 * 32) return launch_neuron_topk<1024, 32, CAP>(At, ld, N, U, K, vals, idx, ldo, st);
    if (N <= 1024 * 64) return launch_neuron_topk<1024, 64, CAP>(At, ld, N, U, K, vals, idx, ldo, st);
    return 1;
}

}  // namespace

extern "C" int mcd_transpose(const float* src, int64_t lds_, int64_t N, int64_t U, float* dst, int64_t ldd,
                             mcd_stream_t stream) {
    MCD_REQUIRE(src && dst, MCD_E_ARG, "mcd_transpose: NULL pointer");
    MCD_REQUIRE(N >= 0 && U >= 0 && lds_ >= U && ldd >= N, MCD_E_ARG, "mcd_transpose: bad shape");
    if (N == 0 || U == 0) return MCD_OK;
    const dim3 grid((unsigned)mcd_cdiv(U, 64), (unsigned)mcd_cdiv(N, 64));
    MCD_REQUIRE(grid.y <= 65535u, MCD_E_UNSUPPORTED, "mcd_transpose: N too large (%lld)", (long long)N);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, lds_, N, U, dst, ldd);
    MCD_LAUNCH_CHECK("transpose_kernel");
    return MCD_OK;
}

static int64_t topk_ws_ld(int64_t N) { return (N + 3) / 4 * 4; }

extern "C" size_t mcd_col_topk_workspace(int64_t N, int64_t U, int64_t stride_n, int64_t stride_u, int K) {
    (void)K;
    if (stride_n == 1 && (stride_u != 1 || U == 1)) return 0;
    return (size_t)U * (size_t)topk_ws_ld(N) * sizeof(float);
}

extern "C" int mcd_col_topk(const float* A, int64_t N, int64_t U, int64_t stride_n, int64_t stride_u, int K,
                            float* vals, int32_t* idx, int64_t ldo, void* ws, size_t ws_bytes, mcd_stream_t stream) {
    MCD_REQUIRE(A && (vals || idx), MCD_E_ARG, "mcd_col_topk: NULL pointer");
    MCD_REQUIRE(N > 0 && U >= 0, MCD_E_ARG, "mcd_col_topk: bad shape N=%lld U=%lld", (long long)N, (long long)U);
    MCD_REQUIRE(K >= 1 && K <= N, MCD_E_RANGE, "selected index k out of range (k=%d, N=%lld)", K, (long long)N);
    MCD_REQUIRE(ldo >= K, MCD_E_ARG, "mcd_col_topk: ldo < K");
    MCD_REQUIRE(N < 0x7fffffffLL, MCD_E_UNSUPPORTED, "mcd_col_topk: N too large");
    if (U == 0) return MCD_OK;
    hipStream_t st = (hipStream_t)stream;
    const float* At;
    int64_t ld;
    if (stride_n == 1 && (stride_u != 1 || U == 1)) {
        MCD_REQUIRE(stride_u >= N || U == 1, MCD_E_ARG, "mcd_col_topk: neuron-major stride_u < N");
        At = A;
        ld = stride_u;
    } else {
        MCD_REQUIRE(stride_u == 1 && stride_n >= U, MCD_E_ARG,
                    "mcd_col_topk: need image-major (stride_u==1) or neuron-major (stride_n==1) input");
        const size_t need = mcd_col_topk_workspace(N, U, stride_n, stride_u, K);
        MCD_REQUIRE(ws && ws_bytes >= need, MCD_E_WORKSPACE, "mcd_col_topk: workspace %zu < %zu bytes", ws_bytes, need);
        ld = topk_ws_ld(N);
        const int rc = mcd_transpose(A, stride_n, N, U, (float*)ws, ld, stream);
        if (rc) return rc;
        At = (const float*)ws;
    }
    int rc;
    if (K <= 128)
        rc = dispatch_neuron_topk<256>(At, ld, N, U, K, vals, idx, ldo, st);
    else if (K <= 1024)
        rc = dispatch_neuron_topk<1024>(At, ld, N, U, K, vals, idx, ldo, st);
    else
        return mcd_fail(MCD_E_UNSUPPORTED, "mcd_col_topk: K=%d > 1024 not supported", K);
    MCD_REQUIRE(rc == 0, MCD_E_UNSUPPORTED, "mcd_col_topk: N=%lld above the register-resident limit (65536)",
                (long long)N);
    MCD_LAUNCH_CHECK("neuron_topk_kernel");
    return MCD_OK;
}

extern "C" int mcd_row_topk(const float* sim, int64_t ld, int64_t U, int64_t C, int k, float* vals, int32_t* idx,
                            mcd_stream_t stream) {
    MCD_REQUIRE(sim && vals && idx, MCD_E_ARG, "mcd_row_topk: NULL pointer");
    MCD_REQUIRE(U >= 0 && C > 0 && ld >= C, MCD_E_ARG, "mcd_row_topk: bad shape");
    MCD_REQUIRE(k >= 1 && k <= C, MCD_E_RANGE, "selected index k out of range (k=%d, C=%lld)", k, (long long)C);
    MCD_REQUIRE(k <= 16, MCD_E_UNSUPPORTED, "mcd_row_topk: k=%d > 16 not supported", k);
    if (U == 0) return MCD_OK;
    const dim3 grid((unsigned)mcd_cdiv(U, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (k == 1)
        hipLaunchKernelGGL(row_topk_kernel<1>, grid, block, 0, st, sim, ld, U, C, k, vals, idx);
    else if (k <= 4)
        hipLaunchKernelGGL(row_topk_kernel<4>, grid, block, 0, st, sim, ld, U, C, k, vals, idx);
    else if (k <= 10)
        hipLaunchKernelGGL(row_topk_kernel<10>, grid, block, 0, st, sim, ld, U, C, k, vals, idx);
    else
        hipLaunchKernelGGL(row_topk_kernel<16>, grid, block, 0, st, sim, ld, U, C, k, vals, idx);
    MCD_LAUNCH_CHECK("row_topk_kernel");
    return MCD_OK;
}
