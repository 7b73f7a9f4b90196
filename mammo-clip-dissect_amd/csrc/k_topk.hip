// k_topk.hip -- selection kernels (integer/bit work on fp32 keys; HBM-bound by the one read of A).
//   K3  mcd_col_topk   concept_vit/similarity.py:55  torch.topk(target_feats, dim=0, k)
//   K6  mcd_row_topk   describe_clip_neurons.py:64 (max), describe_broad_neurons.py:101 (topk k=10)
//       mcd_transpose  image-major [N,U] -> neuron-major [U,N]
//
// K3 design (one workgroup per neuron, activations of that neuron contiguous over images): neuron_topk_fast_kernel
//   1. the neuron's N activations are read ONCE into registers as floats: all of a thread's 16-byte buffer loads are issued
//      before anything waits for one (no bounds branch: dwords past the row's end come back as 0 and become pads);
//   2. every thread takes the maximum of its own values; one wave finds (about) the K-th largest of those THREADS maxima
//      by bisection on the key bits with ballot counts -- a lower bound of the K-th largest value that ~1.1 K values pass;
//   3. the values at or above the bound are compacted into LDS as (key, ~index) pairs (wave ballots + one LDS atomic per
//      wave) and ranked by counting (every thread compares); the first K are the answer, ties by the lower image index.
//   A row with a NaN (torch.topk ranks it above +inf), with more than CAP values at the bound (heavy ties, e.g. a dead
//   ReLU channel) or with K beyond the class is only flagged and finished by neuron_topk_stream_kernel (exact, slow).
//   Rows longer than 51 200 values take neuron_topk_twopass_kernel (the row is read twice, the second time from L2).
#include "mcd_common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

#ifndef MCD_K3_CHUNK_ABOVE
#define MCD_K3_CHUNK_ABOVE 32   // items per thread up to which the compaction runs over all items at once (profiles/r04_k3_notes.txt (e))
#endif
#ifndef MCD_K3_WAVES_512_13
#define MCD_K3_WAVES_512_13 6       // waves per SIMD the 25 000-image class is compiled for (69 registers as it comes: 3 workgroups per CU)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// transpose: 64x64 tiles through LDS (65-float rows: conflict-free column reads)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int64_t lds_, int64_t N,
                                                         int64_t U, float* __restrict__ dst, int64_t ldd) {
    __shared__ float tile[64][65];
    const int64_t n0 = (int64_t)blockIdx.y * 64, u0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 row groups
    // all 16 loads first, on clamped (always valid) addresses: behind a bounds test each load is waited for before the next is
    // issued; the edge tiles' out-of-range slots hold copies that the bounds test of the stores drops
    float v[16];
    const int64_t uc = (u0 + tx < U) ? u0 + tx : U - 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t n = n0 + ty + 4 * i;
        v[i] = src[(n < N ? n : N - 1) * lds_ + uc];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[ty + 4 * i][tx] = v[i];
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

__device__ __forceinline__ unsigned long long pack_entry(uint32_t key, uint32_t n) {
    return ((unsigned long long)key << 32) | (uint32_t)(0xffffffffu - n);  // larger = better; ties: lower index
}

// Order the c <= CAP candidates of the LDS list and write the best K.  The entries are unique (they embed
// the image index), so "number of entries greater than e" is e's position.  Each lane keeps CAP/64 entries
// in registers; a wave ranks one candidate with CAP/64 compares + ballot popcounts (the count is wave-wide,
// no reduction), and the waves of the workgroup take the candidates round-robin.
template <int THREADS, int CAP>
__device__ __forceinline__ void rank_and_store(const unsigned long long* s_list, int c, int K, float* vals,
                                               int32_t* idx, int64_t obase) {
    constexpr int NW = THREADS / 64;
    constexpr int PER = (CAP + 63) / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long mine[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) mine[k] = (lane + 64 * k < c) ? s_list[lane + 64 * k] : 0ull;  // 0 < every entry
    // this wave's candidates are t = wave + NW*i; lane i fetches candidate i once, the loop broadcasts it with
    // v_readlane (no LDS latency inside the loop), and lane i keeps the rank so the stores go out together
    static_assert(CAP <= 64 * NW, "a wave ranks at most 64 candidates");
    const int nmine = (c - wave + NW - 1) / NW;
    const unsigned long long my = (lane < nmine) ? s_list[wave + NW * lane] : 0ull;
    const uint32_t my_lo = (uint32_t)my, my_hi = (uint32_t)(my >> 32);
    int myrank = 0x7fffffff;
    for (int i = 0; i < nmine; ++i) {
        const unsigned long long e = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)my_hi, i) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)my_lo, i);
        int r = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) r += __popcll(__ballot(mine[k] > e));
        if (lane == i) myrank = r;
    }
    if (myrank < K) {
        if (vals) vals[obase + myrank] = mcd_key2f(my_hi);
        if (idx) idx[obase + myrank] = (int32_t)(0xffffffffu - my_lo);
    }
}

// rank_and_store with every thread at work: the candidates are split over `parts` groups of threads (parts = THREADS /
// the power of two that holds c); thread (part, j) counts how many entries of its part's share of the list beat
// candidate j -- the list entry comes as an LDS broadcast (one address per wave) -- and the partial ranks meet in LDS
// atomics on distinct addresses.  s_rank[0..CAP) must be zero on entry (one barrier before, at least).  Ends after a
// barrier; 3 600 -> ~1 200 cycles per neuron at c ~ 110 against one candidate per wave and ballot counts.
template <int THREADS, int CAP>
__device__ __forceinline__ void rank_all_and_store(const unsigned long long* s_list, int* s_rank, int c, int K, float* vals,
                                                   int32_t* idx, int64_t obase) {
    static_assert(CAP <= THREADS && CAP % 64 == 0, "one thread per candidate at least");
    int cp = 64;
    while (cp < c) cp <<= 1;                                   // c <= CAP <= THREADS
    const int parts = THREADS / cp;
    const int j = threadIdx.x & (cp - 1);
    const int part = __builtin_amdgcn_readfirstlane((int)threadIdx.x / cp);   // cp is a multiple of 64: wave-uniform
    const int len = (c + parts - 1) / parts;
    const int i0 = part * len, i1 = min(c, i0 + len);
    const unsigned long long e = (j < c) ? s_list[j] : ~0ull;  // nothing beats the padding lanes' value
    int r = 0;
#pragma unroll 4
    for (int i = i0; i < i1; ++i) r += (s_list[i] > e) ? 1 : 0;
    if (r) atomicAdd(&s_rank[j], r);
    __syncthreads();
    if ((int)threadIdx.x < c) {
        const int rank = s_rank[threadIdx.x];
        if (rank < K) {
            const unsigned long long my = s_list[threadIdx.x];
            if (vals) vals[obase + rank] = mcd_key2f((uint32_t)(my >> 32));
            if (idx) idx[obase + rank] = (int32_t)(0xffffffffu - (uint32_t)my);
        }
    }
}

// wave-wide AND / OR of a 32-bit value, result wave-uniform: 4 DPP row rotations reduce the 16-lane rows, 4 readlanes
// join the rows (both operations are idempotent, so rotating by 1, 2, 4, 8 is an all-reduce)
template <bool IS_AND>
__device__ __forceinline__ uint32_t wave_and_or(uint32_t v) {
#define MCD_ROR(n)                                                                                                       \
    {                                                                                                                    \
        const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + (n), 0xf, 0xf, false);               \
        v = IS_AND ? (v & o) : (v | o);                                                                                  \
    }
    MCD_ROR(1) MCD_ROR(2) MCD_ROR(4) MCD_ROR(8)
#undef MCD_ROR
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                   c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return IS_AND ? ((a & b) & (c & d)) : ((a | b) | (c | d));
}

// T = (about) the K-th largest of the 64 * NW per-thread maxima in s_max; ONE wave calls this.  Bisection on the key
// bits with wave-wide ballot counts; it ends as soon as between K and K + K/8 maxima pass (a few extra survivors cost
// less than more bits).  The bits that ALL maxima share need no search: the loop starts at the highest bit in which
// two maxima differ (on continuous data that skips the sign and most of the exponent: ~8 of ~20 iterations).
// (Regula falsi on the counts with a bisection safeguard -- fewer steps, each with float arithmetic on the one wave's critical
// path -- measured SLOWER: 0.255 -> 0.268 ms at 9 216 x 25 000, 0.110 -> 0.122 for K6's long rows.)
template <int NW>
__device__ __forceinline__ uint32_t bound_from_maxima(const uint32_t* s_max, int K, int lane) {
    uint32_t mx[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) mx[j] = s_max[j * 64 + lane];
    uint32_t Tw = 0;
    int b = 31;
    if (K <= 64 * NW) {          // (with fewer than K maxima the search must end at T = 0: every key survives, the caller flags the neuron)
        uint32_t a = mx[0], o = mx[0];
#pragma unroll
        for (int j = 1; j < NW; ++j) {
            a &= mx[j];
            o |= mx[j];
        }
        a = wave_and_or<true>(a);
        o = wave_and_or<false>(o);
        const uint32_t diff = a ^ o;
        if (diff == 0u) return a;                               // all maxima equal
        b = 31 - __builtin_clz(diff);
        Tw = b == 31 ? 0u : (a & ~((2u << b) - 1u));            // the shared bits above b
    }
    for (; b >= 0; --b) {
        const uint32_t cand = Tw | (1u << b);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < NW; ++j) cnt += __popcll(__ballot(mx[j] >= cand));
        if (cnt >= K) {
            Tw = cand;
            if (cnt <= K + (K >> 3)) break;
        }
    }
    return Tw;
}

// Order c <= CAP candidates with a bitonic sort in LDS (descending; the entries are unique) and write the best K.
// For K beyond what rank_and_store covers (a wave ranks at most 64 candidates): rank_reorder's top 5 % of 50 000
// images is K = 2500.
template <int THREADS, int CAP>
__device__ __forceinline__ void bitonic_and_store(unsigned long long* s_list, int c, int K, float* vals, int32_t* idx,
                                                  int64_t obase) {
    int npad = 2;
    while (npad < c) npad <<= 1;
    for (int i = c + (int)threadIdx.x; i < npad; i += THREADS) s_list[i] = 0ull;   // 0 < every entry: sorts last
    __syncthreads();
    for (int k2 = 2; k2 <= npad; k2 <<= 1) {
        for (int st = k2 >> 1; st > 0; st >>= 1) {
            for (int e = threadIdx.x; e < (npad >> 1); e += THREADS) {
                const int i = ((e / st) * (st << 1)) + (e % st);
                const int l = i + st;
                const bool desc = ((i & k2) == 0);
                const unsigned long long x = s_list[i], y = s_list[l];
                if ((x < y) == desc) {
                    s_list[i] = y;
                    s_list[l] = x;
                }
            }
            __syncthreads();
        }
    }
    for (int r = threadIdx.x; r < K; r += THREADS) {
        const unsigned long long e = s_list[r];
        if (vals) vals[obase + r] = mcd_key2f((uint32_t)(e >> 32));
        if (idx) idx[obase + r] = (int32_t)(0xffffffffu - (uint32_t)(e & 0xffffffffu));
    }
}

// "This row is left to the streaming kernel" mark: -1 in the row's flag word.  flag_stride == 1: an int array of its own,
// where a finished row writes 0.  flag_stride > 1 (K6 on long rows): the flag word IS the row's first output index -- a
// finished row's best index (>= 0) lands there, so only the -1 is written.
__device__ __forceinline__ void mark_slow(int* slow_flag, int flag_stride, bool slow) {
    if (slow) slow_flag[(int64_t)blockIdx.x * flag_stride] = -1;
    else if (flag_stride == 1) slow_flag[blockIdx.x] = 0;
}

// FAST path: the neuron's keys live in registers (4*QUADS per thread, 16-byte loads).
//   Lower bound without a data pass: every thread takes the max of its own keys; the K-th largest of those
//   THREADS maxima (a subset of the keys) is <= the K-th largest key, and because the maxima are the top of
//   disjoint groups the bound is tight (about 1.1 K keys pass it on continuous data).  One wave finds that
//   value from the LDS copy of the maxima (THREADS / 64 values per lane, wave-wide ballot counts), the keys
//   above the bound are compacted into LDS and ordered by rank.
//   A neuron with more than CAP keys at or above the bound (heavy ties, e.g. a dead ReLU channel; or
//   K > THREADS) is only flagged here and is finished by neuron_topk_stream_kernel.
//   Occupancy: 4 k + 17 registers for k quads per thread; the 25 000- and 32 768-image classes (512 threads x 13 / 16 quads)
//   are compiled for 6 waves per SIMD = 3 workgroups per CU (at 8 the 25 000-image class spills: 0.196 against 0.165 ms).
template <int THREADS, int QUADS, int CAP>
__global__ __launch_bounds__(THREADS, (THREADS == 512 && (QUADS == 13 || QUADS == 16)) ? MCD_K3_WAVES_512_13 : 1) void neuron_topk_fast_kernel(const float* __restrict__ At, int64_t ld,
                                                                    int64_t N, int K, float* __restrict__ vals,
                                                                    int32_t* __restrict__ idx, int64_t ldo,
                                                                    int* __restrict__ slow_flag, int flag_stride,
                                                                    int vec_ok) {
    constexpr int NW = THREADS / 64;
    constexpr int ITEMS = 4 * QUADS;
    __shared__ unsigned long long s_list[CAP + 1];   // [CAP]: where the entries of an overflowing list go
    __shared__ uint32_t s_max[THREADS];
    __shared__ int s_rank[CAP];
    __shared__ int s_n;
    __shared__ int s_nan[NW];      // per wave: a NaN among its elements
    __shared__ uint32_t s_T;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const float* row = At + (int64_t)blockIdx.x * ld;
    if (tid < CAP) s_rank[tid] = 0;
#ifdef MCD_K3_STAMPS
    unsigned long long stamp[6];
    stamp[0] = __builtin_amdgcn_s_memtime();
#define MCD_K3_STAMP(i) stamp[i] = __builtin_amdgcn_s_memtime()
#else
#define MCD_K3_STAMP(i)
#endif

    // ---- 1. one coalesced read of the neuron's activations, kept in registers AS FLOATS ----------------------------
    //         Per element: one v_max_f32 (thread maximum) and one NaN test; the order-preserving integer keys (6
    //         instructions each) are made for the thread maximum and, in step 3, for the ~1.1 K survivors only.  A NaN
    //         ranks above +inf for torch.topk but loses every float comparison, so a row that holds one is flagged and
    //         left to the streaming kernel (which works on keys).  Slots past the end of the row hold NaN: they never pass
    //         a comparison, v_max ignores them, and they are not counted as NaNs of the row.
    //         NaN test without a compare per element: the elements are summed (two per v_pk_add_f32); a NaN anywhere makes
    //         the sum NaN.  So does +inf beside -inf: such a row takes the streaming kernel without needing to -- exact there too.
    //         ALL the row's 16-byte loads are issued before anything waits for one: no branch and no use of a loaded value sits
    //         between them (until round 4 the NaN compares sat inside the per-quad bounds branch, the compiler waited for each
    //         load there, and a wave had ONE load in flight: five memory round trips in a row per neuron at 10 000 images).
    //         They are buffer loads on a descriptor of the row: one offset register for all quads (the quad's start rides in
    //         the scalar offset), and the dwords at or past the row's end come back as 0 -- no bounds branch, no access past
    //         the row -- and are set to the pad value afterwards, in the one or two quads that can have any.
    float x[ITEMS];
    const int Ni = (int)N;                                 // N < 2^31 (checked by the host): 32-bit index arithmetic
    f32x2 nsum = {0.f, 0.f};
    const float pad = __uint_as_float(0x7fc00000u);
    if (__builtin_amdgcn_readfirstlane(vec_ok)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, Ni * 4, 0x00020000);
#pragma unroll
        for (int q = 0; q < QUADS; ++q) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, q * THREADS * 16, 0));
            x[4 * q + 0] = v[0];
            x[4 * q + 1] = v[1];
            x[4 * q + 2] = v[2];
            x[4 * q + 3] = v[3];
        }
    } else {                                               // rows that are not 16-byte aligned: element loads, 0 past the end too
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int e = (((i >> 2) * THREADS + tid) << 2) + (i & 3);
            x[i] = (e < Ni) ? row[e] : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        if ((q + 1) * THREADS * 4 <= Ni) {                 // uniform: every thread's quad is whole (all but the last one or two)
            nsum += f32x2{x[4 * q + 0], x[4 * q + 1]};
            nsum += f32x2{x[4 * q + 2], x[4 * q + 3]};
        } else {
            const int e = (q * THREADS + tid) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                nsum.x += x[4 * q + j];                    // (a slot past the end still holds the 0 it was loaded as)
                x[4 * q + j] = (e + j < Ni) ? x[4 * q + j] : pad;
            }
        }
    }
    const float nchk = nsum.x + nsum.y;
    const bool has_nan = nchk != nchk;
    float fmax_ = -INFINITY;       // a thread without elements: -inf, at or below every real key
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) fmax_ = fmaxf(fmax_, x[i]);   // v_max_f32: a NaN operand is dropped
    const uint32_t tmax = mcd_f2key(fmax_);
    const unsigned long long nan_mask = __ballot(has_nan);          // (outside the lane-0 branch: a ballot counts the active lanes)
    if (lane == 0) s_nan[tid >> 6] = nan_mask != 0ull ? 1 : 0;
    MCD_K3_STAMP(1);
    s_max[tid] = tmax;
    if (tid == 0) s_n = 0;
    __syncthreads();
    MCD_K3_STAMP(2);

    // ---- 2. ONE wave: T = (about) the K-th largest of the THREADS maxima; the others wait at the barrier -------
    //         (every wave searching on its own needs no barrier, but with 8 waves per SIMD resident the 32 x NW
    //         ballots per wave are what the SIMDs spend most of the kernel on)
    if (tid < 64) {
        const uint32_t Tw = bound_from_maxima<NW>(s_max, K, lane);
        if (lane == 0) s_T = Tw;
    }
    __syncthreads();
    MCD_K3_STAMP(3);
    const uint32_t T = s_T;

    // ---- 3. compact the keys >= T into LDS (about 1.1 K of them): a wave counts its survivors with ballots, reserves
    //         its range of the list with ONE LDS atomic and places them by prefix popcounts (an atomic per survivor, all
    //         on one address, serialised: 4 500 cycles per neuron at N = 10 000, a fifth of the workgroup's lifetime)
    {
        // the bound as a float: key order refines float order (it only separates -0 from +0), so x >= Tf holds for every
        // element whose key is >= T; T == 0 (fewer than K maxima) lets every element through
        const float Tf = T == 0u ? -INFINITY : mcd_key2f(T);
        // survivor of item i -> list position base + (survivors in lower lanes): a v_mbcnt pair with the base riding in its
        // addend; a list that overflows (the row is then left to the streaming kernel) piles up in the spare slot s_list[CAP].
        // The key is that of a non-NaN value (3 instructions; a row with a NaN is flagged above and its list is not used).
        const auto place = [&](int i, unsigned long long m, int base) {
            const uint32_t slot = min(__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)base)),
                                      (uint32_t)CAP);
            const uint32_t n = (uint32_t)((((i >> 2) * THREADS + tid) << 2) + (i & 3));
            const uint32_t b = __float_as_uint(x[i]);
            s_list[slot] = pack_entry(b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u), n);
        };
        if constexpr (ITEMS <= MCD_K3_CHUNK_ABOVE) {
            // all items at once; "no survivor in this wave" (most (wave, item) pairs) is tested on the re-made ballot
            int wcnt = 0;
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) wcnt += __popcll(__ballot(x[i] >= Tf));
            int base = 0;
            if (lane == 0 && wcnt) base = atomicAdd(&s_n, wcnt);
            base = __builtin_amdgcn_readfirstlane(base);
            if (wcnt) {
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) {
                    const bool pred = x[i] >= Tf;
                    const unsigned long long m = __ballot(pred);
                    const int pc = __popcll(m);
                    if (pc) {
                        if (pred) place(i, m, base);
                        base += pc;
                    }
                }
            }
        } else {
            // in chunks of 20 items: a chunk's ballot masks stay in scalar registers between its count and its placement
            // (40 or 52 pairs would be spilled into vector-register lanes), and a chunk reserves its own range
            constexpr int CH = 20;
#pragma unroll
            for (int c0 = 0; c0 < ITEMS; c0 += CH) {
                unsigned long long mk[CH];
                int wcnt = 0;
#pragma unroll
                for (int i = c0; i < c0 + CH && i < ITEMS; ++i) {
                    mk[i - c0] = __ballot(x[i] >= Tf);
                    wcnt += __popcll(mk[i - c0]);
                }
                if (wcnt) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&s_n, wcnt);
                    base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                    for (int i = c0; i < c0 + CH && i < ITEMS; ++i) {
                        const unsigned long long m = mk[i - c0];
                        if (m) {
                            if (x[i] >= Tf) place(i, m, base);
                            base += __popcll(m);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    MCD_K3_STAMP(4);
    const int c = s_n;
    int any_nan = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) any_nan |= s_nan[w];
    const bool slow = c > CAP || c < K || any_nan != 0;  // too many ties at the bound, K > THREADS, or a NaN in the row: the streaming kernel
    if (tid == 0) mark_slow(slow_flag, flag_stride, slow);
    if (slow) return;

    // ---- 4. order the <= CAP survivors by rank and write the best K ---------------------------
    rank_all_and_store<THREADS, CAP>(s_list, s_rank, c, K, vals, idx, (int64_t)blockIdx.x * ldo);
#ifdef MCD_K3_STAMPS
    __syncthreads();
    MCD_K3_STAMP(5);
    if (tid == 0 && vals) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(vals + (int64_t)blockIdx.x * ldo);
        for (int i = 0; i < 6; ++i) o[i] = stamp[i];
    }
#endif
#undef MCD_K3_STAMP
}

// TWO-PASS variant of the fast path: the keys are not kept in registers.  Pass 1 finds the per-thread maxima, one wave
// derives the bound T from them, pass 2 re-reads the row (L2 / Infinity Cache) and compacts the survivors.  No limit on
// N.  Measured: slower than the register-resident classes up to 25 000 images (0.158 vs 0.111 ms at 10 000), faster
// beyond (0.27 vs 0.35 ms at 50 000; 0.30 vs 3.7 ms at 100 000, where only the streaming kernel applied before).
template <int CAP>
__global__ __launch_bounds__(256) void neuron_topk_twopass_kernel(const float* __restrict__ At, int64_t ld, int64_t N,
                                                                   int K, float* __restrict__ vals,
                                                                   int32_t* __restrict__ idx, int64_t ldo,
                                                                   int* __restrict__ slow_flag, int flag_stride) {
    constexpr int THREADS = 256, NW = 4;
    __shared__ unsigned long long s_list[CAP];
    __shared__ uint32_t s_max[THREADS];
    __shared__ int s_n;
    __shared__ uint32_t s_T;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const float* row = At + (int64_t)blockIdx.x * ld;
    const int Ni = (int)N, N4 = Ni & ~3;      // ld % 4 == 0 and a 16-byte aligned base (host-checked)

    uint32_t tmax = 0;
#pragma unroll 4
    for (int e = 4 * tid; e < N4; e += 4 * THREADS) {
        const float4 v = *reinterpret_cast<const float4*>(row + e);
        const uint32_t k0 = mcd_f2key(v.x), k1 = mcd_f2key(v.y), k2 = mcd_f2key(v.z), k3 = mcd_f2key(v.w);
        const uint32_t a = k0 > k1 ? k0 : k1, b = k2 > k3 ? k2 : k3;
        const uint32_t m = a > b ? a : b;
        tmax = m > tmax ? m : tmax;
    }
    for (int e = N4 + tid; e < Ni; e += THREADS) {
        const uint32_t k = mcd_f2key(row[e]);
        tmax = k > tmax ? k : tmax;
    }
    s_max[tid] = tmax;
    if (tid == 0) s_n = 0;
    __syncthreads();
    if (tid < 64) {
        uint32_t mx[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) mx[j] = s_max[j * 64 + lane];
        uint32_t Tw = 0;
        for (int b = 31; b >= 0; --b) {
            const uint32_t cand = Tw | (1u << b);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < NW; ++j) cnt += __popcll(__ballot(mx[j] >= cand));
            if (cnt >= K) {
                Tw = cand;
                if (cnt <= K + (K >> 3)) break;
            }
        }
        if (lane == 0) s_T = Tw;
    }
    __syncthreads();
    const uint32_t T = s_T;
    auto take = [&](uint32_t key, int n) {
        if (key >= T && key != 0u) {
            const int slot = atomicAdd(&s_n, 1);
            if (slot < CAP) s_list[slot] = pack_entry(key, (uint32_t)n);
        }
    };
#pragma unroll 4
    for (int e = 4 * tid; e < N4; e += 4 * THREADS) {
        const float4 v = *reinterpret_cast<const float4*>(row + e);
        take(mcd_f2key(v.x), e);
        take(mcd_f2key(v.y), e + 1);
        take(mcd_f2key(v.z), e + 2);
        take(mcd_f2key(v.w), e + 3);
    }
    for (int e = N4 + tid; e < Ni; e += THREADS) take(mcd_f2key(row[e]), e);
    __syncthreads();
    const int c = s_n;
    const bool slow = c > CAP || c < K;
    if (tid == 0) mark_slow(slow_flag, flag_stride, slow);
    if (slow) return;
    rank_and_store<THREADS, CAP>(s_list, c, K, vals, idx, (int64_t)blockIdx.x * ldo);
}

// STREAMING path: any N, any tie pattern; keys are re-read from memory (L2) on every pass.  Used for the
// neurons the fast kernel flagged and for N beyond the register-resident limit.
template <int THREADS, int CAP>
__global__ __launch_bounds__(THREADS) void neuron_topk_stream_kernel(const float* __restrict__ At, int64_t ld,
                                                                      int64_t N, int K, float* __restrict__ vals,
                                                                      int32_t* __restrict__ idx, int64_t ldo,
                                                                      const int* __restrict__ slow_flag, int flag_stride) {
    constexpr int NW = THREADS / 64;
    __shared__ unsigned long long s_list[CAP];
    __shared__ int s_cnt[2 * NW];
    __shared__ int s_n;
    __shared__ int s_wtot[NW];
    if (slow_flag && slow_flag[(int64_t)blockIdx.x * flag_stride] != -1) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const float* row = At + (int64_t)blockIdx.x * ld;
    if (tid == 0) s_n = 0;

    uint32_t T = 0;
    int64_t c = N;
    int phase = 0;
    for (int b = 31; b >= 0 && c > CAP; --b) {
        const uint32_t cand = T | (1u << b);
        int wc = 0;
        for (int64_t n0 = 0; n0 < N; n0 += THREADS) {
            const int64_t n = n0 + tid;
            wc += __popcll(__ballot(n < N && mcd_f2key(row[n]) >= cand));
        }
        const int cnt = block_sum_sgpr<THREADS>(wc, s_cnt, phase++);
        if (cnt >= K) {
            T = cand;
            c = cnt;
        }
    }
    __syncthreads();
    if (c <= CAP) {
        for (int64_t n0 = 0; n0 < N; n0 += THREADS) {
            const int64_t n = n0 + tid;
            if (n < N) {
                const uint32_t k = mcd_f2key(row[n]);
                if (k >= T) s_list[atomicAdd(&s_n, 1)] = pack_entry(k, (uint32_t)n);
            }
        }
    } else {
        // T is exact (all 32 bits resolved).  Winners above T, then the `need` tied entries of lowest image
        // index, ranked by an ordered block scan over ascending n.
        int wc = 0;
        for (int64_t n0 = 0; n0 < N; n0 += THREADS) {
            const int64_t n = n0 + tid;
            wc += __popcll(__ballot(n < N && mcd_f2key(row[n]) > T));
        }
        const int need = K - block_sum_sgpr<THREADS>(wc, s_cnt, phase++);
        int seen = 0;  // tied entries before this chunk (uniform)
        for (int64_t n0 = 0; n0 < N; n0 += THREADS) {
            const int64_t n = n0 + tid;
            const uint32_t k = (n < N) ? mcd_f2key(row[n]) : 0u;
            if (n < N && k > T) s_list[atomicAdd(&s_n, 1)] = pack_entry(k, (uint32_t)n);
            if (seen < need) {  // uniform
                const bool eq = (n < N) && k == T;
                const unsigned long long m = __ballot(eq);
                __syncthreads();
                if (lane == 0) s_wtot[tid >> 6] = __popcll(m);
                __syncthreads();
                int woff = 0, tot = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    if (w < (tid >> 6)) woff += s_wtot[w];
                    tot += s_wtot[w];
                }
                const int rank = seen + woff + __popcll(m & ((1ull << lane) - 1ull));
                if (eq && rank < need) s_list[atomicAdd(&s_n, 1)] = pack_entry(k, (uint32_t)n);
                seen += tot;
            }
        }
    }
    __syncthreads();
    if constexpr (CAP <= 64 * NW) rank_and_store<THREADS, CAP>(s_list, s_n, K, vals, idx, (int64_t)blockIdx.x * ldo);
    else bitonic_and_store<THREADS, CAP>(s_list, s_n, K, vals, idx, (int64_t)blockIdx.x * ldo);
}

// ------------------------------------------------------------------------------------------------
// K6: one wave per row of sim.
//   1. every lane takes the maximum key of its columns (c = lane, lane + 64, ...);
//   2. T = the k-th largest of the 64 lane maxima (bisection on the key bits, one ballot per bit): k distinct
//      elements are >= T, so the row's top k are among the elements >= T -- on continuous data about 1.1 k of them;
//   3. second pass over the row (L1/L2-resident): the elements >= T are compacted into 64 LDS slots with ballot
//      prefix counts, as (key << 32 | ~column) so that equal values order by the lower column;
//   4. one candidate per lane, a 21-step bitonic sort across the wave, lanes 0..k-1 write.
// More than 64 candidates (heavy ties: e.g. a constant row) take the insertion-list path of the first version.
// ------------------------------------------------------------------------------------------------
template <int KK>
__device__ __forceinline__ void row_topk_insert(const float* __restrict__ sr, int64_t C, int k, int lane, int64_t row,
                                                float* __restrict__ vals, int32_t* __restrict__ idx) {
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

template <int KK>
__global__ __launch_bounds__(256) void row_topk_kernel(const float* __restrict__ sim, int64_t ld, int64_t U,
                                                        int64_t C, int k, float* __restrict__ vals,
                                                        int32_t* __restrict__ idx) {
    __shared__ unsigned long long s_cand[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= U) return;   // whole wave; no workgroup barrier below
    const float* sr = sim + row * ld;
    if constexpr (KK == 1) {   // torch.max(sim, dim=1): one list head per lane and one wave arg-max is already minimal
        row_topk_insert<1>(sr, C, k, lane, row, vals, idx);
        return;
    }
    // 1. lane maxima.  Long rows (10 000 concepts: 40 KB) are read 16 bytes per lane: one dword per lane moves a quarter of
    //    the bytes per vector-memory instruction (0.28 ms for 9 216 rows of 10 000 against 0.10 vectorised).
    const bool vec = C >= 1024 && (ld % 4 == 0) && (((uintptr_t)sim) % 16 == 0);   // wave-uniform
    const int64_t C4 = vec ? (C & ~(int64_t)3) : 0;
    uint32_t lmax = 0u;     // below every valid key
    for (int64_t c = 4 * lane; c < C4; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(sr + c);
        const uint32_t k0 = mcd_f2key(v.x), k1 = mcd_f2key(v.y), k2 = mcd_f2key(v.z), k3 = mcd_f2key(v.w);
        const uint32_t m01 = k0 > k1 ? k0 : k1, m23 = k2 > k3 ? k2 : k3, m = m01 > m23 ? m01 : m23;
        lmax = m > lmax ? m : lmax;
    }
    for (int64_t c = C4 + lane; c < C; c += 64) {
        const uint32_t key = mcd_f2key(sr[c]);
        lmax = key > lmax ? key : lmax;
    }
    // 2. k-th largest lane maximum (k <= min(C, 16) <= number of lanes that hold a column)
    uint32_t T = 0u;
    for (int b = 31; b >= 0; --b) {
        const uint32_t cand = T | (1u << b);
        if ((int)__popcll(__ballot(lmax >= cand)) >= k) T = cand;
    }
    // 3. compact the elements >= T
    unsigned long long* cand_list = s_cand[wave];
    int base = 0;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int64_t c0 = 0; c0 < C4; c0 += 256) {           // vectorised part: the slot order inside the list does not matter
        const int64_t c = c0 + 4 * lane;                 // (the list is sorted afterwards; ties carry their column)
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool in = c < C4;
        if (in) v = *reinterpret_cast<const float4*>(sr + c);
        const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t key = in ? mcd_f2key(x[j]) : 0u;
            const bool pred = key >= T;
            const unsigned long long m = __ballot(pred);
            if (pred) {
                const int slot = base + __popcll(m & lt_mask);
                if (slot < 64) cand_list[slot] = ((unsigned long long)key << 32) | (uint32_t)(0xffffffffu - (uint32_t)(c + j));
            }
            base += __popcll(m);
        }
    }
    for (int64_t c0 = C4; c0 < C; c0 += 64) {
        const int64_t c = c0 + lane;
        const uint32_t key = (c < C) ? mcd_f2key(sr[c]) : 0u;
        const bool pred = key >= T;                       // T >= 1: lanes past the row never qualify
        const unsigned long long m = __ballot(pred);
        if (pred) {
            const int slot = base + __popcll(m & lt_mask);
            if (slot < 64) cand_list[slot] = ((unsigned long long)key << 32) | (uint32_t)(0xffffffffu - (uint32_t)c);
        }
        base += __popcll(m);
    }
    if (base > 64) {                                       // wave-uniform
        row_topk_insert<KK>(sr, C, k, lane, row, vals, idx);
        return;
    }
    // 4. one candidate per lane (LDS operations of one wave execute in order), bitonic sort, descending
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    unsigned long long e = (lane < base) ? cand_list[lane] : 0ull;
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const unsigned long long other = __shfl_xor(e, j, 64);
            const bool desc = (lane & k2) == 0;            // k2 == 64: every lane
            const bool lower = (lane & j) == 0;
            const bool keep_max = lower == desc;
            const unsigned long long mx = e > other ? e : other, mn = e > other ? other : e;
            e = keep_max ? mx : mn;
        }
    }
    if (lane < k) {
        vals[row * k + lane] = mcd_key2f((uint32_t)(e >> 32));
        idx[row * k + lane] = (int32_t)(0xffffffffu - (uint32_t)(e & 0xffffffffu));
    }
}

// K6 for rows of up to 1 024 concepts (763): a wave per row, the row in registers.  The wave-per-row kernel above walks a row
// twice with one dword load per lane and step, each waited for before the next is issued -- 2 x 12 memory round trips in a
// row per wave at 763 concepts, which is what its 0.019 ms were (0.19 of the HBM rate).  Here the row arrives as four
// 16-byte buffer loads per lane issued together (dwords past the row's end come back as 0 and get key 0, below every real
// key), and the bound, the compaction and the sort work on the 16 keys a lane holds.  Same order as above: NaN on top, then
// the value, then the lower column.
template <int KK>
__global__ __launch_bounds__(256) void row_topk_short_kernel(const float* __restrict__ sim, int64_t ld, int64_t U, int C, int k,
                                                              float* __restrict__ vals, int32_t* __restrict__ idx) {
    __shared__ unsigned long long s_cand[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // uniform: the row's descriptor stays in scalar registers
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= U) return;   // whole wave; no workgroup barrier below
    const float* sr = sim + row * ld;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sr, 0, C * 4, 0x00020000);
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, j * 1024, 0));
    uint32_t key[16];
    uint32_t lmax = 0u;     // below every valid key
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = 256 * (i >> 2) + 4 * lane + (i & 3);
        key[i] = c < C ? mcd_f2key(v[i >> 2][i & 3]) : 0u;
        lmax = key[i] > lmax ? key[i] : lmax;
    }
    // k-th largest lane maximum.  A lane holds 4 adjacent columns per load, so a row shorter than 4 k columns may have fewer
    // than k lanes with a maximum at all: the search then ends at T = 0, which the empty slots' key 0 would pass -- such a
    // row (and one with heavy ties, below) takes the per-lane insertion lists instead.
    uint32_t T = 0u;
    for (int b = 31; b >= 0; --b) {
        const uint32_t cand = T | (1u << b);
        if ((int)__popcll(__ballot(lmax >= cand)) >= k) T = cand;
    }
    if (T == 0u) {                                         // wave-uniform
        row_topk_insert<KK>(sr, C, k, lane, row, vals, idx);
        return;
    }
    // the keys >= T (T >= 1: slots past the row never qualify); the slot order does not matter, the list is sorted below
    unsigned long long* cand_list = s_cand[wave];
    int base = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const bool pred = key[i] >= T;
        const unsigned long long m = __ballot(pred);
        if (m) {
            if (pred) {
                const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)base));
                const uint32_t c = (uint32_t)(256 * (i >> 2) + 4 * lane + (i & 3));
                if (slot < 64u) cand_list[slot] = ((unsigned long long)key[i] << 32) | (0xffffffffu - c);
            }
            base += __popcll(m);
        }
    }
    if (base > 64) {                                       // wave-uniform: heavy ties at the bound
        row_topk_insert<KK>(sr, C, k, lane, row, vals, idx);
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (LDS operations of one wave execute in order)
    unsigned long long e = (lane < base) ? cand_list[lane] : 0ull;
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const unsigned long long other = __shfl_xor(e, j, 64);
            const bool desc = (lane & k2) == 0;
            const bool lower = (lane & j) == 0;
            const bool keep_max = lower == desc;
            const unsigned long long mx = e > other ? e : other, mn = e > other ? other : e;
            e = keep_max ? mx : mn;
        }
    }
    if (lane < k) {
        vals[row * k + lane] = mcd_key2f((uint32_t)(e >> 32));
        idx[row * k + lane] = (int32_t)(0xffffffffu - (uint32_t)(e & 0xffffffffu));
    }
}

template <int THREADS, int QUADS, int CAP>
void launch_topk_fast(const float* At, int64_t ld, int64_t N, int64_t U, int K, float* vals, int32_t* idx, int64_t ldo,
                      int* flag, int flag_stride, int vec_ok, hipStream_t st) {
    hipLaunchKernelGGL((neuron_topk_fast_kernel<THREADS, QUADS, CAP>), dim3((unsigned)U), dim3(THREADS), 0, st, At, ld,
                       N, K, vals, idx, ldo, flag, flag_stride, vec_ok);
}

// returns false when (N, K) is outside the register-resident kernels (everything then streams)
bool dispatch_topk_fast(const float* At, int64_t ld, int64_t N, int64_t U, int K, float* vals, int32_t* idx,
                        int64_t ldo, int* flag, int flag_stride, int vec_ok, hipStream_t st) {
#define MCD_TOPK_FAST(T, Q, CAP) launch_topk_fast<T, Q, CAP>(At, ld, N, U, K, vals, idx, ldo, flag, flag_stride, vec_ok, st)
    // beyond the register-resident classes (and for any N: 100 000 images run at 2.7 TB/s, the streaming kernel at 0.2)
    static const int force_twopass = mcd_dev_knob("MCD_TOPK_TWOPASS", 0);   // dev knob: rows longer than this
    if (K <= 128 && vec_ok && (N > 512 * 100 || (force_twopass > 0 && N > force_twopass)) && N < (1 << 30)) {
        hipLaunchKernelGGL((neuron_topk_twopass_kernel<256>), dim3((unsigned)U), dim3(256), 0, st, At, ld, N, K, vals, idx, ldo,
                           flag, flag_stride);
        return true;
    }
    static const char* force = mcd_dev_env("MCD_TOPK_CLASS");   // dev knob: "threads,quads" among the classes below
    if (force && K <= 128) {
        int t = 0, q = 0;
        if (sscanf(force, "%d,%d", &t, &q) == 2 && N <= (int64_t)t * q * 4) {
            if (t == 256 && q == 10) { MCD_TOPK_FAST(256, 10, 256); return true; }
            if (t == 256 && q == 12) { MCD_TOPK_FAST(256, 12, 256); return true; }
            if (t == 512 && q == 5) { MCD_TOPK_FAST(512, 5, 256); return true; }
            if (t == 512 && q == 6) { MCD_TOPK_FAST(512, 6, 256); return true; }
            if (t == 512 && q == 10) { MCD_TOPK_FAST(512, 10, 256); return true; }
            if (t == 512 && q == 13) { MCD_TOPK_FAST(512, 13, 256); return true; }
            if (t == 768 && q == 9) { MCD_TOPK_FAST(768, 9, 256); return true; }
            if (t == 768 && q == 7) { MCD_TOPK_FAST(768, 7, 256); return true; }
            if (t == 1024 && q == 3) { MCD_TOPK_FAST(1024, 3, 256); return true; }
            if (t == 1024 && q == 5) { MCD_TOPK_FAST(1024, 5, 256); return true; }
            if (t == 1024 && q == 7) { MCD_TOPK_FAST(1024, 7, 256); return true; }
            if (t == 1024 && q == 8) { MCD_TOPK_FAST(1024, 8, 256); return true; }
        }
    }
    if (K <= 128) {  // <= 256 survivors expected (about 1.1-1.3 K)
        if (N <= 256 * 4) MCD_TOPK_FAST(256, 1, 256);
        else if (N <= 256 * 8) MCD_TOPK_FAST(256, 2, 256);
        else if (N <= 256 * 16) MCD_TOPK_FAST(256, 4, 256);
        else if (N <= 256 * 24) MCD_TOPK_FAST(256, 6, 256);
        else if (N <= 512 * 16) MCD_TOPK_FAST(512, 4, 256);
        else if (N <= 256 * 40) MCD_TOPK_FAST(256, 10, 256);  // 8 workgroups of 4 waves per CU: 0.077 ms against 0.083 (512 x 5) at N = 10 000
        else if (N <= 512 * 24) MCD_TOPK_FAST(512, 6, 256);   // 512-thread workgroups: 4 (2) per CU overlap their load and
        else if (N <= 512 * 32) MCD_TOPK_FAST(512, 8, 256);   // selection phases; 1024-thread ones run one per CU
        else if (N <= 512 * 40) MCD_TOPK_FAST(512, 10, 256);  // 0.194 ms against 0.206 (512 x 13) at N = 20 000
        else if (N <= 512 * 52) MCD_TOPK_FAST(512, 13, 256);  // 3 workgroups per CU: 0.165 ms at N = 25 000 (768 x 9: 0.197, 1024 x 7: 0.34)
        else if (N <= 512 * 64) MCD_TOPK_FAST(512, 16, 256);
        else if (N <= 512 * 80) MCD_TOPK_FAST(512, 20, 256);
        else if (N <= 512 * 100) MCD_TOPK_FAST(512, 25, 256);
        else return false;
    } else if (K <= 448) {  // needs >= K thread maxima: 1024-thread blocks
        if (N <= 1024 * 4) MCD_TOPK_FAST(1024, 1, 1024);
        else if (N <= 1024 * 8) MCD_TOPK_FAST(1024, 2, 1024);
        else if (N <= 1024 * 12) MCD_TOPK_FAST(1024, 3, 1024);
        else if (N <= 1024 * 16) MCD_TOPK_FAST(1024, 4, 1024);
        else if (N <= 1024 * 32) MCD_TOPK_FAST(1024, 8, 1024);
        else return false;
    } else {
        return false;
    }
#undef MCD_TOPK_FAST
    return true;
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
static size_t topk_flag_bytes(int64_t U) { return ((size_t)U * sizeof(int) + 255) / 256 * 256; }
static bool topk_is_neuron_major(int64_t N, int64_t U, int64_t stride_n, int64_t stride_u) {
    return stride_n == 1 && (stride_u != 1 || U == 1 || N == 1);   // N == 1: rows of one element, either reading is the same memory
}

extern "C" size_t mcd_col_topk_workspace(int64_t N, int64_t U, int64_t stride_n, int64_t stride_u, int K) {
    (void)K;
    size_t b = topk_flag_bytes(U);  // one "needs the streaming path" word per neuron
    if (!topk_is_neuron_major(N, U, stride_n, stride_u)) b += (size_t)U * (size_t)topk_ws_ld(N) * sizeof(float);
    return b;
}

extern "C" int mcd_col_topk(const float* A, int64_t N, int64_t U, int64_t stride_n, int64_t stride_u, int K,
                            float* vals, int32_t* idx, int64_t ldo, void* ws, size_t ws_bytes, mcd_stream_t stream) {
    MCD_REQUIRE(A && (vals || idx), MCD_E_ARG, "mcd_col_topk: NULL pointer");
    MCD_REQUIRE(N > 0 && U >= 0, MCD_E_ARG, "mcd_col_topk: bad shape N=%lld U=%lld", (long long)N, (long long)U);
    MCD_REQUIRE(K >= 1 && K <= N, MCD_E_RANGE, "selected index k out of range (k=%d, N=%lld)", K, (long long)N);
    MCD_REQUIRE(ldo >= K, MCD_E_ARG, "mcd_col_topk: ldo < K");
    MCD_REQUIRE(N < 0x7fffffffLL, MCD_E_UNSUPPORTED, "mcd_col_topk: N too large");
    MCD_REQUIRE(K <= 4096, MCD_E_UNSUPPORTED, "mcd_col_topk: K=%d > 4096 not supported", K);
    if (U == 0) return MCD_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t need = mcd_col_topk_workspace(N, U, stride_n, stride_u, K);
    MCD_REQUIRE(ws && ws_bytes >= need, MCD_E_WORKSPACE, "mcd_col_topk: workspace %zu < %zu bytes", ws_bytes, need);
    MCD_REQUIRE(((uintptr_t)ws) % 16 == 0, MCD_E_ARG, "mcd_col_topk: workspace must be 16-byte aligned");
    int* flag = (int*)ws;
    const float* At;
    int64_t ld;
    if (topk_is_neuron_major(N, U, stride_n, stride_u)) {
        MCD_REQUIRE(stride_u >= N || U == 1, MCD_E_ARG, "mcd_col_topk: neuron-major stride_u < N");
        At = A;
        ld = stride_u;
    } else {
        MCD_REQUIRE(stride_u == 1 && stride_n >= U, MCD_E_ARG,
                    "mcd_col_topk: need image-major (stride_u==1) or neuron-major (stride_n==1) input");
        float* tbuf = (float*)((char*)ws + topk_flag_bytes(U));
        ld = topk_ws_ld(N);
        const int rc = mcd_transpose(A, stride_n, N, U, tbuf, ld, stream);
        if (rc) return rc;
        At = tbuf;
    }
    const int vec_ok = (ld % 4 == 0) && (((uintptr_t)At) % 16 == 0);
    if (K > 1024) {  // every neuron streams; the survivors (<= 4096) are bitonic-sorted in LDS
        hipLaunchKernelGGL((neuron_topk_stream_kernel<1024, 4096>), dim3((unsigned)U), dim3(1024), 0, st, At, ld, N, K, vals,
                           idx, ldo, (const int*)nullptr, 1);
        MCD_LAUNCH_CHECK("neuron_topk_stream_kernel");
        return MCD_OK;
    }
    const bool fast = dispatch_topk_fast(At, ld, N, U, K, vals, idx, ldo, flag, 1, vec_ok, st);
    MCD_LAUNCH_CHECK("neuron_topk_fast_kernel");
    const int* fl = fast ? flag : nullptr;  // nullptr: every neuron takes the streaming path
    if (K <= 128)
        hipLaunchKernelGGL((neuron_topk_stream_kernel<256, 128>), dim3((unsigned)U), dim3(256), 0, st, At, ld, N, K, vals,
                           idx, ldo, fl, 1);
    else if (K <= 256)
        hipLaunchKernelGGL((neuron_topk_stream_kernel<256, 256>), dim3((unsigned)U), dim3(256), 0, st, At, ld, N, K, vals,
                           idx, ldo, fl, 1);
    else
        hipLaunchKernelGGL((neuron_topk_stream_kernel<1024, 1024>), dim3((unsigned)U), dim3(1024), 0, st, At, ld, N, K, vals,
                           idx, ldo, fl, 1);
    MCD_LAUNCH_CHECK("neuron_topk_stream_kernel");
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
    // Long rows (the stress chain's 10 000 concepts): a row IS a neuron-major K3 problem (same order: ties to the lower index,
    // NaN on top), and K3's register-resident workgroup-per-row kernel reads it once where the wave-per-row kernel below reads
    // it twice: 0.107 against 0.175 ms for 9 216 rows of 10 000.  The "left to the streaming kernel" mark lives in the row's
    // first output index (no workspace in this entry point).  Short rows (763 concepts: 0.018 ms) stay with the wave kernel.
    static const int long_rows = mcd_dev_knob("MCD_ROW_TOPK_LONG", 4096);   // dev knob: threshold
    if (C >= long_rows && C < 0x7fffffffLL) {
        const int vec_ok = (ld % 4 == 0) && (((uintptr_t)sim) % 16 == 0);
        if (dispatch_topk_fast(sim, ld, C, U, k, vals, idx, k, idx, k, vec_ok, st)) {
            MCD_LAUNCH_CHECK("neuron_topk_fast_kernel");
            hipLaunchKernelGGL((neuron_topk_stream_kernel<256, 128>), dim3((unsigned)U), dim3(256), 0, st, sim, ld, C, k, vals, idx,
                               (int64_t)k, (const int*)idx, k);
            MCD_LAUNCH_CHECK("neuron_topk_stream_kernel");
            return MCD_OK;
        }
    }
    if (C <= 1024) {   // the row in registers (buffer loads: a row needs dword alignment only -- the [U, 763] result of K5 has none better)
        if (k <= 4)
            hipLaunchKernelGGL(row_topk_short_kernel<4>, grid, block, 0, st, sim, ld, U, (int)C, k, vals, idx);
        else if (k <= 10)
            hipLaunchKernelGGL(row_topk_short_kernel<10>, grid, block, 0, st, sim, ld, U, (int)C, k, vals, idx);
        else
            hipLaunchKernelGGL(row_topk_short_kernel<16>, grid, block, 0, st, sim, ld, U, (int)C, k, vals, idx);
        MCD_LAUNCH_CHECK("row_topk_short_kernel");
        return MCD_OK;
    }
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
