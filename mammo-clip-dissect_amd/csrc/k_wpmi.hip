// k_wpmi.hip -- the soft-WPMI / WPMI scoring kernels.
//   K4  mcd_wpmi_score      concept_vit/similarity.py:59-65 (soft_wpmi), :84-88 (wpmi)
//   K5  mcd_logsumexp_sub   concept_vit/similarity.py:70-72, :92-96
//
// Compiled with -ffp-contract=off: every fp32 operation of the reference's expression
//     log((1 + p*(g - 1)) + min_prob)
// rounds on its own, and the sum over the K selected images follows ATen's CPU
// torch.sum(dim=0) (SumKernel.cpp): columns c < split in the cascade order (16-row chunks, 4
// levels), columns c >= split in the row_sum order (4 row-interleaved partials, each cascaded).
//
// K4 mapping: one 64-lane wave per (neuron, 128-concept slab); lane l owns concepts 2l, 2l+1 of the
// slab, so one gathered row of S is read as 512 contiguous bytes per wave-instruction (8 B/lane)
// and a 16-row chunk (one cascade step) is in flight at once.  Image indices and p[j] are
// wave-uniform and come through scalar loads.  The few row_sum-order columns are handled by a
// second launch that packs several neurons per wave.
#include "mcd_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

// ATen multi_row_sum state for one output element, level_power = 4 (K < 2^19)
struct Cascade {
    float a0, a1, a2, a3;
    __device__ __forceinline__ void init() { a0 = a1 = a2 = a3 = 0.f; }
    // called after a complete 16-addend chunk has gone into a0; i = addends consumed so far
    __device__ __forceinline__ void flush(int i) {
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

// ---- accurate natural log (default) -----------------------------------------------------------------
// torch.log on CPU is MKL's high-accuracy vsLn, i.e. practically correctly rounded, so the closer this log
// is to correctly rounded the more sums land on the reference's bits.  Table method (scripts/gen_log_table.py):
//   x = 2^e * m, i = top 7 mantissa bits, inv_i = 8-bit reciprocal of the interval centre:
//   log x = V[e,i] + log1p(r),  r = m*inv_i - 1 = x*(inv_i 2^-e) - 1,  |r| < 2^-7, a multiple of 2^-31:
//   ONE fma yields r exactly (no mantissa extraction, no product residual).
//   V = e*ln2 - log(inv_i) is tabulated as a float pair for x in [2^-25, 2), indexed (with inv_i 2^-e) by the
//   top 16 bits of x; log1p(r) - r by a quartic; the pieces are summed as a double-float.
//   Emulated on the host (tests/test_log_table_cpu.py): 99.9999 % correctly rounded, never > 1 ulp off;
//   agreement with torch.log: 99.993 % (torch.log itself is 99.993 % correctly rounded).
// Arguments outside the table (zero, negative, denormal, >= 2, inf, nan) take libm's logf.
#include "log_table.inc"
constexpr unsigned MCD_LOG_BASE = (unsigned)(127 + MCD_LOG_E_MIN) * MCD_LOG_NI;
constexpr unsigned MCD_LOG_N = MCD_LOG_ROWS * MCD_LOG_NI;
// stride between the hi / lo / inv arrays in LDS: NOT a multiple of 64 dwords, so the compiler keeps three plain
// ds_read_b32 per component (a merged ds_read2st64 would return hi and lo of ONE component in adjacent registers
// and cost v_mov shuffles to form the packed (component 0, component 1) operands)
constexpr unsigned MCD_LOG_STRIDE = MCD_LOG_N + 1;

typedef __attribute__((address_space(3))) const float lds_cfloat;

struct LogTab {
    unsigned base;  // LDS byte address of hi[0], minus 4 * MCD_LOG_BASE (so that 4 * (top 16 bits of x) indexes directly)
};

// workgroup-wide copy of the table into LDS (39 KB, as three float arrays so that the two components of a
// packed operand can be read into adjacent registers); ends with a barrier
__device__ __forceinline__ LogTab load_log_tables(float* s_t) {
    for (unsigned t = threadIdx.x; t < MCD_LOG_N; t += blockDim.x) {
        s_t[t] = __uint_as_float(g_log_rec_bits[t][0]);
        s_t[MCD_LOG_STRIDE + t] = __uint_as_float(g_log_rec_bits[t][1]);
        s_t[2 * MCD_LOG_STRIDE + t] = __uint_as_float(g_log_rec_bits[t][2]);
    }
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)s_t;
    return LogTab{lds0 - 4u * MCD_LOG_BASE};
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// LDS address of w's table entry in ONE instruction: the top 16 bits of w (its high half-register, selected with
// op_sel) are the raw table index; v_mad_u32_u16 multiplies them by 4 and adds the pre-offset base.
__device__ __forceinline__ lds_cfloat* log_entry(float w, const LogTab& T) {
    unsigned a;
#if defined(MCD_K4_ABL_NOCONFLICT)
    // TIMING EXPERIMENT ONLY (scripts/k4_lds_ablation.sh; wrong results): every lane reads the entry of its own lane id --
    // 64 consecutive dwords, no two lanes of a ds_read_b32 on one bank -- through the same single instruction, still
    // data-dependent on w: what K4 would take if the table lookups never conflicted.
    const unsigned mine = T.base + 4u * (MCD_LOG_BASE + 1024u + (threadIdx.x & 63u));
    asm("v_mad_u32_u16 %0, %1, 0, %2 op_sel:[1,0,0,0]" : "=v"(a) : "v"(w), "v"(mine));
#else
    asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(a) : "v"(w), "s"(T.base));
#endif
    return (lds_cfloat*)(uintptr_t)a;
}
__device__ __forceinline__ bool log_in_table(unsigned bits_min, unsigned bits_max) {
    // 2^MCD_LOG_E_MIN <= w < 2 for every w of the group (as unsigned bit patterns: negatives, NaN, inf, 0 fail)
    return bits_min >= ((unsigned)(127 + MCD_LOG_E_MIN) << 23) && bits_max < 0x40000000u;
}
// table log of a pair whose arguments are known to be in the table's range
__device__ __forceinline__ v2f log_tab2(v2f w, const LogTab& T) {
    lds_cfloat* e0 = log_entry(w.x, T);
    lds_cfloat* e1 = log_entry(w.y, T);
    const v2f vh = v2f{e0[0], e1[0]}, vl = v2f{e0[MCD_LOG_STRIDE], e1[MCD_LOG_STRIDE]},
              inv = v2f{e0[2 * MCD_LOG_STRIDE], e1[2 * MCD_LOG_STRIDE]};
    const v2f r = __builtin_elementwise_fma(w, inv, (v2f)(-1.0f));      // exact
    v2f q = __builtin_elementwise_fma(r, (v2f)(-0.25f), (v2f)(0x1.555556p-2f));
    q = __builtin_elementwise_fma(r, q, (v2f)(-0.5f));
    v2f low = __builtin_elementwise_fma(r * r, q, vl);                  // (log1p(r) - r) + V.lo
    const v2f H = vh + r;
    const v2f err = r - (H - vh);                                       // fast two-sum: |vh| >= |r| or vh == 0
    low = low + err;
    return H + low;
}
__device__ __forceinline__ v2f log_acc2(v2f w, const LogTab& T) {
    const unsigned b0 = __float_as_uint(w.x), b1 = __float_as_uint(w.y);
    if (__builtin_expect(!log_in_table(min(b0, b1), max(b0, b1)), 0)) return v2f{logf(w.x), logf(w.y)};
    return log_tab2(w, T);
}
__device__ __forceinline__ float log_acc(float w, const LogTab& T) { return log_acc2(v2f{w, w}, T).x; }
// three pairs (one gathered row of the sliced kernel) behind ONE range check; TRUSTED: the caller guarantees
// 2^MCD_LOG_E_MIN <= w < 2 (S is a softmax output and min_prob >= 2^MCD_LOG_E_MIN), no check at all
template <bool TRUSTED>
__device__ __forceinline__ void log_acc2x3(v2f w0, v2f w1, v2f w2, const LogTab& T, v2f& t0, v2f& t1, v2f& t2) {
    if constexpr (TRUSTED) {
        t0 = log_tab2(w0, T);
        t1 = log_tab2(w1, T);
        t2 = log_tab2(w2, T);
        return;
    }
    const unsigned a0 = __float_as_uint(w0.x), a1 = __float_as_uint(w0.y), b0 = __float_as_uint(w1.x),
                   b1 = __float_as_uint(w1.y), c0 = __float_as_uint(w2.x), c1 = __float_as_uint(w2.y);
    const unsigned mn = min(min(min(a0, a1), min(b0, b1)), min(c0, c1));
    const unsigned mx = max(max(max(a0, a1), max(b0, b1)), max(c0, c1));
    if (__builtin_expect(!log_in_table(mn, mx), 0)) {
        t0 = v2f{logf(w0.x), logf(w0.y)};
        t1 = v2f{logf(w1.x), logf(w1.y)};
        t2 = v2f{logf(w2.x), logf(w2.y)};
        return;
    }
    t0 = log_tab2(w0, T);
    t1 = log_tab2(w1, T);
    t2 = log_tab2(w2, T);
}

template <bool SOFT, bool SAFE_LOG>
__device__ __forceinline__ float wpmi_term(float g, float pj, float min_prob, const LogTab& T) {
    float w;
    if constexpr (SOFT) {
        const float d = g - 1.0f;
        const float y = pj * d;
        const float z = 1.0f + y;
        w = z + min_prob;
    } else {
        w = g + min_prob;
    }
    if constexpr (SAFE_LOG) return log_acc(w, T);  // SAFE_LOG == accurate (default); false == fast v_log_f32 path
    return mcd_log_pos(w);
}

// ---- K4 main: cascade-order columns [0, ncols), VEC concepts per lane ---------------------------
template <int VEC, bool SOFT, bool SAFE_LOG>
__global__ __launch_bounds__(256) void wpmi_main_kernel(const float* __restrict__ S, int64_t ldS,
                                                         const int32_t* __restrict__ idx, int64_t ldidx, int64_t U,
                                                         int K, const float* __restrict__ p, float min_prob,
                                                         int ncols, int nslab, float* __restrict__ out, int64_t ldo) {
    __shared__ float s_logtab[SAFE_LOG ? 3 * MCD_LOG_STRIDE : 1];
    LogTab T{0u};
    if constexpr (SAFE_LOG) T = load_log_tables(s_logtab);  // before any early exit: it ends in a barrier
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item >= U * nslab) return;
    const int64_t u = item / nslab;
    const int slab = (int)(item - u * nslab);
    const int c0 = (slab * 64 + lane) * VEC;
    if (c0 >= ncols) return;
    const int32_t* my_idx = idx + u * ldidx;
    const float* Sc = S + c0;

    Cascade acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v].init();

    int i = 0;
    for (; i + 16 <= K; i += 16) {
        float g[16][VEC];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = my_idx[i + r];
            const float* src = Sc + row * ldS;
            if constexpr (VEC == 2) {
                const float2 t = *reinterpret_cast<const float2*>(src);
                g[r][0] = t.x;
                g[r][1] = t.y;
            } else {
                g[r][0] = *src;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pj = SOFT ? p[i + r] : 0.f;
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v].a0 += wpmi_term<SOFT, SAFE_LOG>(g[r][v], pj, min_prob, T);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v].flush(i + 16);
    }
    for (; i < K; ++i) {
        const int64_t row = my_idx[i];
        const float pj = SOFT ? p[i] : 0.f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v].a0 += wpmi_term<SOFT, SAFE_LOG>(Sc[row * ldS + v], pj, min_prob, T);
    }
    float* o = out + u * ldo + c0;
#pragma unroll
    for (int v = 0; v < VEC; ++v)
        if (c0 + v < ncols) o[v] = acc[v].total();
}

// ---- K4 sliced: the same sums, laid out for the 8 XCD-private L2s and for packed fp32 math -----------
// Requires ldS % 96 == 0 (C = 763 -> ldS = 768 = 8 slices of 96 concepts).  blockIdx.x % 8 selects the slice of a
// round, and workgroups are dealt round-robin over the XCDs, so (for 8 slices) every XCD keeps gathering from
// the SAME 96-column slice of S: N x 384 B (3.8 MB at N = 10 000) instead of all of S (30.7 MB) competes for
// its 4 MiB L2.  Placement is a speed assumption only; results do not depend on it.
// A wave covers 4 neurons x 16 lanes; lane q of a neuron owns the concept PAIRS (32k + 2q, 32k + 2q + 1),
// k = 0..2, so each of its three 8-byte loads per gathered row is part of one full 128-byte line, and every
// arithmetic step runs on a pair as one packed instruction (v_pk_add/mul/fma_f32: gfx950's fp32 VALU rate
// is 16 lanes/clk/SIMD unpacked, twice that packed -- the kernel is VALU-bound, not bandwidth-bound).
struct Cascade2 {
    v2f a0, a1, a2, a3;
    __device__ __forceinline__ void init() { a0 = a1 = a2 = a3 = (v2f)(0.f); }
    __device__ __forceinline__ void flush(int i) {
        a1 += a0;
        a0 = (v2f)(0.f);
        if ((i & 0xF0) != 0) return;
        a2 += a1;
        a1 = (v2f)(0.f);
        if ((i & 0xF00) != 0) return;
        a3 += a2;
        a2 = (v2f)(0.f);
    }
    __device__ __forceinline__ v2f total() const { return ((a0 + a1) + a2) + a3; }
};

template <bool SOFT>
__device__ __forceinline__ v2f wpmi_arg2(v2f g, float pj, float min_prob) {
    if constexpr (SOFT) {
        const v2f d = g - (v2f)(1.0f);
        const v2f y = (v2f)(pj) * d;
        const v2f z = (v2f)(1.0f) + y;
        return z + (v2f)(min_prob);
    } else {
        return g + (v2f)(min_prob);
    }
}

template <bool SOFT, bool SAFE_LOG>
__device__ __forceinline__ v2f wpmi_term2(v2f g, float pj, float min_prob, const LogTab& T) {
    v2f w;
    if constexpr (SOFT) {
        const v2f d = g - (v2f)(1.0f);
        const v2f y = (v2f)(pj) * d;
        const v2f z = (v2f)(1.0f) + y;
        w = z + (v2f)(min_prob);
    } else {
        w = g + (v2f)(min_prob);
    }
    if constexpr (SAFE_LOG) {
        return log_acc2(w, T);
    } else {
        v2f r;
        r.x = __builtin_amdgcn_logf(w.x);  // log2, 1 ulp (v_log_f32)
        r.y = __builtin_amdgcn_logf(w.y);
        const v2f c = (v2f)(0x1.62e42ep-1f), cc = (v2f)(0x1.efa39ep-25f);
        const v2f ph = r * c;
        v2f pl = __builtin_elementwise_fma(r, c, -ph);  // exact product residual (deliberate fma)
        pl = __builtin_elementwise_fma(r, cc, pl);
        return ph + pl;
    }
}

// Two-level ATen cascade state for a concept pair (K < 256: levels 2 and 3 of multi_row_sum stay zero,
// and adding those zeros at the end changes nothing).
struct Pair2 {
    v2f a0, a1;
    __device__ __forceinline__ void init() { a0 = a1 = (v2f)(0.f); }
    __device__ __forceinline__ void flush() { a1 += a0; a0 = (v2f)(0.f); }
    __device__ __forceinline__ v2f total() const { return a0 + a1; }
};

template <bool SOFT, bool SAFE_LOG, bool TRUSTED>
__device__ __forceinline__ void wpmi_term2x3(v2f g0, v2f g1, v2f g2, float pj, float min_prob, const LogTab& T,
                                             v2f& t0, v2f& t1, v2f& t2) {
    if constexpr (SAFE_LOG) {
        log_acc2x3<TRUSTED>(wpmi_arg2<SOFT>(g0, pj, min_prob), wpmi_arg2<SOFT>(g1, pj, min_prob),
                            wpmi_arg2<SOFT>(g2, pj, min_prob), T, t0, t1, t2);
    } else {
        t0 = wpmi_term2<SOFT, false>(g0, pj, min_prob, T);
        t1 = wpmi_term2<SOFT, false>(g1, pj, min_prob, T);
        t2 = wpmi_term2<SOFT, false>(g2, pj, min_prob, T);
    }
}

// Lane layout inside a 96-concept slice (16 lanes per neuron, 4 neurons per wave):
//   lane q loads ONE 16-byte quad  (concepts 4q .. 4q+3 of the slice: 256 contiguous bytes per neuron-row)
//            and ONE  8-byte pair  (concepts 64+2q, 64+2q+1:          128 contiguous bytes per neuron-row),
// i.e. three concept pairs per lane from 2 vector-memory instructions (the L1 address/data path -- TA/TD --
// is the busiest unit of this kernel; dwordx4 moves twice the bytes per TA cycle of dwordx2).
// RS: the slice's last 32-column group (the dwordx2 pair) lies at/after `split` and sums in ATen's row_sum
// order (4 row-interleaved partials).  split is a multiple of 32 and C - split < 32; the host only takes
// this kernel when the row_sum group, if any, is that last group of its slice (C = 763: split = 736 =
// 7*96 + 64), so the choice is workgroup-uniform.
template <bool SOFT, bool SAFE_LOG, bool OFF32, bool RS, bool TRUSTED>
__device__ __forceinline__ void wpmi_slice_body(const float* __restrict__ S, int64_t ldS,
                                                const int32_t* __restrict__ my_idx, int K,
                                                const float* __restrict__ p, float min_prob, int cs, int q,
                                                int ncols, float* __restrict__ o, bool live, const LogTab& T) {
    // OFF32 (S smaller than 4 GiB, rows < 2^24, row pitch < 2^24 B): the row offset is one 24-bit multiply
    // and the load takes a uniform base + 32-bit lane offset; otherwise full 64-bit addressing.
    const char* Sb = reinterpret_cast<const char*>(S);
    const uint32_t pitch = (uint32_t)(ldS * 4);
    const uint32_t off4 = (uint32_t)(cs + 4 * q) * 4u, off2 = (uint32_t)(cs + 64 + 2 * q) * 4u;
    auto load_row = [&](int32_t row, v2f& ga, v2f& gb, v2f& gc) {
        const v4f* p4;
        const v2f* p2;
        if constexpr (OFF32) {
            const uint32_t ro = __umul24((uint32_t)row, pitch);
            p4 = reinterpret_cast<const v4f*>(Sb + (size_t)(ro + off4));
            p2 = reinterpret_cast<const v2f*>(Sb + (size_t)(ro + off2));
        } else {
            const float* r = S + (int64_t)row * ldS;
            p4 = reinterpret_cast<const v4f*>(r + cs + 4 * q);
            p2 = reinterpret_cast<const v2f*>(r + cs + 64 + 2 * q);
        }
        const v4f t = *p4;
        ga = t.xy;
        gb = t.zw;
        gc = *p2;
    };
    constexpr int RB = 8;  // rows in flight per batch
    Pair2 acc[3];
    Pair2 part[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k].init();
#pragma unroll
    for (int m = 0; m < 4; ++m) part[m].init();

    // one 16-row chunk (one level-0 cascade step of columns in the cascade order)
    auto chunk16 = [&](int i) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 16 / RB; ++h) {  // batches of RB rows in flight
            v2f g[RB][3];
#pragma unroll
            for (int r = 0; r < RB; ++r) load_row(my_idx[i + RB * h + r], g[r][0], g[r][1], g[r][2]);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float pj = SOFT ? p[i + RB * h + r] : 0.f;
                v2f ta, tb, t;
                wpmi_term2x3<SOFT, SAFE_LOG, TRUSTED>(g[r][0], g[r][1], g[r][2], pj, min_prob, T, ta, tb, t);
                acc[0].a0 += ta;
                acc[1].a0 += tb;
                if (RS) part[r & 3].a0 += t;   // (i + RB*h + r) & 3 == r & 3 (RB is a multiple of 4)
                else acc[2].a0 += t;
                // no range-check branch separates the rows here: a fence after every second row keeps the scheduler from
                // interleaving the table look-ups of all RB rows (measured: every row 0.293 ms, every second row 0.280,
                // none 0.284)
                if (TRUSTED && (r & 1)) __builtin_amdgcn_sched_barrier(0);
            }
        }
        acc[0].flush();
        acc[1].flush();
        if (!RS) acc[2].flush();
    };
    int i = 0;
    if (RS) {  // row_sum order: each partial has consumed 16 of its own rows after 64 rows
        for (; i + 64 <= K; i += 64) {
#pragma unroll 1
            for (int c = 0; c < 64; c += 16) chunk16(i + c);
#pragma unroll
            for (int m = 0; m < 4; ++m) part[m].flush();
        }
    }
#pragma unroll 1
    for (; i + 16 <= K; i += 16) chunk16(i);
    for (; i < K; i += 4) {  // remainder: K % 16 rows, a multiple of 4 (host-checked), 4 rows per pass
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v2f ga, gb, gc;
            load_row(my_idx[i + r], ga, gb, gc);
            const float pj = SOFT ? p[i + r] : 0.f;
            v2f ta, tb, t;
            wpmi_term2x3<SOFT, SAFE_LOG, TRUSTED>(ga, gb, gc, pj, min_prob, T, ta, tb, t);
            acc[0].a0 += ta;
            acc[1].a0 += tb;
            if (RS) part[r].a0 += t;   // (i + r) & 3 == r: i is a multiple of 4
            else acc[2].a0 += t;
        }
    }
    if (!live) return;
    v2f t2;
    if (RS) {
        t2 = part[0].total();
        t2 += part[1].total();
        t2 += part[2].total();
        t2 += part[3].total();
    } else {
        t2 = acc[2].total();
    }
    const v2f t0 = acc[0].total(), t1 = acc[1].total();
    const int ca = cs + 4 * q, cc = cs + 64 + 2 * q;
    if (ca + 0 < ncols) o[ca + 0] = t0.x;
    if (ca + 1 < ncols) o[ca + 1] = t0.y;
    if (ca + 2 < ncols) o[ca + 2] = t1.x;
    if (ca + 3 < ncols) o[ca + 3] = t1.y;
    if (cc + 0 < ncols) o[cc + 0] = t2.x;
    if (cc + 1 < ncols) o[cc + 1] = t2.y;
}

template <bool SOFT, bool SAFE_LOG, bool OFF32, bool TRUSTED>
__global__ __launch_bounds__(256, (SAFE_LOG && !TRUSTED) ? 3 : 4) void wpmi_slice_kernel(const float* __restrict__ S, int64_t ldS,
                                                          const int32_t* __restrict__ idx, int64_t ldidx, int64_t U,
                                                          int K, const float* __restrict__ p, float min_prob,
                                                          int ncols, int split, int n_slices, int wg_per_slice,
                                                          float* __restrict__ out, int64_t ldo) {
    __shared__ float s_logtab[SAFE_LOG ? 3 * MCD_LOG_STRIDE : 1];
    LogTab T{0u};
    if constexpr (SAFE_LOG) T = load_log_tables(s_logtab);  // before any early exit: it ends in a barrier
    const int lane = threadIdx.x & 63;
    // Workgroup -> (slice, neuron-group lane).  Slices go out in ROUNDS of 8 (one per XCD: slice % 8 == blockIdx % 8),
    // all workgroups of a round before the next round's, so only ~8-16 slices are being gathered from at any time:
    // at C = 10 000 (104 slices, S = 1 GB) the active slices (9.6 MB each at 25 000 images) stay in the Infinity
    // Cache instead of every neuron's 100 gathers per slice going to HBM.  With 8 slices this is the plain mapping.
    const int per_round = 8 * wg_per_slice;
    const int round = blockIdx.x / per_round;
    const int within = blockIdx.x - round * per_round;
    const int slice = round * 8 + (within & 7);
    if (slice >= n_slices) return;                        // last round of a slice count that is not a multiple of 8
    const int cs = slice * 96;
    // workgroup-uniform: this slice's last 32-column group is the row_sum group
    const bool rs = split < ncols && split == cs + 64;
    const int64_t n_groups = (U + 15) / 16;               // 16 neurons per workgroup pass
    for (int64_t ng = within >> 3; ng < n_groups; ng += wg_per_slice) {  // persistent: the tables load once
        const int64_t u_raw = (ng * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
        const bool live = u_raw < U;
        const int64_t u = live ? u_raw : U - 1;        // keep the wave convergent; dead lanes redo the last neuron
        const int32_t* my_idx = idx + u * ldidx;
        float* o = out + u * ldo;
        if (rs)
            wpmi_slice_body<SOFT, SAFE_LOG, OFF32, true, TRUSTED>(S, ldS, my_idx, K, p, min_prob, cs, lane & 15, ncols, o, live, T);
        else
            wpmi_slice_body<SOFT, SAFE_LOG, OFF32, false, TRUSTED>(S, ldS, my_idx, K, p, min_prob, cs, lane & 15, ncols, o, live, T);
    }
}

// ---- K4 tail: row_sum-order columns [c_lo, c_hi), GW lanes per neuron, 64/GW neurons per wave -----
template <bool SOFT, bool SAFE_LOG>
__global__ __launch_bounds__(256) void wpmi_tail_kernel(const float* __restrict__ S, int64_t ldS,
                                                         const int32_t* __restrict__ idx, int64_t ldidx, int64_t U,
                                                         int K, const float* __restrict__ p, float min_prob, int c_lo,
                                                         int c_hi, int gw_log2, float* __restrict__ out,
                                                         int64_t ldo) {
    __shared__ float s_logtab[SAFE_LOG ? 3 * MCD_LOG_STRIDE : 1];
    LogTab T{0u};
    if constexpr (SAFE_LOG) T = load_log_tables(s_logtab);  // before any early exit: it ends in a barrier
    const int lane = threadIdx.x & 63;
    const int gw = 1 << gw_log2;
    const int per_wave = 64 >> gw_log2;
    const int64_t u = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * per_wave + (lane >> gw_log2);
    const int c = c_lo + (lane & (gw - 1));
    if (u >= U || c >= c_hi) return;
    const int32_t* my_idx = idx + u * ldidx;
    const float* Sc = S + c;

    // row_sum: partial k takes rows k, k+4, ... (q = K/4 of them), each partial cascaded on its own
    const int q = K >> 2;
    Cascade part[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) part[k].init();
    int m = 0;  // own-rows consumed per partial
    for (; m + 16 <= q; m += 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = 4 * (m + r) + k;
                const float g = Sc[(int64_t)my_idx[j] * ldS];
                part[k].a0 += wpmi_term<SOFT, SAFE_LOG>(g, SOFT ? p[j] : 0.f, min_prob, T);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) part[k].flush(m + 16);
    }
    for (; m < q; ++m) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = 4 * m + k;
            const float g = Sc[(int64_t)my_idx[j] * ldS];
            part[k].a0 += wpmi_term<SOFT, SAFE_LOG>(g, SOFT ? p[j] : 0.f, min_prob, T);
        }
    }
    float tot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tot[k] = part[k].total();
    for (int j = 4 * q; j < K; ++j) {
        const float g = Sc[(int64_t)my_idx[j] * ldS];
        tot[0] += wpmi_term<SOFT, SAFE_LOG>(g, SOFT ? p[j] : 0.f, min_prob, T);
    }
    tot[0] += tot[1];
    tot[0] += tot[2];
    tot[0] += tot[3];
    out[u * ldo + c] = tot[0];
}

// ---- K4s: the stress chain's K4 -- bf16 similarity matrix, per-row reciprocal, v_log_f32 (no parity claim) ---------
// Input: E = bf16(exp(a (P - 1))) and rinv[n] = 1 / sum_c exp(a (P[n,c] - 1)) from K1s, i.e. S[n,c] = E[n,c] * rinv[n]
// without S ever being written.  Per gathered row the argument of the log is
//     soft:  w = (1 - p_j + min_prob) + (p_j * rinv[n]) * E[n,c]      (= 1 + p_j (S - 1) + min_prob, regrouped)
//     hard:  w = min_prob + rinv[n] * E[n,c]
// one fma per concept; the sum runs in the log2 domain (v_log_f32, 1 ulp) and is scaled by ln 2 once at the end.
// Layout: slices of 128 concepts (256 bytes of a bf16 row = two full lines); wave = 4 neurons x 16 lanes, a lane
// loads ONE 16-byte piece (8 concepts) per gathered row -- half the bytes and half the vector-memory instructions of the
// fp32 kernel per concept.  Slices go out in rounds of 8 (slice % 8 == blockIdx % 8 -> one slice per XCD L2 at a
// time), as in wpmi_slice_kernel.  (64-concept slices, whose 3.2 MB per slice would fit an XCD's L2, measured slower twice:
// 1.93 against 1.80 ms, 1.84 against 1.68.)
//
// What the kernel runs on is the L1 address/data path and the vector ALU together (TA busy 86 %, VALU ~80 %: profiles/r03_k4s_pmc.txt;
// the gather pattern alone, without arithmetic, runs at 18 TB/s = 0.98 ms, profiles/r03_gather_path.txt), so everything
// that is not the gather itself is kept off it:
//  * what a gathered row needs besides its bytes -- the row index and the scale p_j rinv[row] -- does not depend on the slice:
//    wpmi_meta_kernel writes it once per call as meta[u][j] = {row, p_j rinv[row]} (8 U K bytes of workspace), and per batch of 16
//    rows lane q of a neuron loads ONE 8-byte entry -- row i + q: a 128-byte line per neuron -- plus p[i + q] for the constant
//    c_j; the 16 lanes hand the values to each other with DPP row broadcasts (v_mov_b32 row_newbcast:r, VALU only).
//    Round 2 loaded the index and gathered rinv[row] in every slice (a dependent 4-byte gather, 64 different lines per wave
//    and batch against the 128 lines of the batch's row pieces: a third of the kernel's L2 requests, 1.59 -> 1.38 ms without
//    them, profiles/r03_k4s_pmc.txt); round 1 had every lane load all 16 indices and gather rinv per row piece.
//  * 16 rows (16 KB per wave) are in flight per batch, at 4 waves per SIMD.
//  * GROUP = 4: log2(x0) + log2(x1) + log2(x2) + log2(x3) is ONE v_log_f32 of the product: three packed multiplies per four
//    rows replace three of four half-rate transcendentals, and the product's rounding (3 x 2^-24 relative) is below
//    v_log_f32's own 1 ulp at |log2| ~ 8.  Every argument is >= min_prob (p <= 1), so the host takes GROUP = 4 only for
//    min_prob >= 2^-30: products stay normal.  Rows that pad a batch are neutral: scale 0, constant 1, log2(1) = 0.
//    (Products of EIGHT for soft WPMI -- safe where every 1 - p_j + min_prob >= 2^-15, decided on the device -- measured no faster:
//    1.395 against 1.379 ms; the logs are not what the kernel waits for.)
// meta[u][j] = {image row idx[u][j], p_j * rinv[row]} (hard WPMI: rinv[row]): what K4s needs per gathered row besides the row
__global__ __launch_bounds__(256) void wpmi_meta_kernel(const int32_t* __restrict__ idx, int64_t ldidx, int64_t U, int K,
                                                         const float* __restrict__ rinv, const float* __restrict__ p, int soft,
                                                         int2* __restrict__ meta) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= U * K) return;
    const int64_t u = t / K;
    const int j = (int)(t - u * K);
    const int32_t row = idx[u * ldidx + j];
    const float sc = rinv[row] * (soft ? p[j] : 1.0f);
    meta[t] = make_int2(row, __float_as_int(sc));
}

template <int R>
__device__ __forceinline__ int bcast16(int v) {   // lane R of every 16-lane row to all lanes of that row
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + R, 0xf, 0xf, false);
}
template <int R>
__device__ __forceinline__ float bcast16(float v) { return __int_as_float(bcast16<R>(__float_as_int(v))); }
template <int R, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (R < N) {
        f(std::integral_constant<int, R>{});
        static_for<R + 1, N>(f);
    }
}

// (The gathers as `buffer_load ... lds` into a per-wave LDS ring + ds_read_b128 were built and measured in round 3: bit-identical and
// no faster -- the gather path delivers the same 18 TB/s to registers and to LDS, scripts/micro/gather_path.hip,
// profiles/r03_gather_path.txt -- so the registers stay.)
template <bool SOFT, int GROUP, bool OFF32>
__global__ __launch_bounds__(256) void wpmi_bf16_kernel(const uint16_t* __restrict__ E, int64_t ldE,
                                                         const int2* __restrict__ meta, int64_t U, int K,
                                                         const float* __restrict__ p, float min_prob, int ncols, int n_slices,
                                                         int groups, float* __restrict__ out, int64_t ldo) {
    static_assert(GROUP == 1 || GROUP == 4, "rows per log");
    const int lane = threadIdx.x & 63;
    const int per_round = 8 * groups;
    const int round = blockIdx.x / per_round;
    const int within = blockIdx.x - round * per_round;
    const int slice = round * 8 + (within & 7);
    if (slice >= n_slices) return;
    const int q = lane & 15;
    const int c0 = slice * 128 + 8 * q;
    const int64_t u_raw = ((int64_t)(within >> 3) * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = u_raw < U;
    const int64_t u = live ? u_raw : U - 1;            // keep the wave convergent; dead lanes redo the last neuron
    const char* Eb = reinterpret_cast<const char*>(E);
    const uint32_t pitch = (uint32_t)(ldE * 2), coloff = (uint32_t)c0 * 2u;   // OFF32: E < 4 GiB, rows and pitch < 2^24
    v2f a0[4], a1[4];                                  // concept pairs (2k, 2k+1): packed fma / mul / add (v_pk_*_f32)
#pragma unroll
    for (int k = 0; k < 4; ++k) a0[k] = a1[k] = (v2f)(0.f);

    // lane q's row of a batch comes from the META array wpmi_meta_kernel wrote ({image row, p_j * rinv[row]} per (neuron, j):
    // it does not depend on the slice), 8 bytes per lane = one 128-byte line per neuron and batch, fetched TWO batches ahead
    // (unconditionally, from a clamped position: a branch around the loads would cost the counted waits below their exactness).
    // (Before: index -> rinv[index] as a dependent 4-byte gather per lane, 64 different lines per wave and batch -- half as many
    // L2 requests as the 128 lines of the batch's row pieces themselves, in each of the 79 slices.)
    const int2* my_meta = meta + u * K;
    struct Meta { int32_t row; float sc, pj; };
    auto meta_load = [&](int i) __attribute__((always_inline)) -> Meta {
        const int j = i + q < K ? i + q : K - 1;
        const int2 m = my_meta[j];
        return Meta{m.x, __int_as_float(m.y), SOFT ? p[j] : 1.0f};
    };
    // what the batch at row i needs from lane q: its row's byte offset, scale and constant (rows past K: neutral -- any valid
    // address, scale 0, constant 1)
    auto derive = [&](const Meta& m, int i, uint32_t& off, float& sc, float& cj) __attribute__((always_inline)) {
        const bool mine = i + q < K;
        const uint32_t row = mine ? (uint32_t)m.row : 0u;
        off = OFF32 ? __umul24(row, pitch) : row;
        sc = mine ? m.sc : 0.f;
        cj = mine ? (SOFT ? (1.0f - m.pj) + min_prob : min_prob) : 1.0f;
    };
    uint4 g[16];
    auto load4 = [&](auto r4c, uint32_t off_b) __attribute__((always_inline)) {
        constexpr int r4 = decltype(r4c)::value;
        static_for<0, 4>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = 4 * r4 + decltype(rc)::value;
            const uint32_t o = (uint32_t)bcast16<r>((int)off_b);
            if constexpr (OFF32) g[r] = *reinterpret_cast<const uint4*>(Eb + (size_t)(o + coloff));
            else g[r] = *reinterpret_cast<const uint4*>(Eb + ((int64_t)o * ldE * 2 + coloff));
        });
    };
    auto term4 = [&](auto r4c, float s_q, float c_q) __attribute__((always_inline)) {
        constexpr int r4 = decltype(r4c)::value;
        v2f pr[4];
        static_for<0, 4>([&](auto rc) __attribute__((always_inline)) {
            constexpr int rr = decltype(rc)::value, r = 4 * r4 + rr;
            const float s = bcast16<r>(s_q), cj = bcast16<r>(c_q);
            const unsigned w[4] = {g[r].x, g[r].y, g[r].z, g[r].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v2f e = {__uint_as_float(w[k] << 16), __uint_as_float(w[k] & 0xffff0000u)};
                const v2f x = __builtin_elementwise_fma(e, (v2f)(s), (v2f)(cj));
                if constexpr (GROUP == 1) {
                    a0[k] += v2f{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)};
                } else {
#if defined(MCD_K4S_ABL) && (MCD_K4S_ABL & 2)
                    pr[k] = rr == 0 ? x : pr[k] + x;
#else
                    pr[k] = rr == 0 ? x : pr[k] * x;
#endif
                }
            }
        });
        if constexpr (GROUP == 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#if defined(MCD_K4S_ABL) && (MCD_K4S_ABL & 2)
                a0[k] += pr[k];
#else
                a0[k] += v2f{__builtin_amdgcn_logf(pr[k].x), __builtin_amdgcn_logf(pr[k].y)};
#endif
            }
        }
    };
    auto fold_batch = [&]() __attribute__((always_inline)) {   // two-level sum: level 0 holds at most 16 rows' terms
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a1[k] += a0[k];
            a0[k] = (v2f)(0.f);
        }
    };
    // Software pipeline over the batches: while the terms of quad r4 of batch i are computed, the row pieces of quads r4+1 .. 3 of
    // batch i and of quads 0 .. r4-1 of batch i+16 are in flight -- quad r4 of batch i+16 is issued into the registers its
    // terms have just left.  The steady loop (this batch and the next one full) has no conditions inside, so that hipcc's
    // counted vmcnt waits stay exact (each quad waits with the 12 younger row pieces and the 2 metadata loads outstanding);
    // the first batch of a short list and the last one or two batches run the guarded copy below.
    uint32_t off_c, off_n;
    float s_c, c_c, s_n, c_n;
    {
        const Meta m0 = meta_load(0), m1 = meta_load(16);
        derive(m0, 0, off_c, s_c, c_c);
        derive(m1, 16, off_n, s_n, c_n);
    }
    int i = 0;
    if (K >= 32) {
        static_for<0, 4>([&](auto r4c) __attribute__((always_inline)) { load4(r4c, off_c); });
        for (; i + 32 <= K; i += 16) {
            const Meta m2 = meta_load(i + 32);
            static_for<0, 4>([&](auto r4c) __attribute__((always_inline)) {
                term4(r4c, s_c, c_c);
                __builtin_amdgcn_sched_barrier(0);   // (left alone, the scheduler sinks all 16 loads below the last quad's terms)
                load4(r4c, off_n);
                __builtin_amdgcn_sched_barrier(0);
            });
            fold_batch();
            off_c = off_n;
            s_c = s_n;
            c_c = c_n;
            derive(m2, i + 32, off_n, s_n, c_n);
        }
    } else {
        static_for<0, 4>([&](auto r4c) __attribute__((always_inline)) {
            if (4 * decltype(r4c)::value < K) load4(r4c, off_c);
        });
    }
    {   // batch i is in the registers (the quads of it that exist); at most one more batch follows
        const int left = K - i, left_n = left - 16;
        static_for<0, 4>([&](auto r4c) __attribute__((always_inline)) {
            constexpr int r4 = decltype(r4c)::value;
            if (4 * r4 < left) term4(r4c, s_c, c_c);
            if (4 * r4 < left_n) load4(r4c, off_n);
        });
        fold_batch();
        if (left_n > 0) {
            static_for<0, 4>([&](auto r4c) __attribute__((always_inline)) {
                if (4 * decltype(r4c)::value < left_n) term4(r4c, s_n, c_n);
            });
            fold_batch();
        }
    }
    if (!live) return;
    float* o = out + u * ldo + c0;
    const float ln2 = 0x1.62e430p-1f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const v2f t = a1[k] * (v2f)(ln2);
        if (c0 + 2 * k < ncols) o[2 * k] = t.x;
        if (c0 + 2 * k + 1 < ncols) o[2 * k + 1] = t.y;
    }
}

// ---- K5 -------------------------------------------------------------------------------------------
// prob_d = logsumexp_u(pdge) - log U per layer ("segment") and column, out = pdge - lam*prob_d, with the sum over
// rows in ATen's order.  That order only chains MICRO-CHUNK sums: every 64-row super-chunk yields 4 sums of 16
// rows per column (cascade order: rows 16q..16q+15; row_sum order: rows q, q+4, ..., q+60), each started from 0;
// the chain over them is short (U/16 terms).  So the work splits into three fully parallel launches:
//   A  column maxima of row splits                          -> ws.pmax[seg][split][c]
//   B  micro-chunk sums of exp(x - max)                     -> ws.msum[micro][c]
//   C  per workgroup: fold the micro sums in ATen's order (+ the rows beyond the last super-chunk), prob_d,
//      then subtract on its chunk of rows.
// torch.exp / torch.log inside logsumexp are MKL's high-accuracy vsExp / vsLn (practically correctly rounded), and
// one ulp of prob_d moves a whole column of the layer's scores by one ulp.  K5 evaluates 7 M exps and 9 K logs per
// run -- nothing next to K4 -- so both are taken in double precision and rounded once (correctly rounded up to the
// double library's own error): with ocml's fp32 expf / logf 2 of 763 columns came out one ulp off the reference.
// (the correctly rounded exponential through double is not what K5 waits for: with __expf the stress shape's 0.18 ms become 0.17)
__device__ __forceinline__ float k5_exp(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float k5_log(float x) { return (float)log((double)x); }

struct SegTable {
    int64_t off[65];    // row offsets of the segments
    int32_t msoff[65];  // micro-sum row offsets of the segments (4 per complete 64-row super-chunk)
};
constexpr int K5_RS = 8;   // row splits per segment for the maxima / the final subtraction
constexpr int K5_SB = 4;   // super-chunks per workgroup in B

__device__ __forceinline__ float k5_colmax(const float* pmax, int seg, int64_t Cp, int64_t c) {
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K5_RS; ++k) m = fmaxf(m, pmax[((int64_t)seg * K5_RS + k) * Cp + c]);
    return isinf(m) ? 0.f : m;  // torch.logsumexp: maxes.masked_fill_(maxes.abs() == inf, 0)
}

__global__ __launch_bounds__(256) void lse_max_kernel(const float* __restrict__ pdge, int64_t ld, int64_t C,
                                                       SegTable seg, float* __restrict__ pmax, int64_t Cp) {
    __shared__ float s_red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int sg = blockIdx.y, sp = blockIdx.z;
    const int64_t r0 = seg.off[sg], U = seg.off[sg + 1] - r0;
    const int64_t per = (U + K5_RS - 1) / K5_RS;
    const int64_t lo = sp * per, hi = (lo + per < U) ? lo + per : U;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const float* x = pdge + r0 * ld + (c < C ? c : 0);
    float m = -INFINITY;
    for (int64_t r = lo + w; r < hi; r += 4) m = fmaxf(m, x[r * ld]);
    s_red[w][lane] = m;
    __syncthreads();
    if (w == 0) pmax[((int64_t)sg * K5_RS + sp) * Cp + c] = fmaxf(fmaxf(s_red[0][lane], s_red[1][lane]),
                                                                 fmaxf(s_red[2][lane], s_red[3][lane]));
}

__global__ __launch_bounds__(256) void lse_microsum_kernel(const float* __restrict__ pdge, int64_t ld, int64_t C,
                                                            SegTable seg, int split, const float* __restrict__ pmax,
                                                            float* __restrict__ msum, int64_t Cp) {
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;  // wave q sums micro-chunk q of each super-chunk
    const int sg = blockIdx.y;
    const int64_t r0 = seg.off[sg], U = seg.off[sg + 1] - r0;
    const int64_t n_super = U >> 6;
    const int64_t sc0 = (int64_t)blockIdx.z * K5_SB;
    if (sc0 >= n_super) return;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const bool rs = c >= split;
    const float* x = pdge + r0 * ld + (c < C ? c : 0);
    const float m = k5_colmax(pmax, sg, Cp, c);
    for (int64_t sc = sc0; sc < sc0 + K5_SB && sc < n_super; ++sc) {
        const int64_t base = sc * 64;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = rs ? (base + q + 4 * r) : (base + 16 * q + r);
            s += k5_exp(x[row * ld] - m);
        }
        msum[((int64_t)seg.msoff[sg] + sc * 4 + q) * Cp + c] = s;
    }
}

__global__ __launch_bounds__(256) void lse_finish_kernel(const float* pdge, int64_t ld, int64_t C, SegTable seg,
                                                          float lam, int split, const float* __restrict__ pmax,
                                                          const float* __restrict__ msum, int64_t Cp, float* out,
                                                          int64_t ldo) {
    __shared__ float s_prob[64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int sg = blockIdx.y, sp = blockIdx.z;
    const int64_t r0 = seg.off[sg], U = seg.off[sg + 1] - r0;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const bool live = c < C;
    const bool rs = c >= split;
    const float* x = pdge + r0 * ld + (live ? c : 0);
    if (w == 0) {
        const float m = k5_colmax(pmax, sg, Cp, c);
        const int64_t n_super = U >> 6;
        const float* ms = msum + (int64_t)seg.msoff[sg] * Cp + c;
        Cascade st[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) st[k].init();
        for (int64_t sc = 0; sc < n_super; ++sc) {
            const float m0 = ms[(sc * 4 + 0) * Cp], m1 = ms[(sc * 4 + 1) * Cp], m2 = ms[(sc * 4 + 2) * Cp],
                        m3 = ms[(sc * 4 + 3) * Cp];
            if (rs) {
                st[0].a0 = m0; st[0].flush((int)((sc + 1) * 16));
                st[1].a0 = m1; st[1].flush((int)((sc + 1) * 16));
                st[2].a0 = m2; st[2].flush((int)((sc + 1) * 16));
                st[3].a0 = m3; st[3].flush((int)((sc + 1) * 16));
            } else {
                st[0].a0 = m0; st[0].flush((int)(sc * 64 + 16));
                st[0].a0 = m1; st[0].flush((int)(sc * 64 + 32));
                st[0].a0 = m2; st[0].flush((int)(sc * 64 + 48));
                st[0].a0 = m3; st[0].flush((int)(sc * 64 + 64));
            }
        }
        float s;
        if (!rs) {
            // cascade remainder: complete 16-row chunks beyond the last super-chunk, then the last rows
            int64_t i = n_super * 64;
            for (; i + 16 <= U; i += 16) {
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) a += k5_exp(x[(i + r) * ld] - m);
                st[0].a0 = a;
                st[0].flush((int)(i + 16));
            }
            for (; i < U; ++i) st[0].a0 += k5_exp(x[i * ld] - m);
            s = st[0].total();
        } else {
            const int64_t qn = U >> 2;
            float tot[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                for (int64_t mm = n_super * 16; mm < qn; ++mm) st[k].a0 += k5_exp(x[(4 * mm + k) * ld] - m);
                tot[k] = st[k].total();
            }
            for (int64_t i = 4 * qn; i < U; ++i) tot[0] += k5_exp(x[i * ld] - m);
            tot[0] += tot[1];
            tot[0] += tot[2];
            tot[0] += tot[3];
            s = tot[0];
        }
        const float lse = k5_log(s) + m;
        const float prob_d = lse - k5_log((float)U);
        s_prob[lane] = lam * prob_d;
    }
    __syncthreads();
    if (!live) return;
    const float prob_scaled = s_prob[lane];
    const int64_t per = (U + K5_RS - 1) / K5_RS;
    const int64_t lo = sp * per, hi = (lo + per < U) ? lo + per : U;
    float* o = out + r0 * ldo + c;
    for (int64_t r = lo + w; r < hi; r += 4) o[r * ldo] = x[r * ld] - prob_scaled;
}

// ---- K5, one pass: a (segment, W-column panel) of pdge is held in LDS --------------------------------------
// The three launches above read pdge three times (162 MB of traffic for 56 MB of algorithmic bytes at config 2).
// When a segment's U x W floats fit in LDS (W = 16 columns up to 1920 neurons, 8 up to 3840; measured at 12 x 768:
// 16 columns 0.029 ms, 32 columns 0.043, 8 columns 0.037, three launches 0.037) one
// workgroup reads its panel once, finds the column maxima, forms the 16-row micro-chunk sums of exp(x - max) in
// parallel, folds them in ATen's order (the same arithmetic, operation for operation, as lse_finish_kernel: the two
// paths give identical bits), and writes pdge - lam*prob_d: one read and one write of every element.
template <int W>
__global__ __launch_bounds__(256) void lse_panel_kernel(const float* pdge, int64_t ld, int64_t C, SegTable seg, float lam,
                                                         int split, float* out, int64_t ldo, int n_panels, int n_seg, int vec_ok) {
    extern __shared__ float s_x[];                 // [U][W] panel, then [n_micro][W] micro sums, then scratch
    constexpr int TR = 256 / W;                    // thread rows
    const int w = threadIdx.x % W, tr = threadIdx.x / W;
    // Workgroup -> (segment, panel).  A 16-column panel is HALF of each 128-byte line of its rows, so the two panels of a line go
    // to the SAME XCD in consecutive dispatch slots (ids 16 k + x and 16 k + 8 + x share XCD x): the second one's reads hit the
    // lines the first one pulled into that L2 and their stores merge into whole lines there.  Dealt round-robin (the plain 2-D
    // grid) the halves landed on different XCDs and every line crossed the HBM interface twice (82 MB for 56 at configs[1]).
    // Speed only: the result does not depend on the mapping.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int64_t pair = (int64_t)(slot >> 1) * 8 + xcd;
    const int pairs_per_seg = (n_panels + 1) >> 1;
    const int sg = (int)(pair / pairs_per_seg);
    const int panel = (int)(pair - (int64_t)sg * pairs_per_seg) * 2 + (slot & 1);
    if (sg >= n_seg || panel >= n_panels) return;  // padding of the grid (whole workgroups, before any barrier)
    const int64_t r0 = seg.off[sg];
    const int U = (int)(seg.off[sg + 1] - r0);
    const int64_t c = (int64_t)panel * W + w;
    const bool live = c < C;
    const bool rs = c >= split;
    float* s_ms = s_x + (size_t)U * W;             // micro sums: 4 per complete 64-row super-chunk
    const int n_super = U >> 6;
    float* s_red = s_ms + (size_t)4 * n_super * W; // [TR][W] maxima, then [W] lam*prob_d
    const float* x = pdge + r0 * ld + (live ? c : 0);

    // 1. the panel, and the column maxima (a maximum is exact in any order, NaNs dropped by v_max either way)
    // vec (W == 16, the whole panel inside C, pitches and bases 16-byte aligned): 16 bytes per lane -- a wave instruction
    // moves 16 rows x 64 B instead of 4, a thread has U/64 loads in flight instead of U/16 (0.21 -> 0.15 ms at 12 x 768 x 10 000)
    const bool vec = W == 16 && vec_ok && (int64_t)panel * W + W <= C;      // workgroup-uniform
    float m = -INFINITY;
    if (vec) {
        const int w4 = threadIdx.x & 3, tr4 = threadIdx.x >> 2;
        const float* xb = pdge + r0 * ld + (int64_t)panel * W + 4 * w4;
        float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
#pragma unroll 4   // (6 deep: the same; all 12 loads of a thread up front: 0.18 -> 0.25 ms)
        for (int u = tr4; u < U; u += 64) {
            const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)u * ld);
            *reinterpret_cast<float4*>(s_x + u * W + 4 * w4) = v;
            m0 = fmaxf(m0, v.x);
            m1 = fmaxf(m1, v.y);
            m2 = fmaxf(m2, v.z);
            m3 = fmaxf(m3, v.w);
        }
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) {
            m0 = fmaxf(m0, __shfl_xor(m0, off, 64));
            m1 = fmaxf(m1, __shfl_xor(m1, off, 64));
            m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
            m3 = fmaxf(m3, __shfl_xor(m3, off, 64));
        }
        if ((threadIdx.x & 63) < 4) {       // lane w4 of each wave: the wave's maxima of columns 4 w4 .. 4 w4 + 3
            float* d = s_red + (threadIdx.x >> 6) * W + 4 * w4;
            d[0] = m0;
            d[1] = m1;
            d[2] = m2;
            d[3] = m3;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) m = fmaxf(m, s_red[k * W + w]);
    } else {
#pragma unroll 8   // (16 / 24 / 48 deep: the same 28-29 us at configs[1] -- round 4)
        for (int u = tr; u < U; u += TR) {
            const float v = x[(int64_t)u * ld];
            s_x[u * W + w] = v;
            m = fmaxf(m, v);
        }
        s_red[tr * W + w] = m;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TR; ++k) m = fmaxf(m, s_red[k * W + w]);
    }
    if (isinf(m)) m = 0.f;                         // torch.logsumexp: maxes.masked_fill_(maxes.abs() == inf, 0)

    // 2. micro-chunk sums of exp(x - max): item (micro, column), 16 sequential adds each
    for (int mi = tr; mi < 4 * n_super; mi += TR) {
        const int sc = mi >> 2, q = mi & 3;
        const int base = sc * 64;
        float sacc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rs ? (base + q + 4 * r) : (base + 16 * q + r);
            sacc += k5_exp(s_x[row * W + w] - m);
        }
        s_ms[mi * W + w] = sacc;
    }
    __syncthreads();

    // 3. fold in ATen's order (one thread per column)
    if (tr == 0) {
        Cascade st[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) st[k].init();
        for (int sc = 0; sc < n_super; ++sc) {
            const float m0 = s_ms[(sc * 4 + 0) * W + w], m1 = s_ms[(sc * 4 + 1) * W + w], m2 = s_ms[(sc * 4 + 2) * W + w],
                        m3 = s_ms[(sc * 4 + 3) * W + w];
            if (rs) {
                st[0].a0 = m0; st[0].flush((sc + 1) * 16);
                st[1].a0 = m1; st[1].flush((sc + 1) * 16);
                st[2].a0 = m2; st[2].flush((sc + 1) * 16);
                st[3].a0 = m3; st[3].flush((sc + 1) * 16);
            } else {
                st[0].a0 = m0; st[0].flush(sc * 64 + 16);
                st[0].a0 = m1; st[0].flush(sc * 64 + 32);
                st[0].a0 = m2; st[0].flush(sc * 64 + 48);
                st[0].a0 = m3; st[0].flush(sc * 64 + 64);
            }
        }
        float ssum;
        if (!rs) {
            int i = n_super * 64;
            for (; i + 16 <= U; i += 16) {
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) a += k5_exp(s_x[(i + r) * W + w] - m);
                st[0].a0 = a;
                st[0].flush(i + 16);
            }
            for (; i < U; ++i) st[0].a0 += k5_exp(s_x[i * W + w] - m);
            ssum = st[0].total();
        } else {
            const int qn = U >> 2;
            float tot[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                for (int mm = n_super * 16; mm < qn; ++mm) st[k].a0 += k5_exp(s_x[(4 * mm + k) * W + w] - m);
                tot[k] = st[k].total();
            }
            for (int i = 4 * qn; i < U; ++i) tot[0] += k5_exp(s_x[i * W + w] - m);
            tot[0] += tot[1];
            tot[0] += tot[2];
            tot[0] += tot[3];
            ssum = tot[0];
        }
        const float lse = k5_log(ssum) + m;
        const float prob_d = lse - k5_log((float)U);
        s_red[w] = lam * prob_d;
    }
    __syncthreads();

    // 4. subtract and write
    if (vec) {
        const int w4 = threadIdx.x & 3, tr4 = threadIdx.x >> 2;
        const float p0 = s_red[4 * w4], p1 = s_red[4 * w4 + 1], p2 = s_red[4 * w4 + 2], p3 = s_red[4 * w4 + 3];
        float* ob = out + r0 * ldo + (int64_t)panel * W + 4 * w4;
#pragma unroll 4
        for (int u = tr4; u < U; u += 64) {
            const float4 v = *reinterpret_cast<const float4*>(s_x + u * W + 4 * w4);
            *reinterpret_cast<float4*>(ob + (int64_t)u * ldo) = make_float4(v.x - p0, v.y - p1, v.z - p2, v.w - p3);
        }
        return;
    }
    if (!live) return;
    const float prob_scaled = s_red[w];
    float* o = out + r0 * ldo + c;
    for (int u = tr; u < U; u += TR) o[(int64_t)u * ldo] = s_x[u * W + w] - prob_scaled;
}

static size_t k5_panel_lds(int64_t Umax, int W) {
    return (size_t)(Umax * W + 4 * (Umax >> 6) * W + 256) * sizeof(float);
}

int ilog2_ceil(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

extern "C" int mcd_wpmi_score(const float* S, int64_t ldS, int64_t N, int64_t C, const int32_t* idx, int64_t ldidx,
                              int64_t U, int K, const float* p, float min_prob, int soft, int split, float* pdge,
                              int64_t ldo, mcd_stream_t stream) {
    MCD_REQUIRE(S && idx && pdge, MCD_E_ARG, "mcd_wpmi_score: NULL pointer");
    MCD_REQUIRE(!(soft & 1) || p, MCD_E_ARG, "mcd_wpmi_score: soft scoring needs p[K]");
    MCD_REQUIRE(N > 0 && C > 0 && U >= 0 && K >= 1 && ldS >= C && ldidx >= K && ldo >= C, MCD_E_ARG,
                "mcd_wpmi_score: bad shape N=%lld C=%lld U=%lld K=%d", (long long)N, (long long)C, (long long)U, K);
    MCD_REQUIRE(K < (1 << 19), MCD_E_UNSUPPORTED, "mcd_wpmi_score: K=%d >= 2^19 changes ATen's chunk size", K);
    MCD_REQUIRE(C < (1 << 30), MCD_E_UNSUPPORTED, "mcd_wpmi_score: C too large");
    if (U == 0) return MCD_OK;
    if (split < 0) split = (int)(C >= 8 ? (C / 32) * 32 : (C / 4) * 4);
    if (split > C) split = (int)C;
    hipStream_t st = (hipStream_t)stream;
    // soft bit 1 (MCD_WPMI_FAST_LOG): the v_log_f32 based log (<= ~1.5 ulp) instead of the accurate table log;
    // honoured only when min_prob keeps every log argument normal
    static const int env_fast = mcd_dev_knob("MCD_FAST_LOG", 0);  // dev knob
    const bool fast_log = (((soft & 2) != 0) || env_fast) && (min_prob >= 1.17549435e-38f);
    const bool safe = !fast_log;  // template flag: accurate log
    // soft bit 2 (MCD_WPMI_S_IS_PROB): the caller guarantees S in [0,1] and p in [0,1]; with 2^MCD_LOG_E_MIN <= min_prob < 1 every
    // log argument then lies inside the table and the per-row range check is dropped
    const bool trusted = ((soft & 4) != 0) && min_prob >= ldexpf(1.0f, MCD_LOG_E_MIN) && min_prob < 1.0f;
    soft &= 1;
    const bool vec2 = (ldS % 2 == 0) && (((uintptr_t)S) % 8 == 0) && (split % 2 == 0);

#define MCD_WPMI_MAIN(VEC, SOFT, SAFE)                                                                          \
    hipLaunchKernelGGL((wpmi_main_kernel<VEC, SOFT, SAFE>), dim3(grid), dim3(256), 0, st, S, ldS, idx, ldidx, U, K, \
                       p, min_prob, split, nslab, pdge, ldo)
    static const int no_slice = mcd_dev_knob("MCD_WPMI_NO_SLICE", 0);  // dev knob
    // the sliced kernel does both summation orders itself; it needs 96-float slices, K % 4 == 0 (no row_sum
    // leftovers) and K < 256 (two cascade levels)
    const bool rs_ok = (split >= C) || (split % 96 == 64 && C - split < 32);  // row_sum group = a slice's last group
    // otherwise, when `split` falls on a slice boundary (C = 10 000: split = 9984 = 104 * 96), the sliced kernel takes
    // the cascade-order columns [0, split) and the tail kernel below the rest
    const bool rs_cut = !rs_ok && split > 0 && split % 96 == 0;
    if (ldS % 96 == 0 && (((uintptr_t)S) % 16 == 0) && K % 4 == 0 && K < 256 && (rs_ok || rs_cut) && !no_slice) {
        const int64_t C_all = C;
        if (rs_cut) C = split;
        const int n_slices = (int)mcd_cdiv(C, 96);
        const int64_t groups = mcd_cdiv(U, 16);
        const bool off32_ = (N < (1 << 24)) && (ldS * 4 < (1 << 24)) && ((double)N * (double)ldS * 4.0 < 4294967296.0);
        // accurate log: persistent workgroups (3 or 4 per CU x 256 CUs / 8 slices = 96 or 128 neuron groups in flight
        // per slice) so the 39 KB log tables are loaded once per workgroup; fast log: one pass per workgroup
        const int64_t gcap = safe ? ((trusted && off32_) ? 128 : 96) : groups;
        const int wg_per_slice = (int)(groups < gcap ? groups : gcap);
        const unsigned grid = (unsigned)(mcd_cdiv(n_slices, 8) * 8 * wg_per_slice);
        const bool off32 = off32_;
#define MCD_WPMI_SLICE_L(SOFT, SAFE, O32, TR)                                                                     \
    hipLaunchKernelGGL((wpmi_slice_kernel<SOFT, SAFE, O32, TR>), dim3(grid), dim3(256), 0, st, S, ldS, idx, ldidx, \
                       U, K, p, min_prob, (int)C, split, n_slices, wg_per_slice, pdge, ldo)
#define MCD_WPMI_SLICE(SOFT, SAFE)                                                                               \
    do {                                                                                                         \
        if (off32) {                                                                                             \
            if (trusted && SAFE) MCD_WPMI_SLICE_L(SOFT, SAFE, true, true);                                       \
            else MCD_WPMI_SLICE_L(SOFT, SAFE, true, false);                                                      \
        } else {                                                                                                 \
            MCD_WPMI_SLICE_L(SOFT, SAFE, false, false);                                                          \
        }                                                                                                        \
    } while (0)
        if (soft) { if (safe) MCD_WPMI_SLICE(true, true); else MCD_WPMI_SLICE(true, false); }
        else      { if (safe) MCD_WPMI_SLICE(false, true); else MCD_WPMI_SLICE(false, false); }
#undef MCD_WPMI_SLICE
#undef MCD_WPMI_SLICE_L
        MCD_LAUNCH_CHECK("wpmi_slice_kernel");
        if (!rs_cut) return MCD_OK;
        C = C_all;
    } else if (split > 0) {
        const int vec = vec2 ? 2 : 1;
        const int nslab = (int)mcd_cdiv(split, 64 * vec);
        const unsigned grid = (unsigned)mcd_cdiv(U * nslab, 4);
        if (vec2) {
            if (soft) { if (safe) MCD_WPMI_MAIN(2, true, true); else MCD_WPMI_MAIN(2, true, false); }
            else      { if (safe) MCD_WPMI_MAIN(2, false, true); else MCD_WPMI_MAIN(2, false, false); }
        } else {
            if (soft) { if (safe) MCD_WPMI_MAIN(1, true, true); else MCD_WPMI_MAIN(1, true, false); }
            else      { if (safe) MCD_WPMI_MAIN(1, false, true); else MCD_WPMI_MAIN(1, false, false); }
        }
        MCD_LAUNCH_CHECK("wpmi_main_kernel");
    }
#undef MCD_WPMI_MAIN
    if (split < C) {
        const int wt = (int)(C - split);
        MCD_REQUIRE(wt <= 64, MCD_E_UNSUPPORTED, "mcd_wpmi_score: %d row_sum-order columns (> 64)", wt);
        const int gl = ilog2_ceil(wt);
        const int per_wave = 64 >> gl;
        const unsigned grid = (unsigned)mcd_cdiv(mcd_cdiv(U, per_wave), 4);
#define MCD_WPMI_TAIL(SOFT, SAFE)                                                                              \
    hipLaunchKernelGGL((wpmi_tail_kernel<SOFT, SAFE>), dim3(grid), dim3(256), 0, st, S, ldS, idx, ldidx, U, K, p, \
                       min_prob, split, (int)C, gl, pdge, ldo)
        if (soft) { if (safe) MCD_WPMI_TAIL(true, true); else MCD_WPMI_TAIL(true, false); }
        else      { if (safe) MCD_WPMI_TAIL(false, true); else MCD_WPMI_TAIL(false, false); }
#undef MCD_WPMI_TAIL
        MCD_LAUNCH_CHECK("wpmi_tail_kernel");
    }
    return MCD_OK;
}

static int64_t k5_cp(int64_t C) { return (C + 63) / 64 * 64; }

extern "C" size_t mcd_logsumexp_sub_workspace(int64_t U_total, int64_t C, int n_seg) {
    // column maxima of the row splits + micro-chunk sums (at most U/16 rows)
    return (size_t)((int64_t)n_seg * K5_RS + U_total / 16 + 4) * (size_t)k5_cp(C) * sizeof(float);
}

extern "C" int mcd_logsumexp_sub(const float* pdge, int64_t ld, int64_t C, const int64_t* seg_offsets, int n_seg,
                                 float lam, int split, float* out, int64_t ldo, void* ws, size_t ws_bytes,
                                 mcd_stream_t stream) {
    MCD_REQUIRE(pdge && out && seg_offsets, MCD_E_ARG, "mcd_logsumexp_sub: NULL pointer");
    MCD_REQUIRE(C > 0 && ld >= C && ldo >= C, MCD_E_ARG, "mcd_logsumexp_sub: bad shape");
    MCD_REQUIRE(n_seg >= 0 && n_seg <= 64, MCD_E_UNSUPPORTED, "mcd_logsumexp_sub: n_seg=%d not in [0,64]", n_seg);
    if (n_seg == 0) return MCD_OK;
    SegTable seg;
    int64_t max_super = 0;
    seg.msoff[0] = 0;
    for (int s = 0; s <= n_seg; ++s) {
        seg.off[s] = seg_offsets[s];
        MCD_REQUIRE(s == 0 || seg.off[s] > seg.off[s - 1], MCD_E_ARG, "mcd_logsumexp_sub: empty or unordered segment %d",
                    s - 1);
        MCD_REQUIRE(s == 0 || seg.off[s] - seg.off[s - 1] < (1 << 19), MCD_E_UNSUPPORTED,
                    "mcd_logsumexp_sub: segment of 2^19 rows or more changes ATen's chunk size");
        if (s > 0) {
            const int64_t ns = (seg.off[s] - seg.off[s - 1]) >> 6;
            seg.msoff[s] = seg.msoff[s - 1] + (int32_t)(4 * ns);
            if (ns > max_super) max_super = ns;
        }
    }
    const int64_t U_total = seg.off[n_seg] - seg.off[0];
    const size_t need = mcd_logsumexp_sub_workspace(U_total, C, n_seg);
    MCD_REQUIRE(ws && ws_bytes >= need, MCD_E_WORKSPACE, "mcd_logsumexp_sub: workspace %zu < %zu bytes", ws_bytes, need);
    if (split < 0) split = (int)(C >= 8 ? (C / 32) * 32 : (C / 4) * 4);
    hipStream_t st = (hipStream_t)stream;
    {   // one-pass panel kernel when the largest segment fits in LDS
        int64_t Umax = 0;
        for (int s = 0; s < n_seg; ++s)
            if (seg.off[s + 1] - seg.off[s] > Umax) Umax = seg.off[s + 1] - seg.off[s];
        static const int no_panel = mcd_dev_knob("MCD_LSE_NO_PANEL", 0);  // dev knob
        const int W = Umax <= 1920 ? 16 : (Umax <= 3840 ? 8 : 0);   // 16 columns: 52 KB at 768 neurons, 3 workgroups per CU
        if (W && !no_panel) {
            const size_t shmem = k5_panel_lds(Umax, W);
            static bool attr_done_dev[MCD_MAX_DEVICES];
            bool& attr_done = attr_done_dev[mcd_cur_device()];
            if (!attr_done) {
                hipError_t e2 = hipFuncSetAttribute((const void*)lse_panel_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
                hipError_t e3 = hipFuncSetAttribute((const void*)lse_panel_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
                MCD_REQUIRE(e2 == hipSuccess && e3 == hipSuccess, MCD_E_LAUNCH, "mcd_logsumexp_sub: cannot reserve LDS");
                attr_done = true;
            }
            const int n_panels = (int)mcd_cdiv(C, W);
            const int64_t n_pairs = (int64_t)((n_panels + 1) / 2) * n_seg;
            const dim3 grid((unsigned)(mcd_cdiv(n_pairs, 8) * 16));          // pairs in rounds of 8 (one per XCD), two slots each
            // 16-byte accesses: every row of every segment starts a multiple of 16 bytes into both matrices
            static const int no_vec = mcd_dev_knob("MCD_LSE_NO_VEC", 0);   // dev knob
            const int vec_ok = !no_vec && ld % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)pdge & 15) == 0 && ((uintptr_t)out & 15) == 0;
            if (W == 16) hipLaunchKernelGGL(lse_panel_kernel<16>, grid, dim3(256), shmem, st, pdge, ld, C, seg, lam, split, out, ldo, n_panels, (int)n_seg, vec_ok);
            else hipLaunchKernelGGL(lse_panel_kernel<8>, grid, dim3(256), shmem, st, pdge, ld, C, seg, lam, split, out, ldo, n_panels, (int)n_seg, 0);
            MCD_LAUNCH_CHECK("lse_panel_kernel");
            return MCD_OK;
        }
    }
    const int64_t Cp = k5_cp(C);
    float* pmax = (float*)ws;
    float* msum = pmax + (int64_t)n_seg * K5_RS * Cp;
    const unsigned panels = (unsigned)(Cp / 64);
    hipLaunchKernelGGL(lse_max_kernel, dim3(panels, (unsigned)n_seg, K5_RS), dim3(256), 0, st, pdge, ld, C, seg, pmax, Cp);
    MCD_LAUNCH_CHECK("lse_max_kernel");
    if (max_super > 0) {
        hipLaunchKernelGGL(lse_microsum_kernel, dim3(panels, (unsigned)n_seg, (unsigned)mcd_cdiv(max_super, K5_SB)),
                           dim3(256), 0, st, pdge, ld, C, seg, split, pmax, msum, Cp);
        MCD_LAUNCH_CHECK("lse_microsum_kernel");
    }
    hipLaunchKernelGGL(lse_finish_kernel, dim3(panels, (unsigned)n_seg, K5_RS), dim3(256), 0, st, pdge, ld, C, seg, lam,
                       split, pmax, msum, Cp, out, ldo);
    MCD_LAUNCH_CHECK("lse_finish_kernel");
    return MCD_OK;
}


extern "C" size_t mcd_wpmi_score_bf16_workspace(int64_t U, int K) {
    return (size_t)(U > 0 ? U : 0) * (size_t)(K > 0 ? K : 0) * sizeof(int2);   // meta[u][j] = {row, p_j * rinv[row]}
}

extern "C" int mcd_wpmi_score_bf16(const uint16_t* E, int64_t ldE, int64_t N, int64_t C, const float* rinv,
                                   const int32_t* idx, int64_t ldidx, int64_t U, int K, const float* p, float min_prob,
                                   int soft, float* pdge, int64_t ldo, void* ws, size_t ws_bytes, mcd_stream_t stream) {
    MCD_REQUIRE(E && rinv && idx && pdge, MCD_E_ARG, "mcd_wpmi_score_bf16: NULL pointer");
    MCD_REQUIRE(N > 0 && C > 0 && U >= 0 && K >= 1 && ldidx >= K && ldo >= C, MCD_E_ARG, "mcd_wpmi_score_bf16: bad shape");
    MCD_REQUIRE(ldE % 128 == 0 && ldE >= C && ((uintptr_t)E) % 16 == 0, MCD_E_ARG,
                "mcd_wpmi_score_bf16: E rows must be padded to a multiple of 128 concepts and 16-byte aligned");
    MCD_REQUIRE(!(soft & 1) || p, MCD_E_ARG, "mcd_wpmi_score_bf16: soft-WPMI needs p[K]");
    MCD_REQUIRE(min_prob >= 1.17549435e-38f, MCD_E_ARG, "mcd_wpmi_score_bf16: min_prob must keep the log arguments normal");
    MCD_REQUIRE(C < (1 << 30), MCD_E_UNSUPPORTED, "mcd_wpmi_score_bf16: C too large");
    if (U == 0) return MCD_OK;
    MCD_REQUIRE(ws && ((uintptr_t)ws & 7) == 0 && ws_bytes >= mcd_wpmi_score_bf16_workspace(U, K), MCD_E_WORKSPACE,
                "mcd_wpmi_score_bf16: workspace %zu < %zu bytes (or not 8-byte aligned)", ws_bytes, mcd_wpmi_score_bf16_workspace(U, K));
    MCD_REQUIRE(U * (int64_t)K < (1LL << 38), MCD_E_UNSUPPORTED, "mcd_wpmi_score_bf16: U * K too large");   // wpmi_meta_kernel: one thread each
    int2* meta = (int2*)ws;
    const int n_slices = (int)mcd_cdiv(C, 128);
    const int64_t groups = mcd_cdiv(U, 16);                // 16 neurons per workgroup
    const int64_t grid64 = mcd_cdiv(n_slices, 8) * 8 * groups;
    MCD_REQUIRE(grid64 < (1LL << 31) && groups < (1 << 27), MCD_E_UNSUPPORTED, "mcd_wpmi_score_bf16: too many workgroups");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wpmi_meta_kernel, dim3((unsigned)mcd_cdiv(U * K, 256)), dim3(256), 0, st, idx, ldidx, U, K, rinv, p, soft & 1, meta);
    MCD_LAUNCH_CHECK("wpmi_meta_kernel");
    static const int env_group = mcd_dev_knob("MCD_WPMI_BF16_GROUP", 0);   // dev knob: 1 or 4
    const bool group4 = env_group != 1 && min_prob >= 0x1p-30f;   // products of four arguments stay normal numbers
    const bool off32 = N < (1 << 24) && ldE * 2 < (1 << 24) && N * ldE * 2 < (1LL << 32);
#define MCD_WB(SOFT, GROUP, OFF32)                                                                                     \
    hipLaunchKernelGGL((wpmi_bf16_kernel<SOFT, GROUP, OFF32>), dim3((unsigned)grid64), dim3(256), 0, st, E, ldE, meta, U, K, p, \
                       min_prob, (int)C, n_slices, (int)groups, pdge, ldo)
#define MCD_WB2(SOFT, GROUP) do { if (off32) MCD_WB(SOFT, GROUP, true); else MCD_WB(SOFT, GROUP, false); } while (0)
    if (soft & 1) { if (group4) MCD_WB2(true, 4); else MCD_WB2(true, 1); }
    else          { if (group4) MCD_WB2(false, 4); else MCD_WB2(false, 1); }
#undef MCD_WB2
#undef MCD_WB
    MCD_LAUNCH_CHECK("wpmi_bf16_kernel");
    return MCD_OK;
}
