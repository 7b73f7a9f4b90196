// mcd_common.h -- shared helpers for the gfx950 kernels of libmcd_hip.so (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/mcd_hip.h"

#define MCD_WAVE 64

// thread-local error text, set by mcd_fail(), read through mcd_last_error()
extern thread_local char g_mcd_err[512];
int mcd_fail(int code, const char* fmt, ...);

#define MCD_REQUIRE(cond, code, ...)                 \
    do {                                             \
        if (!(cond)) return mcd_fail(code, __VA_ARGS__); \
    } while (0)

#define MCD_LAUNCH_CHECK(name)                                                                   \
    do {                                                                                         \
        hipError_t e__ = hipGetLastError();                                                      \
        if (e__ != hipSuccess) return mcd_fail(MCD_E_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline int64_t mcd_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Development knobs: environment variables that pick kernel variants for scripts/ (and for one test that compares two of
// them).  They exist in the DEV build only (`make dev`, -DMCD_DEV_KNOBS -> libmcd_hip_dev.so); the product library never
// reads the environment.
#ifdef MCD_DEV_KNOBS
static inline int mcd_dev_knob(const char* name, int def) { const char* v = getenv(name); return v ? atoi(v) : def; }
static inline const char* mcd_dev_env(const char* name) { return getenv(name); }
#else
static inline int mcd_dev_knob(const char*, int def) { return def; }
static inline const char* mcd_dev_env(const char*) { return nullptr; }
#endif

// One-time per-DEVICE state (hipFuncSetAttribute applies to the current device only; CU counts are per device):
// host-side caches are arrays indexed by the current HIP device, never process-wide flags.
#define MCD_MAX_DEVICES 64
static inline int mcd_cur_device(void) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MCD_MAX_DEVICES) d = 0;
    return d;
}

// ---- device helpers ---------------------------------------------------------------------------
// Order-preserving map fp32 -> u32 (larger float <=> larger key); every NaN maps to the top key,
// which is torch.topk's rule (NaN ranks above +inf).
__device__ __forceinline__ uint32_t mcd_f2key(float f) {
    uint32_t b = __float_as_uint(f);
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float mcd_key2f(uint32_t k) {
    if (k == 0xffffffffu) return __uint_as_float(0x7fc00000u);
    uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

__device__ __forceinline__ float mcd_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float mcd_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int mcd_wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// natural log for arguments that are normal, finite and positive (every call site guarantees it):
// v_log_f32 (1 ulp log2) times ln2 carried in two floats -- the ocml algorithm without its
// denormal / inf / nan handling.  The fma here is deliberate (error-free product), unlike the
// data path, which is compiled with -ffp-contract=off.
__device__ __forceinline__ float mcd_log_pos(float x) {
    const float r = __builtin_amdgcn_logf(x);  // log2(x)
    const float c = 0x1.62e42ep-1f, cc = 0x1.efa39ep-25f;
    const float ph = r * c;
    float pl = __builtin_fmaf(r, c, -ph);
    pl = __builtin_fmaf(r, cc, pl);
    return ph + pl;
}
