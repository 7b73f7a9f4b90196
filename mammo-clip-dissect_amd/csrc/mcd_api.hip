// mcd_api.hip -- error reporting and version of the C ABI (include/mcd_hip.h).
#include "mcd_common.h"

thread_local char g_mcd_err[512] = "";

int mcd_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_mcd_err, sizeof(g_mcd_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* mcd_last_error(void) { return g_mcd_err; }
extern "C" int mcd_abi_version(void) { return 9; }  // 2: mcd_embed_gemm takes a workspace; mcd_rank_reorder.  3: mcd_vit_attention.  4: mcd_layer_norm.  5: mcd_patchify.  6: mcd_embed_gemm_exp, mcd_wpmi_score_bf16 (the stress chain).  7: mcd_wpmi_score_bf16 takes a workspace.  8: mcd_embed_gemm_exp_time_kernel / _kernel_ms (measurement hook).  9: mcd_embed_gemm_exp is one kernel for every shape -- ldE must be a multiple of 16 (MCD_E_UNSUPPORTED otherwise), rinv = 1 / the sum of the STORED bf16 values
