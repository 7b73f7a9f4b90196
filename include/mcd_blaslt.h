/* mcd_blaslt.h -- C ABI of libmcd_blaslt.so: encoder-side "linear + bias + residual" as one hipBLASLt call.
 *
 * Optional companion of libmcd_hip.so (include/mcd_hip.h), kept in its own shared object so that the dissection
 * core carries no dependency on hipBLASLt.  Not part of the dissection path proper: it serves the ViT blocks of the
 * image tower inside the forwards that the reference's extraction loop drives (concept_vit/utils.py:117-148), where
 *     x = x + proj(attention(...))          (ViTSelfOutput / nn.MultiheadAttention.out_proj + the residual add)
 *     x = x + fc2(gelu(fc1(...)))           (ViTOutput / the MLP's c_proj + the residual add)
 * are, in PyTorch, a GEMM with a bias epilogue followed by a separate elementwise add over the whole residual stream.
 * Same conventions as mcd_hip.h: device pointers, caller-owned buffers, a hipStream_t, status 0 = ok.
 */
#ifndef MCD_BLASLT_H
#define MCD_BLASLT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* mcd_blaslt_stream_t; /* hipStream_t */

/* text of the last error of this thread (libmcd_blaslt.so keeps its own) */
const char* mcd_blaslt_last_error(void);

/* bytes of workspace the caller should provide (32 MiB) */
size_t mcd_linear_residual_workspace(void);

/* out[M,N] = res[M,N] + h[M,K] . W[N,K]^T + bias[N], all fp32 row-major with leading dimensions ld*.
 * res may be NULL (no residual) or equal to out (in place); bias may be NULL.  The first call for a shape times up
 * to 32 hipBLASLt candidates on `stream` (it synchronises and allocates a scratch D of M*ldo floats for that);
 * later calls are asynchronous.
 * replaces  nn.Linear + the residual add of a transformer block (see above). */
int mcd_linear_residual(const float* h, int64_t ldh, const float* W, int64_t ldw, const float* bias, const float* res,
                        int64_t ldr, float* out, int64_t ldo, int64_t M, int64_t N, int64_t K, void* ws, size_t ws_bytes,
                        mcd_blaslt_stream_t stream);

/* average time (ms) of the algorithm kept for shape (M, N, K) and the number of candidates that were timed */
int mcd_linear_residual_plan_info(int64_t M, int64_t N, int64_t K, float* ms, int* tried);

/* Reproducible algorithm choice.  A timed pick may differ between processes (and different algorithms sum in different
 * orders); the env MCD_BLASLT_PICK=heuristic takes the first usable heuristic candidate instead, and these two calls let
 * one process's picks be forced in another (the multi-rank host code broadcasts rank 0's):
 *   get_picks: up to `cap` records of 5 int64 {M, N, K, has_res, pick} into `out` (host memory); returns the number held.
 *   set_pick:  force `pick` (index into the heuristic's candidate list; < 0 un-forces) for a shape; an existing plan for
 *              it is rebuilt on the next call. */
int mcd_linear_residual_get_picks(int64_t* out, int cap);
int mcd_linear_residual_set_pick(int64_t M, int64_t N, int64_t K, int has_res, int pick);

#ifdef __cplusplus
}
#endif
#endif /* MCD_BLASLT_H */
