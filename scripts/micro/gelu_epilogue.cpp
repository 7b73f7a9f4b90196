// Is hipBLASLt's GELU epilogue the exact (erf) GELU or the tanh approximation?  D = GELU(h . I + 0) on a grid of values.
// hipcc --offload-arch=gfx950 -O2 -o gelu_epilogue gelu_epilogue.cpp -L/opt/rocm/lib -lhipblaslt -Wl,-rpath,/opt/rocm/lib
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define CK(x) do { auto s_ = (x); if ((int)s_ != 0) { printf("fail %s -> %d\n", #x, (int)s_); return 1; } } while (0)
int main() {
    const int M = 256, N = 64, K = 64;
    std::vector<float> h(M * K), W(N * K, 0.f), out(M * N);
    for (int i = 0; i < M * K; ++i) h[i] = -6.f + 12.f * i / (M * K - 1);
    for (int i = 0; i < N; ++i) W[i * K + i] = 1.f;
    float *dh, *dW, *dout; void* ws;
    CK(hipMalloc(&dh, h.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dout, out.size() * 4)); CK(hipMalloc(&ws, 32 << 20));
    CK(hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    hipblasLtHandle_t hd; CK(hipblasLtCreate(&hd));
    hipblasLtMatmulDesc_t desc; CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    int32_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N; uint32_t ep = HIPBLASLT_EPILOGUE_GELU;
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, 4));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, 4));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, 4));
    hipblasLtMatrixLayout_t a, b, c;
    CK(hipblasLtMatrixLayoutCreate(&a, HIP_R_32F, K, N, K)); CK(hipblasLtMatrixLayoutCreate(&b, HIP_R_32F, K, M, K)); CK(hipblasLtMatrixLayoutCreate(&c, HIP_R_32F, N, M, N));
    hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
    uint64_t mw = 32 << 20; CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &mw, 8));
    hipblasLtMatmulHeuristicResult_t r[4]; int n = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(hd, desc, a, b, c, c, pref, 4, r, &n));
    if (n < 1) { printf("no algo\n"); return 1; }
    float one = 1.f, zero = 0.f;
    CK(hipblasLtMatmul(hd, desc, &one, dW, a, dh, b, &zero, dout, c, dout, c, &r[0].algo, ws, r[0].workspaceSize, 0));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    double e_erf = 0, e_tanh = 0;
    for (int i = 0; i < M * N; ++i) {
        const double x = h[i];
        const double g_erf = 0.5 * x * (1.0 + erf(x / sqrt(2.0)));
        const double g_tanh = 0.5 * x * (1.0 + tanh(0.7978845608028654 * (x + 0.044715 * x * x * x)));
        e_erf = fmax(e_erf, fabs(out[i] - g_erf));
        e_tanh = fmax(e_tanh, fabs(out[i] - g_tanh));
    }
    printf("hipBLASLt GELU epilogue: max |diff| vs erf GELU %.3e, vs tanh GELU %.3e\n", e_erf, e_tanh);
    return 0;
}
