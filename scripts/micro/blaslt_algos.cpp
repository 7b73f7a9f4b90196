// Which fp32 hipBLASLt solutions exist WITHOUT stream-K / split-K for the encoder's GEMM shapes, and what do they cost?
// (a pick whose per-row summation order cannot depend on M.)   out[M,N] = h[M,K] . W[N,K]^T + bias, res
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -o blaslt_algos scripts/micro/blaslt_algos.cpp -lhipblaslt && ./blaslt_algos
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <cstdio>
#include <string>
#include <vector>
#include <algorithm>
#define CK(x) do { auto s_ = (x); if ((int)s_ != 0) { printf("FAIL %s -> %d\n", #x, (int)s_); return 1; } } while (0)

static std::string field(const std::string& n, const char* key) {   // "_SK3_" -> "3"
    auto p = n.find(std::string("_") + key);
    if (p == std::string::npos) return "?";
    p += 1 + strlen(key);
    auto e = n.find('_', p);
    return n.substr(p, e - p);
}

int main() {
    hipblasLtHandle_t h;
    CK(hipblasLtCreate(&h));
    const int64_t shapes[][3] = {{492500, 2304, 768}, {492500, 768, 768}, {492500, 3072, 768}, {492500, 768, 3072}, {246250, 768, 3072}, {59100, 3072, 768}};
    float *A, *B, *C, *D, *bias;
    void* ws;
    const size_t wsz = 32u << 20;
    CK(hipMalloc(&A, (size_t)3072 * 3072 * 4)); CK(hipMalloc(&B, (size_t)492500 * 3072 * 4)); CK(hipMalloc(&C, (size_t)492500 * 3072 * 4));
    CK(hipMalloc(&D, (size_t)492500 * 3072 * 4)); CK(hipMalloc(&bias, 3072 * 4)); CK(hipMalloc(&ws, wsz));
    CK(hipMemset(A, 0, (size_t)3072 * 3072 * 4)); CK(hipMemset(B, 0, (size_t)492500 * 3072 * 4)); CK(hipMemset(C, 0, (size_t)492500 * 3072 * 4)); CK(hipMemset(bias, 0, 3072 * 4));
    std::vector<hipblasLtMatmulHeuristicResult_t> all;
    CK(hipblaslt_ext::getAllAlgos(h, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, HIPBLAS_OP_T, HIPBLAS_OP_N, HIP_R_32F, HIP_R_32F, HIP_R_32F, HIP_R_32F,
                                  HIPBLAS_COMPUTE_32F, all));
    printf("getAllAlgos(fp32 TN): %zu solutions\n", all.size());
    for (auto& sh : shapes) {
        const int64_t M = sh[0], N = sh[1], K = sh[2];
        hipblasLtMatmulDesc_t desc;
        hipblasLtMatrixLayout_t a, b, c, d;
        CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        const int32_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
        const uint32_t ep = HIPBLASLT_EPILOGUE_BIAS;
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
        CK(hipblasLtMatrixLayoutCreate(&a, HIP_R_32F, K, N, K)); CK(hipblasLtMatrixLayoutCreate(&b, HIP_R_32F, K, M, K));
        CK(hipblasLtMatrixLayoutCreate(&c, HIP_R_32F, N, M, N)); CK(hipblasLtMatrixLayoutCreate(&d, HIP_R_32F, N, M, N));
        const float one = 1.f;
        struct R { double ms; std::string sk, gsu, mt; int idx; };
        std::vector<R> res;
        int supported = 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (auto& r : all) {
            size_t need = 0;
            if (hipblaslt_ext::matmulIsAlgoSupported(h, desc, &one, a, b, &one, c, d, r.algo, need) != HIPBLAS_STATUS_SUCCESS || need > wsz) continue;
            ++supported;
            const std::string name = hipblaslt_ext::getKernelNameFromAlgo(h, r.algo);
            const std::string sk = field(name, "SK"), gsu = field(name, "GSU"), mt = field(name, "MT");
            if (sk != "0") continue;                       // stream-K off only
            bool ok = true;
            for (int rep = 0; rep < 3 && ok; ++rep) {
                if (rep == 1) hipEventRecord(e0, 0);
                ok = hipblasLtMatmul(h, desc, &one, A, a, B, b, &one, C, c, D, d, &r.algo, ws, need, 0) == HIPBLAS_STATUS_SUCCESS;
            }
            hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) ok = false;
            float ms = 0.f;
            if (ok) hipEventElapsedTime(&ms, e0, e1);
            if (ok) res.push_back({ms / 2.0, sk, gsu, mt, hipblaslt_ext::getIndexFromAlgo(r.algo)});
        }
        // the heuristic's first pick, for comparison
        hipblasLtMatmulPreference_t pref; hipblasLtMatmulPreferenceCreate(&pref);
        hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz));
        std::vector<hipblasLtMatmulHeuristicResult_t> cand(32); int n = 0;
        hipblasLtMatmulAlgoGetHeuristic(h, desc, a, b, c, d, pref, 32, cand.data(), &n);
        double best_h = 1e9; std::string best_name;
        for (int i = 0; i < n; ++i) {
            bool ok = true;
            for (int rep = 0; rep < 3 && ok; ++rep) {
                if (rep == 1) hipEventRecord(e0, 0);
                ok = hipblasLtMatmul(h, desc, &one, A, a, B, b, &one, C, c, D, d, &cand[i].algo, ws, cand[i].workspaceSize, 0) == HIPBLAS_STATUS_SUCCESS;
            }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
            if (ok && ms / 2.0 < best_h) { best_h = ms / 2.0; best_name = hipblaslt_ext::getKernelNameFromAlgo(h, cand[i].algo); }
        }
        std::sort(res.begin(), res.end(), [](const R& x, const R& y) { return x.ms < y.ms; });
        const double gf = 2.0 * M * N * K / 1e9;
        printf("M=%lld N=%lld K=%lld: %d supported, %zu with SK0; heuristic best of %d: %.3f ms (%.1f TF) MT%s SK%s GSU%s\n", (long long)M, (long long)N, (long long)K,
               supported, res.size(), n, best_h, gf / best_h, field(best_name, "MT").c_str(), field(best_name, "SK").c_str(), field(best_name, "GSU").c_str());
        for (size_t i = 0; i < res.size() && i < 5; ++i)
            printf("    SK0 #%zu: index %d  MT%s GSU%s  %.3f ms (%.1f TF)\n", i, res[i].idx, res[i].mt.c_str(), res[i].gsu.c_str(), res[i].ms, gf / res[i].ms);
    }
    return 0;
}
