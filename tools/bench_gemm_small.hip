// Launch-loop timing of the fp32 GEMM entry points at small M (no interpreter between launches).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bench_gemm_small tools/bench_gemm_small.hip \
//          -Ltext-style-transfer_amd -ltst_hip -Wl,-rpath,'$ORIGIN/../text-style-transfer_amd'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/tst_hip.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

int main() {
    const int64_t Mmax = 16384, Nmax = 3072, Kmax = 2048;
    std::vector<float> h((size_t)Mmax * Kmax);
    srand(3);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *A, *W, *C, *R, *bias;
    CHECK(hipMalloc(&A, (size_t)Mmax * Kmax * 4)); CHECK(hipMalloc(&W, (size_t)Kmax * Nmax * 4));
    CHECK(hipMalloc(&C, (size_t)Mmax * Nmax * 4)); CHECK(hipMalloc(&R, (size_t)Mmax * Nmax * 4)); CHECK(hipMalloc(&bias, Nmax * 4));
    CHECK(hipMemcpy(A, h.data(), (size_t)Mmax * Kmax * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(W, h.data(), (size_t)Kmax * Nmax * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(R, h.data(), (size_t)Mmax * 1024 * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(bias, h.data(), Nmax * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct Shape { const char* name; int64_t N, K; bool res; int act; };
    const Shape shapes[] = {{"qkv", 768, 256, false, 0}, {"proj+res", 256, 256, true, 0}, {"ff1 relu", 1024, 256, false, 1},
                            {"ff2+res", 256, 1024, true, 0}, {"gru gates", 3072, 1024, false, 0}, {"lm qkv", 1536, 512, false, 0},
                            {"lm ff2", 512, 2048, true, 0}};
    const int64_t Ms[] = {16, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384};
    for (const Shape& s : shapes)
        for (int64_t M : Ms) {
            printf("%-9s M=%-5lld N=%-4lld K=%-4lld", s.name, (long long)M, (long long)s.N, (long long)s.K);
            for (int variant = 0; variant < 4; ++variant) {   // 0: tst_gemm_mfma_f32 (its own choice); 1, 2, 3: strips 1, 2, 4
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(e0));
                    for (int i = 0; i < 30; ++i) {
                        const int rc = variant == 0
                            ? tst_gemm_mfma_f32(A, s.K, W, s.N, bias, s.res ? R : nullptr, s.N, C, s.N, M, s.N, s.K, s.act, nullptr)
                            : tst_gemm_skinny_f32(A, s.K, W, s.N, bias, s.res ? R : nullptr, s.N, C, s.N, M, s.N, s.K, s.act,
                                                  1 << (variant - 1), nullptr);
                        if (rc != 0) { printf("error: %s\n", tst_last_error()); return 1; }
                    }
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                if (variant == 0) printf("  auto %7.1f us", best * 1e3f / 30); else printf(" | %d: %7.1f", 1 << (variant - 1), best * 1e3f / 30);
            }
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
