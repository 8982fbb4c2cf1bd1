// Probe: what bounds a one-wave-per-16x16-tile fp32 GEMM (v_mfma_f32_16x16x4_f32) at small M?
// Variants: MODE 0 = dword loads of A and W; 1 = no A loads; 2 = no W loads; 3 = no loads at all (MFMA chain only);
//           4 = A as float4 + 4x4 (row, register) transpose with permlane swaps, W dword loads
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o skinny_probe skinny_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int KC>
__global__ __launch_bounds__(64) void k_skinny(const float* __restrict__ A, const float* __restrict__ W, float* C, int M, int N, int K) {
    constexpr int STEPS = KC / 4;
    const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;
    const int tiles_n = N / 16;
    const int m0 = (blockIdx.x / tiles_n) * 16, n0 = (blockIdx.x % tiles_n) * 16;
    const float* a_row = A + (int64_t)(m0 + r) * K;
    const float* w_col = W + n0 + r;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float a_cur[STEPS], b_cur[STEPS], a_nxt[STEPS], b_nxt[STEPS];
    auto load = [&](float* av, float* bv, int k0) {
        if (MODE == 4) {
#pragma unroll
            for (int c = 0; c < STEPS / 4; ++c) {
                float4 v = *reinterpret_cast<const float4*>(a_row + k0 + 16 * c + 4 * kq);
                av[4 * c + 0] = v.x; av[4 * c + 1] = v.y; av[4 * c + 2] = v.z; av[4 * c + 3] = v.w;
            }
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int k = k0 + 4 * s + kq;
            if (MODE == 0 || MODE == 2) av[s] = a_row[k];
            else if (MODE != 4) av[s] = 1.0f + lane;
            if (MODE == 0 || MODE == 1 || MODE == 4) bv[s] = w_col[(int64_t)k * N];
            else bv[s] = 0.5f + s;
        }
    };
    auto mfma = [&](float* av, float* bv) {
        if (MODE == 4) {
#pragma unroll
            for (int c = 0; c < STEPS / 4; ++c) {
                // registers av[4c+i] hold component i of row kq's float4; transpose (row, register)
                unsigned x0 = __float_as_uint(av[4 * c]), x1 = __float_as_uint(av[4 * c + 1]), x2 = __float_as_uint(av[4 * c + 2]),
                         x3 = __float_as_uint(av[4 * c + 3]);
                auto p02 = __builtin_amdgcn_permlane32_swap(x0, x2, false, false);
                auto p13 = __builtin_amdgcn_permlane32_swap(x1, x3, false, false);
                x0 = p02[0]; x2 = p02[1]; x1 = p13[0]; x3 = p13[1];
                auto q01 = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);
                auto q23 = __builtin_amdgcn_permlane16_swap(x2, x3, false, false);
                av[4 * c] = __uint_as_float(q01[0]); av[4 * c + 1] = __uint_as_float(q01[1]);
                av[4 * c + 2] = __uint_as_float(q23[0]); av[4 * c + 3] = __uint_as_float(q23[1]);
            }
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
    };
    load(a_cur, b_cur, 0);
    for (int k0 = 0; k0 < K; k0 += 2 * KC) {
        if (k0 + KC < K) load(a_nxt, b_nxt, k0 + KC);
        mfma(a_cur, b_cur);
        if (k0 + KC < K) {
            if (k0 + 2 * KC < K) load(a_cur, b_cur, k0 + 2 * KC);
            mfma(a_nxt, b_nxt);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) C[(int64_t)(m0 + 4 * kq + i) * N + n0 + r] = acc[i];
}

template <int MODE, int KC>
float run(const float* A, const float* W, float* C, int M, int N, int K, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = (M / 16) * (N / 16);
    k_skinny<MODE, KC><<<grid, 64>>>(A, W, C, M, N, K);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_skinny<MODE, KC><<<grid, 64>>>(A, W, C, M, N, K);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main() {
    const int Mmax = 2048, Nmax = 1024, Kmax = 4096;
    std::vector<float> hA((size_t)Mmax * Kmax), hW((size_t)Kmax * Nmax);
    srand(3);
    for (auto& v : hA) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hW) v = (float)rand() / RAND_MAX - 0.5f;
    float *A, *W, *C, *C2;
    CHECK(hipMalloc(&A, hA.size() * 4)); CHECK(hipMalloc(&W, hW.size() * 4));
    CHECK(hipMalloc(&C, (size_t)Mmax * Nmax * 4)); CHECK(hipMalloc(&C2, (size_t)Mmax * Nmax * 4));
    CHECK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    // exactness of the transposed-load variant against the dword variant
    {
        const int M = 256, N = 256, K = 1024;
        k_skinny<0, 128><<<(M / 16) * (N / 16), 64>>>(A, W, C, M, N, K);
        k_skinny<4, 128><<<(M / 16) * (N / 16), 64>>>(A, W, C2, M, N, K);
        std::vector<float> h1((size_t)M * N), h2((size_t)M * N);
        CHECK(hipMemcpy(h1.data(), C, h1.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h2.data(), C2, h2.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (size_t i = 0; i < h1.size(); ++i) bad += h1[i] != h2[i];
        printf("float4+swap variant vs dword variant: %d mismatches of %zu\n", bad, h1.size());
    }
    const int shapes[][3] = {{256, 256, 256}, {256, 256, 1024}, {256, 256, 4096}, {256, 1024, 256}, {1024, 256, 1024}, {64, 256, 1024}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        printf("M=%d N=%d K=%d:", M, N, K);
        printf("  dword %.1f us", run<0, 128>(A, W, C, M, N, K, 50));
        printf("  noA %.1f", run<1, 128>(A, W, C, M, N, K, 50));
        printf("  noW %.1f", run<2, 128>(A, W, C, M, N, K, 50));
        printf("  none %.1f", run<3, 128>(A, W, C, M, N, K, 50));
        printf("  A-float4 %.1f", run<4, 128>(A, W, C, M, N, K, 50));
        printf("  A-float4/KC64 %.1f", run<4, 64>(A, W, C, M, N, K, 50));
        printf("  dword/KC32 %.1f\n", run<0, 32>(A, W, C, M, N, K, 50));
    }
    return 0;
}
