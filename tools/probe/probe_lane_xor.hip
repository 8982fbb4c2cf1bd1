// Probe: lane ^ off exchanges without ds_bpermute (DPP for off = 1, 2, 4, 8; v_permlane16_swap / v_permlane32_swap for 16, 32)
// against __shfl_xor, and their timing in a 64-lane sum butterfly.
// Build: hipcc --offload-arch=gfx950 -O3 -o probe_lane_xor probe_lane_xor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int OFF> __device__ __forceinline__ float lane_xor(float v) {
    const int x = __float_as_int(v);
    if (OFF == 1) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    if (OFF == 2) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    if (OFF == 4) {
        int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false);   // row_shl:4 into banks 0, 2: lane i <- i + 4
        t = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);       // row_shr:4 into banks 1, 3: lane i <- i - 4
        return __int_as_float(t);
    }
    if (OFF == 8) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false));   // row_ror:8
    if (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        return __int_as_float((int)(((threadIdx.x >> 4) & 1) ? r[0] : r[1]));
    }
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    return __int_as_float((int)(((threadIdx.x >> 5) & 1) ? r[0] : r[1]));
}

__global__ void k_map(int* out) {
    const int l = threadIdx.x & 63;
    const float v = (float)l;
    out[0 * 64 + l] = (int)lane_xor<1>(v);
    out[1 * 64 + l] = (int)lane_xor<2>(v);
    out[2 * 64 + l] = (int)lane_xor<4>(v);
    out[3 * 64 + l] = (int)lane_xor<8>(v);
    out[4 * 64 + l] = (int)lane_xor<16>(v);
    out[5 * 64 + l] = (int)lane_xor<32>(v);
}

template <bool FAST>
__global__ void k_time(const float* in, float* out, int iters) {
    float v = in[threadIdx.x + blockIdx.x * blockDim.x];
    for (int it = 0; it < iters; ++it) {
        if (FAST) {
            v = v + lane_xor<32>(v); v = v + lane_xor<16>(v); v = v + lane_xor<8>(v);
            v = v + lane_xor<4>(v); v = v + lane_xor<2>(v); v = v + lane_xor<1>(v);
        } else {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
        }
        v = v * 0.015625f + 1.0f;
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

int main() {
    int* d;
    CHECK(hipMalloc(&d, 6 * 64 * 4));
    k_map<<<1, 64>>>(d);
    int h[6 * 64];
    CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const int offs[6] = {1, 2, 4, 8, 16, 32};
    int bad = 0;
    for (int q = 0; q < 6; ++q)
        for (int l = 0; l < 64; ++l)
            if (h[q * 64 + l] != (l ^ offs[q])) { if (bad < 10) printf("off %d lane %d reads %d\n", offs[q], l, h[q * 64 + l]); ++bad; }
    printf("lane maps: %d mismatches\n", bad);
    const int n = 256 * 1024;
    float *in, *o1, *o2;
    CHECK(hipMalloc(&in, n * 4)); CHECK(hipMalloc(&o1, n * 4)); CHECK(hipMalloc(&o2, n * 4));
    float* hin = (float*)malloc(n * 4);
    for (int i = 0; i < n; ++i) hin[i] = (float)rand() / RAND_MAX;
    CHECK(hipMemcpy(in, hin, n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int fast = 0; fast < 2; ++fast) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            if (fast) k_time<true><<<n / 256, 256>>>(in, o2, 200); else k_time<false><<<n / 256, 256>>>(in, o1, 200);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
        }
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.3f ms for 200 butterflies x %d waves\n", fast ? "dpp / permlane" : "__shfl_xor", ms, n / 64);
    }
    float *h1 = (float*)malloc(n * 4), *h2 = (float*)malloc(n * 4);
    CHECK(hipMemcpy(h1, o1, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h2, o2, n * 4, hipMemcpyDeviceToHost));
    int diff = 0;
    for (int i = 0; i < n; ++i) diff += h1[i] != h2[i];
    printf("butterfly results: %d of %d differ\n", diff, n);
    return 0;
}
