// Measures what this GPU reaches on three micro-loops, to stand beside the vendor peaks in the bench line:
//   copy   : float4 device-to-device copy of a buffer far larger than the 256 MiB memory-side cache (read + written bytes / time)
//   read   : read-only pass over the same buffer
//   f32    : v_mfma_f32_32x32x2_f32 back to back, operands in registers, four independent accumulators per wave
//   bf16   : v_mfma_f32_32x32x16_bf16, same structure
// Operands are pseudo-random (the chip holds a lower clock on random data than on zeros).  Prints one JSON line.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/peaks tools/peaks.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one 16-byte element per thread, nontemporal load and store, a grid of n / 256 workgroups: the fastest of the shapes tried
// (tools/probe/copy_variants.hip: 6.45 TB/s; workgroups that copy 16-32 KiB pieces 5.4-5.8, grid-stride loops 4.5-4.9,
// hipMemcpyAsync 4.9)
__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// read-only pass (sum of every element): what a kernel that only reads, like the selection pass over the logits, can reach
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ src, float* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
        s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
    }
    const float s = s0.x + s0.y + s0.z + s0.w + s1.x + s1.y + s1.z + s1.w + s2.x + s2.y + s2.z + s2.w + s3.x + s3.y + s3.z + s3.w;
    if (s == 12345.678f) out[0] = s;   // keeps the loads alive
}

// the same as one workgroup per contiguous 128 KB row (the access pattern of the selection pass): 8 independent 16-byte loads
// per thread and trip, rows handed to workgroups in launch order
__global__ __launch_bounds__(256) void read_rows_kernel(const float4* __restrict__ src, float* __restrict__ out, int64_t rows) {
    constexpr int ROW4 = 8192;   // float4 per row
    float4 s0 = {0, 0, 0, 0}, s1 = s0;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float4* p = src + r * ROW4 + threadIdx.x;
#pragma unroll
        for (int c = 0; c < ROW4 / 256; c += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(c + u) * 256];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                s0.x += v[u].x; s0.y += v[u].y; s0.z += v[u].z; s0.w += v[u].w;
                s1.x += v[u + 1].x; s1.y += v[u + 1].y; s1.z += v[u + 1].z; s1.w += v[u + 1].w;
            }
        }
    }
    const float s = s0.x + s0.y + s0.z + s0.w + s1.x + s1.y + s1.z + s1.w;
    if (s == 12345.678f) out[0] = s;
}

__device__ inline float lane_rand(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(x & 0xffff) / 65536.0f - 0.5f;
}

__global__ __launch_bounds__(256) void mfma_f32_kernel(float* out, int iters) {
    const uint32_t id = blockIdx.x * 256 + threadIdx.x;
    const float a0 = lane_rand(id * 2 + 1), b0 = lane_rand(id * 2 + 2);
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, b0, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
    out[id] = s;
}

__global__ __launch_bounds__(256) void mfma_bf16_kernel(float* out, int iters) {
    const uint32_t id = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)lane_rand(id * 16 + j);
        b[j] = (__bf16)lane_rand(id * 16 + 8 + j);
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
    out[id] = s;
}

template <typename F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;

    const int64_t bytes = (int64_t)2 << 30;   // 2 GiB each way
    float4 *src, *dst;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMalloc(&dst, bytes));
    CHECK(hipMemset(src, 1, bytes));
    const int64_t n = bytes / 16;
    const float copy_ms = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const f32x4*)src, (f32x4*)dst, n); }, 10);
    const double copy_gbs = 2.0 * bytes / (copy_ms * 1e-3) / 1e9;
    const float read_ms = time_ms([&] { hipLaunchKernelGGL(read_kernel, dim3(cus * 16), dim3(256), 0, 0, src, (float*)dst, n); }, 10);
    const int64_t rows = bytes / (8192 * 16);
    const float rows_ms = time_ms([&] { hipLaunchKernelGGL(read_rows_kernel, dim3((unsigned)rows), dim3(256), 0, 0, src, (float*)dst, rows); }, 10);
    const double read_gbs = (double)bytes / ((read_ms < rows_ms ? read_ms : rows_ms) * 1e-3) / 1e9;   // the better of the two patterns
    CHECK(hipFree(src));
    CHECK(hipFree(dst));

    const int blocks = cus * 2;   // two 4-wave workgroups per CU: two waves per SIMD
    float* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const int iters = 20000;
    const double waves = (double)blocks * 4;
    const float f32_ms = time_ms([&] { hipLaunchKernelGGL(mfma_f32_kernel, dim3(blocks), dim3(256), 0, 0, out, iters); }, 3);
    const double f32_tf = waves * iters * 4 * (2.0 * 32 * 32 * 2) / (f32_ms * 1e-3) / 1e12;
    const int iters16 = 40000;
    const float bf16_ms = time_ms([&] { hipLaunchKernelGGL(mfma_bf16_kernel, dim3(blocks), dim3(256), 0, 0, out, iters16); }, 3);
    const double bf16_tf = waves * iters16 * 4 * (2.0 * 32 * 32 * 16) / (bf16_ms * 1e-3) / 1e12;
    CHECK(hipFree(out));

    printf("{\"device\": \"%s\", \"compute_units\": %d, \"copy_GBps\": %.1f, \"read_GBps\": %.1f, \"copy_bytes_each_way\": %lld, "
           "\"mfma_f32_TFLOPs\": %.1f, \"mfma_bf16_TFLOPs\": %.1f, "
           "\"method\": \"float4 copy kernel (read + written bytes, nontemporal, one element per thread) and read-only sum kernels (grid-stride, and one workgroup per contiguous 128 KB row with 8 loads in flight: the better of the two); v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16 loops, "
           "register operands (pseudo-random), 4 accumulators per wave, 2 waves per SIMD\"}\n",
           prop.name, cus, copy_gbs, read_gbs, (long long)bytes, f32_tf, bf16_tf);
    return 0;
}
