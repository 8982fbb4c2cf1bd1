// Probe: device-to-device copy rate (read + written bytes) of a 2 GiB buffer for a few kernel shapes, to choose the one
// tools/peaks.hip reports.   hipcc --offload-arch=gfx950 -O3 -w tools/probe/copy_variants.hip -o abl_out/copy_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void one_per_thread(const f32x4* __restrict__ s, f32x4* __restrict__ d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const f32x4 v = NT_LD ? __builtin_nontemporal_load(s + i) : s[i];
        if (NT_ST) __builtin_nontemporal_store(v, d + i); else d[i] = v;
    }
}
// a workgroup copies a contiguous piece of U * 4 KiB, U loads in flight per thread
template <int U, bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void chunk_per_block(const f32x4* __restrict__ s, f32x4* __restrict__ d, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * (256 * U) + threadIdx.x;
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT_LD ? __builtin_nontemporal_load(s + base + u * 256) : s[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT_ST) __builtin_nontemporal_store(v[u], d + base + u * 256); else d[base + u * 256] = v[u]; }
}
// grid-stride with U loads in flight
template <int U, bool NT>
__global__ __launch_bounds__(256) void grid_stride(const f32x4* __restrict__ s, f32x4* __restrict__ d, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
    }
}
template <typename F> float ms_of(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 10; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    const int64_t bytes = (int64_t)2 << 30, n = bytes / 16;
    f32x4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes);
#define R(name, call) printf("%-46s %.0f GB/s\n", name, 2.0 * bytes / (ms_of([&] { call; }) * 1e-3) / 1e9)
    R("one float4 per thread", (one_per_thread<false, false><<<dim3(n / 256), dim3(256)>>>(s, d, n)));
    R("one float4 per thread, nt stores", (one_per_thread<false, true><<<dim3(n / 256), dim3(256)>>>(s, d, n)));
    R("one float4 per thread, nt loads + stores", (one_per_thread<true, true><<<dim3(n / 256), dim3(256)>>>(s, d, n)));
    R("16 KiB per workgroup (4 in flight)", (chunk_per_block<4, false, false><<<dim3(n / 1024), dim3(256)>>>(s, d, n)));
    R("16 KiB per workgroup, nt stores", (chunk_per_block<4, false, true><<<dim3(n / 1024), dim3(256)>>>(s, d, n)));
    R("16 KiB per workgroup, nt both", (chunk_per_block<4, true, true><<<dim3(n / 1024), dim3(256)>>>(s, d, n)));
    R("32 KiB per workgroup (8 in flight), nt both", (chunk_per_block<8, true, true><<<dim3(n / 2048), dim3(256)>>>(s, d, n)));
    R("32 KiB per workgroup (8 in flight)", (chunk_per_block<8, false, false><<<dim3(n / 2048), dim3(256)>>>(s, d, n)));
    R("grid-stride 4096 wg, 4 in flight, nt", (grid_stride<4, true><<<dim3(4096), dim3(256)>>>(s, d, n)));
    R("grid-stride 4096 wg, 4 in flight", (grid_stride<4, false><<<dim3(4096), dim3(256)>>>(s, d, n)));
    R("grid-stride 2048 wg, 8 in flight, nt", (grid_stride<8, true><<<dim3(2048), dim3(256)>>>(s, d, n)));
    R("grid-stride 8192 wg, 2 in flight", (grid_stride<2, false><<<dim3(8192), dim3(256)>>>(s, d, n)));
    float ms = ms_of([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); });
    printf("%-46s %.0f GB/s\n", "hipMemcpyAsync device to device", 2.0 * bytes / (ms * 1e-3) / 1e9);
    return 0;
}
