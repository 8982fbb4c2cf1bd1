// Probe: what takes the fp32 matrix pipe from the ~146 TFLOP/s of a register-only MFMA stream (tools/probe/mfma_rate.hip,
// one accumulator per wave, two waves per SIMD) to the ~120 TFLOP/s at which every real fp32 product kernel ends?  The loop of
// the weights-in-registers kernel (128 B operands in registers, one dependent chain of 128 v_mfma_f32_32x32x2_f32 per "tile")
// is rebuilt piece by piece: A operands from registers / from LDS (one ds_read_b128 per four MFMAs, two groups ahead), a
// workgroup barrier per tile, LDS writes per tile, global loads per tile, global stores per tile.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_feed.hip -o tools/probe/bin/mfma_feed && tools/probe/bin/mfma_feed
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float rnd_val(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

constexpr int LDA = 260;

// FEED: 0 = A operands in registers, 1 = from LDS.  BAR: workgroup barrier per tile.  LW: 4 ds_write_b128 per tile.
// GL: 4 global_load_dwordx4 per tile (consumed by the LDS writes when LW).  GS: 16 global_store_dword per tile.
template <int FEED, int BAR, int LW, int GL, int GS>
__global__ __launch_bounds__(512, 1) void k(const float* __restrict__ src, float* __restrict__ dst, float* out, int tiles) {
    __shared__ __attribute__((aligned(16))) float As[2][32 * LDA];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5;
    const unsigned tid = blockIdx.x * blockDim.x + t;
    float w[128];
#pragma unroll
    for (int j = 0; j < 128; ++j) w[j] = rnd_val(tid * 131u + j) * 0.0625f;
    for (int i = t; i < 2 * 32 * LDA; i += 512) (&As[0][0])[i] = rnd_val(tid * 7u + i);
    __syncthreads();
    float4 av[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) av[j] = make_float4(rnd_val(tid + j), rnd_val(tid + 10 + j), rnd_val(tid + 20 + j), rnd_val(tid + 30 + j));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 st[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    int buf = 0;
    const float* gsrc = src + ((size_t)blockIdx.x * 32 + wave * 4) * 256 + 4 * lane;
    float* gdst = dst + ((size_t)blockIdx.x * 32 + 4 * h) * 256 + wave * 32 + (lane & 31);
    for (int tile = 0; tile < tiles; ++tile) {
        // the loads sit in a block of their own (as in the real kernel, where the last tile requests nothing): written
        // unconditionally the compiler sinks every load to its use behind the MFMA loop and the tile ends with exposed round trips
        if (GL && tile + 1 < tiles) {
#pragma unroll
            for (int i = 0; i < 4; ++i) st[i] = *reinterpret_cast<const float4*>(gsrc + ((size_t)(tile & 7) * 8192 * 32 + i) * 256);
        }
        const float* ap = &As[buf][(lane & 31) * LDA + 4 * h];
        float4 a0, a1, a2;
        if (FEED) { a0 = *reinterpret_cast<const float4*>(ap); a1 = *reinterpret_cast<const float4*>(ap + 8); a2 = a1; }
        else { a0 = av[0]; a1 = av[1]; a2 = av[2]; }
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            if (FEED) { if (q + 2 < 32) a2 = *reinterpret_cast<const float4*>(ap + (q + 2) * 8); }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, w[4 * q + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, w[4 * q + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, w[4 * q + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, w[4 * q + 3], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (FEED) { a0 = a1; a1 = a2; } else { const float4 tmp = a0; a0 = a1; a1 = a2; a2 = tmp; }
        }
        if (LW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&As[buf ^ 1][(wave * 4 + i) * LDA + 4 * lane]) = GL ? st[i] : av[i % 3];
        }
        if (GS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gdst[((size_t)(tile & 7) * 8192 * 32 + (r & 3) + 8 * (r >> 2)) * 256] = acc[r];
        }
        if (BAR) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (LW) buf ^= 1;
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[tid] = s + st[0].x;
}

template <int FEED, int BAR, int LW, int GL, int GS>
void run(const char* what, const float* src, float* dst, float* out) {
    const int tiles = 160;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FEED, BAR, LW, GL, GS>), dim3(256), dim3(512), 0, 0, src, dst, out, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<FEED, BAR, LW, GL, GS>), dim3(256), dim3(512), 0, 0, src, dst, out, tiles);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 5 * 2.0 * 32 * 32 * 2 * 128.0 * tiles * 256.0 * 8;
    printf("%-78s %.1f TFLOP/s\n", what, flop / ms / 1e9);
}

int main() {
    float *src, *dst, *out;
    const size_t n = (size_t)8 * 8192 * 32 * 256 + 256 * 32 * 256;
    hipMalloc(&src, n * 4); hipMalloc(&dst, n * 4); hipMalloc(&out, 256 * 512 * 4);
    hipMemset(src, 0, n * 4);
    run<0, 0, 0, 0, 0>("A operands in registers", src, dst, out);
    run<1, 0, 0, 0, 0>("A operands from LDS (ds_read_b128 per 4 MFMAs, two groups ahead)", src, dst, out);
    run<1, 1, 0, 0, 0>("  + workgroup barrier per 128 MFMAs", src, dst, out);
    run<1, 1, 1, 0, 0>("  + barrier + 4 ds_write_b128 per tile", src, dst, out);
    run<1, 1, 1, 1, 0>("  + barrier + LDS writes + 4 global_load_dwordx4 per tile", src, dst, out);
    run<1, 1, 1, 1, 1>("  + barrier + LDS writes + loads + 16 global_store_dword per tile", src, dst, out);
    run<1, 0, 0, 0, 1>("A from LDS + 16 global_store_dword per tile only", src, dst, out);
    run<0, 0, 0, 0, 1>("A in registers + 16 global_store_dword per tile only", src, dst, out);
    run<0, 1, 0, 0, 0>("A in registers + barrier per tile only", src, dst, out);
    return 0;
}
