// Probe: fp32 MFMA issue rate (v_mfma_f32_32x32x2_f32) as a function of independent accumulators per wave and waves per SIMD,
// register operands only.  One workgroup per CU (256 workgroups), 4 * WPS waves each.  Second table: the same with operands
// that CHANGE from one MFMA to the next (64 B registers and 8 A registers of random values, as a kernel that keeps a weight slab
// in registers issues them): what the matrix pipe sustains on real data, at the clock the chip then holds.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__device__ __forceinline__ float rnd_val(unsigned x) {     // a float in [-1, 1) with a random mantissa
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

template <int NACC>
__global__ void k_data(float* out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float wv[64], av[8];
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 64; ++j) wv[j] = rnd_val(tid * 131u + j) * 0.0625f;
#pragma unroll
    for (int j = 0; j < 8; ++j) av[j] = rnd_val(tid * 977u + 100 + j);
    for (int i = 0; i < iters / 64; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j % 8], wv[j], acc[j % NACC], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[tid] = s;
}

template <int NACC>
void run_data(int wps, float* out) {
    const int iters = 20480;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_data<NACC>, dim3(256), dim3(256 * wps), 0, 0, out, 128);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k_data<NACC>, dim3(256), dim3(256 * wps), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 5 * 2.0 * 32 * 32 * 2 * (double)iters * 256.0 * 4 * wps;
    printf("changing operands: acc/wave %d  waves/SIMD %d : %.1f TFLOP/s\n", NACC, wps, flop / ms / 1e9);
}

template <int NACC>
void run(int wps, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256 * wps), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256 * wps), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 32 * 32 * 2 * (double)NACC * iters * 256.0 * 4 * wps;
    printf("acc/wave %d  waves/SIMD %d : %.1f TFLOP/s\n", NACC, wps, flop / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    for (int wps = 1; wps <= 4; ++wps) {
        run<1>(wps, out); run<2>(wps, out); run<4>(wps, out); run<8>(wps, out);
    }
    for (int wps = 1; wps <= 4; ++wps) {
        run_data<1>(wps, out); run_data<2>(wps, out); run_data<4>(wps, out);
    }
    return 0;
}
