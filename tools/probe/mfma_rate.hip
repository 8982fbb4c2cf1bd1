// Probe: fp32 MFMA issue rate (v_mfma_f32_32x32x2_f32) as a function of independent accumulators per wave and waves per SIMD,
// register operands only.  One workgroup per CU (256 workgroups), 4 * WPS waves each.
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
    return 0;
}
