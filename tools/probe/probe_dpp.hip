// Prints which lane each lane reads under the DPP controls used for 16-lane butterflies.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL>
__device__ int dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
__global__ void k(int* out) {
    const int l = threadIdx.x;
    out[l] = dpp<0xB1>(l);          // quad_perm [1,0,3,2]
    out[64 + l] = dpp<0x4E>(l);     // quad_perm [2,3,0,1]
    out[128 + l] = dpp<0x141>(l);   // row_half_mirror
    out[192 + l] = dpp<0x140>(l);   // row_mirror
    // 16-lane sum butterfly through DPP
    float v = (float)(1 << (l & 15));
    v += __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false);
    out[256 + l] = (int)v;
}
int main() {
    int* d; hipMalloc(&d, 320 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"quad_perm[1,0,3,2]", "quad_perm[2,3,0,1]", "row_half_mirror", "row_mirror", "sum16"};
    for (int q = 0; q < 5; ++q) { printf("%-20s", names[q]); for (int l = 0; l < 20; ++l) printf(" %d", h[64 * q + l]); printf(" ... %d %d\n", h[64*q+62], h[64*q+63]); }
    return 0;
}
