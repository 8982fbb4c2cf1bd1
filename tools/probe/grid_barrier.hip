// Probe: cost of a grid-wide barrier (one atomic counter in device memory, every workgroup resident) on this GPU, as a
// function of the number of workgroups: what a decode step written as ONE kernel would pay between its phases.
// Every spin is bounded: a workgroup that does not see the others within SPIN_MAX polls sets a flag and leaves, and so does
// everybody after it, so the kernel always ends.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int SPIN_MAX = 1 << 22;

// MODE 0: one counter, system-scope builtins (__atomic_*); 1: one counter, agent scope; 2: agent scope, two levels (a counter per
// XCD, blockIdx & 7, whose last arrival adds to the global counter that everybody polls)
template <int MODE>
__global__ __launch_bounds__(256) void barrier_loop(unsigned* counter, int* failed, float* sink, int rounds, int work) {
    const unsigned nwg = gridDim.x;
    float acc = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < work; ++i) acc = acc * 1.0001f + 0.5f;     // a little per-phase work
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned target;
            if (MODE == 0) {
                __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);
                target = (unsigned)(r + 1) * nwg;
            } else if (MODE == 1) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                target = (unsigned)(r + 1) * nwg;
            } else {
                const unsigned per_xcd = nwg >> 3;
                unsigned* mine = counter + 64 + 64 * (blockIdx.x & 7);   // counters 256 bytes apart
                const unsigned old = __hip_atomic_fetch_add(mine, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == (unsigned)(r + 1) * per_xcd) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                target = (unsigned)(r + 1) * 8;
            }
            int spins = 0;
            while ((MODE == 0 ? __atomic_load_n(counter, __ATOMIC_ACQUIRE)
                              : __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < target) {
                if (++spins > SPIN_MAX || __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned* counter; int* failed; float* sink;
    CHECK(hipMalloc(&counter, 4096)); CHECK(hipMalloc(&failed, 4)); CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int rounds = 2000;
    for (int mode = 0; mode < 3; ++mode)
    for (int per_cu = 1; per_cu <= 2; ++per_cu)
        for (int work : {0, 2000}) {
            const int nwg = cus * per_cu;
            float best = 1e30f;
            int fail = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemset(counter, 0, 4096)); CHECK(hipMemset(failed, 0, 4));
                CHECK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(barrier_loop<0>, dim3(nwg), dim3(256), 0, 0, counter, failed, sink, rounds, work);
                else if (mode == 1) hipLaunchKernelGGL(barrier_loop<1>, dim3(nwg), dim3(256), 0, 0, counter, failed, sink, rounds, work);
                else hipLaunchKernelGGL(barrier_loop<2>, dim3(nwg), dim3(256), 0, 0, counter, failed, sink, rounds, work);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipMemcpy(&fail, failed, 4, hipMemcpyDeviceToHost));
                if (ms < best) best = ms;
            }
            printf("{\"mode\": %d, \"workgroups\": %d, \"per_cu\": %d, \"work_iterations\": %d, \"rounds\": %d, \"us_per_round\": %.3f, \"failed\": %d}\n", mode, nwg, per_cu,
                   work, rounds, best * 1e3 / rounds, fail);
        }
    return 0;
}
