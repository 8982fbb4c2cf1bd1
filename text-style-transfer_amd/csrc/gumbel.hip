// Row-wise operations on token distributions: the soft ("gumbel") decoding step and the mixture of an ensemble.
// Soft decoding step: out = softmax((softmax(logits) + G) / temperature) per row, with
// G = -log(-log(U + 1e-20) + 1e-20) from caller-provided uniform draws U.  One workgroup per row; the row
// is walked five times (max, sum, noisy scores + their max, exponentials + their sum, normalisation) with
// coalesced 4-byte accesses; after the first walk it is served from L2.  fp32 throughout, the same
// exp / log polynomials as the other kernels; results are compared with a tolerance, so the order of the
// workgroup reductions is free.
#include "tst_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum_butterfly(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(256) void gumbel_softmax_kernel(const float* __restrict__ x, int64_t ld, int V,
                                                             const float* __restrict__ u, int64_t ldu, float temperature,
                                                             float* __restrict__ out, int64_t ldo) {
    __shared__ float red[4];
    const float* xr = x + (int64_t)blockIdx.x * ld;
    const float* ur = u + (int64_t)blockIdx.x * ldu;
    float* yr = out + (int64_t)blockIdx.x * ldo;
    const int t = threadIdx.x;
    float m = -INFINITY;
    for (int i = t; i < V; i += 256) m = fmaxf(m, xr[i]);
    m = block_max(m, red);
    float s = 0.0f;
    for (int i = t; i < V; i += 256) s = s + d_exp_le0(xr[i] - m);
    s = block_sum(s, red);
    float ym = -INFINITY;
    for (int i = t; i < V; i += 256) {
        const float p = d_exp_le0(xr[i] - m) / s;
        const float g = -d_log(-d_log(ur[i] + 1e-20f) + 1e-20f);
        const float y = (p + g) / temperature;
        yr[i] = y;
        ym = fmaxf(ym, y);
    }
    ym = block_max(ym, red);
    float ys = 0.0f;
    for (int i = t; i < V; i += 256) {   // a thread re-reads only what it wrote itself
        const float e = d_exp_le0(yr[i] - ym);
        yr[i] = e;
        ys = ys + e;
    }
    ys = block_sum(ys, red);
    for (int i = t; i < V; i += 256) yr[i] = yr[i] / ys;
}

// out[v] = log(sum_i w[i] * softmax(x_i)[v]) for n_models logit rows x_i of one batch row: a weighted mixture of the
// members' distributions, returned as log-probabilities.  One workgroup per batch row; first the max and the sum of
// exponentials of every member (kept in LDS), then one pass that mixes, members in ascending order.
constexpr int MIX_MAX = 8;

__global__ __launch_bounds__(256) void mix_logprobs_kernel(const float* __restrict__ x, int64_t model_stride, int64_t ld,
                                                           int n_models, const float* __restrict__ w, int V,
                                                           float* __restrict__ out, int64_t ldo) {
    __shared__ float red[4];
    __shared__ float mm[MIX_MAX], ss[MIX_MAX], ww[MIX_MAX];
    const int t = threadIdx.x;
    const float* xr = x + (int64_t)blockIdx.x * ld;
    for (int i = 0; i < n_models; ++i) {
        const float* xi = xr + i * model_stride;
        float m = -INFINITY;
        for (int v = t; v < V; v += 256) m = fmaxf(m, xi[v]);
        m = block_max(m, red);
        float s = 0.0f;
        for (int v = t; v < V; v += 256) s = s + d_exp_le0(xi[v] - m);
        s = block_sum(s, red);
        if (t == 0) { mm[i] = m; ss[i] = s; ww[i] = w[i]; }
    }
    __syncthreads();
    float* yr = out + (int64_t)blockIdx.x * ldo;
    for (int v = t; v < V; v += 256) {
        float acc = 0.0f;
        for (int i = 0; i < n_models; ++i) acc = acc + ww[i] * (d_exp_le0(xr[i * model_stride + v] - mm[i]) / ss[i]);
        yr[v] = d_log(acc);
    }
}

}  // namespace

extern "C" int tst_mix_logprobs_f32(const float* logits, int64_t model_stride, int64_t ld, int64_t n_models,
                                    const float* weights, int64_t R, int64_t V, float* out, int64_t ldo, void* stream) {
    TST_REQUIRE(tst_dims_ok({n_models, R, V}), TST_ERR_UNSUPPORTED, "tst_mix_logprobs_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(logits && weights && out, TST_ERR_ARG, "tst_mix_logprobs_f32: null pointer");
    TST_REQUIRE(R >= 0 && V > 0 && ld >= V && ldo >= V && V < (1ll << 30) && R < (1ll << 31) && model_stride >= 0, TST_ERR_ARG,
                "tst_mix_logprobs_f32: bad sizes");
    TST_REQUIRE(n_models >= 1 && n_models <= MIX_MAX, TST_ERR_UNSUPPORTED, "tst_mix_logprobs_f32: 1 to %d models", MIX_MAX);
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(mix_logprobs_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, model_stride, ld,
                       (int)n_models, weights, (int)V, out, ldo);
    return tst_after_launch("tst_mix_logprobs_f32");
}

extern "C" int tst_gumbel_softmax_f32(const float* logits, int64_t ld, int64_t R, int64_t V, const float* u, int64_t ldu,
                                      float temperature, float* out, int64_t ldo, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, V}), TST_ERR_UNSUPPORTED, "tst_gumbel_softmax_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(logits && u && out, TST_ERR_ARG, "tst_gumbel_softmax_f32: null pointer");
    TST_REQUIRE(R >= 0 && V > 0 && ld >= V && ldu >= V && ldo >= V && V < (1ll << 30) && R < (1ll << 31), TST_ERR_ARG,
                "tst_gumbel_softmax_f32: bad sizes");
    TST_REQUIRE(temperature > 0.0f, TST_ERR_ARG, "tst_gumbel_softmax_f32: temperature must be positive");
    TST_REQUIRE(out != logits && out != u, TST_ERR_ARG, "tst_gumbel_softmax_f32: out must not alias an input");
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(gumbel_softmax_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, ld, (int)V, u, ldu,
                       temperature, out, ldo);
    return tst_after_launch("tst_gumbel_softmax_f32");
}
