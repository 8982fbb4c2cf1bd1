// Row-wise HBM-bound kernels: embedding gather + positional add, LayerNorm, GRU cell update, row gather.
// All of them move 16 bytes per lane per access; none has cross-workgroup reuse.
#include "tst_common.h"

namespace {

// one thread per float4 of the output
__global__ __launch_bounds__(256) void embed_posadd_kernel(const int32_t* __restrict__ tok, const float* __restrict__ emb,
                                                           const float* __restrict__ pe, const int32_t* __restrict__ pos,
                                                           float scale, float* __restrict__ x, int64_t R, int d4, int64_t V,
                                                           const float* __restrict__ z = nullptr, const int32_t* __restrict__ zidx = nullptr,
                                                           unsigned long long* __restrict__ clear_keys = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * d4) return;
    const int64_t r = i / d4;
    const int c4 = (int)(i - r * d4);
    if (clear_keys != nullptr && c4 < TST_ARGMAX_SLOTS) clear_keys[r * TST_ARGMAX_SLOTS + c4] = 0ull;   // the step's arg-max keys start at "no candidate"
    int64_t id = tok[r];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    float4 e = reinterpret_cast<const float4*>(emb + id * (int64_t)d4 * 4)[c4];
    float4 o;
    o.x = e.x * scale; o.y = e.y * scale; o.z = e.z * scale; o.w = e.w * scale;
    if (pe != nullptr) {
        const float4 p = reinterpret_cast<const float4*>(pe)[(pos != nullptr ? (int64_t)pos[r] * d4 : 0) + c4];
        o.x = o.x + p.x; o.y = o.y + p.y; o.z = o.z + p.z; o.w = o.w + p.w;
    }
    if (z != nullptr) {   // + z[zidx[r]]: the add_rows_kernel step, in the same order
        const float4 a = reinterpret_cast<const float4*>(z)[(int64_t)zidx[r] * d4 + c4];
        o.x = o.x + a.x; o.y = o.y + a.y; o.z = o.z + a.z; o.w = o.w + a.w;
    }
    reinterpret_cast<float4*>(x)[i] = o;
}

// x = x * scale + pe (one thread per float4): the tail of an embedding that was computed as a product
// (token distributions times the embedding matrix) instead of a row lookup
__global__ __launch_bounds__(256) void scale_posadd_kernel(float* __restrict__ x, float scale, const float* __restrict__ pe,
                                                           int64_t R, int d4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * d4) return;
    float4 o = reinterpret_cast<float4*>(x)[i];
    o.x = o.x * scale; o.y = o.y * scale; o.z = o.z * scale; o.w = o.w * scale;
    if (pe != nullptr) {
        const float4 p = reinterpret_cast<const float4*>(pe)[i % d4];
        o.x = o.x + p.x; o.y = o.y + p.y; o.z = o.z + p.z; o.w = o.w + p.w;
    }
    reinterpret_cast<float4*>(x)[i] = o;
}

__device__ __forceinline__ void store4(float* y, int64_t i4, const float4& o) { reinterpret_cast<float4*>(y)[i4] = o; }
__device__ __forceinline__ void store4(__bf16* y, int64_t i4, const float4& o) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 v;
    v[0] = (__bf16)o.x; v[1] = (__bf16)o.y; v[2] = (__bf16)o.z; v[3] = (__bf16)o.w;
    reinterpret_cast<bf16x4*>(y)[i4] = v;
}

// one wave per row; lane l owns the float4 chunks l, l + 64, ... (element c -> partial (c / 4) % 64);
// TO = float, or __bf16 when the result feeds a bf16 GEMM (statistics and affine map stay fp32, one rounding)
template <typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, float eps, TO* __restrict__ y,
                                                        int64_t R, int d4) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float4* xr = reinterpret_cast<const float4*>(x) + r * d4;
    float p = 0.0f;
    for (int c = lane; c < d4; c += 64) {
        const float4 v = xr[c];
        p = p + v.x; p = p + v.y; p = p + v.z; p = p + v.w;
    }
    const float d = (float)(d4 * 4);
    const float mean = wave_sum_butterfly(p) / d;
    p = 0.0f;
    for (int c = lane; c < d4; c += 64) {
        const float4 v = xr[c];
        float t;
        t = v.x - mean; p = __builtin_fmaf(t, t, p);
        t = v.y - mean; p = __builtin_fmaf(t, t, p);
        t = v.z - mean; p = __builtin_fmaf(t, t, p);
        t = v.w - mean; p = __builtin_fmaf(t, t, p);
    }
    const float var = wave_sum_butterfly(p) / d;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int c = lane; c < d4; c += 64) {
        const float4 v = xr[c];
        const float4 gg = reinterpret_cast<const float4*>(g)[c];
        const float4 bb = reinterpret_cast<const float4*>(b)[c];
        float4 o;
        o.x = __builtin_fmaf((v.x - mean) * rstd, gg.x, bb.x);
        o.y = __builtin_fmaf((v.y - mean) * rstd, gg.y, bb.y);
        o.z = __builtin_fmaf((v.z - mean) * rstd, gg.z, bb.z);
        o.w = __builtin_fmaf((v.w - mean) * rstd, gg.w, bb.w);
        store4(y, r * d4 + c, o);
    }
}

// The same for rows of 256 * CH floats (CH = 1, 2: the decoder sizes): the lane's CH chunks stay in registers, so a row is
// loaded once instead of three times and the two statistics passes have no memory latency between them.  Same operations in
// the same order as layernorm_kernel, hence the same bits.
template <typename TO, int CH>
__global__ __launch_bounds__(256) void layernorm_reg_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ b, float eps, TO* __restrict__ y, int64_t R) {
    constexpr int d4 = 64 * CH;
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float4* xr = reinterpret_cast<const float4*>(x) + r * d4;
    float4 v[CH], gg[CH], bb[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = xr[lane + 64 * c];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        gg[c] = reinterpret_cast<const float4*>(g)[lane + 64 * c];
        bb[c] = reinterpret_cast<const float4*>(b)[lane + 64 * c];
    }
    float p = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) { p = p + v[c].x; p = p + v[c].y; p = p + v[c].z; p = p + v[c].w; }
    const float d = (float)(d4 * 4);
    const float mean = wave_sum_butterfly(p) / d;
    p = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        float t;
        t = v[c].x - mean; p = __builtin_fmaf(t, t, p);
        t = v[c].y - mean; p = __builtin_fmaf(t, t, p);
        t = v[c].z - mean; p = __builtin_fmaf(t, t, p);
        t = v[c].w - mean; p = __builtin_fmaf(t, t, p);
    }
    const float var = wave_sum_butterfly(p) / d;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        float4 o;
        o.x = __builtin_fmaf((v[c].x - mean) * rstd, gg[c].x, bb[c].x);
        o.y = __builtin_fmaf((v[c].y - mean) * rstd, gg[c].y, bb[c].y);
        o.z = __builtin_fmaf((v[c].z - mean) * rstd, gg[c].z, bb[c].z);
        o.w = __builtin_fmaf((v[c].w - mean) * rstd, gg[c].w, bb[c].w);
        store4(y, r * d4 + lane + 64 * c, o);
    }
}

template <typename TO>
void launch_layernorm(const float* x, const float* gamma, const float* beta, float eps, TO* y, int64_t R, int64_t d, hipStream_t st) {
    const dim3 grid((unsigned)((R + 3) / 4)), block(256);
    if (d == 256) hipLaunchKernelGGL((layernorm_reg_kernel<TO, 1>), grid, block, 0, st, x, gamma, beta, eps, y, R);
    else if (d == 512) hipLaunchKernelGGL((layernorm_reg_kernel<TO, 2>), grid, block, 0, st, x, gamma, beta, eps, y, R);
    else hipLaunchKernelGGL(layernorm_kernel<TO>, grid, block, 0, st, x, gamma, beta, eps, y, R, (int)(d / 4));
}

// one thread per float4 of h.  gi_row (when not NULL): row r takes its input pre-activations from row gi_row[r] (clamped to
// 0 .. V - 1) of gi, a table with one row per vocabulary entry (ldgi4 float4 per row), instead of row r
__global__ __launch_bounds__(256) void gru_cell_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                       const float* h, float* h_out, int64_t R, int H4,
                                                       const int32_t* __restrict__ gi_row = nullptr, int V = 0, int64_t ldgi4 = 0) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * H4) return;
    const int64_t r = i / H4;
    const int c4 = (int)(i - r * H4);
    const float4* a = gi_row != nullptr ? reinterpret_cast<const float4*>(gi) + (int64_t)min(max(gi_row[r], 0), V - 1) * ldgi4
                                        : reinterpret_cast<const float4*>(gi) + r * 3 * H4;
    const float4* b = reinterpret_cast<const float4*>(gh) + r * 3 * H4;
    const float4 ir = a[c4], iz = a[H4 + c4], in = a[2 * H4 + c4];
    const float4 hr = b[c4], hz = b[H4 + c4], hn = b[2 * H4 + c4];
    const float4 hv = reinterpret_cast<const float4*>(h)[i];
    float4 o;
    o.x = gru_elem(ir.x, iz.x, in.x, hr.x, hz.x, hn.x, hv.x);
    o.y = gru_elem(ir.y, iz.y, in.y, hr.y, hz.y, hn.y, hv.y);
    o.z = gru_elem(ir.z, iz.z, in.z, hr.z, hz.z, hn.z, hv.z);
    o.w = gru_elem(ir.w, iz.w, in.w, hr.w, hz.w, hn.w, hv.w);
    reinterpret_cast<float4*>(h_out)[i] = o;
}

// x[r][:] += z[idx[r]][:]
__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ x, const float* __restrict__ z,
                                                       const int32_t* __restrict__ idx, int64_t R, int d4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * d4) return;
    const int64_t r = i / d4;
    const int c4 = (int)(i - r * d4);
    float4 v = reinterpret_cast<float4*>(x)[i];
    const float4 a = reinterpret_cast<const float4*>(z)[(int64_t)idx[r] * d4 + c4];
    v.x = v.x + a.x; v.y = v.y + a.y; v.z = v.z + a.z; v.w = v.w + a.w;
    reinterpret_cast<float4*>(x)[i] = v;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                          float* __restrict__ dst, int64_t R, int w4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * w4) return;
    const int64_t r = i / w4;
    const int c4 = (int)(i - r * w4);
    reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[(int64_t)idx[r] * w4 + c4];
}

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// dst[idx[r]][:] = src[r][:] (idx without repeats)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                           float* __restrict__ dst, int64_t R, int w4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * w4) return;
    const int64_t r = i / w4;
    const int c4 = (int)(i - r * w4);
    reinterpret_cast<float4*>(dst)[(int64_t)idx[r] * w4 + c4] = reinterpret_cast<const float4*>(src)[i];
}

// ---- packing of a padded source batch: only the non-padding positions become rows -------------------------------------
// one wave per sentence: number of non-padding tokens
__global__ __launch_bounds__(256) void pack_count_kernel(const int64_t* __restrict__ src, int64_t pad, int64_t B, int S,
                                                         int32_t* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int n = 0;
    for (int p0 = 0; p0 < S; p0 += 64) {
        const int p = p0 + lane;
        n += __popcll(__ballot(p < S && src[b * S + p] != pad));
    }
    if (lane == 0) count[b] = n;
}

// one workgroup: exclusive prefix sum of count[0..B) into offs, total into n_live
__global__ __launch_bounds__(1024) void pack_scan_kernel(const int32_t* __restrict__ count, int32_t* __restrict__ offs, int64_t B,
                                                         int32_t* __restrict__ n_live) {
    __shared__ int wsum[16];
    const int t = threadIdx.x;
    const int64_t per = (B + 1023) / 1024;
    const int64_t lo = (int64_t)t * per, hi = lo + per < B ? lo + per : B;
    int local = 0;
    for (int64_t i = lo; i < hi; ++i) local += count[i];
    int inc = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if ((t & 63) >= off) inc += o;
    }
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    __syncthreads();
    int wbase = 0;
    for (int q = 0; q < (t >> 6); ++q) wbase += wsum[q];
    int run = wbase + inc - local;
    for (int64_t i = lo; i < hi; ++i) { offs[i] = run; run += count[i]; }
    if (t == 1023) *n_live = wbase + inc;
}

// one wave per sentence: mask, row map and the packed rows' token, position, sentence and padded row
__global__ __launch_bounds__(256) void pack_write_kernel(const int64_t* __restrict__ src, int64_t pad, int64_t B, int S,
                                                         const int32_t* __restrict__ offs, uint8_t* __restrict__ mask,
                                                         int32_t* __restrict__ kvrow, int32_t* __restrict__ tok,
                                                         int32_t* __restrict__ pos, int32_t* __restrict__ sent,
                                                         int32_t* __restrict__ padded_row) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int at = offs[b];
    for (int p0 = 0; p0 < S; p0 += 64) {
        const int p = p0 + lane;
        const int64_t tk = p < S ? src[b * S + p] : pad;
        const bool live = p < S && tk != pad;
        const unsigned long long bal = __ballot(live);
        const int row = at + __popcll(bal & ((1ull << lane) - 1ull));
        if (p < S) {
            mask[b * S + p] = live ? 1 : 0;
            kvrow[b * S + p] = live ? row : -1;
        }
        if (live) {
            tok[row] = (int32_t)tk;
            pos[row] = p;
            sent[row] = (int32_t)b;
            padded_row[row] = (int32_t)(b * S + p);
        }
        at += __popcll(bal);
    }
}

}  // namespace

extern "C" int tst_embed_posadd_f32(const int32_t* tok, const float* emb, const float* pe, const int32_t* pos, float scale,
                                    float* x, int64_t R, int64_t d, int64_t V, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, d, V}), TST_ERR_UNSUPPORTED, "tst_embed_posadd_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(tok && emb && x && (pe || !pos), TST_ERR_ARG, "tst_embed_posadd_f32: null pointer");
    TST_REQUIRE(R >= 0 && d > 0 && d % 4 == 0 && V > 0, TST_ERR_ARG, "tst_embed_posadd_f32: need d %% 4 == 0, V > 0");
    TST_REQUIRE(tst_aligned16(emb) && tst_aligned16(x) && tst_aligned16(pe), TST_ERR_ARG, "tst_embed_posadd_f32: alignment");
    TST_REQUIRE(R * (d / 4) < (1ll << 31) * 256, TST_ERR_UNSUPPORTED, "tst_embed_posadd_f32: too large");
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(embed_posadd_kernel, dim3(blocks_for(R * (d / 4), 256)), dim3(256), 0, (hipStream_t)stream, tok, emb,
                       pe, pos, scale, x, R, (int)(d / 4), V);
    return tst_after_launch("tst_embed_posadd_f32");
}

extern "C" int tst_embed_posadd_rows_f32(const int32_t* tok, const float* emb, const float* pe, const int32_t* pos, float scale,
                                         const float* z, const int32_t* zidx, float* x, int64_t R, int64_t d, int64_t V, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, d, V}), TST_ERR_UNSUPPORTED, "tst_embed_posadd_rows_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(tok && emb && x && (pe || !pos) && z && zidx, TST_ERR_ARG, "tst_embed_posadd_rows_f32: null pointer");
    TST_REQUIRE(R >= 0 && d > 0 && d % 4 == 0 && V > 0, TST_ERR_ARG, "tst_embed_posadd_rows_f32: need d %% 4 == 0, V > 0");
    TST_REQUIRE(tst_aligned16(emb) && tst_aligned16(x) && tst_aligned16(pe) && tst_aligned16(z), TST_ERR_ARG, "tst_embed_posadd_rows_f32: alignment");
    TST_REQUIRE(R * (d / 4) < (1ll << 31) * 256, TST_ERR_UNSUPPORTED, "tst_embed_posadd_rows_f32: too large");
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(embed_posadd_kernel, dim3(blocks_for(R * (d / 4), 256)), dim3(256), 0, (hipStream_t)stream, tok, emb, pe, pos, scale,
                       x, R, (int)(d / 4), V, z, zidx);
    return tst_after_launch("tst_embed_posadd_rows_f32");
}

int tst_embed_posadd_clear_f32(const int32_t* tok, const float* emb, const float* pe, float scale, const float* z, const int32_t* zidx,
                               float* x, uint64_t* clear_keys, int64_t R, int64_t d, int64_t V, void* stream) {
    TST_REQUIRE(tok && emb && x && (z == nullptr || zidx != nullptr), TST_ERR_ARG, "tst_embed_posadd_clear_f32: null pointer");
    TST_REQUIRE(R >= 0 && d > 0 && d % 4 == 0 && V > 0 && (clear_keys == nullptr || d / 4 >= TST_ARGMAX_SLOTS), TST_ERR_ARG,
                "tst_embed_posadd_clear_f32: need d %% 4 == 0, V > 0 (and d >= 4 * TST_ARGMAX_SLOTS to clear keys)");
    TST_REQUIRE(tst_aligned16(emb) && tst_aligned16(x) && tst_aligned16(pe) && tst_aligned16(z) && ((uintptr_t)clear_keys & 7u) == 0, TST_ERR_ARG,
                "tst_embed_posadd_clear_f32: alignment");
    TST_REQUIRE(R * (d / 4) < (1ll << 31) * 256, TST_ERR_UNSUPPORTED, "tst_embed_posadd_clear_f32: too large");
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(embed_posadd_kernel, dim3(blocks_for(R * (d / 4), 256)), dim3(256), 0, (hipStream_t)stream, tok, emb, pe,
                       (const int32_t*)nullptr, scale, x, R, (int)(d / 4), V, z, zidx, (unsigned long long*)clear_keys);
    return tst_after_launch("tst_embed_posadd_clear_f32");
}

extern "C" int tst_scale_posadd_f32(float* x, float scale, const float* pe, int64_t R, int64_t d, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, d}), TST_ERR_UNSUPPORTED, "tst_scale_posadd_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(x, TST_ERR_ARG, "tst_scale_posadd_f32: null pointer");
    TST_REQUIRE(R >= 0 && d > 0 && d % 4 == 0, TST_ERR_ARG, "tst_scale_posadd_f32: need d %% 4 == 0");
    TST_REQUIRE(tst_aligned16(x) && tst_aligned16(pe), TST_ERR_ARG, "tst_scale_posadd_f32: alignment");
    TST_REQUIRE(R * (d / 4) < (1ll << 31) * 256, TST_ERR_UNSUPPORTED, "tst_scale_posadd_f32: too large");
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(scale_posadd_kernel, dim3(blocks_for(R * (d / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, scale, pe,
                       R, (int)(d / 4));
    return tst_after_launch("tst_scale_posadd_f32");
}

extern "C" int tst_layernorm_f32(const float* x, const float* gamma, const float* beta, float eps, float* y, int64_t R,
                                 int64_t d, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, d}), TST_ERR_UNSUPPORTED, "tst_layernorm_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(x && gamma && beta && y, TST_ERR_ARG, "tst_layernorm_f32: null pointer");
    TST_REQUIRE(R >= 0 && d > 0 && d % 4 == 0, TST_ERR_ARG, "tst_layernorm_f32: need d %% 4 == 0");
    TST_REQUIRE(tst_aligned16(x) && tst_aligned16(y) && tst_aligned16(gamma) && tst_aligned16(beta), TST_ERR_ARG,
                "tst_layernorm_f32: alignment");
    if (R == 0) return TST_OK;
    launch_layernorm<float>(x, gamma, beta, eps, y, R, d, (hipStream_t)stream);
    return tst_after_launch("tst_layernorm_f32");
}

extern "C" int tst_layernorm_f32_bf16(const float* x, const float* gamma, const float* beta, float eps, void* y, int64_t R,
                                      int64_t d, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, d}), TST_ERR_UNSUPPORTED, "tst_layernorm_f32_bf16: a size exceeds 2^31 - 1");
    TST_REQUIRE(x && gamma && beta && y, TST_ERR_ARG, "tst_layernorm_f32_bf16: null pointer");
    TST_REQUIRE(R >= 0 && d > 0 && d % 4 == 0, TST_ERR_ARG, "tst_layernorm_f32_bf16: need d %% 4 == 0");
    TST_REQUIRE(tst_aligned16(x) && tst_aligned16(y) && tst_aligned16(gamma) && tst_aligned16(beta), TST_ERR_ARG,
                "tst_layernorm_f32_bf16: alignment");
    if (R == 0) return TST_OK;
    launch_layernorm<__bf16>(x, gamma, beta, eps, (__bf16*)y, R, d, (hipStream_t)stream);
    return tst_after_launch("tst_layernorm_f32_bf16");
}

extern "C" int tst_add_rows_f32(float* x, const float* z, const int32_t* idx, int64_t R, int64_t d, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, d}), TST_ERR_UNSUPPORTED, "tst_add_rows_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(x && z && idx, TST_ERR_ARG, "tst_add_rows_f32: null pointer");
    TST_REQUIRE(R >= 0 && d > 0 && d % 4 == 0, TST_ERR_ARG, "tst_add_rows_f32: need d %% 4 == 0");
    TST_REQUIRE(tst_aligned16(x) && tst_aligned16(z), TST_ERR_ARG, "tst_add_rows_f32: alignment");
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(add_rows_kernel, dim3(blocks_for(R * (d / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, z, idx, R,
                       (int)(d / 4));
    return tst_after_launch("tst_add_rows_f32");
}

extern "C" int tst_gru_cell_step_tab_f32(const float* gi_tab, int64_t ld_tab, const int32_t* tok, int64_t V, const float* gh, const float* h,
                                         float* h_out, int64_t R, int64_t H, void* stream) {
    TST_REQUIRE(tst_dims_ok({V, R, H}), TST_ERR_UNSUPPORTED, "tst_gru_cell_step_tab_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(gi_tab && tok && gh && h && h_out, TST_ERR_ARG, "tst_gru_cell_step_tab_f32: null pointer");
    TST_REQUIRE(R >= 0 && H > 0 && H % 4 == 0 && V > 0 && V < (1ll << 31) && ld_tab >= 3 * H && ld_tab % 4 == 0, TST_ERR_ARG,
                "tst_gru_cell_step_tab_f32: need H %% 4 == 0, V > 0, ld_tab %% 4 == 0, ld_tab >= 3 H");
    TST_REQUIRE(tst_aligned16(gi_tab) && tst_aligned16(gh) && tst_aligned16(h) && tst_aligned16(h_out), TST_ERR_ARG,
                "tst_gru_cell_step_tab_f32: alignment");
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(gru_cell_kernel, dim3(blocks_for(R * (H / 4), 256)), dim3(256), 0, (hipStream_t)stream, gi_tab, gh, h, h_out, R,
                       (int)(H / 4), tok, (int)V, ld_tab / 4);
    return tst_after_launch("tst_gru_cell_step_tab_f32");
}

extern "C" int tst_gru_cell_step_f32(const float* gi, const float* gh, const float* h, float* h_out, int64_t R, int64_t H,
                                     void* stream) {
    TST_REQUIRE(tst_dims_ok({R, H}), TST_ERR_UNSUPPORTED, "tst_gru_cell_step_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(gi && gh && h && h_out, TST_ERR_ARG, "tst_gru_cell_step_f32: null pointer");
    TST_REQUIRE(R >= 0 && H > 0 && H % 4 == 0, TST_ERR_ARG, "tst_gru_cell_step_f32: need H %% 4 == 0");
    TST_REQUIRE(tst_aligned16(gi) && tst_aligned16(gh) && tst_aligned16(h) && tst_aligned16(h_out), TST_ERR_ARG,
                "tst_gru_cell_step_f32: alignment");
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(gru_cell_kernel, dim3(blocks_for(R * (H / 4), 256)), dim3(256), 0, (hipStream_t)stream, gi, gh, h,
                       h_out, R, (int)(H / 4));
    return tst_after_launch("tst_gru_cell_step_f32");
}

extern "C" int tst_gather_rows_f32(const float* src, const int32_t* idx, float* dst, int64_t R, int64_t width,
                                   void* stream) {
    TST_REQUIRE(tst_dims_ok({R, width}), TST_ERR_UNSUPPORTED, "tst_gather_rows_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(src && idx && dst && src != dst, TST_ERR_ARG, "tst_gather_rows_f32: null or aliased pointer");
    TST_REQUIRE(R >= 0 && width > 0 && width % 4 == 0, TST_ERR_ARG, "tst_gather_rows_f32: need width %% 4 == 0");
    TST_REQUIRE(tst_aligned16(src) && tst_aligned16(dst), TST_ERR_ARG, "tst_gather_rows_f32: alignment");
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_for(R * (width / 4), 256)), dim3(256), 0, (hipStream_t)stream, src,
                       idx, dst, R, (int)(width / 4));
    return tst_after_launch("tst_gather_rows_f32");
}

extern "C" int tst_scatter_rows_f32(const float* src, const int32_t* idx, float* dst, int64_t R, int64_t width, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, width}), TST_ERR_UNSUPPORTED, "tst_scatter_rows_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(src && idx && dst && src != dst, TST_ERR_ARG, "tst_scatter_rows_f32: null or aliased pointer");
    TST_REQUIRE(R >= 0 && width > 0 && width % 4 == 0, TST_ERR_ARG, "tst_scatter_rows_f32: need width %% 4 == 0");
    TST_REQUIRE(tst_aligned16(src) && tst_aligned16(dst), TST_ERR_ARG, "tst_scatter_rows_f32: alignment");
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(blocks_for(R * (width / 4), 256)), dim3(256), 0, (hipStream_t)stream, src, idx, dst, R,
                       (int)(width / 4));
    return tst_after_launch("tst_scatter_rows_f32");
}

extern "C" int tst_pack_sources(const int64_t* src, int64_t pad, int64_t B, int64_t S, uint8_t* mask, int32_t* kvrow, int32_t* tok,
                                int32_t* pos, int32_t* sent, int32_t* padded_row, int32_t* work, int32_t* n_live, void* stream) {
    TST_REQUIRE(tst_dims_ok({B, S}), TST_ERR_UNSUPPORTED, "tst_pack_sources: a size exceeds 2^31 - 1");
    TST_REQUIRE(src && mask && kvrow && tok && pos && sent && padded_row && work && n_live, TST_ERR_ARG, "tst_pack_sources: null pointer");
    TST_REQUIRE(B >= 0 && S > 0 && B * S < (1ll << 31), TST_ERR_ARG, "tst_pack_sources: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    int32_t* count = work;
    int32_t* offs = work + B;
    if (B > 0) hipLaunchKernelGGL(pack_count_kernel, dim3(blocks_for(B, 4)), dim3(256), 0, st, src, pad, B, (int)S, count);
    hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(1024), 0, st, count, offs, B, n_live);
    if (B > 0)
        hipLaunchKernelGGL(pack_write_kernel, dim3(blocks_for(B, 4)), dim3(256), 0, st, src, pad, B, (int)S, offs, mask, kvrow, tok, pos, sent,
                           padded_row);
    return tst_after_launch("tst_pack_sources");
}
