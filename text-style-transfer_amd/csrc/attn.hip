// Single-query attention for one decode step: one wave per (row, head).
//
// Memory-bound: per (row, head) the wave streams n key rows and n value rows of dh floats, each a
// contiguous 4*dh-byte segment (self-attention finds them through the per-row ancestor table, so a beam
// reorder never moves the cache).  Scores use 16 lanes per key position (4 positions per wave pass,
// float4 per lane, fma chain then a 16-lane butterfly); the softmax keeps one score per lane (n <= 256);
// the value sum runs over positions in ascending order with one fma chain per output element.
// This fixes the arithmetic order completely (see tst_common.h).
#include "tst_common.h"
#include <math.h>

namespace {

constexpr int MAXP = 256;  // positions per row supported by the one-score-per-lane softmax

// 4 consecutive elements as floats (fp32: one 16-byte load; bf16: one 8-byte load, widened exactly)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __bf16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 ld2(const __bf16* p) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
    return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }

template <typename T, bool SELF, int NCH, int VEC>
__global__ __launch_bounds__(256) void attn_step_kernel(
    // SELF: cache [n_pos][slab_rows][3d], anc; else q [R][d], kv [n_sent][S][2d], mask, sent
    const T* __restrict__ base, int64_t slab_stride, const int32_t* __restrict__ anc, int64_t anc_ld,
    const T* __restrict__ qbuf, const uint8_t* __restrict__ mask, const int32_t* __restrict__ sent, int n, T* __restrict__ out,
    int64_t R, int d, int heads, int dh, float scale) {
    __shared__ float sc[4][MAXP];   // score, then softmax weight, per position
    __shared__ int slot[4][MAXP];   // SELF: cache row of every position; else: the list of unmasked positions
    __shared__ int kvr[4][MAXP];    // cross: the k|v row of every listed position
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t gw = (int64_t)blockIdx.x * 4 + w;
    const int64_t total = R * heads;
    if (gw >= total) gw = total - 1;  // surplus waves redo the last (row, head): same values, no early exit before barriers
    const int64_t r = gw / heads;
    const int h = (int)(gw - r * heads);
    const int pos = n - 1;

    // Only unmasked ("live") positions are ever loaded: a masked position has weight exactly 0 and
    // fma(0, v, acc) == acc, so skipping it changes nothing.  nl = number of live positions (wave-uniform).
    const T* qp;
    int64_t ent = 0;
    int nl;
    if (SELF) {
        for (int p = lane; p < n; p += 64) slot[w][p] = (p < pos) ? anc[r * anc_ld + p] : (int)r;
        qp = base + (int64_t)pos * slab_stride + r * 3 * d + h * dh;
        nl = n;
    } else {
        ent = sent[r];
        qp = qbuf + r * d + h * dh;
        nl = 0;
        for (int p0 = 0; p0 < n; p0 += 64) {
            const int p = p0 + lane;
            const bool live = p < n && (mask == nullptr || mask[ent * n + p] != 0);
            const unsigned long long bal = __ballot(live);
            if (live) {
                const int at = nl + __popcll(bal & ((1ull << lane) - 1ull));
                slot[w][at] = p;
                kvr[w][at] = anc != nullptr ? anc[ent * n + p] : (int)(ent * n + p);   // anc: optional row map of a packed k|v
            } else if (p < n) sc[w][p] = -INFINITY;
            nl += __popcll(bal);
        }
    }
    const int j = lane & 15, g = lane >> 4;
    float4 qv[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = j + 16 * c;
        qv[c] = (ch * 4 < dh) ? ld4(qp + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    // key row of the i-th live position; the value row follows d elements later
    auto krow = [&](int i) -> const T* {
        if (SELF) return base + (int64_t)i * slab_stride + (int64_t)slot[w][i] * 3 * d + d + h * dh;
        return base + (int64_t)kvr[w][i] * 2 * d + h * dh;
    };

    // scores: 16 lanes per position, 4 positions per pass, 4 passes in flight (independent loads)
    for (int i0 = 0; i0 < nl; i0 += 16) {
        // three phases, so that the pass costs one LDS wait and one memory wait instead of one of each per position:
        // the rows of the 4 positions of this lane group, then all their loads, then the dot products
        const T* kp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) kp[u] = krow(min(i0 + 4 * u + g, nl - 1));
        float4 kv[4][NCH];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = j + 16 * c;
                kv[u][c] = ld4(kp[u] + 4 * ((ch * 4 < dh) ? ch : 0));   // always a load (no branch); unused when beyond the head
            }
        float part[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = j + 16 * c;
                if (ch * 4 < dh) {
                    acc = __builtin_fmaf(qv[c].x, kv[u][c].x, acc);
                    acc = __builtin_fmaf(qv[c].y, kv[u][c].y, acc);
                    acc = __builtin_fmaf(qv[c].z, kv[u][c].z, acc);
                    acc = __builtin_fmaf(qv[c].w, kv[u][c].w, acc);
                }
            }
            part[u] = acc;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 4 * u + g;
            const float tot = group16_sum_butterfly(part[u]);
            if (j == 0 && i < nl) sc[w][SELF ? i : slot[w][i]] = tot * scale;
        }
    }
    __syncthreads();

    // softmax weights, one score per lane and 64-position block
    float s[MAXP / 64];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXP / 64; ++i) {
        const int p = lane + 64 * i;
        s[i] = p < n ? sc[w][p] : -INFINITY;
        m = fmaxf(m, s[i]);
    }
    m = wave_max(m);
    float part = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXP / 64; ++i) {
        const int p = lane + 64 * i;
        const float e = p < n ? d_exp_le0(s[i] - m) : 0.0f;
        s[i] = e;
        part = part + e;
    }
    const float sum = wave_sum_butterfly(part);
#pragma unroll
    for (int i = 0; i < MAXP / 64; ++i) {
        const int p = lane + 64 * i;
        if (p < n) sc[w][p] = s[i] / sum;
    }
    __syncthreads();

    // weighted value sum over the live positions in ascending order, 8 loads in flight per lane
    T* orow = out + r * d + h * dh;
    for (int c0 = lane * VEC; c0 < dh; c0 += 64 * VEC) {
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.0f;
        for (int i0 = 0; i0 < nl; i0 += 8) {
            float wgt[8], v[8][VEC];
            const T* vrow[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vrow[u] = krow(min(i0 + u, nl - 1)) + d + c0;   // all LDS reads of the pass first
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u, nl - 1);
                const float wv = sc[w][SELF ? i : slot[w][i]];   // read unconditionally (no branch per position)
                wgt[u] = i0 + u < nl ? wv : 0.0f;                // beyond the list: weight 0, no effect
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const T* vp = vrow[u];
                if (VEC == 4) {
                    const float4 x = ld4(vp);
                    v[u][0] = x.x; v[u][1 % VEC] = x.y; v[u][2 % VEC] = x.z; v[u][3 % VEC] = x.w;
                } else if (VEC == 2) {
                    const float2 x = ld2(vp);
                    v[u][0] = x.x; v[u][1 % VEC] = x.y;
                } else {
                    v[u][0] = ld1(vp);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = __builtin_fmaf(wgt[u], v[u][e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) st1(orow + c0 + e, acc[e]);
    }
}

__device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4(__bf16* p, const float (&v)[4]) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 o;
    o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}

// Heads of 64 elements: one wave per (row, 4 adjacent heads), 16 lanes per head, lane j of a head owns elements
// 4 j .. 4 j + 3.  The 4 heads of a cache row are adjacent in memory, so every key or value access of the wave
// is one contiguous 256-element segment (the one-wave-per-head kernel above would read a 64-element head with
// one element per lane in its value pass).  Same arithmetic, in the same order, as that kernel: a score is the
// 16-partial butterfly sum; the softmax sum keeps the 64 interleaved partials (position p -> partial p % 64),
// four per lane, and folds them exactly as the 64-lane xor-butterfly would (32, 16 inside the lane; 8, 4, 2, 1
// across the 16 lanes); the value sum runs over the positions in ascending order.
template <typename T, bool SELF>
__global__ __launch_bounds__(256) void attn_step_h64_kernel(
    const T* __restrict__ base, int64_t slab_stride, const int32_t* __restrict__ anc, int64_t anc_ld,
    const T* __restrict__ qbuf, const uint8_t* __restrict__ mask, const int32_t* __restrict__ sent, int n, T* __restrict__ out,
    int64_t R, int d, int heads, float scale) {
    __shared__ float sc[4][4][MAXP];   // [wave][head of the group][position]: score, then softmax weight
    __shared__ int slot[4][MAXP];      // SELF: cache row of every position; else: the list of unmasked positions
    __shared__ int kvr[4][MAXP];       // cross: the k|v row of every listed position
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hgroups = heads >> 2;
    int64_t gw = (int64_t)blockIdx.x * 4 + w;
    const int64_t total = R * hgroups;
    if (gw >= total) gw = total - 1;   // surplus waves redo the last item: same values, no early exit before barriers
    const int64_t r = gw / hgroups;
    const int g = lane >> 4, j = lane & 15;
    const int eoff = ((int)(gw - r * hgroups) * 4 + g) * 64 + 4 * j;   // this lane's 4 elements inside a row of d
    const int pos = n - 1;

    const T* qp;
    int64_t ent = 0;
    int nl;
    if (SELF) {
        for (int p = lane; p < n; p += 64) slot[w][p] = (p < pos) ? anc[r * anc_ld + p] : (int)r;
        qp = base + (int64_t)pos * slab_stride + r * 3 * d + eoff;
        nl = n;
    } else {
        ent = sent[r];
        qp = qbuf + r * d + eoff;
        nl = 0;
        for (int p0 = 0; p0 < n; p0 += 64) {
            const int p = p0 + lane;
            const bool live = p < n && (mask == nullptr || mask[ent * n + p] != 0);
            const unsigned long long bal = __ballot(live);
            if (live) {
                const int at = nl + __popcll(bal & ((1ull << lane) - 1ull));
                slot[w][at] = p;
                kvr[w][at] = anc != nullptr ? anc[ent * n + p] : (int)(ent * n + p);
            } else if (p < n) { sc[w][0][p] = -INFINITY; sc[w][1][p] = -INFINITY; sc[w][2][p] = -INFINITY; sc[w][3][p] = -INFINITY; }
            nl += __popcll(bal);
        }
    }
    const float4 qv = ld4(qp);
    __syncthreads();

    auto krow = [&](int i) -> const T* {   // this lane's 4 key elements of the i-th live position; values d further
        if (SELF) return base + (int64_t)i * slab_stride + (int64_t)slot[w][i] * 3 * d + d + eoff;
        return base + (int64_t)kvr[w][i] * 2 * d + eoff;
    };

    for (int i0 = 0; i0 < nl; i0 += 8) {   // 8 positions in flight
        float part[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 kv = ld4(krow(min(i0 + u, nl - 1)));
            float acc = 0.0f;
            acc = __builtin_fmaf(qv.x, kv.x, acc);
            acc = __builtin_fmaf(qv.y, kv.y, acc);
            acc = __builtin_fmaf(qv.z, kv.z, acc);
            acc = __builtin_fmaf(qv.w, kv.w, acc);
            part[u] = acc;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u;
            const float tot = group16_sum_butterfly(part[u]);
            if (j == 0 && i < nl) sc[w][g][SELF ? i : slot[w][i]] = tot * scale;
        }
    }
    __syncthreads();

    // softmax of head g: this lane holds the positions j + 16 q + 64 i, i.e. partials j + 16 q of the 64
    float s[MAXP / 64][4];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXP / 64; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = j + 16 * q + 64 * i;
            s[i][q] = p < n ? sc[w][g][p] : -INFINITY;
            m = fmaxf(m, s[i][q]);
        }
    m = group16_max(m);
    float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < MAXP / 64; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = j + 16 * q + 64 * i;
            const float e = p < n ? d_exp_le0(s[i][q] - m) : 0.0f;
            s[i][q] = e;
            part[q] = part[q] + e;
        }
    const float sum = group16_sum_butterfly((part[0] + part[2]) + (part[1] + part[3]));
#pragma unroll
    for (int i = 0; i < MAXP / 64; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = j + 16 * q + 64 * i;
            if (p < n) sc[w][g][p] = s[i][q] / sum;
        }
    __syncthreads();

    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i0 = 0; i0 < nl; i0 += 8) {
        float wgt[8];
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u, nl - 1);
            wgt[u] = i0 + u < nl ? sc[w][g][SELF ? i : slot[w][i]] : 0.0f;   // beyond the list: weight 0, no effect
            v[u] = ld4(krow(i) + d);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = __builtin_fmaf(wgt[u], v[u].x, acc[0]);
            acc[1] = __builtin_fmaf(wgt[u], v[u].y, acc[1]);
            acc[2] = __builtin_fmaf(wgt[u], v[u].z, acc[2]);
            acc[3] = __builtin_fmaf(wgt[u], v[u].w, acc[3]);
        }
    }
    st4(out + r * d + eoff, acc);
}

template <typename T, bool SELF>
int launch_attn(const T* base, int64_t slab_stride, const int32_t* anc, int64_t anc_ld, const T* q,
                const uint8_t* mask, const int32_t* sent, int64_t n, T* out, int64_t R, int64_t d, int64_t heads,
                hipStream_t stream, const char* name) {
    const int64_t dh = d / heads;
    const float scale = (float)(1.0 / sqrt((double)dh));  // rounded once, from the double-precision value
    if (dh == 64 && heads % 4 == 0) {
        const unsigned grid4 = (unsigned)((R * (heads / 4) + 3) / 4);
        hipLaunchKernelGGL((attn_step_h64_kernel<T, SELF>), dim3(grid4), dim3(256), 0, stream, base, slab_stride, anc, anc_ld, q,
                           mask, sent, (int)n, out, R, (int)d, (int)heads, scale);
        return tst_after_launch(name);
    }
    const unsigned grid = (unsigned)((R * heads + 3) / 4);
    const int nch = (int)((dh + 63) / 64);
    const int vec = (dh % 256 == 0) ? 4 : (dh % 128 == 0) ? 2 : 1;
#define TST_ATTN_CASE(NCH_, VEC_)                                                                                        \
    hipLaunchKernelGGL((attn_step_kernel<T, SELF, NCH_, VEC_>), dim3(grid), dim3(256), 0, stream, base, slab_stride, anc,  \
                       anc_ld, q, mask, sent, (int)n, out, R, (int)d, (int)heads, (int)dh, scale)
    if (nch == 1 && vec == 1) TST_ATTN_CASE(1, 1);
    else if (nch == 2 && vec == 2) TST_ATTN_CASE(2, 2);
    else if (nch == 2 && vec == 1) TST_ATTN_CASE(2, 1);
    else if (nch == 3 && vec == 1) TST_ATTN_CASE(3, 1);
    else if (nch == 4 && vec == 4) TST_ATTN_CASE(4, 4);
    else if (nch == 4 && vec == 1) TST_ATTN_CASE(4, 1);
    else {
        tst_set_error("%s: head size %lld not supported (multiple of 4, at most 256)", name, (long long)dh);
        return TST_ERR_UNSUPPORTED;
    }
#undef TST_ATTN_CASE
    return tst_after_launch(name);
}

// Cross-attention for beam search: the KB hypotheses of one sentence are adjacent rows and attend to the SAME keys and
// values, so one wave serves (KB rows, head): every key / value element is loaded once and used for KB queries.
// Per row the arithmetic and its order are those of attn_step_kernel (scores: 16 partials + butterfly; softmax: one
// score per lane and 64-position block; value sum over the listed positions in ascending order), so results are
// bit-identical.  A group whose rows do not all name the same sentence (the first step of a decode call runs one
// row per sentence) is computed row by row.  No workgroup barrier: all LDS state is private to the wave.
template <typename T, int KB, int NCH, int VEC>
__device__ __forceinline__ void attend_rows(const T* __restrict__ base, const int32_t* __restrict__ kvrow, const T* __restrict__ qbuf,
                                            const uint8_t* __restrict__ mask, int64_t ent, int n, T* __restrict__ out, int64_t r0,
                                            int d, int h, int dh, float scale, float (*sc)[MAXP], int* slot, int* kvr, int lane) {
    int nl = 0;
    for (int p0 = 0; p0 < n; p0 += 64) {
        const int p = p0 + lane;
        const bool live = p < n && (mask == nullptr || mask[ent * n + p] != 0);
        const unsigned long long bal = __ballot(live);
        if (live) {
            const int at = nl + __popcll(bal & ((1ull << lane) - 1ull));
            slot[at] = p;
            kvr[at] = kvrow != nullptr ? kvrow[ent * n + p] : (int)(ent * n + p);
        } else if (p < n) {
#pragma unroll
            for (int b = 0; b < KB; ++b) sc[b][p] = -INFINITY;
        }
        nl += __popcll(bal);
    }
    const int j = lane & 15, g = lane >> 4;
    float4 qv[KB][NCH];
#pragma unroll
    for (int b = 0; b < KB; ++b)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = j + 16 * c;
            qv[b][c] = (ch * 4 < dh) ? ld4(qbuf + (r0 + b) * d + h * dh + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    auto krow = [&](int i) -> const T* { return base + (int64_t)kvr[i] * 2 * d + h * dh; };

    for (int i0 = 0; i0 < nl; i0 += 16) {
        float part[4][KB];
        const T* kpp[4];    // rows first (LDS), then all loads, then the products
#pragma unroll
        for (int u = 0; u < 4; ++u) kpp[u] = krow(min(i0 + 4 * u + g, nl - 1));
        float4 kvv[4][NCH];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = j + 16 * c;
                kvv[u][c] = ld4(kpp[u] + 4 * ((ch * 4 < dh) ? ch : 0));   // always a load; unused when beyond the head
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float acc[KB];
#pragma unroll
            for (int b = 0; b < KB; ++b) acc[b] = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = j + 16 * c;
                if (ch * 4 < dh) {
                    const float4 kv = kvv[u][c];
#pragma unroll
                    for (int b = 0; b < KB; ++b) {
                        acc[b] = __builtin_fmaf(qv[b][c].x, kv.x, acc[b]);
                        acc[b] = __builtin_fmaf(qv[b][c].y, kv.y, acc[b]);
                        acc[b] = __builtin_fmaf(qv[b][c].z, kv.z, acc[b]);
                        acc[b] = __builtin_fmaf(qv[b][c].w, kv.w, acc[b]);
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < KB; ++b) part[u][b] = acc[b];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 4 * u + g;
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const float tot = group16_sum_butterfly(part[u][b]);
                if (j == 0 && i < nl) sc[b][slot[i]] = tot * scale;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();

#pragma unroll
    for (int b = 0; b < KB; ++b) {
        float sv[MAXP / 64];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < MAXP / 64; ++i) {
            const int p = lane + 64 * i;
            sv[i] = p < n ? sc[b][p] : -INFINITY;
            m = fmaxf(m, sv[i]);
        }
        m = wave_max(m);
        float part = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXP / 64; ++i) {
            const int p = lane + 64 * i;
            const float e = p < n ? d_exp_le0(sv[i] - m) : 0.0f;
            sv[i] = e;
            part = part + e;
        }
        const float sum = wave_sum_butterfly(part);
#pragma unroll
        for (int i = 0; i < MAXP / 64; ++i) {
            const int p = lane + 64 * i;
            if (p < n) sc[b][p] = sv[i] / sum;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();

    for (int c0 = lane * VEC; c0 < dh; c0 += 64 * VEC) {
        float acc[KB][VEC];
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[b][e] = 0.0f;
        for (int i0 = 0; i0 < nl; i0 += 8) {
            float v[8][VEC];
            int pos[8];
            const T* vrow[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u, nl - 1);
                const int sl = slot[i];
                pos[u] = i0 + u < nl ? sl : -1;
                vrow[u] = krow(i) + d + c0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const T* vp = vrow[u];
                if (VEC == 4) {
                    const float4 x = ld4(vp);
                    v[u][0] = x.x; v[u][1 % VEC] = x.y; v[u][2 % VEC] = x.z; v[u][3 % VEC] = x.w;
                } else if (VEC == 2) {
                    const float2 x = ld2(vp);
                    v[u][0] = x.x; v[u][1 % VEC] = x.y;
                } else {
                    v[u][0] = ld1(vp);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int b = 0; b < KB; ++b) {
                    const float wv = sc[b][max(pos[u], 0)];                  // read unconditionally (no branch per position)
                    const float wgt = pos[u] >= 0 ? wv : 0.0f;               // beyond the list: weight 0, no effect
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[b][e] = __builtin_fmaf(wgt, v[u][e], acc[b][e]);
                }
        }
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int e = 0; e < VEC; ++e) st1(out + (r0 + b) * d + h * dh + c0 + e, acc[b][e]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

template <typename T, int KB, int NCH, int VEC>
__global__ __launch_bounds__(256) void attn_cross_group_kernel(const T* __restrict__ base, const int32_t* __restrict__ kvrow,
                                                               const T* __restrict__ qbuf, const uint8_t* __restrict__ mask,
                                                               const int32_t* __restrict__ sent, int n, T* __restrict__ out,
                                                               int64_t n_groups, int d, int heads, int dh, float scale) {
    __shared__ float sc[4][KB][MAXP];
    __shared__ int slot[4][MAXP];
    __shared__ int kvr[4][MAXP];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * 4 + w;
    if (gw >= n_groups * heads) return;          // no workgroup barrier below
    const int64_t grp = gw / heads;
    const int h = (int)(gw - grp * heads);
    const int64_t r0 = grp * KB;
    const int64_t ent = sent[r0];
    bool same = true;
#pragma unroll
    for (int b = 1; b < KB; ++b) same = same && sent[r0 + b] == ent;
    if (same) {
        attend_rows<T, KB, NCH, VEC>(base, kvrow, qbuf, mask, ent, n, out, r0, d, h, dh, scale, sc[w], slot[w], kvr[w], lane);
    } else {
        for (int b = 0; b < KB; ++b)
            attend_rows<T, 1, NCH, VEC>(base, kvrow, qbuf, mask, sent[r0 + b], n, out, r0 + b, d, h, dh, scale, sc[w], slot[w], kvr[w],
                                        lane);
    }
}

template <typename T, int KB>
int launch_cross_group(const T* kv, const int32_t* kvrow, const T* q, const uint8_t* mask, const int32_t* sent, int64_t n, T* out,
                       int64_t R, int64_t d, int64_t heads, hipStream_t stream, const char* name) {
    const int64_t dh = d / heads;
    const float scale = (float)(1.0 / sqrt((double)dh));
    const int64_t n_groups = R / KB;
    const unsigned grid = (unsigned)((n_groups * heads + 3) / 4);
    const int nch = (int)((dh + 63) / 64);
    const int vec = (dh % 256 == 0) ? 4 : (dh % 128 == 0) ? 2 : 1;
#define TST_GROUP_CASE(NCH_, VEC_)                                                                                       \
    hipLaunchKernelGGL((attn_cross_group_kernel<T, KB, NCH_, VEC_>), dim3(grid), dim3(256), 0, stream, kv, kvrow, q, mask, sent, \
                       (int)n, out, n_groups, (int)d, (int)heads, (int)dh, scale)
    if (nch == 1 && vec == 1) TST_GROUP_CASE(1, 1);
    else if (nch == 2 && vec == 2) TST_GROUP_CASE(2, 2);
    else if (nch == 2 && vec == 1) TST_GROUP_CASE(2, 1);
    else return 1;   // not built for this head size: the caller takes the row-per-wave kernel
#undef TST_GROUP_CASE
    return tst_after_launch(name);
}

}  // namespace

extern "C" int tst_attn_self_step_f32(const float* cache, int64_t slab_rows, const int32_t* anc, int64_t anc_ld,
                                      int64_t pos, float* out, int64_t R, int64_t d, int64_t heads, void* stream) {
    TST_REQUIRE(tst_dims_ok({slab_rows, pos, R, d, heads}), TST_ERR_UNSUPPORTED, "tst_attn_self_step_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(cache && out && (anc || pos == 0), TST_ERR_ARG, "tst_attn_self_step_f32: null pointer");
    TST_REQUIRE(R >= 0 && R <= slab_rows && d > 0 && heads > 0 && d % heads == 0 && (d / heads) % 4 == 0 && pos >= 0,
                TST_ERR_ARG, "tst_attn_self_step_f32: bad sizes");
    TST_REQUIRE(pos + 1 <= MAXP && pos <= anc_ld, TST_ERR_UNSUPPORTED, "tst_attn_self_step_f32: at most %d positions", MAXP);
    TST_REQUIRE(tst_aligned16(cache) && tst_aligned16(out), TST_ERR_ARG, "tst_attn_self_step_f32: alignment");
    if (R == 0) return TST_OK;
    return launch_attn<float, true>(cache, slab_rows * 3 * d, anc, anc_ld, nullptr, nullptr, nullptr, pos + 1, out, R, d, heads,
                             (hipStream_t)stream, "tst_attn_self_step_f32");
}

extern "C" int tst_attn_cross_step_f32(const float* q, const float* kv, const uint8_t* mask, const int32_t* sent,
                                       int64_t S, float* out, int64_t R, int64_t d, int64_t heads, void* stream) {
    TST_REQUIRE(tst_dims_ok({S, R, d, heads}), TST_ERR_UNSUPPORTED, "tst_attn_cross_step_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(q && kv && sent && out, TST_ERR_ARG, "tst_attn_cross_step_f32: null pointer");
    TST_REQUIRE(R >= 0 && d > 0 && heads > 0 && d % heads == 0 && (d / heads) % 4 == 0 && S > 0, TST_ERR_ARG,
                "tst_attn_cross_step_f32: bad sizes");
    TST_REQUIRE(S <= MAXP, TST_ERR_UNSUPPORTED, "tst_attn_cross_step_f32: at most %d source positions", MAXP);
    TST_REQUIRE(tst_aligned16(q) && tst_aligned16(kv) && tst_aligned16(out), TST_ERR_ARG, "tst_attn_cross_step_f32: alignment");
    if (R == 0) return TST_OK;
    return launch_attn<float, false>(kv, 0, nullptr, 0, q, mask, sent, S, out, R, d, heads, (hipStream_t)stream,
                              "tst_attn_cross_step_f32");
}

extern "C" int tst_attn_cross_packed_f32(const float* q, const float* kv, const uint8_t* mask, const int32_t* sent,
                                         const int32_t* kvrow, int64_t S, float* out, int64_t R, int64_t d, int64_t heads,
                                         void* stream) {
    TST_REQUIRE(tst_dims_ok({S, R, d, heads}), TST_ERR_UNSUPPORTED, "tst_attn_cross_packed_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(q && kv && sent && out && kvrow, TST_ERR_ARG, "tst_attn_cross_packed_f32: null pointer");
    TST_REQUIRE(R >= 0 && d > 0 && heads > 0 && d % heads == 0 && (d / heads) % 4 == 0 && S > 0, TST_ERR_ARG,
                "tst_attn_cross_packed_f32: bad sizes");
    TST_REQUIRE(S <= MAXP, TST_ERR_UNSUPPORTED, "tst_attn_cross_packed_f32: at most %d source positions", MAXP);
    TST_REQUIRE(tst_aligned16(q) && tst_aligned16(kv) && tst_aligned16(out), TST_ERR_ARG, "tst_attn_cross_packed_f32: alignment");
    if (R == 0) return TST_OK;
    return launch_attn<float, false>(kv, 0, kvrow, 0, q, mask, sent, S, out, R, d, heads, (hipStream_t)stream,
                              "tst_attn_cross_packed_f32");
}

extern "C" int tst_attn_cross_group_f32(const float* q, const float* kv, const uint8_t* mask, const int32_t* sent,
                                        const int32_t* kvrow, int64_t S, float* out, int64_t R, int64_t d, int64_t heads,
                                        int64_t group, void* stream) {
    TST_REQUIRE(tst_dims_ok({S, R, d, heads, group}), TST_ERR_UNSUPPORTED, "tst_attn_cross_group_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(q && kv && sent && out, TST_ERR_ARG, "tst_attn_cross_group_f32: null pointer");
    TST_REQUIRE(R >= 0 && d > 0 && heads > 0 && d % heads == 0 && (d / heads) % 4 == 0 && S > 0 && group >= 1, TST_ERR_ARG,
                "tst_attn_cross_group_f32: bad sizes");
    TST_REQUIRE(kvrow == nullptr || mask != nullptr, TST_ERR_ARG, "tst_attn_cross_group_f32: a packed k|v needs the mask");
    TST_REQUIRE(S <= MAXP, TST_ERR_UNSUPPORTED, "tst_attn_cross_group_f32: at most %d source positions", MAXP);
    TST_REQUIRE(tst_aligned16(q) && tst_aligned16(kv) && tst_aligned16(out), TST_ERR_ARG, "tst_attn_cross_group_f32: alignment");
    if (R == 0) return TST_OK;
    int rc = 1;   // 1: no grouped kernel for this case
    if (R % group == 0) {
        const hipStream_t st = (hipStream_t)stream;
        if (group == 2) rc = launch_cross_group<float, 2>(kv, kvrow, q, mask, sent, S, out, R, d, heads, st, "tst_attn_cross_group_f32");
        else if (group == 4) rc = launch_cross_group<float, 4>(kv, kvrow, q, mask, sent, S, out, R, d, heads, st, "tst_attn_cross_group_f32");
        else if (group == 8) rc = launch_cross_group<float, 8>(kv, kvrow, q, mask, sent, S, out, R, d, heads, st, "tst_attn_cross_group_f32");
    }
    if (rc != 1) return rc;
    return launch_attn<float, false>(kv, 0, kvrow, 0, q, mask, sent, S, out, R, d, heads, (hipStream_t)stream,
                              "tst_attn_cross_group_f32");
}

// bf16 storage (cache, q, k|v and output), fp32 arithmetic in the same order as the fp32 kernels
extern "C" int tst_attn_self_step_bf16(const void* cache, int64_t slab_rows, const int32_t* anc, int64_t anc_ld,
                                       int64_t pos, void* out, int64_t R, int64_t d, int64_t heads, void* stream) {
    TST_REQUIRE(tst_dims_ok({slab_rows, pos, R, d, heads}), TST_ERR_UNSUPPORTED, "tst_attn_self_step_bf16: a size exceeds 2^31 - 1");
    TST_REQUIRE(cache && out && (anc || pos == 0), TST_ERR_ARG, "tst_attn_self_step_bf16: null pointer");
    TST_REQUIRE(R >= 0 && R <= slab_rows && d > 0 && heads > 0 && d % heads == 0 && (d / heads) % 4 == 0 && pos >= 0,
                TST_ERR_ARG, "tst_attn_self_step_bf16: bad sizes");
    TST_REQUIRE(pos + 1 <= MAXP && pos <= anc_ld, TST_ERR_UNSUPPORTED, "tst_attn_self_step_bf16: at most %d positions", MAXP);
    TST_REQUIRE(tst_aligned16(cache) && tst_aligned16(out), TST_ERR_ARG, "tst_attn_self_step_bf16: alignment");
    if (R == 0) return TST_OK;
    return launch_attn<__bf16, true>((const __bf16*)cache, slab_rows * 3 * d, anc, anc_ld, nullptr, nullptr, nullptr, pos + 1,
                                     (__bf16*)out, R, d, heads, (hipStream_t)stream, "tst_attn_self_step_bf16");
}

extern "C" int tst_attn_cross_step_bf16(const void* q, const void* kv, const uint8_t* mask, const int32_t* sent, int64_t S,
                                        void* out, int64_t R, int64_t d, int64_t heads, void* stream) {
    TST_REQUIRE(tst_dims_ok({S, R, d, heads}), TST_ERR_UNSUPPORTED, "tst_attn_cross_step_bf16: a size exceeds 2^31 - 1");
    TST_REQUIRE(q && kv && sent && out, TST_ERR_ARG, "tst_attn_cross_step_bf16: null pointer");
    TST_REQUIRE(R >= 0 && d > 0 && heads > 0 && d % heads == 0 && (d / heads) % 4 == 0 && S > 0, TST_ERR_ARG,
                "tst_attn_cross_step_bf16: bad sizes");
    TST_REQUIRE(S <= MAXP, TST_ERR_UNSUPPORTED, "tst_attn_cross_step_bf16: at most %d source positions", MAXP);
    TST_REQUIRE(tst_aligned16(q) && tst_aligned16(kv) && tst_aligned16(out), TST_ERR_ARG, "tst_attn_cross_step_bf16: alignment");
    if (R == 0) return TST_OK;
    return launch_attn<__bf16, false>((const __bf16*)kv, 0, nullptr, 0, (const __bf16*)q, mask, sent, S, (__bf16*)out, R, d,
                                      heads, (hipStream_t)stream, "tst_attn_cross_step_bf16");
}

// The same step with the beams of a sentence as the rows of matrix products on the matrix cores (attn_mfma.hip): scores in
// chain order, so results differ from tst_attn_cross_group_f32 in the last bits.  Measured slower than that kernel at the
// benchmark sizes (97 against 47 us per launch at 8192 sentences, beam 4, 2 heads of 128): exported for tests and tuning, not
// used by tst_tfm_step_run.
extern "C" int tst_attn_cross_mfma_f32(const float* q, const float* kv, const uint8_t* mask, const int32_t* sent, const int32_t* kvrow,
                                       int64_t S, float* out, int64_t R, int64_t d, int64_t heads, int64_t group, void* stream) {
    TST_REQUIRE(tst_dims_ok({S, R, d, heads, group}), TST_ERR_UNSUPPORTED, "tst_attn_cross_mfma_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(q && kv && sent && out, TST_ERR_ARG, "tst_attn_cross_mfma_f32: null pointer");
    TST_REQUIRE(R >= 0 && d > 0 && heads > 0 && d % heads == 0 && S > 0 && group >= 1, TST_ERR_ARG, "tst_attn_cross_mfma_f32: bad sizes");
    TST_REQUIRE(kvrow == nullptr || mask != nullptr, TST_ERR_ARG, "tst_attn_cross_mfma_f32: a packed k|v needs the mask");
    TST_REQUIRE(R % group == 0 && tst_cross_mfma_ok(S, d / heads, group), TST_ERR_UNSUPPORTED,
                "tst_attn_cross_mfma_f32: needs R %% group == 0, group 2..8, at most 64 positions and heads of 4, 8, .., 64 or 128 elements");
    TST_REQUIRE(tst_aligned16(q) && tst_aligned16(kv) && tst_aligned16(out), TST_ERR_ARG, "tst_attn_cross_mfma_f32: alignment");
    if (R == 0) return TST_OK;
    return tst_launch_cross_mfma(kv, kvrow, q, mask, sent, S, out, R, d, heads, group, (hipStream_t)stream);
}
