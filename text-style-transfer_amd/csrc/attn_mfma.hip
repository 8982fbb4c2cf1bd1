// Cross-attention step for beam search on the matrix cores: the hypotheses of a sentence share its keys and values, so
// for one (sentence, head) the scores of all beams are ONE small matrix product Q [beams x dh] * K^T [dh x S] and the
// outputs another, W [beams x S] * V [S x dh].
//
// v_mfma_f32_4x4x1_16B_f32 multiplies 16 independent blocks, block b: D_b [4 x 4] += A_b [4 x 1] * B_b [1 x 4], with lane
// 4 b + i holding A_b[i] / B_b[i] and register r of lane 4 b + j holding D_b[r][j].  Scores: the four rows of every A block
// are the queries of four beams (element e), block b's B row is element e of the keys of positions 4 b .. 4 b + 3: lane p IS
// position p, register r its score for beam r; dh instructions accumulate the dot products in ascending e, one fma per
// element and instruction: score = fma chain over e from 0, then * 1/sqrt(dh) ("chain order": the row-per-wave kernels sum
// 16 partials instead).  Outputs: A rows = the weights of the four beams for position p, block b's B row = elements
// 4 b .. 4 b + 3 of the value row of position p; S instructions per 64 columns, p ascending: the value sum of the other
// kernels.  Softmax between the two: lane = position, so maxima and sums are the usual butterflies over the wave.
//
// The key tile of the (sentence, head), [S][dh] floats, is staged once in LDS (row stride dh + 4 floats: a lane reads ITS
// row 16 bytes at a time without bank conflicts), the queries and the weights pass through LDS as well (the A operand is
// a per-block broadcast); value rows are read straight from memory, 64 consecutive floats per instruction.
// One wave per (group of beams, head); no workgroup barrier: all LDS state is private to the wave.
// Status: bit-exact against a CPU statement of the same arithmetic, but about twice the time of attn_cross_group_kernel (VALU dot products,
// 16 waves per CU) at the benchmark sizes: the two dependent chains of dh and S small MFMAs per wave and the four waves per
// CU that the key tiles leave room for do not hide the memory latency.  Kept as tst_attn_cross_mfma_f32, not on the decode path.
#include "tst_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MP = 64;       // source positions per sentence supported (one per lane)
constexpr int MDH = 128;     // head size supported

// per-wave LDS, carved from dynamic shared memory: keys [n][dh + 4], queries [8][dh] (unused rows are zero), softmax
// weights [8][64]
struct WaveSmem {
    float* ks_;
    float* qs_;
    float* ws_;
    int kstride, dh;
    __device__ __forceinline__ float* ks(int p) const { return ks_ + p * kstride; }
    __device__ __forceinline__ float* qs(int b) const { return qs_ + b * dh; }
    __device__ __forceinline__ float* ws(int b) const { return ws_ + b * MP; }
};

__host__ __device__ inline int wave_smem_floats(int n, int dh) { return n * (dh + 4) + 8 * dh + 8 * MP; }

// rows r0 .. r0 + nb - 1 (nb <= 4 * NQ) attend to sentence `ent`
template <int NQ>
__device__ __forceinline__ void attend_group(const float* __restrict__ kv, const int32_t* __restrict__ kvrow, const float* __restrict__ qbuf,
                                             const uint8_t* __restrict__ mask, int64_t ent, int n, float* __restrict__ out, int64_t r0,
                                             int nb, int d, int h, int dh, float scale, const WaveSmem& sm, int lane) {
    const int dh4 = dh >> 2;
    // this lane's position: live or masked, and its k|v row
    const bool live = lane < n && (mask == nullptr || mask[ent * n + lane] != 0);
    const int my_row = live ? (kvrow != nullptr ? kvrow[ent * n + lane] : (int)(ent * n + lane)) : -1;
    const unsigned long long live_mask = __ballot(live);

    // value rows: V[p][lane] and V[p][64 + lane] of every position into registers, issued before anything else so that the
    // loads are in flight while the keys are staged and the scores are computed (p < 64 is a compile-time bound: static
    // register indices; rows of masked positions are not read and count as zero)
    const float* vbase = kv + d + h * dh;
    float v0[MP], v1[MP];
    const bool col0 = lane < dh, col1 = 64 + lane < dh;
#pragma unroll
    for (int p = 0; p < MP; ++p) {
        const int row = __builtin_amdgcn_readlane(my_row, p);          // wave-uniform
        const bool ok = p < n && row >= 0;
        v0[p] = (ok && col0) ? vbase[(int64_t)row * 2 * d + lane] : 0.0f;
        v1[p] = (ok && col1) ? vbase[(int64_t)row * 2 * d + 64 + lane] : 0.0f;
    }

    // stage the keys: 64 / dh4 rows per pass, 16 bytes per lane, four passes of loads in flight; masked rows are zero
    const int lpr = dh4;                       // lanes per row
    const int rpp = 64 / lpr;                  // rows per pass
    for (int p0 = 0; p0 < n; p0 += 4 * rpp) {
        float4 kreg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * rpp + lane / lpr, c = lane % lpr;
            const int row = __shfl(my_row, p < MP ? p : 0, 64);
            kreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < n && row >= 0) kreg[u] = *reinterpret_cast<const float4*>(kv + (int64_t)row * 2 * d + h * dh + 4 * c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * rpp + lane / lpr, c = lane % lpr;
            if (p < n) *reinterpret_cast<float4*>(sm.ks(p) + 4 * c) = kreg[u];
        }
    }
    for (int i = lane; i < 4 * NQ * dh4; i += 64) {
        const int b = i / dh4, c = i % dh4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < nb) v = *reinterpret_cast<const float4*>(qbuf + (r0 + b) * d + h * dh + 4 * c);
        *reinterpret_cast<float4*>(sm.qs(b) + 4 * c) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();

    // scores: acc[q][r] = sum_e Q[4 q + r][e] * K[lane][e], e ascending
    f32x4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kp = lane < n ? lane : 0;        // lanes beyond n read a valid row; their scores are discarded
    for (int c = 0; c < dh4; ++c) {
        const float4 kb = *reinterpret_cast<const float4*>(sm.ks(kp) + 4 * c);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 qa = *reinterpret_cast<const float4*>(sm.qs(4 * q + (lane & 3)) + 4 * c);
            acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa.x, kb.x, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa.y, kb.y, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa.z, kb.z, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa.w, kb.w, acc[q], 0, 0, 0);
        }
    }
    // softmax over the positions (= lanes) per beam: w = exp(s - max) / sum, masked positions have weight exactly 0
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s = live ? acc[q][r] * scale : -INFINITY;
            const float mx = wave_max(s);
            const float e = live ? d_exp_le0(s - mx) : 0.0f;
            const float sum = wave_sum_butterfly(e);
            sm.ws(4 * q + r)[lane] = e / sum;
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();

    // outputs: o[q][r] = sum_p W[4 q + r][p] * V[p][column], p ascending (weights of masked positions are exactly 0 and
    // their value rows count as 0: the same sums as skipping them)
    f32x4 o0[NQ], o1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { o0[q] = f32x4{0.f, 0.f, 0.f, 0.f}; o1[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int p = 0; p < MP; ++p) {
        if (p < n && ((live_mask >> p) & 1ull)) {      // wave-uniform
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float wq = sm.ws(4 * q + (lane & 3))[p];
                o0[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq, v0[p], o0[q], 0, 0, 0);
                if (dh > 64) o1[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq, v1[p], o1[q], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < nb) {
                float* op = out + (r0 + 4 * q + r) * d + h * dh;
                if (col0) op[lane] = o0[q][r];
                if (col1) op[64 + lane] = o1[q][r];
            }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

template <int NQ>
__global__ __launch_bounds__(128) void attn_cross_mfma_kernel(const float* __restrict__ kv, const int32_t* __restrict__ kvrow,
                                                              const float* __restrict__ qbuf, const uint8_t* __restrict__ mask,
                                                              const int32_t* __restrict__ sent, int n, float* __restrict__ out,
                                                              int64_t n_groups, int group, int d, int heads, int dh, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* mine = smem + w * wave_smem_floats(n, dh);
    const WaveSmem sm{mine, mine + n * (dh + 4), mine + n * (dh + 4) + 8 * dh, dh + 4, dh};
    const int64_t gw = (int64_t)blockIdx.x * 2 + w;
    if (gw >= n_groups * heads) return;          // no workgroup barrier below
    const int64_t grp = gw / heads;
    const int h = (int)(gw - grp * heads);
    const int64_t r0 = grp * group;
    const int64_t ent = sent[r0];
    bool same = true;
    for (int b = 1; b < group; ++b) same = same && sent[r0 + b] == ent;
    if (same) {
        attend_group<NQ>(kv, kvrow, qbuf, mask, ent, n, out, r0, group, d, h, dh, scale, sm, lane);
    } else {   // rows of different sentences (the grouping is a hint): one row at a time, same arithmetic
        for (int b = 0; b < group; ++b)
            attend_group<1>(kv, kvrow, qbuf, mask, sent[r0 + b], n, out, r0 + b, 1, d, h, dh, scale, sm, lane);
    }
}

}  // namespace

// 1 when the matrix-core kernel handles the case (the caller falls back to the row-per-wave arithmetic otherwise)
int tst_cross_mfma_ok(int64_t S, int64_t dh, int64_t group) { return (S <= MP && dh <= MDH && dh % 4 == 0 && 64 % (dh / 4) == 0 && group >= 2 && group <= 8) ? 1 : 0; }

int tst_launch_cross_mfma(const float* kv, const int32_t* kvrow, const float* q, const uint8_t* mask, const int32_t* sent, int64_t S,
                          float* out, int64_t R, int64_t d, int64_t heads, int64_t group, hipStream_t stream) {
    const int64_t dh = d / heads;
    const float scale = (float)(1.0 / sqrt((double)dh));
    const int64_t n_groups = R / group;
    const unsigned grid = (unsigned)((n_groups * heads + 1) / 2);
    const size_t lds = (size_t)2 * wave_smem_floats((int)S, (int)dh) * sizeof(float);
    if (group <= 4)
        hipLaunchKernelGGL(attn_cross_mfma_kernel<1>, dim3(grid), dim3(128), lds, stream, kv, kvrow, q, mask, sent, (int)S, out, n_groups,
                           (int)group, (int)d, (int)heads, (int)dh, scale);
    else
        hipLaunchKernelGGL(attn_cross_mfma_kernel<2>, dim3(grid), dim3(128), lds, stream, kv, kvrow, q, mask, sent, (int)S, out, n_groups,
                           (int)group, (int)d, (int)heads, (int)dh, scale);
    return tst_after_launch("tst_attn_cross_mfma_f32");
}
