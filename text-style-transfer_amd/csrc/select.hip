// Token selection and decode-loop bookkeeping on the device: arg-max, log-sum-exp + top-k over the
// vocabulary (the HBM-bound pass over the [rows, V] logits), beam merge, retirement of finished
// sentences and batch compaction.  No MFMA here: coalesced 16-byte reads, wave butterflies, LDS only
// to combine the four waves of a workgroup.
#include "tst_common.h"
#include <limits.h>
#include <math.h>

namespace {

// (value desc, index asc) ordering
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(i, off, 64);
        if (better(ov, oi, v, i)) { v = ov; i = oi; }
    }
}

// block-wide (256 threads) argmax; result valid in every thread. red_* are LDS arrays of 4.
__device__ __forceinline__ void block_argmax(float& v, int& i, float* red_v, int* red_i) {
    wave_argmax(v, i);
    const int w = threadIdx.x >> 6;
    __syncthreads();  // protects red_* against the previous round's readers
    if ((threadIdx.x & 63) == 0) { red_v[w] = v; red_i[w] = i; }
    __syncthreads();
    v = red_v[0]; i = red_i[0];
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (better(red_v[q], red_i[q], v, i)) { v = red_v[q]; i = red_i[q]; }
}

// Logits come as fp32 or bf16; a "chunk" is 16 bytes of a row: EPC = 4 fp32 or 8 bf16 elements.  All
// arithmetic is fp32 on the exactly widened values.
template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int EPC = 4; };
template <> struct Elem<__bf16> { static constexpr int EPC = 8; };

// loads chunk c of a row as floats; elements at or beyond V read as -inf
__device__ __forceinline__ void load_chunk(const float* xr, int c, int V, bool vec_ok, float (&o)[4]) {
    if (vec_ok && 4 * c + 3 < V) {
        const float4 v = reinterpret_cast<const float4*>(xr)[c];
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = 4 * c + q < V ? xr[4 * c + q] : -INFINITY;
}

__device__ __forceinline__ void load_chunk(const __bf16* xr, int c, int V, bool vec_ok, float (&o)[8]) {
    if (vec_ok && 8 * c + 7 < V) {
        const uint4 u = reinterpret_cast<const uint4*>(xr)[c];
        o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
        o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
        o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xffff0000u);
        o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xffff0000u);
        return;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = 8 * c + q < V ? (float)xr[8 * c + q] : -INFINITY;
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_rows_kernel(const T* __restrict__ x, int64_t ld, int V,
                                                          int32_t* __restrict__ tok, int vec_ok) {
    constexpr int EPC = Elem<T>::EPC;
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    const T* xr = x + (int64_t)blockIdx.x * ld;
    const int nch = (V + EPC - 1) / EPC;
    float bv = -INFINITY;
    int bi = INT_MAX;
    for (int c = threadIdx.x; c < nch; c += 256) {
        float e[EPC];
        load_chunk(xr, c, V, vec_ok != 0, e);
#pragma unroll
        for (int q = 0; q < EPC; ++q)
            if (e[q] > bv) { bv = e[q]; bi = EPC * c + q; }
    }
    block_argmax(bv, bi, red_v, red_i);
    if (threadIdx.x == 0) tok[blockIdx.x] = bi == INT_MAX ? 0 : bi;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Gumbel-max sampling: argmax_v (x_v / temperature - log(-log(u_v))), u_v from a counter-based hash of
// (seed, step, sentence, v); first maximal index wins.  One workgroup per row.
template <typename T>
__global__ __launch_bounds__(256) void sample_rows_kernel(const T* __restrict__ x, int64_t ld, int V, float temperature,
                                                          uint32_t seed, uint32_t step, const int32_t* __restrict__ sent,
                                                          int32_t* __restrict__ tok, int vec_ok) {
    constexpr int EPC = Elem<T>::EPC;
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    const T* xr = x + (int64_t)blockIdx.x * ld;
    const uint32_t rowkey = mix32((uint32_t)sent[blockIdx.x] * 0x9E3779B9u + mix32(step * 0x85EBCA6Bu + seed));
    const int nch = (V + EPC - 1) / EPC;
    float bv = -INFINITY;
    int bi = INT_MAX;
    for (int c = threadIdx.x; c < nch; c += 256) {
        float e[EPC];
        load_chunk(xr, c, V, vec_ok != 0, e);
#pragma unroll
        for (int q = 0; q < EPC; ++q) {
            const int id = EPC * c + q;
            const uint32_t h = mix32((uint32_t)id ^ rowkey);
            const float u = (float)(h >> 9) * 1.1920928955078125e-07f + 5.9604644775390625e-08f;
            const float key = e[q] / temperature + (-d_log(-d_log(u)));
            if (id < V && key > bv) { bv = key; bi = id; }
        }
    }
    block_argmax(bv, bi, red_v, red_i);
    if (threadIdx.x == 0) tok[blockIdx.x] = bi == INT_MAX ? 0 : bi;
}

// ---- log-sum-exp + top-k --------------------------------------------------------------------------------
// Canonical log-sum-exp of a row (the same definition whichever kernel produces the partials):
//   blocks of 16 consecutive elements (elements at or beyond V count as -inf): m_b = max,
//   s_b = sum of exp(x - m_b) by the xor-butterfly 1, 2, 4, 8;  64 accumulators, accumulator q takes the blocks
//   q, q + 64, ... in order:  m' = max(m_q, m_b);  a_q = a_q * exp(m_q - m') + s_b * exp(m_b - m');  m_q = m';
//   M = max_q m_q;  total = xor-butterfly (32..1) over q of a_q * exp(m_q - M);  lse = M + log(total).
// logsoftmax_topk_kernel computes it in one pass over the logits.  Alternatively the (m_b, s_b) pairs ("parts",
// [row][ceil(V / 16)][2]) come from the epilogue of the vocabulary GEMM and topk_from_parts_kernel finishes them.

// sum of the EPC exponentials of a lane's chunk by the first butterfly levels (offsets 1, 2 and, for 8 elements, 4)
__device__ __forceinline__ float chunk_sum_asc(const float (&e)[4]) { return (e[0] + e[1]) + (e[2] + e[3]); }
__device__ __forceinline__ float chunk_sum_asc(const float (&e)[8]) {
    return ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
}

__device__ __forceinline__ float elem_at(const float* xr, int i) { return xr[i]; }
__device__ __forceinline__ float elem_at(const __bf16* xr, int i) { return (float)xr[i]; }

// Stand-alone kernel: ONE pass over the logits.  A row is handled by TPR = 64 * LPB threads, LPB = lanes per
// 16-element block (4 for fp32, 2 for bf16: one 16-byte chunk per lane), i.e. by 4 or 2 waves; a 256-thread
// workgroup takes 256 / TPR rows.  Thread u of a row owns the chunks u, u + TPR, ..., so the LPB lanes of a
// group hold the blocks g, g + 64, ... of accumulator g = u / LPB: they combine their chunks by DPP inside the
// quad and all keep the accumulator's online state (m, a).  Four chunks are loaded per iteration to keep
// enough bytes in flight.  The row's first wave then finishes: lse from the 64 accumulators, tau = the K-th
// largest accumulator maximum; the threads whose maximum reaches tau re-read their own chunks and push the
// elements >= tau to an LDS list, from which the first wave picks the K largest (value desc, index asc).
// Three workgroup barriers in all.
template <typename T, int K, bool VEC>
__global__ __launch_bounds__(256) void logsoftmax_topk_kernel(const T* __restrict__ x, int64_t ld, int V, int64_t R,
                                                              float* __restrict__ top_val, int32_t* __restrict__ top_idx,
                                                              float* __restrict__ lse) {
    constexpr int EPC = Elem<T>::EPC, LPB = 16 / EPC, TPR = 64 * LPB, RPB = 256 / TPR, CAP = 1024 / RPB;
    __shared__ float acc_m[RPB][64], acc_a[RPB][64], tau_s[RPB];
    __shared__ float cand_v[RPB][CAP];
    __shared__ int cand_i[RPB][CAP];
    __shared__ int cand_n[RPB];
    const int t = threadIdx.x, u = t % TPR, rl = t / TPR, lane = t & 63;
    const bool first_wave = u < 64;
    int64_t r = (int64_t)blockIdx.x * RPB + rl;
    const bool row_ok = r < R;
    if (!row_ok) r = R - 1;                      // a surplus half-workgroup redoes the last row and writes nothing
    const T* xr = x + r * ld;
    const int nb = (V + 15) >> 4, nch = nb * LPB;  // whole blocks: a chunk past the row reads as -inf
    float mq = -INFINITY, aq = 0.0f;
    auto consume = [&](const float (&e0)[EPC]) {   // one chunk of every lane of the group = one block
        float e[EPC];
        float m = e0[0];
#pragma unroll
        for (int q = 1; q < EPC; ++q) m = fmaxf(m, e0[q]);
        m = fmaxf(m, dpp_row<DPP_XOR1>(m));
        if (LPB == 4) m = fmaxf(m, dpp_row<DPP_XOR2>(m));
#pragma unroll
        for (int q = 0; q < EPC; ++q) e[q] = d_exp_le0(e0[q] - m);
        float sb = chunk_sum_asc(e);
        sb = sb + dpp_row<DPP_XOR1>(sb);
        if (LPB == 4) sb = sb + dpp_row<DPP_XOR2>(sb);
        const float mn = fmaxf(mq, m);
        aq = aq * d_exp_le0(mq - mn) + sb * d_exp_le0(m - mn);
        mq = mn;
    };
    int c0 = u;
    if (VEC) {  // blocks that lie fully inside the row (the same bound for all lanes of a group, which run the
                // DPP steps together): unguarded 16-byte loads, four in flight
        const int nfull = (V >> 4) * LPB;
        for (; c0 + 3 * TPR < nfull; c0 += 4 * TPR) {
            float vv[4][EPC];
#pragma unroll
            for (int i = 0; i < 4; ++i) load_chunk(xr, c0 + TPR * i, INT_MAX, true, vv[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) consume(vv[i]);
        }
    }
    for (; c0 < nch; c0 += TPR) {  // ragged end of the row (and rows that are not 16-byte aligned)
        float vv[EPC];
        load_chunk(xr, c0, V, VEC, vv);
        consume(vv);
    }
    if ((u & (LPB - 1)) == 0) { acc_m[rl][u / LPB] = mq; acc_a[rl][u / LPB] = aq; }
    __syncthreads();
    if (first_wave) {
        const float m_l = acc_m[rl][lane], a_l = acc_a[rl][lane];
        const float M = wave_max(m_l);
        const float total = wave_sum_butterfly(a_l * d_exp_le0(m_l - M));
        if (lane == 0 && row_ok) lse[r] = M + d_log(total);
        float tau = -INFINITY, cand = m_l;   // K-th largest accumulator maximum (ties: lower accumulator first)
        for (int o = 0; o < K; ++o) {
            float v = cand;
            int id = lane;
            wave_argmax(v, id);
            tau = v;
            if (id == lane) cand = -INFINITY;
        }
        if (lane == 0) { tau_s[rl] = tau; cand_n[rl] = 0; }
    }
    __syncthreads();
    const float tau = tau_s[rl];
    if (mq >= tau) {  // the lanes of about K groups
        for (int c = u; c < nch; c += TPR) {
            float e[EPC];
            load_chunk(xr, c, V, VEC, e);
#pragma unroll
            for (int q = 0; q < EPC; ++q)
                if (e[q] >= tau && EPC * c + q < V) {
                    const int at = atomicAdd(&cand_n[rl], 1);
                    if (at < CAP) { cand_v[rl][at] = e[q]; cand_i[rl][at] = EPC * c + q; }
                }
        }
    }
    __syncthreads();
    if (!first_wave) return;
    const int ncand = cand_n[rl];
    if (ncand <= CAP) {
        for (int o = 0; o < K; ++o) {
            float bv = -INFINITY;
            int bi = INT_MAX, be = -1;
            for (int e = lane; e < ncand; e += 64)
                if (better(cand_v[rl][e], cand_i[rl][e], bv, bi)) { bv = cand_v[rl][e]; bi = cand_i[rl][e]; be = e; }
            const int mine = bi;
            wave_argmax(bv, bi);
            if (be >= 0 && bi == mine) { cand_v[rl][be] = -INFINITY; cand_i[rl][be] = INT_MAX; }
            if (lane == 0 && row_ok) { top_val[r * K + o] = bv; top_idx[r * K + o] = bi == INT_MAX ? -1 : bi; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
    } else {
        float lastv = INFINITY;
        int lasti = -1;
        for (int o = 0; o < K; ++o) {
            float bv = -INFINITY;
            int bi = INT_MAX;
            for (int id = lane; id < V; id += 64) {
                const float e = elem_at(xr, id);
                const bool after = e < lastv || (e == lastv && id > lasti);
                if (after && better(e, id, bv, bi)) { bv = e; bi = id; }
            }
            wave_argmax(bv, bi);
            lastv = bv; lasti = bi;
            if (lane == 0 && row_ok) { top_val[r * K + o] = bv; top_idx[r * K + o] = bi == INT_MAX ? -1 : bi; }
        }
    }
}

constexpr int CAND_CAP = 256;  // per-wave LDS candidate list of the top-k filter

// One WAVE per row (4 rows per workgroup, no workgroup barriers).  Lane q is accumulator q of the canonical
// log-sum-exp.  Top K: the lane's running maximum is its largest element; tau = the K-th largest of the 64 lane
// maxima; at least K elements are >= tau and they sit in blocks with m_b >= tau of lanes whose maximum is >= tau,
// so only those few blocks of logits are read back (16 elements each) and their elements >= tau go to an LDS
// list, from which K rounds of wave arg-max (value desc, index asc) pick the result.  If the list overflows (very
// many equal values) the K rounds scan the whole row instead.
template <typename T, int K>
__global__ __launch_bounds__(256) void topk_from_parts_kernel(const T* __restrict__ x, int64_t ld, int V,
                                                              const float* __restrict__ parts, int nb, int64_t R,
                                                              float* __restrict__ top_val, int32_t* __restrict__ top_idx,
                                                              float* __restrict__ lse) {
    __shared__ float cand_v[4][CAND_CAP];
    __shared__ int cand_i[4][CAND_CAP];
    __shared__ int cand_n[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + w;
    if (r >= R) return;
    const T* xr = x + r * ld;
    const float2* pr = reinterpret_cast<const float2*>(parts) + r * nb;
    if (lane == 0) cand_n[w] = 0;
    float mq = -INFINITY, aq = 0.0f;
    for (int b = lane; b < nb; b += 64) {
        const float2 p = pr[b];
        const float mn = fmaxf(mq, p.x);
        aq = aq * d_exp_le0(mq - mn) + p.y * d_exp_le0(p.x - mn);
        mq = mn;
    }
    const float M = wave_max(mq);
    const float total = wave_sum_butterfly(aq * d_exp_le0(mq - M));
    if (lane == 0) lse[r] = M + d_log(total);
    // K-th largest lane maximum (ties between lanes: lower lane first)
    float tau = -INFINITY, cand = mq;
    for (int o = 0; o < K; ++o) {
        float v = cand;
        int id = lane;
        wave_argmax(v, id);
        tau = v;
        if (id == lane) cand = -INFINITY;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // cand_n[w] = 0 is visible to the wave's atomics
    if (mq >= tau) {  // about K lanes
        for (int b = lane; b < nb; b += 64) {
            if (pr[b].x < tau) continue;
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int id = 16 * b + q;
                if (id >= V) break;
                const float e = elem_at(xr, id);
                if (e >= tau) {
                    const int at = atomicAdd(&cand_n[w], 1);
                    if (at < CAND_CAP) { cand_v[w][at] = e; cand_i[w][at] = id; }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    const int ncand = cand_n[w];
    if (ncand <= CAND_CAP) {
        for (int o = 0; o < K; ++o) {
            float bv = -INFINITY;
            int bi = INT_MAX, be = -1;
            for (int e = lane; e < ncand; e += 64)
                if (better(cand_v[w][e], cand_i[w][e], bv, bi)) { bv = cand_v[w][e]; bi = cand_i[w][e]; be = e; }
            const int mine = bi;
            wave_argmax(bv, bi);
            if (be >= 0 && bi == mine) { cand_v[w][be] = -INFINITY; cand_i[w][be] = INT_MAX; }
            if (lane == 0) { top_val[r * K + o] = bv; top_idx[r * K + o] = bi == INT_MAX ? -1 : bi; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
    } else {
        float lastv = INFINITY;
        int lasti = -1;
        for (int o = 0; o < K; ++o) {
            float bv = -INFINITY;
            int bi = INT_MAX;
            for (int id = lane; id < V; id += 64) {
                const float e = elem_at(xr, id);
                const bool after = e < lastv || (e == lastv && id > lasti);
                if (after && better(e, id, bv, bi)) { bv = e; bi = id; }
            }
            wave_argmax(bv, bi);
            lastv = bv; lasti = bi;
            if (lane == 0) { top_val[r * K + o] = bv; top_idx[r * K + o] = bi == INT_MAX ? -1 : bi; }
        }
    }
}

__global__ __launch_bounds__(256) void greedy_select_kernel(const int32_t* __restrict__ am, int64_t n, int32_t eos,
                                                            int32_t* parent, int32_t* tok, float* score, int32_t* fin,
                                                            int32_t* nemit, tst_rows cur) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int32_t tk = am[r];
    parent[r] = (int32_t)r;
    tok[r] = tk;
    score[r] = 0.0f;
    fin[r] = tk == eos ? 1 : 0;
    nemit[r] = cur.nemit[r] + 1;
}

// one thread per sentence; offers are enumerated in (beam, rank) order and picked with a strict compare,
// which makes ties go to the lower beam, then the lower rank
__global__ __launch_bounds__(256) void beam_merge_kernel(tst_rows cur, const float* __restrict__ top_val,
                                                         const int32_t* __restrict__ top_idx,
                                                         const float* __restrict__ lse, int64_t n_sent, int k, int k_in,
                                                         int32_t eos, int32_t* parent, int32_t* tok, float* score,
                                                         int32_t* fin, int32_t* nemit) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= n_sent) return;
    const int64_t base = a * k_in;   // input rows of the sentence
    const int64_t obase = a * k;     // output slots
    float cv[64];
    int8_t cj[64];
    int32_t ct[64];
    int nc = 0;
    for (int j = 0; j < k_in; ++j) {
        const int64_t row = base + j;
        if (cur.fin[row]) {
            cv[nc] = cur.score[row]; cj[nc] = (int8_t)j; ct[nc] = -1; ++nc;
        } else {
            const float sc = cur.score[row], ls = lse[row];
            for (int rk = 0; rk < k; ++rk) {
                cv[nc] = sc + (top_val[row * k + rk] - ls);
                cj[nc] = (int8_t)j; ct[nc] = top_idx[row * k + rk]; ++nc;
            }
        }
    }
    unsigned long long taken = 0ull;
    for (int o = 0; o < k; ++o) {
        int best = -1;
        for (int c = 0; c < nc; ++c)
            if (!((taken >> c) & 1ull) && (best < 0 || cv[c] > cv[best])) best = c;
        const int64_t dst = obase + o;
        if (best < 0) {  // fewer offers than beams: pad with an ended, worthless hypothesis
            parent[dst] = (int32_t)base; tok[dst] = eos; score[dst] = -INFINITY; fin[dst] = 1; nemit[dst] = 0;
            continue;
        }
        taken |= 1ull << best;
        const int64_t prow = base + cj[best];
        parent[dst] = (int32_t)prow;
        score[dst] = cv[best];
        if (ct[best] < 0) {
            tok[dst] = eos; fin[dst] = 1; nemit[dst] = cur.nemit[prow];
        } else {
            tok[dst] = ct[best]; fin[dst] = ct[best] == eos ? 1 : 0; nemit[dst] = cur.nemit[prow] + 1;
        }
    }
}

// one thread per sentence: decide whether it ended, and if so write its best beam to the outputs
__global__ __launch_bounds__(256) void finish_kernel(tst_rows cur, const int32_t* __restrict__ parent,
                                                     const int32_t* __restrict__ tok, const float* __restrict__ score,
                                                     const int32_t* __restrict__ fin, const int32_t* __restrict__ nemit,
                                                     const float* __restrict__ lp, int64_t n_sent, int k, int T,
                                                     int last_step, int top_ends, int32_t* out_tok, int32_t* out_len,
                                                     float* out_score, int32_t* alive) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= n_sent) return;
    const int64_t base = a * k;
    bool all_fin = true;
    for (int j = 0; j < k; ++j) all_fin = all_fin && fin[base + j] != 0;
    // top_ends: the best beam (slot 0, the highest score) has ended and nothing can overtake it
    const bool done = all_fin || last_step != 0 || (top_ends != 0 && fin[base] != 0);
    alive[a] = done ? 0 : 1;
    if (!done) return;
    int best = 0;
    float br = score[base] / lp[nemit[base]];
    for (int j = 1; j < k; ++j) {
        const float rj = score[base + j] / lp[nemit[base + j]];
        if (rj > br) { br = rj; best = j; }
    }
    const int64_t sel = base + best;
    const int64_t p = parent[sel];
    const int64_t s = cur.sent[p];
    const int plen = 1 + cur.nemit[p] - cur.fin[p];
    const int olen = 1 + nemit[sel] - fin[sel];
    for (int i = 0; i < plen; ++i) out_tok[s * (T + 1) + i] = cur.hist[p * (T + 1) + i];
    if (olen > plen) out_tok[s * (T + 1) + plen] = tok[sel];
    out_len[s] = olen;
    out_score[s] = score[sel];
}

// exclusive prefix sum of alive[0..n) by one workgroup of 1024 threads; pos[a] = new sentence index
__global__ __launch_bounds__(1024) void scan_alive_kernel(const int32_t* __restrict__ alive, int32_t* __restrict__ pos,
                                                          int64_t n, int32_t* __restrict__ n_alive) {
    __shared__ int wsum[16];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo = (int64_t)t * per, hi = lo + per < n ? lo + per : n;
    int local = 0;
    for (int64_t i = lo; i < hi; ++i) local += alive[i];
    // inclusive scan of `local` across the workgroup
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
    for (int64_t i = lo; i < hi; ++i) { pos[i] = run; run += alive[i]; }
    if (t == 1023) *n_alive = wbase + inc;
}

// one wave per new row slot (a, j) of a surviving sentence
__global__ __launch_bounds__(256) void gather_state_kernel(tst_rows cur, tst_rows nxt, const int32_t* __restrict__ parent,
                                                           const int32_t* __restrict__ tok, const float* __restrict__ score,
                                                           const int32_t* __restrict__ fin, const int32_t* __restrict__ nemit,
                                                           const int32_t* __restrict__ alive, const int32_t* __restrict__ pos,
                                                           int64_t n_rows, int k, int step, int T, int32_t* row_parent) {
    const int lane = threadIdx.x & 63;
    const int64_t slot = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= n_rows) return;
    const int64_t a = slot / k;
    if (!alive[a]) return;
    const int j = (int)(slot - a * k);
    const int64_t dst = (int64_t)pos[a] * k + j;
    const int64_t p = parent[slot];
    const int plen = 1 + cur.nemit[p] - cur.fin[p];
    const int nlen = 1 + nemit[slot] - fin[slot];
    for (int i = lane; i < plen; i += 64) nxt.hist[dst * (T + 1) + i] = cur.hist[p * (T + 1) + i];
    if (cur.anc != nullptr) {
        for (int i = lane; i < step; i += 64) nxt.anc[dst * T + i] = cur.anc[p * T + i];
    }
    if (lane == 0) {
        if (nlen > plen) nxt.hist[dst * (T + 1) + plen] = tok[slot];
        if (cur.anc != nullptr && step < T) nxt.anc[dst * T + step] = (int32_t)p;
        nxt.sent[dst] = cur.sent[p];
        nxt.tok[dst] = tok[slot];
        nxt.score[dst] = score[slot];
        nxt.fin[dst] = fin[slot];
        nxt.nemit[dst] = nemit[slot];
        row_parent[dst] = (int32_t)p;
    }
}

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

template <typename T>
inline int row_vec_ok(const T* x, int64_t ld) { return (tst_aligned16(x) && ld % Elem<T>::EPC == 0) ? 1 : 0; }

template <typename T, int K>
int launch_finish(const T* logits, int64_t ld, int64_t R, int64_t V, const float* parts, float* tv, int32_t* ti, float* lse,
                  hipStream_t st, const char* name) {
    hipLaunchKernelGGL((topk_from_parts_kernel<T, K>), dim3(blocks_for(R, 4)), dim3(256), 0, st, logits, ld, (int)V, parts,
                       (int)((V + 15) / 16), R, tv, ti, lse);
    return tst_after_launch(name);
}

template <typename T>
int finish_entry(const T* logits, int64_t ld, int64_t R, int64_t V, int64_t k, const float* parts, float* top_val,
                 int32_t* top_idx, float* lse, void* stream, const char* name) {
    TST_REQUIRE(logits && parts && top_val && top_idx && lse, TST_ERR_ARG, "%s: null pointer", name);
    TST_REQUIRE(R >= 0 && V > 0 && ld >= V && V < (1ll << 30) && R < (1ll << 31), TST_ERR_ARG, "%s: bad sizes", name);
    TST_REQUIRE(k >= 1 && k <= 8 && k <= V, TST_ERR_UNSUPPORTED, "%s: need 1 <= k <= min(8, V)", name);
    TST_REQUIRE((reinterpret_cast<uintptr_t>(parts) & 7u) == 0, TST_ERR_ARG, "%s: parts must be 8-byte aligned", name);
    if (R == 0) return TST_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: return launch_finish<T, 1>(logits, ld, R, V, parts, top_val, top_idx, lse, st, name);
        case 2: return launch_finish<T, 2>(logits, ld, R, V, parts, top_val, top_idx, lse, st, name);
        case 3: return launch_finish<T, 3>(logits, ld, R, V, parts, top_val, top_idx, lse, st, name);
        case 4: return launch_finish<T, 4>(logits, ld, R, V, parts, top_val, top_idx, lse, st, name);
        case 5: return launch_finish<T, 5>(logits, ld, R, V, parts, top_val, top_idx, lse, st, name);
        case 6: return launch_finish<T, 6>(logits, ld, R, V, parts, top_val, top_idx, lse, st, name);
        case 7: return launch_finish<T, 7>(logits, ld, R, V, parts, top_val, top_idx, lse, st, name);
        default: return launch_finish<T, 8>(logits, ld, R, V, parts, top_val, top_idx, lse, st, name);
    }
}

template <typename T, int K>
int launch_topk(const T* logits, int64_t ld, int64_t R, int64_t V, float* tv, int32_t* ti, float* lse, hipStream_t st,
                const char* name) {
    constexpr int RPB = 256 / (64 * (16 / Elem<T>::EPC));   // rows per workgroup
    const dim3 grid(blocks_for(R, RPB));
    if (row_vec_ok(logits, ld))
        hipLaunchKernelGGL((logsoftmax_topk_kernel<T, K, true>), grid, dim3(256), 0, st, logits, ld, (int)V, R, tv, ti, lse);
    else
        hipLaunchKernelGGL((logsoftmax_topk_kernel<T, K, false>), grid, dim3(256), 0, st, logits, ld, (int)V, R, tv, ti, lse);
    return tst_after_launch(name);
}

template <typename T>
int topk_entry(const T* logits, int64_t ld, int64_t R, int64_t V, int64_t k, float* top_val, int32_t* top_idx, float* lse,
               void* stream, const char* name) {
    TST_REQUIRE(logits && top_val && top_idx && lse, TST_ERR_ARG, "%s: null pointer", name);
    TST_REQUIRE(R >= 0 && V > 0 && ld >= V && V < (1ll << 30) && R < (1ll << 31), TST_ERR_ARG, "%s: bad sizes", name);
    TST_REQUIRE(k >= 1 && k <= 8 && k <= V, TST_ERR_UNSUPPORTED, "%s: need 1 <= k <= min(8, V)", name);
    if (R == 0) return TST_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: return launch_topk<T, 1>(logits, ld, R, V, top_val, top_idx, lse, st, name);
        case 2: return launch_topk<T, 2>(logits, ld, R, V, top_val, top_idx, lse, st, name);
        case 3: return launch_topk<T, 3>(logits, ld, R, V, top_val, top_idx, lse, st, name);
        case 4: return launch_topk<T, 4>(logits, ld, R, V, top_val, top_idx, lse, st, name);
        case 5: return launch_topk<T, 5>(logits, ld, R, V, top_val, top_idx, lse, st, name);
        case 6: return launch_topk<T, 6>(logits, ld, R, V, top_val, top_idx, lse, st, name);
        case 7: return launch_topk<T, 7>(logits, ld, R, V, top_val, top_idx, lse, st, name);
        default: return launch_topk<T, 8>(logits, ld, R, V, top_val, top_idx, lse, st, name);
    }
}

template <typename T>
int argmax_entry(const T* logits, int64_t ld, int64_t R, int64_t V, int32_t* tok, void* stream, const char* name) {
    TST_REQUIRE(logits && tok, TST_ERR_ARG, "%s: null pointer", name);
    TST_REQUIRE(R >= 0 && V > 0 && ld >= V && V < (1ll << 30) && R < (1ll << 31), TST_ERR_ARG, "%s: bad sizes", name);
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(argmax_rows_kernel<T>, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, ld, (int)V, tok,
                       row_vec_ok(logits, ld));
    return tst_after_launch(name);
}

template <typename T>
int sample_entry(const T* logits, int64_t ld, int64_t R, int64_t V, float temperature, uint32_t seed, uint32_t step,
                 const int32_t* sent, int32_t* tok, void* stream, const char* name) {
    TST_REQUIRE(logits && sent && tok, TST_ERR_ARG, "%s: null pointer", name);
    TST_REQUIRE(R >= 0 && V > 0 && ld >= V && V < (1ll << 30) && R < (1ll << 31) && temperature > 0.0f, TST_ERR_ARG,
                "%s: bad sizes or temperature <= 0", name);
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(sample_rows_kernel<T>, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, ld, (int)V,
                       temperature, seed, step, sent, tok, row_vec_ok(logits, ld));
    return tst_after_launch(name);
}

bool rows_ok(const tst_rows* r, bool need_anc) {
    return r && r->hist && r->sent && r->tok && r->score && r->fin && r->nemit && (!need_anc || r->anc);
}

}  // namespace

extern "C" int tst_argmax_rows_f32(const float* logits, int64_t ld, int64_t R, int64_t V, int32_t* tok, void* stream) {
    return argmax_entry<float>(logits, ld, R, V, tok, stream, "tst_argmax_rows_f32");
}

extern "C" int tst_argmax_rows_bf16(const void* logits, int64_t ld, int64_t R, int64_t V, int32_t* tok, void* stream) {
    return argmax_entry<__bf16>((const __bf16*)logits, ld, R, V, tok, stream, "tst_argmax_rows_bf16");
}

extern "C" int tst_sample_rows_f32(const float* logits, int64_t ld, int64_t R, int64_t V, float temperature, uint32_t seed,
                                   uint32_t step, const int32_t* sent, int32_t* tok, void* stream) {
    return sample_entry<float>(logits, ld, R, V, temperature, seed, step, sent, tok, stream, "tst_sample_rows_f32");
}

extern "C" int tst_sample_rows_bf16(const void* logits, int64_t ld, int64_t R, int64_t V, float temperature, uint32_t seed,
                                    uint32_t step, const int32_t* sent, int32_t* tok, void* stream) {
    return sample_entry<__bf16>((const __bf16*)logits, ld, R, V, temperature, seed, step, sent, tok, stream,
                                "tst_sample_rows_bf16");
}

extern "C" int tst_logsoftmax_topk_f32(const float* logits, int64_t ld, int64_t R, int64_t V, int64_t k, float* top_val,
                                       int32_t* top_idx, float* lse, void* stream) {
    return topk_entry<float>(logits, ld, R, V, k, top_val, top_idx, lse, stream, "tst_logsoftmax_topk_f32");
}

extern "C" int tst_logsoftmax_topk_bf16(const void* logits, int64_t ld, int64_t R, int64_t V, int64_t k, float* top_val,
                                        int32_t* top_idx, float* lse, void* stream) {
    return topk_entry<__bf16>((const __bf16*)logits, ld, R, V, k, top_val, top_idx, lse, stream, "tst_logsoftmax_topk_bf16");
}

extern "C" int tst_topk_from_parts_f32(const float* logits, int64_t ld, const float* parts, int64_t R, int64_t V, int64_t k,
                                       float* top_val, int32_t* top_idx, float* lse, void* stream) {
    return finish_entry<float>(logits, ld, R, V, k, parts, top_val, top_idx, lse, stream, "tst_topk_from_parts_f32");
}

extern "C" int tst_topk_from_parts_bf16(const void* logits, int64_t ld, const float* parts, int64_t R, int64_t V, int64_t k,
                                        float* top_val, int32_t* top_idx, float* lse, void* stream) {
    return finish_entry<__bf16>((const __bf16*)logits, ld, R, V, k, parts, top_val, top_idx, lse, stream,
                                "tst_topk_from_parts_bf16");
}

extern "C" int tst_greedy_select(const int32_t* argmax_tok, int64_t n_rows, int32_t eos, int32_t* sel_parent,
                                 int32_t* sel_tok, float* sel_score, int32_t* sel_fin, int32_t* sel_nemit,
                                 const tst_rows* cur, void* stream) {
    TST_REQUIRE(argmax_tok && sel_parent && sel_tok && sel_score && sel_fin && sel_nemit && rows_ok(cur, false), TST_ERR_ARG,
                "tst_greedy_select: null pointer");
    TST_REQUIRE(n_rows >= 0, TST_ERR_ARG, "tst_greedy_select: bad size");
    if (n_rows == 0) return TST_OK;
    hipLaunchKernelGGL(greedy_select_kernel, dim3(blocks_for(n_rows, 256)), dim3(256), 0, (hipStream_t)stream, argmax_tok,
                       n_rows, eos, sel_parent, sel_tok, sel_score, sel_fin, sel_nemit, *cur);
    return tst_after_launch("tst_greedy_select");
}

extern "C" int tst_beam_merge(const tst_rows* cur, const float* top_val, const int32_t* top_idx, const float* lse,
                              int64_t n_sent, int64_t k, int64_t k_in, int32_t eos, int32_t* sel_parent, int32_t* sel_tok,
                              float* sel_score, int32_t* sel_fin, int32_t* sel_nemit, void* stream) {
    TST_REQUIRE(rows_ok(cur, false) && top_val && top_idx && lse && sel_parent && sel_tok && sel_score && sel_fin &&
                    sel_nemit, TST_ERR_ARG, "tst_beam_merge: null pointer");
    TST_REQUIRE(n_sent >= 0 && k >= 1 && k <= 8 && k_in >= 1 && k_in <= k, TST_ERR_ARG,
                "tst_beam_merge: need 1 <= k_in <= k <= 8");
    if (n_sent == 0) return TST_OK;
    hipLaunchKernelGGL(beam_merge_kernel, dim3(blocks_for(n_sent, 256)), dim3(256), 0, (hipStream_t)stream, *cur, top_val,
                       top_idx, lse, n_sent, (int)k, (int)k_in, eos, sel_parent, sel_tok, sel_score, sel_fin, sel_nemit);
    return tst_after_launch("tst_beam_merge");
}

extern "C" int tst_retire_compact(const tst_rows* cur, const tst_rows* nxt, const int32_t* sel_parent,
                                  const int32_t* sel_tok, const float* sel_score, const int32_t* sel_fin,
                                  const int32_t* sel_nemit, const float* lp, int64_t n_sent, int64_t k, int64_t step,
                                  int64_t T, int32_t eos, int last_step, int top_ends, int32_t* out_tok,
                                  int32_t* out_len, float* out_score, int32_t* scratch_alive_pos, int32_t* row_parent,
                                  int32_t* n_alive, void* stream) {
    (void)eos;
    TST_REQUIRE(rows_ok(cur, false) && rows_ok(nxt, cur && cur->anc) && sel_parent && sel_tok && sel_score && sel_fin &&
                    sel_nemit && lp && out_tok && out_len && out_score && scratch_alive_pos && row_parent && n_alive,
                TST_ERR_ARG, "tst_retire_compact: null pointer");
    TST_REQUIRE(n_sent >= 0 && k >= 1 && k <= 8 && step >= 0 && step < T && n_sent * k < (1ll << 31), TST_ERR_ARG,
                "tst_retire_compact: bad sizes (need 0 <= step < T, 1 <= k <= 8)");
    hipStream_t st = (hipStream_t)stream;
    int32_t* alive = scratch_alive_pos;
    int32_t* pos = scratch_alive_pos + n_sent;
    if (n_sent > 0) {
        hipLaunchKernelGGL(finish_kernel, dim3(blocks_for(n_sent, 256)), dim3(256), 0, st, *cur, sel_parent, sel_tok,
                           sel_score, sel_fin, sel_nemit, lp, n_sent, (int)k, (int)T, last_step, top_ends, out_tok, out_len,
                           out_score, alive);
    }
    hipLaunchKernelGGL(scan_alive_kernel, dim3(1), dim3(1024), 0, st, alive, pos, n_sent, n_alive);
    if (n_sent > 0) {
        hipLaunchKernelGGL(gather_state_kernel, dim3(blocks_for(n_sent * k, 4)), dim3(256), 0, st, *cur, *nxt, sel_parent,
                           sel_tok, sel_score, sel_fin, sel_nemit, alive, pos, n_sent * k, (int)k, (int)step, (int)T,
                           row_parent);
    }
    return tst_after_launch("tst_retire_compact");
}
