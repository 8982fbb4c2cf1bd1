// Token selection and decode-loop bookkeeping on the device: arg-max, log-sum-exp + top-k over the
// vocabulary (the HBM-bound pass over the [rows, V] logits), beam merge, retirement of finished
// sentences and batch compaction.  No MFMA here: coalesced 16-byte reads, wave butterflies, LDS only
// to combine the four waves of a workgroup.
#include "tst_common.h"
#include <limits.h>
#include <stdlib.h>
#include <atomic>
#include <math.h>

namespace {

// (value desc, index asc) ordering
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

template <int OFF>
__device__ __forceinline__ void argmax_step(float& v, int& i) {
    const float ov = lane_xor<OFF>(v);
    const int oi = lane_xor_i<OFF>(i);
    if (better(ov, oi, v, i)) { v = ov; i = oi; }
}

__device__ __forceinline__ void wave_argmax(float& v, int& i) {
    argmax_step<32>(v, i);
    argmax_step<16>(v, i);
    argmax_step<8>(v, i);
    argmax_step<4>(v, i);
    argmax_step<2>(v, i);
    argmax_step<1>(v, i);
}

// block-wide argmax over NW waves (256 threads unless stated); result valid in every thread. red_* are LDS arrays of NW.
template <int NW = 4>
__device__ __forceinline__ void block_argmax(float& v, int& i, float* red_v, int* red_i) {
    wave_argmax(v, i);
    const int w = threadIdx.x >> 6;
    __syncthreads();  // protects red_* against the previous round's readers
    if ((threadIdx.x & 63) == 0) { red_v[w] = v; red_i[w] = i; }
    __syncthreads();
    v = red_v[0]; i = red_i[0];
#pragma unroll
    for (int q = 1; q < NW; ++q)
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

// a chunk that lies fully inside the row, first pass.  bf16 rows are read nontemporally (+10 % on that pass: 4.4 -> 4.85 TB/s)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_chunk_nt(const float* xr, int c, float (&o)[4]) {
    load_chunk(xr, c, INT_MAX, true, o);   // fp32 rows: measured equal or slower nontemporal inside the decode loop: ordinary loads
}
__device__ __forceinline__ void load_chunk_nt(const __bf16* xr, int c, float (&o)[8]) {
    const u32x4_t u = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(xr) + c);
    o[0] = __uint_as_float(u[0] << 16); o[1] = __uint_as_float(u[0] & 0xffff0000u);
    o[2] = __uint_as_float(u[1] << 16); o[3] = __uint_as_float(u[1] & 0xffff0000u);
    o[4] = __uint_as_float(u[2] << 16); o[5] = __uint_as_float(u[2] & 0xffff0000u);
    o[6] = __uint_as_float(u[3] << 16); o[7] = __uint_as_float(u[3] & 0xffff0000u);
}

// One workgroup of BLOCK threads per row.  With few rows (one workgroup per CU or less) 1024 threads put four times as
// many loads in flight per row; the first maximal index is the same whatever the split.
template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void argmax_rows_kernel(const T* __restrict__ x, int64_t ld, int V,
                                                            int32_t* __restrict__ tok, int vec_ok) {
    constexpr int EPC = Elem<T>::EPC;
    __shared__ float red_v[BLOCK / 64];
    __shared__ int red_i[BLOCK / 64];
    const T* xr = x + (int64_t)blockIdx.x * ld;
    const int nch = (V + EPC - 1) / EPC;
    float bv = -INFINITY;
    int bi = INT_MAX;
    for (int c = threadIdx.x; c < nch; c += BLOCK) {
        float e[EPC];
        load_chunk(xr, c, V, vec_ok != 0, e);
#pragma unroll
        for (int q = 0; q < EPC; ++q)
            if (e[q] > bv) { bv = e[q]; bi = EPC * c + q; }
    }
    block_argmax<BLOCK / 64>(bv, bi, red_v, red_i);
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

constexpr int CAND_CAP = 1024;  // LDS candidate list of the top-k filter

// exp for arguments <= 0.  fp32 logits: the fixed polynomial (bit-identical on host and device).  bf16 logits
// carry 8 significant bits and are checked against a tolerance, and their rows hold twice the elements per
// byte, so the pass is bound by the exponentials: there the hardware's exp2 (v_exp_f32, ~1 ulp) is used.
template <bool FAST>
__device__ __forceinline__ float exp_le0(float x) {
    if (FAST) return x > -100.0f ? __builtin_amdgcn_exp2f(x * 1.44269504088896341f) : 0.0f;   // false for NaN too
    return d_exp_le0(x);
}

// online log-sum-exp update with a group of 4 chunks (ascending index order within the group)
template <int EPC>
__device__ __forceinline__ void lse_consume(const float (&vv)[4][EPC], float& m, float& sacc) {
    constexpr bool FAST = EPC == 8;
    float mn = m;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < EPC; ++q) mn = fmaxf(mn, vv[u][q]);
    sacc = sacc * exp_le0<FAST>(m - mn);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < EPC; ++q) sacc = sacc + exp_le0<FAST>(vv[u][q] - mn);
    m = mn;
}

__device__ __forceinline__ float elem_at(const float* xr, int i) { return xr[i]; }
__device__ __forceinline__ float elem_at(const __bf16* xr, int i) { return (float)xr[i]; }

// One workgroup per row, ONE pass over the row: each logit is read from HBM once and only a few registers
// are live, so occupancy is high and the loads overlap the arithmetic.  Thread t owns the 16-byte chunks
// t, t + 256, ... and consumes them four at a time (values x_i in ascending index order) into an online
// log-sum-exp state:  m' = max(m, x_i);  s = s * exp(m - m') + sum_i exp(x_i - m');  m = m'.
// The 256 states go to LDS and the FIRST WAVE finishes alone (no further workgroup-wide reductions):
// M = max of the 256 m, total = butterfly sum of s * exp(m - M) per group of 64 states, groups left to right,
// lse = M + log(total).
// Top K: m is also the thread's largest element.  tau = the K-th largest of the 256 thread maxima; at least K
// elements are >= tau and they can only sit in threads whose own maximum is >= tau, so only the chunks of those
// (about K) threads are re-read - by the whole workgroup, one chunk per lane - and their elements >= tau pushed
// to an LDS list, from which the first wave picks the K
// largest (value desc, index asc) by K wave arg-max rounds.  If the list overflows (many equal values) the
// first wave's K rounds run over the whole row instead.  Three workgroup barriers in all.
template <typename T, int K, bool VEC>
__global__ __launch_bounds__(256) void logsoftmax_topk_kernel(const T* __restrict__ x, int64_t ld, int V,
                                                              float* __restrict__ top_val, int32_t* __restrict__ top_idx,
                                                              float* __restrict__ lse) {
    constexpr int EPC = Elem<T>::EPC;
    __shared__ float st_m[256], st_s[256];
    __shared__ float tau_s;
    __shared__ int hot[256], n_hot;   // threads whose maximum reaches tau
    __shared__ float cand_v[CAND_CAP];
    __shared__ int cand_i[CAND_CAP];
    __shared__ int cand_n;
    const int t = threadIdx.x;
    const int64_t r = blockIdx.x;
    const T* xr = x + r * ld;
    const int nch = (V + EPC - 1) / EPC;
    float m = -INFINITY, sacc = 0.0f;
    // a chunk beyond the row reads as -inf and contributes exp(-inf) = 0
    int c0 = t;
    if (VEC) {  // groups that lie fully inside the row: unguarded 16-byte loads
        const int nfull = V / EPC;
        for (; c0 + 768 < nfull; c0 += 1024) {
            float vv[4][EPC];
#pragma unroll
            for (int u = 0; u < 4; ++u) load_chunk_nt(xr, c0 + 256 * u, vv[u]);
            lse_consume<EPC>(vv, m, sacc);
        }
    }
    for (; c0 < nch; c0 += 1024) {  // ragged end of the row (and rows that are not 16-byte aligned)
        float vv[4][EPC];
#pragma unroll
        for (int u = 0; u < 4; ++u) load_chunk(xr, c0 + 256 * u, V, VEC, vv[u]);
        lse_consume<EPC>(vv, m, sacc);
    }
    st_m[t] = m;
    st_s[t] = sacc;
    __syncthreads();
    if (t < 64) {
        float mg[4], sg[4], m0[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { mg[g] = st_m[t + 64 * g]; sg[g] = st_s[t + 64 * g]; m0[g] = mg[g]; }
        const float M = wave_max(fmaxf(fmaxf(mg[0], mg[1]), fmaxf(mg[2], mg[3])));
        float part[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) part[g] = wave_sum_butterfly(sg[g] * d_exp_le0(mg[g] - M));
        if (t == 0) lse[r] = M + d_log(((part[0] + part[1]) + part[2]) + part[3]);
        // K-th largest thread maximum (ties between threads: lower thread first)
        float tau = -INFINITY;
        for (int o = 0; o < K; ++o) {
            float v = mg[0];
            int id = t;
#pragma unroll
            for (int g = 1; g < 4; ++g)
                if (mg[g] > v) { v = mg[g]; id = t + 64 * g; }
            const int mine = id;
            wave_argmax(v, id);
            tau = v;
            if (id == mine) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (id == t + 64 * g) mg[g] = -INFINITY;
            }
        }
        int nh = 0;   // list of the (about K) threads whose maximum reaches tau
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bool is_hot = m0[g] >= tau;
            const unsigned long long bal = __ballot(is_hot);
            if (is_hot) hot[nh + __popcll(bal & ((1ull << t) - 1ull))] = t + 64 * g;
            nh += __popcll(bal);
        }
        if (t == 0) { tau_s = tau; cand_n = 0; n_hot = nh; }
    }
    __syncthreads();
    // the whole workgroup re-reads the chunks of the hot threads (chunk h + 256 i of thread h), one chunk per
    // work item, so that the loads are spread over all lanes instead of running serially in K threads
    const float tau = tau_s;
    const int ipt = (nch + 255) >> 8;   // chunks per thread
    const int items = n_hot * ipt;
    for (int it = t; it < items; it += 256) {
        const int c = hot[it / ipt] + 256 * (it % ipt);
        if (c >= nch) continue;
        float e[EPC];
        load_chunk(xr, c, V, VEC, e);
#pragma unroll
        for (int q = 0; q < EPC; ++q)
            if (e[q] >= tau && EPC * c + q < V) {
                const int at = atomicAdd(&cand_n, 1);
                if (at < CAND_CAP) { cand_v[at] = e[q]; cand_i[at] = EPC * c + q; }
            }
    }
    __syncthreads();
    if (t >= 64) return;
    const int ncand = cand_n;
    if (ncand <= CAND_CAP) {
        for (int o = 0; o < K; ++o) {
            float bv = -INFINITY;
            int bi = INT_MAX, be = -1;
            for (int e = t; e < ncand; e += 64)
                if (better(cand_v[e], cand_i[e], bv, bi)) { bv = cand_v[e]; bi = cand_i[e]; be = e; }
            const int mine = bi;
            wave_argmax(bv, bi);
            if (be >= 0 && bi == mine) { cand_v[be] = -INFINITY; cand_i[be] = INT_MAX; }
            if (t == 0) { top_val[r * K + o] = bv; top_idx[r * K + o] = bi == INT_MAX ? -1 : bi; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the removal is seen by the wave's next round
        }
    } else {
        float lastv = INFINITY;
        int lasti = -1;
        for (int o = 0; o < K; ++o) {
            float bv = -INFINITY;
            int bi = INT_MAX;
            for (int id = t; id < V; id += 64) {
                const float e = elem_at(xr, id);
                const bool after = e < lastv || (e == lastv && id > lasti);
                if (after && better(e, id, bv, bi)) { bv = e; bi = id; }
            }
            wave_argmax(bv, bi);
            lastv = bv; lasti = bi;
            if (t == 0) { top_val[r * K + o] = bv; top_idx[r * K + o] = bi == INT_MAX ? -1 : bi; }
        }
    }
}

// Log-sum-exp + top K of bf16 logits whose vocabulary projection has already reduced every slice of 64 columns to (largest
// value, sum of exp(value - largest)) in its epilogue (tst_vocabproj_stats_bf16): a row then costs 8 * n_slices bytes of
// statistics plus the few slices that can hold one of its K largest elements, instead of its 2 * V bytes.
// One workgroup of 8 waves per block of 32 rows (the unit of the statistics' layout): the block's statistics are staged
// into LDS transposed to [row][slice], then every wave finishes 4 rows on its own (no further workgroup barrier):
//   M = max of the slice maxima;  lse = M + log(sum over slices of sum_s * exp(max_s - M));
//   tau = the K-th largest slice maximum: at least K elements are >= tau and they sit in slices whose maximum is >= tau;
//   those slices are re-read from the logits (128 bytes each), their elements >= tau go to a list, from which K rounds of
//   wave arg-max (value descending, index ascending) pick the result, the same selection as logsoftmax_topk_kernel.
// Many equal values can overflow the lists; the wave then runs its K rounds over the whole row.
constexpr int STAT_ROWS = 32, STAT_NSMAX = 512, STAT_LD = STAT_NSMAX + 1, STAT_HOT = 64, STAT_CAND = 256;

template <int K>
__global__ __launch_bounds__(512) void topk_from_stats_kernel(const __bf16* __restrict__ x, int64_t ld, int V,
                                                              const float* __restrict__ part, int n_slices, int R,
                                                              float* __restrict__ top_val, int32_t* __restrict__ top_idx,
                                                              float* __restrict__ lse) {
    __shared__ float sm[STAT_ROWS * STAT_LD], ss[STAT_ROWS * STAT_LD];
    __shared__ int hot[8][STAT_HOT];
    __shared__ float cand_v[8][STAT_CAND];
    __shared__ int cand_i[8][STAT_CAND];
    __shared__ int cand_n[8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * STAT_ROWS;
    // stage: float4 number q of the block holds rows 4 (q % 16) .. + 3 (first 32: maxima, then sums) of slice q / 16
    const float4* src = reinterpret_cast<const float4*>(part + (int64_t)blockIdx.x * n_slices * 64);
    for (int q = t; q < n_slices * 16; q += 512) {
        const float4 v = src[q];
        const int sl = q >> 4, c = (q & 15) * 4;
        float* dst = (c < 32 ? sm : ss) + (c & 31) * STAT_LD + sl;
        dst[0] = v.x; dst[STAT_LD] = v.y; dst[2 * STAT_LD] = v.z; dst[3 * STAT_LD] = v.w;
    }
    __syncthreads();
    constexpr int NJ = STAT_NSMAX / 64;
    for (int rr = 0; rr < STAT_ROWS / 8; ++rr) {
        const int lr = wave * (STAT_ROWS / 8) + rr;
        const int64_t r = r0 + lr;
        if (r >= R) break;                                   // wave-uniform
        const __bf16* xr = x + r * ld;
        float mg[NJ], m0[NJ];
        float lm = -INFINITY;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int sl = lane + 64 * j;
            mg[j] = sl < n_slices ? sm[lr * STAT_LD + sl] : -INFINITY;
            m0[j] = mg[j];
            lm = fmaxf(lm, mg[j]);
        }
        const float M = wave_max(lm);
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int sl = lane + 64 * j;
            if (sl < n_slices) acc += ss[lr * STAT_LD + sl] * __builtin_amdgcn_exp2f((mg[j] - M) * 1.44269504088896341f);
        }
        const float total = wave_sum_butterfly(acc);
        if (lane == 0) lse[r] = M + d_log(total);
        // K-th largest slice maximum (ties between slices: lower slice first)
        float tau = -INFINITY;
        for (int o = 0; o < K; ++o) {
            float v = mg[0];
            int id = lane;
#pragma unroll
            for (int j = 1; j < NJ; ++j)
                if (mg[j] > v) { v = mg[j]; id = lane + 64 * j; }
            const int mine = id;
            wave_argmax(v, id);
            tau = v;
            if (id == mine) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (id == lane + 64 * j) mg[j] = -INFINITY;
            }
        }
        int nh = 0;   // slices whose maximum reaches tau
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool is_hot = m0[j] >= tau && lane + 64 * j < n_slices;
            const unsigned long long bal = __ballot(is_hot);
            const int at = nh + __popcll(bal & ((1ull << lane) - 1ull));
            if (is_hot && at < STAT_HOT) hot[wave][at] = lane + 64 * j;
            nh += __popcll(bal);
        }
        if (lane == 0) cand_n[wave] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        bool overflow = nh > STAT_HOT;
        if (!overflow) {
            // 8 lanes per hot slice, 16 bytes (8 elements) each
            for (int it = lane; it < nh * 8; it += 64) {
                const int c = hot[wave][it >> 3] * 8 + (it & 7);     // 16-byte chunk of the row
                float e[8];
                load_chunk(xr, c, V, true, e);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (e[q] >= tau && 8 * c + q < V) {
                        const int at = atomicAdd(&cand_n[wave], 1);
                        if (at < STAT_CAND) { cand_v[wave][at] = e[q]; cand_i[wave][at] = 8 * c + q; }
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            overflow = cand_n[wave] > STAT_CAND;
        }
        if (!overflow) {
            const int ncand = cand_n[wave];
            for (int o = 0; o < K; ++o) {
                float bv = -INFINITY;
                int bi = INT_MAX, be = -1;
                for (int e = lane; e < ncand; e += 64)
                    if (better(cand_v[wave][e], cand_i[wave][e], bv, bi)) { bv = cand_v[wave][e]; bi = cand_i[wave][e]; be = e; }
                const int mine = bi;
                wave_argmax(bv, bi);
                if (be >= 0 && bi == mine) { cand_v[wave][be] = -INFINITY; cand_i[wave][be] = INT_MAX; }
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
}

// token of a row's arg-max keys (tst_ln_vocab_argmax_f32: TST_ARGMAX_SLOTS keys per row, 16-byte aligned): the largest key
// decides; 0 = no candidate, else ~column in the lower word
__device__ __forceinline__ int32_t key_token(const unsigned long long* __restrict__ row_keys) {
    unsigned long long key = 0ull;
#pragma unroll
    for (int s = 0; s < TST_ARGMAX_SLOTS; s += 2) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(row_keys + s);
        key = v.x > key ? v.x : key;
        key = v.y > key ? v.y : key;
    }
    return key == 0ull ? 0 : (int32_t)~(uint32_t)key;
}

// am: the selected token of every row; keys (when not NULL): arg-max keys instead
__global__ __launch_bounds__(256) void greedy_select_kernel(const int32_t* __restrict__ am, const unsigned long long* __restrict__ keys,
                                                            int64_t n, int32_t eos, int32_t* parent, int32_t* tok, float* score,
                                                            int32_t* fin, int32_t* nemit, tst_rows cur) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int32_t tk = keys != nullptr ? key_token(keys + r * TST_ARGMAX_SLOTS) : am[r];
    parent[r] = (int32_t)r;
    tok[r] = tk;
    score[r] = 0.0f;
    fin[r] = tk == eos ? 1 : 0;
    nemit[r] = cur.nemit[r] + 1;
}

// One wave per sentence.  Lane j * k + rk holds the offer (beam j, rank rk); a finished beam has one offer (its
// own score, rank 0).  k rounds of wave arg-max (value desc, lane asc) pick the new beams: ties go to the lower
// beam, then the lower rank.  Inactive lanes compete with value -inf and an index past every real lane, so an
// offer whose score is -inf still wins over "no offer".
__global__ __launch_bounds__(256) void beam_merge_kernel(tst_rows cur, const float* __restrict__ top_val,
                                                         const int32_t* __restrict__ top_idx,
                                                         const float* __restrict__ lse, int64_t n_sent, int k, int k_in,
                                                         int32_t eos, int32_t* parent, int32_t* tok, float* score,
                                                         int32_t* fin, int32_t* nemit) {
    const int lane = threadIdx.x & 63;
    const int64_t a = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= n_sent) return;
    const int64_t base = a * k_in;   // input rows of the sentence
    const int64_t obase = a * k;     // output slots
    const int j = lane / k, rk = lane - j * k;
    bool active = lane < k_in * k;
    float val = -INFINITY;
    int32_t tk = -1;
    if (active) {
        const int64_t row = base + j;
        if (cur.fin[row]) {
            active = rk == 0;
            val = cur.score[row];
        } else {
            val = cur.score[row] + (top_val[row * k + rk] - lse[row]);
            tk = top_idx[row * k + rk];
        }
    }
    for (int o = 0; o < k; ++o) {
        float v = active ? val : -INFINITY;
        int id = active ? lane : 1000 + lane;
        wave_argmax(v, id);
        const int64_t dst = obase + o;
        if (id >= 1000) {  // fewer offers than beams: pad with an ended, worthless hypothesis
            if (lane == 0) { parent[dst] = (int32_t)base; tok[dst] = eos; score[dst] = -INFINITY; fin[dst] = 1; nemit[dst] = 0; }
            continue;
        }
        const int32_t wt = __shfl(tk, id, 64);
        if (lane == id) active = false;
        if (lane == 0) {
            const int64_t prow = base + id / k;
            parent[dst] = (int32_t)prow;
            score[dst] = v;
            if (wt < 0) {
                tok[dst] = eos; fin[dst] = 1; nemit[dst] = cur.nemit[prow];
            } else {
                tok[dst] = wt; fin[dst] = wt == eos ? 1 : 0; nemit[dst] = cur.nemit[prow] + 1;
            }
        }
    }
}

// one thread per sentence: decide whether it ended, and if so write its best beam to the outputs
__global__ __launch_bounds__(256) void finish_kernel(tst_rows cur, const int32_t* __restrict__ parent,
                                                     const int32_t* __restrict__ tok, const float* __restrict__ score,
                                                     const int32_t* __restrict__ fin, const int32_t* __restrict__ nemit,
                                                     const float* __restrict__ lp, int64_t n_sent, int k, int T,
                                                     int last_step, int top_ends, int32_t* out_tok, int32_t* out_len,
                                                     float* out_score, int32_t* alive, const int32_t* __restrict__ n_true) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= n_sent) return;
    // n_sent may be an upper bound (a host that does not wait for the previous step's count): sentences beyond the true
    // count, kept on the device, do not exist
    if (n_true != nullptr && a >= *n_true) { alive[a] = 0; return; }
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
                                                          int64_t n, int32_t* __restrict__ n_alive, int32_t* n_alive_host) {
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
    if (t == 1023) {
        *n_alive = wbase + inc;
        if (n_alive_host != nullptr) *n_alive_host = wbase + inc;
    }
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

// Small batches: greedy selection (optional), retirement, the prefix sum and the gather in ONE workgroup of 1024
// threads (at most 1024 sentences and 1024 rows), the same arithmetic as the four kernels above, phase by phase.
template <bool GREEDY>
__global__ __launch_bounds__(1024) void tail_small_kernel(tst_rows cur, tst_rows nxt, const int32_t* __restrict__ am,
                                                          const unsigned long long* __restrict__ keys, int32_t* parent,
                                                          int32_t* tok, float* score, int32_t* fin, int32_t* nemit,
                                                          const float* __restrict__ lp, int n_sent, int k, int step, int T,
                                                          int32_t eos, int last_step, int top_ends, int32_t* out_tok,
                                                          int32_t* out_len, float* out_score, int32_t* row_parent,
                                                          int32_t* n_alive, const int32_t* __restrict__ n_true, int32_t* n_alive_host) {
    __shared__ int alive_s[1024], pos_s[1024], wsum[16];
    const int t = threadIdx.x;
    if (n_true != nullptr) n_sent = min(n_sent, (int)*n_true);   // n_sent may be an upper bound: see finish_kernel
    const int n_rows = n_sent * k;
    if (GREEDY) {   // greedy_select_kernel (k == 1)
        if (t < n_rows) {
            const int32_t tk = keys != nullptr ? key_token(keys + (int64_t)t * TST_ARGMAX_SLOTS) : am[t];
            parent[t] = t;
            tok[t] = tk;
            score[t] = 0.0f;
            fin[t] = tk == eos ? 1 : 0;
            nemit[t] = cur.nemit[t] + 1;
        }
        __syncthreads();
    }
    int al = 0;
    if (t < n_sent) {   // finish_kernel
        const int base = t * k;
        bool all_fin = true;
        for (int j = 0; j < k; ++j) all_fin = all_fin && fin[base + j] != 0;
        const bool done = all_fin || last_step != 0 || (top_ends != 0 && fin[base] != 0);
        al = done ? 0 : 1;
        if (done) {
            int best = 0;
            float br = score[base] / lp[nemit[base]];
            for (int j = 1; j < k; ++j) {
                const float rj = score[base + j] / lp[nemit[base + j]];
                if (rj > br) { br = rj; best = j; }
            }
            const int sel = base + best;
            const int64_t p = parent[sel];
            const int64_t s = cur.sent[p];
            const int plen = 1 + cur.nemit[p] - cur.fin[p];
            const int olen = 1 + nemit[sel] - fin[sel];
            for (int i = 0; i < plen; ++i) out_tok[s * (T + 1) + i] = cur.hist[p * (T + 1) + i];
            if (olen > plen) out_tok[s * (T + 1) + plen] = tok[sel];
            out_len[s] = olen;
            out_score[s] = score[sel];
        }
    }
    // scan_alive_kernel with one sentence per thread
    int inc = al;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if ((t & 63) >= off) inc += o;
    }
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    __syncthreads();
    int wbase = 0;
    for (int q = 0; q < (t >> 6); ++q) wbase += wsum[q];
    alive_s[t] = al;
    pos_s[t] = wbase + inc - al;
    if (t == 1023) {
        *n_alive = wbase + inc;
        if (n_alive_host != nullptr) *n_alive_host = wbase + inc;
    }
    __syncthreads();
    // gather_state_kernel: per-slot scalars by the slot's thread, then the history and ancestor rows element by element
    // over the whole workgroup (every load independent of the others)
    __shared__ int dst_s[1024], src_s[1024], plen_s[1024];
    {
        int dsti = -1, srci = 0, pl = 0;
        if (t < n_rows) {
            const int slot = t;
            const int a = slot / k;
            if (alive_s[a]) {
                const int j = slot - a * k;
                const int64_t dst = (int64_t)pos_s[a] * k + j;
                const int64_t p = parent[slot];
                const int plen = 1 + cur.nemit[p] - cur.fin[p];
                const int nlen = 1 + nemit[slot] - fin[slot];
                if (nlen > plen) nxt.hist[dst * (T + 1) + plen] = tok[slot];
                if (cur.anc != nullptr && step < T) nxt.anc[dst * T + step] = (int32_t)p;
                nxt.sent[dst] = cur.sent[p];
                nxt.tok[dst] = tok[slot];
                nxt.score[dst] = score[slot];
                nxt.fin[dst] = fin[slot];
                nxt.nemit[dst] = nemit[slot];
                row_parent[dst] = (int32_t)p;
                dsti = (int)dst; srci = (int)p; pl = plen;
            }
        }
        dst_s[t] = dsti; src_s[t] = srci; plen_s[t] = pl;
    }
    __syncthreads();
    const int W = T + 1;
    for (int idx = t; idx < n_rows * W; idx += 1024) {
        const int slot = idx / W, i = idx - slot * W;
        const int dst = dst_s[slot];
        if (dst >= 0 && i < plen_s[slot]) nxt.hist[(int64_t)dst * W + i] = cur.hist[(int64_t)src_s[slot] * W + i];
    }
    if (cur.anc != nullptr && step > 0) {
        for (int idx = t; idx < n_rows * step; idx += 1024) {
            const int slot = idx / step, i = idx - slot * step;
            const int dst = dst_s[slot];
            if (dst >= 0) nxt.anc[(int64_t)dst * T + i] = cur.anc[(int64_t)src_s[slot] * T + i];
        }
    }
}

// rows up to which one workgroup does the whole tail (measured crossover against the separate kernels);
// TST_SMALL_TAIL_ROWS=<n> in the environment moves it (measurements; at most 1024: one thread per row)
int64_t small_tail_rows() {
    static const int64_t v = [] {
        const char* e = getenv("TST_SMALL_TAIL_ROWS");
        const long long n = e != nullptr ? atoll(e) : 160;
        return (int64_t)(n < 0 ? 0 : n > 1024 ? 1024 : n);
    }();
    return v;
}
#define SMALL_TAIL_ROWS small_tail_rows()

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

template <typename T>
inline int row_vec_ok(const T* x, int64_t ld) { return (tst_aligned16(x) && ld % Elem<T>::EPC == 0) ? 1 : 0; }

template <typename T, int K>
int launch_topk(const T* logits, int64_t ld, int64_t R, int64_t V, float* tv, int32_t* ti, float* lse, hipStream_t st,
                const char* name) {
    if (row_vec_ok(logits, ld))
        hipLaunchKernelGGL((logsoftmax_topk_kernel<T, K, true>), dim3((unsigned)R), dim3(256), 0, st, logits, ld, (int)V, tv, ti, lse);
    else
        hipLaunchKernelGGL((logsoftmax_topk_kernel<T, K, false>), dim3((unsigned)R), dim3(256), 0, st, logits, ld, (int)V, tv, ti, lse);
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
    if (R <= 512 && V >= 8192)
        hipLaunchKernelGGL((argmax_rows_kernel<T, 1024>), dim3((unsigned)R), dim3(1024), 0, (hipStream_t)stream, logits, ld, (int)V, tok,
                           row_vec_ok(logits, ld));
    else
        hipLaunchKernelGGL((argmax_rows_kernel<T, 256>), dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, ld, (int)V, tok,
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
    TST_REQUIRE(tst_dims_ok({R, V}), TST_ERR_UNSUPPORTED, "tst_argmax_rows_f32: a size exceeds 2^31 - 1");
    return argmax_entry<float>(logits, ld, R, V, tok, stream, "tst_argmax_rows_f32");
}

extern "C" int tst_argmax_rows_bf16(const void* logits, int64_t ld, int64_t R, int64_t V, int32_t* tok, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, V}), TST_ERR_UNSUPPORTED, "tst_argmax_rows_bf16: a size exceeds 2^31 - 1");
    return argmax_entry<__bf16>((const __bf16*)logits, ld, R, V, tok, stream, "tst_argmax_rows_bf16");
}

extern "C" int tst_sample_rows_f32(const float* logits, int64_t ld, int64_t R, int64_t V, float temperature, uint32_t seed,
                                   uint32_t step, const int32_t* sent, int32_t* tok, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, V}), TST_ERR_UNSUPPORTED, "tst_sample_rows_f32: a size exceeds 2^31 - 1");
    return sample_entry<float>(logits, ld, R, V, temperature, seed, step, sent, tok, stream, "tst_sample_rows_f32");
}

extern "C" int tst_sample_rows_bf16(const void* logits, int64_t ld, int64_t R, int64_t V, float temperature, uint32_t seed,
                                    uint32_t step, const int32_t* sent, int32_t* tok, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, V}), TST_ERR_UNSUPPORTED, "tst_sample_rows_bf16: a size exceeds 2^31 - 1");
    return sample_entry<__bf16>((const __bf16*)logits, ld, R, V, temperature, seed, step, sent, tok, stream,
                                "tst_sample_rows_bf16");
}

extern "C" int tst_logsoftmax_topk_f32(const float* logits, int64_t ld, int64_t R, int64_t V, int64_t k, float* top_val,
                                       int32_t* top_idx, float* lse, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, V, k}), TST_ERR_UNSUPPORTED, "tst_logsoftmax_topk_f32: a size exceeds 2^31 - 1");
    return topk_entry<float>(logits, ld, R, V, k, top_val, top_idx, lse, stream, "tst_logsoftmax_topk_f32");
}

extern "C" int tst_logsoftmax_topk_bf16(const void* logits, int64_t ld, int64_t R, int64_t V, int64_t k, float* top_val,
                                        int32_t* top_idx, float* lse, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, V, k}), TST_ERR_UNSUPPORTED, "tst_logsoftmax_topk_bf16: a size exceeds 2^31 - 1");
    return topk_entry<__bf16>((const __bf16*)logits, ld, R, V, k, top_val, top_idx, lse, stream, "tst_logsoftmax_topk_bf16");
}

extern "C" int tst_topk_from_stats_bf16(const void* logits, int64_t ld, const float* part, int64_t n_slices, int64_t R, int64_t V,
                                        int64_t k, float* top_val, int32_t* top_idx, float* lse, void* stream) {
    TST_REQUIRE(tst_dims_ok({n_slices, R, V, k}), TST_ERR_UNSUPPORTED, "tst_topk_from_stats_bf16: a size exceeds 2^31 - 1");
    TST_REQUIRE(logits && part && top_val && top_idx && lse, TST_ERR_ARG, "tst_topk_from_stats_bf16: null pointer");
    TST_REQUIRE(R >= 0 && R < (1ll << 31) && V > 0 && ld >= V && k >= 1 && k <= 8 && k <= V, TST_ERR_ARG, "tst_topk_from_stats_bf16: bad sizes");
    TST_REQUIRE(V % 8 == 0 && ld % 8 == 0 && tst_aligned16(logits) && tst_aligned16(part), TST_ERR_ARG,
                "tst_topk_from_stats_bf16: V and ld must be multiples of 8, logits and part 16-byte aligned");
    TST_REQUIRE(n_slices == (V + 63) / 64, TST_ERR_ARG, "tst_topk_from_stats_bf16: n_slices must be ceil(V / 64)");
    TST_REQUIRE(n_slices <= STAT_NSMAX, TST_ERR_UNSUPPORTED, "tst_topk_from_stats_bf16: at most %d slices (V <= %d)", STAT_NSMAX,
                STAT_NSMAX * 64);
    if (R == 0) return TST_OK;
    const dim3 grid((unsigned)((R + STAT_ROWS - 1) / STAT_ROWS));
#define TST_STATS_TOPK(KK)                                                                                                        \
    hipLaunchKernelGGL(topk_from_stats_kernel<KK>, grid, dim3(512), 0, (hipStream_t)stream, (const __bf16*)logits, ld, (int)V, part, \
                       (int)n_slices, (int)R, top_val, top_idx, lse)
    switch (k) {
        case 1: TST_STATS_TOPK(1); break;
        case 2: TST_STATS_TOPK(2); break;
        case 3: TST_STATS_TOPK(3); break;
        case 4: TST_STATS_TOPK(4); break;
        case 5: TST_STATS_TOPK(5); break;
        case 6: TST_STATS_TOPK(6); break;
        case 7: TST_STATS_TOPK(7); break;
        default: TST_STATS_TOPK(8); break;
    }
#undef TST_STATS_TOPK
    return tst_after_launch("tst_topk_from_stats_bf16");
}

extern "C" int tst_greedy_select(const int32_t* argmax_tok, int64_t n_rows, int32_t eos, int32_t* sel_parent,
                                 int32_t* sel_tok, float* sel_score, int32_t* sel_fin, int32_t* sel_nemit,
                                 const tst_rows* cur, void* stream) {
    TST_REQUIRE(tst_dims_ok({n_rows}), TST_ERR_UNSUPPORTED, "tst_greedy_select: a size exceeds 2^31 - 1");
    TST_REQUIRE(argmax_tok && sel_parent && sel_tok && sel_score && sel_fin && sel_nemit && rows_ok(cur, false), TST_ERR_ARG,
                "tst_greedy_select: null pointer");
    TST_REQUIRE(n_rows >= 0, TST_ERR_ARG, "tst_greedy_select: bad size");
    if (n_rows == 0) return TST_OK;
    hipLaunchKernelGGL(greedy_select_kernel, dim3(blocks_for(n_rows, 256)), dim3(256), 0, (hipStream_t)stream, argmax_tok,
                       (const unsigned long long*)nullptr, n_rows, eos, sel_parent, sel_tok, sel_score, sel_fin, sel_nemit, *cur);
    return tst_after_launch("tst_greedy_select");
}

extern "C" int tst_beam_merge(const tst_rows* cur, const float* top_val, const int32_t* top_idx, const float* lse,
                              int64_t n_sent, int64_t k, int64_t k_in, int32_t eos, int32_t* sel_parent, int32_t* sel_tok,
                              float* sel_score, int32_t* sel_fin, int32_t* sel_nemit, void* stream) {
    TST_REQUIRE(tst_dims_ok({n_sent, k, k_in}), TST_ERR_UNSUPPORTED, "tst_beam_merge: a size exceeds 2^31 - 1");
    TST_REQUIRE(rows_ok(cur, false) && top_val && top_idx && lse && sel_parent && sel_tok && sel_score && sel_fin &&
                    sel_nemit, TST_ERR_ARG, "tst_beam_merge: null pointer");
    TST_REQUIRE(n_sent >= 0 && k >= 1 && k <= 8 && k_in >= 1 && k_in <= k, TST_ERR_ARG,
                "tst_beam_merge: need 1 <= k_in <= k <= 8");
    if (n_sent == 0) return TST_OK;
    hipLaunchKernelGGL(beam_merge_kernel, dim3(blocks_for(n_sent, 4)), dim3(256), 0, (hipStream_t)stream, *cur, top_val,
                       top_idx, lse, n_sent, (int)k, (int)k_in, eos, sel_parent, sel_tok, sel_score, sel_fin, sel_nemit);
    return tst_after_launch("tst_beam_merge");
}

namespace {

int retire_compact_impl(const tst_rows* cur, const tst_rows* nxt, const int32_t* sel_parent, const int32_t* sel_tok,
                        const float* sel_score, const int32_t* sel_fin, const int32_t* sel_nemit, const float* lp, int64_t n_sent,
                        int64_t k, int64_t step, int64_t T, int32_t eos, int last_step, int top_ends, int32_t* out_tok,
                        int32_t* out_len, float* out_score, int32_t* scratch_alive_pos, int32_t* row_parent, int32_t* n_alive,
                        void* stream, const char* name, const int32_t* n_true = nullptr, int32_t* n_alive_host = nullptr) {
    TST_REQUIRE(rows_ok(cur, false) && rows_ok(nxt, cur && cur->anc) && sel_parent && sel_tok && sel_score && sel_fin &&
                    sel_nemit && lp && out_tok && out_len && out_score && scratch_alive_pos && row_parent && n_alive,
                TST_ERR_ARG, "%s: null pointer", name);
    TST_REQUIRE(n_sent >= 0 && k >= 1 && k <= 8 && step >= 0 && step < T && n_sent * k < (1ll << 31), TST_ERR_ARG,
                "%s: bad sizes (need 0 <= step < T, 1 <= k <= 8)", name);
    hipStream_t st = (hipStream_t)stream;
    if (n_sent * k <= SMALL_TAIL_ROWS) {   // one workgroup does all three steps
        hipLaunchKernelGGL(tail_small_kernel<false>, dim3(1), dim3(1024), 0, st, *cur, *nxt, (const int32_t*)nullptr,
                           (const unsigned long long*)nullptr, const_cast<int32_t*>(sel_parent),
                           const_cast<int32_t*>(sel_tok), const_cast<float*>(sel_score), const_cast<int32_t*>(sel_fin),
                           const_cast<int32_t*>(sel_nemit), lp, (int)n_sent, (int)k, (int)step, (int)T, eos, last_step, top_ends,
                           out_tok, out_len, out_score, row_parent, n_alive, n_true, n_alive_host);
        return tst_after_launch(name);
    }
    int32_t* alive = scratch_alive_pos;
    int32_t* pos = scratch_alive_pos + n_sent;
    if (n_sent > 0) {
        hipLaunchKernelGGL(finish_kernel, dim3(blocks_for(n_sent, 256)), dim3(256), 0, st, *cur, sel_parent, sel_tok,
                           sel_score, sel_fin, sel_nemit, lp, n_sent, (int)k, (int)T, last_step, top_ends, out_tok, out_len,
                           out_score, alive, n_true);
    }
    hipLaunchKernelGGL(scan_alive_kernel, dim3(1), dim3(1024), 0, st, alive, pos, n_sent, n_alive, n_alive_host);
    if (n_sent > 0) {
        hipLaunchKernelGGL(gather_state_kernel, dim3(blocks_for(n_sent * k, 4)), dim3(256), 0, st, *cur, *nxt, sel_parent,
                           sel_tok, sel_score, sel_fin, sel_nemit, alive, pos, n_sent * k, (int)k, (int)step, (int)T,
                           row_parent);
    }
    return tst_after_launch(name);
}

}  // namespace

extern "C" int tst_retire_compact(const tst_rows* cur, const tst_rows* nxt, const int32_t* sel_parent,
                                  const int32_t* sel_tok, const float* sel_score, const int32_t* sel_fin,
                                  const int32_t* sel_nemit, const float* lp, int64_t n_sent, int64_t k, int64_t step,
                                  int64_t T, int32_t eos, int last_step, int top_ends, int32_t* out_tok,
                                  int32_t* out_len, float* out_score, int32_t* scratch_alive_pos, int32_t* row_parent,
                                  int32_t* n_alive, void* stream) {
    TST_REQUIRE(tst_dims_ok({n_sent, k, step, T}), TST_ERR_UNSUPPORTED, "tst_retire_compact: a size exceeds 2^31 - 1");
    return retire_compact_impl(cur, nxt, sel_parent, sel_tok, sel_score, sel_fin, sel_nemit, lp, n_sent, k, step, T, eos, last_step,
                               top_ends, out_tok, out_len, out_score, scratch_alive_pos, row_parent, n_alive, stream,
                               "tst_retire_compact");
}

// ---- the entry point that allocates nothing ------------------------------------------------------------------------------
// The alive flags and the prefix sums of tst_retire_compact (2 * n_sent int32) live in this array of the code object instead
// of a caller's buffer.  Every GPU holds its own copy at its own address; `scratch` in host code is only a handle for
// hipGetSymbolAddress, never an address a kernel may be given.
static __device__ __align__(256) int32_t scratch[2 * TST_NOALLOC_MAX_SENT];

namespace {

constexpr int MAX_DEVICES = 64;
std::atomic<int32_t*> g_scratch_ptr[MAX_DEVICES];   // device address of `scratch` per device, resolved at the first call there

int scratch_on_current_device(int32_t** out, const char* name) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    TST_REQUIRE(e == hipSuccess && dev >= 0 && dev < MAX_DEVICES, TST_ERR_LAUNCH, "%s: hipGetDevice: %s", name, hipGetErrorString(e));
    int32_t* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (p == nullptr) {
        void* sym = nullptr;
        e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));   // the current device's copy
        TST_REQUIRE(e == hipSuccess && sym != nullptr, TST_ERR_LAUNCH, "%s: hipGetSymbolAddress(scratch) on device %d: %s", name, dev,
                    hipGetErrorString(e));
        p = (int32_t*)sym;
        g_scratch_ptr[dev].store(p, std::memory_order_release);   // a racing thread resolves the same address
    }
    *out = p;
    return TST_OK;
}

}  // namespace

extern "C" int tst_retire_compact_noalloc(const tst_rows* cur, const tst_rows* nxt, const int32_t* sel_parent,
                                          const int32_t* sel_tok, const float* sel_score, const int32_t* sel_fin,
                                          const int32_t* sel_nemit, const float* lp, int64_t n_sent, int64_t k, int64_t step,
                                          int64_t T, int32_t eos, int last_step, int top_ends, int32_t* out_tok,
                                          int32_t* out_len, float* out_score, int32_t* row_parent, int32_t* n_alive,
                                          void* stream) {
    TST_REQUIRE(tst_dims_ok({n_sent, k, step, T}), TST_ERR_UNSUPPORTED, "tst_retire_compact_noalloc: a size exceeds 2^31 - 1");
    TST_REQUIRE(n_sent <= TST_NOALLOC_MAX_SENT, TST_ERR_UNSUPPORTED,
                "tst_retire_compact_noalloc: %lld sentences, the built-in scratch holds %d (use tst_retire_compact with a buffer)",
                (long long)n_sent, TST_NOALLOC_MAX_SENT);
    int32_t* ptr = nullptr;
    const int rc = scratch_on_current_device(&ptr, "tst_retire_compact_noalloc");
    if (rc != TST_OK) return rc;
    // nothing to clear: finish_kernel writes alive[0..n_sent) and scan_alive_kernel pos[0..n_sent) before anything reads them
    return retire_compact_impl(cur, nxt, sel_parent, sel_tok, sel_score, sel_fin, sel_nemit, lp, n_sent, k, step, T, eos, last_step,
                               top_ends, out_tok, out_len, out_score, ptr, row_parent, n_alive, stream, "tst_retire_compact_noalloc");
}

namespace {

// argmax_tok: the selected tokens, or NULL with keys: the arg-max keys of tst_ln_vocab_argmax_f32
int greedy_retire_compact_impl(const int32_t* argmax_tok, const tst_rows* cur, const tst_rows* nxt, int32_t* sel_parent, int32_t* sel_tok,
                               float* sel_score, int32_t* sel_fin, int32_t* sel_nemit, const float* lp, int64_t n_sent, int64_t step,
                               int64_t T, int32_t eos, int last_step, int32_t* out_tok, int32_t* out_len, float* out_score,
                               int32_t* scratch_alive_pos, int32_t* row_parent, int32_t* n_alive, void* stream, const int32_t* n_true,
                               const uint64_t* keys = nullptr, int32_t* n_alive_host = nullptr) {
    const unsigned long long* kp = (const unsigned long long*)keys;
    TST_REQUIRE(tst_aligned16(keys), TST_ERR_ARG, "tst_greedy_retire_compact: arg-max keys must be 16-byte aligned");
    if (n_sent > SMALL_TAIL_ROWS || n_sent <= 0) {
        TST_REQUIRE((argmax_tok || kp) && sel_parent && sel_tok && sel_score && sel_fin && sel_nemit && rows_ok(cur, false), TST_ERR_ARG,
                    "tst_greedy_retire_compact: null pointer");
        if (n_sent > 0) {
            hipLaunchKernelGGL(greedy_select_kernel, dim3(blocks_for(n_sent, 256)), dim3(256), 0, (hipStream_t)stream, argmax_tok, kp, n_sent,
                               eos, sel_parent, sel_tok, sel_score, sel_fin, sel_nemit, *cur);
            const int rc = tst_after_launch("tst_greedy_retire_compact");
            if (rc != TST_OK) return rc;
        }
        return retire_compact_impl(cur, nxt, sel_parent, sel_tok, sel_score, sel_fin, sel_nemit, lp, n_sent, 1, step, T, eos, last_step, 0,
                                   out_tok, out_len, out_score, scratch_alive_pos, row_parent, n_alive, stream, "tst_greedy_retire_compact",
                                   n_true, n_alive_host);
    }
    TST_REQUIRE((argmax_tok || kp) && rows_ok(cur, false) && rows_ok(nxt, cur && cur->anc) && sel_parent && sel_tok && sel_score && sel_fin &&
                    sel_nemit && lp && out_tok && out_len && out_score && row_parent && n_alive,
                TST_ERR_ARG, "tst_greedy_retire_compact: null pointer");
    TST_REQUIRE(step >= 0 && step < T, TST_ERR_ARG, "tst_greedy_retire_compact: bad sizes (need 0 <= step < T)");
    hipLaunchKernelGGL(tail_small_kernel<true>, dim3(1), dim3(1024), 0, (hipStream_t)stream, *cur, *nxt, argmax_tok, kp, sel_parent, sel_tok,
                       sel_score, sel_fin, sel_nemit, lp, (int)n_sent, 1, (int)step, (int)T, eos, last_step, 0, out_tok, out_len,
                       out_score, row_parent, n_alive, n_true, n_alive_host);
    return tst_after_launch("tst_greedy_retire_compact");
}

}  // namespace

extern "C" int tst_greedy_retire_compact(const int32_t* argmax_tok, const tst_rows* cur, const tst_rows* nxt, int32_t* sel_parent,
                                         int32_t* sel_tok, float* sel_score, int32_t* sel_fin, int32_t* sel_nemit, const float* lp,
                                         int64_t n_sent, int64_t step, int64_t T, int32_t eos, int last_step, int32_t* out_tok,
                                         int32_t* out_len, float* out_score, int32_t* scratch_alive_pos, int32_t* row_parent,
                                         int32_t* n_alive, void* stream) {
    TST_REQUIRE(tst_dims_ok({n_sent, step, T}), TST_ERR_UNSUPPORTED, "tst_greedy_retire_compact: a size exceeds 2^31 - 1");
    return greedy_retire_compact_impl(argmax_tok, cur, nxt, sel_parent, sel_tok, sel_score, sel_fin, sel_nemit, lp, n_sent, step, T, eos,
                                      last_step, out_tok, out_len, out_score, scratch_alive_pos, row_parent, n_alive, stream, nullptr);
}

int tst_greedy_retire_compact_bounded(const int32_t* argmax_tok, const tst_rows* cur, const tst_rows* nxt, int32_t* sel_parent,
                                      int32_t* sel_tok, float* sel_score, int32_t* sel_fin, int32_t* sel_nemit, const float* lp,
                                      int64_t n_sent, int64_t step, int64_t T, int32_t eos, int last_step, int32_t* out_tok,
                                      int32_t* out_len, float* out_score, int32_t* scratch_alive_pos, int32_t* row_parent,
                                      int32_t* n_alive, const int32_t* n_true, const uint64_t* keys, int32_t* n_alive_host, void* stream) {
    return greedy_retire_compact_impl(argmax_tok, cur, nxt, sel_parent, sel_tok, sel_score, sel_fin, sel_nemit, lp, n_sent, step, T, eos,
                                      last_step, out_tok, out_len, out_score, scratch_alive_pos, row_parent, n_alive, stream, n_true, keys,
                                      n_alive_host);
}

// The two retirement entry points with the true sentence count read from device memory (n_sent is then an upper bound):
// used by tst_select_step_run when tst_select_step.n_sent_dev is set.
int tst_retire_compact_bounded(const tst_rows* cur, const tst_rows* nxt, const int32_t* sel_parent, const int32_t* sel_tok,
                               const float* sel_score, const int32_t* sel_fin, const int32_t* sel_nemit, const float* lp, int64_t n_sent,
                               int64_t k, int64_t step, int64_t T, int32_t eos, int last_step, int top_ends, int32_t* out_tok,
                               int32_t* out_len, float* out_score, int32_t* scratch_alive_pos, int32_t* row_parent, int32_t* n_alive,
                               const int32_t* n_true, int32_t* n_alive_host, void* stream) {
    return retire_compact_impl(cur, nxt, sel_parent, sel_tok, sel_score, sel_fin, sel_nemit, lp, n_sent, k, step, T, eos, last_step,
                               top_ends, out_tok, out_len, out_score, scratch_alive_pos, row_parent, n_alive, stream,
                               "tst_select_step_run", n_true, n_alive_host);
}
