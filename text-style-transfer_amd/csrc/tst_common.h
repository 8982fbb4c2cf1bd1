// Shared definitions of the gfx950 kernels: error reporting, wave helpers and the canonical fp32
// functions (fixed polynomials; only +, *, fma, /, rint and bit operations, so that a CPU
// implementation of the same formulas gives the same bits).  Compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <initializer_list>
#include "../../include/tst_hip.h"

void tst_set_error(const char* fmt, ...);

// select.hip: the retirement entry points with the true sentence count read from device memory (n_sent an upper bound),
// an optional second (host-visible) target of the survivor count, and arg-max keys in place of the token list
int tst_retire_compact_bounded(const tst_rows* cur, const tst_rows* nxt, const int32_t* sel_parent, const int32_t* sel_tok,
                               const float* sel_score, const int32_t* sel_fin, const int32_t* sel_nemit, const float* lp, int64_t n_sent,
                               int64_t k, int64_t step, int64_t T, int32_t eos, int last_step, int top_ends, int32_t* out_tok,
                               int32_t* out_len, float* out_score, int32_t* scratch_alive_pos, int32_t* row_parent, int32_t* n_alive,
                               const int32_t* n_true, int32_t* n_alive_host, void* stream);
int tst_greedy_retire_compact_bounded(const int32_t* argmax_tok, const tst_rows* cur, const tst_rows* nxt, int32_t* sel_parent,
                                      int32_t* sel_tok, float* sel_score, int32_t* sel_fin, int32_t* sel_nemit, const float* lp,
                                      int64_t n_sent, int64_t step, int64_t T, int32_t eos, int last_step, int32_t* out_tok,
                                      int32_t* out_len, float* out_score, int32_t* scratch_alive_pos, int32_t* row_parent,
                                      int32_t* n_alive, const int32_t* n_true, const uint64_t* keys, int32_t* n_alive_host, void* stream);
// rowops.hip: tst_embed_posadd_f32 / tst_embed_posadd_rows_f32 (z may be NULL) that also clears clear_keys[0..R) (arg-max keys)
int tst_embed_posadd_clear_f32(const int32_t* tok, const float* emb, const float* pe, float scale, const float* z, const int32_t* zidx,
                               float* x, uint64_t* clear_keys, int64_t R, int64_t d, int64_t V, void* stream);

// gemm_f32_wreg.hip: the K = 256 product with the weights held in registers (used by tst_gemm_mfma_f32)
bool tst_gemm_wreg_ok(int64_t M, int64_t N, int64_t K);
void tst_launch_gemm_wreg(const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias, const float* res, int64_t ldr,
                          float* C, int64_t ldc, int64_t M, int64_t N, int act, int nontemporal, hipStream_t st);

void tst_launch_gemm_wreg_ln(const float* x, const float* gamma, const float* beta, float eps, const float* Wt, int64_t ldw,
                             const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, int act, hipStream_t st);

// attn_mfma.hip: the matrix-core cross-attention of beam groups (used by tst_attn_cross_group_f32)
int tst_cross_mfma_ok(int64_t S, int64_t dh, int64_t group);
int tst_launch_cross_mfma(const float* kv, const int32_t* kvrow, const float* q, const uint8_t* mask, const int32_t* sent, int64_t S,
                          float* out, int64_t R, int64_t d, int64_t heads, int64_t group, hipStream_t stream);

#define TST_REQUIRE(cond, code, ...)                 \
    do {                                             \
        if (!(cond)) {                               \
            tst_set_error(__VA_ARGS__);              \
            return (code);                           \
        }                                            \
    } while (0)

static inline int tst_after_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tst_set_error("%s: %s", what, hipGetErrorString(e));
        return TST_ERR_LAUNCH;
    }
    return TST_OK;
}

// Row counts, widths, vocabulary sizes and other counts are at most 2^31 - 1: the kernels count rows and columns in 32-bit
// integers and build their grids from them.  Offsets in elements or bytes are formed in 64 bits, so leading dimensions and
// whole buffers (rows x width) may be larger.  Every entry point checks this first.
static inline bool tst_dims_ok(std::initializer_list<int64_t> sizes) {
    for (int64_t v : sizes)
        if (v > (int64_t)0x7fffffff) return false;
    return true;
}

static inline bool tst_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define TST_WAVE 64

// Value of lane (lane ^ OFF), OFF a power of two below 64, without the LDS crossbar of ds_bpermute (__shfl_xor):
// DPP controls inside a row of 16 lanes, v_permlane16_swap / v_permlane32_swap (gfx950) across rows and halves.
// Checked lane by lane against __shfl_xor on the device (tools/probe/probe_lane_xor.hip).
template <int OFF>
__device__ __forceinline__ int lane_xor_i(int x) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "lane_xor: 1, 2, 4, 8, 16 or 32");
    if (OFF == 1) return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    if (OFF == 2) return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    if (OFF == 4) {
        int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false);           // row_shl:4 into banks 0, 2: lane i <- i + 4
        return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);            // row_shr:4 into banks 1, 3: lane i <- i - 4
    }
    if (OFF == 8) return __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false);  // row_ror:8
    if (OFF == 16) {   // rows 0 <-> 1 and 2 <-> 3
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        return (int)(((threadIdx.x >> 4) & 1) ? r[0] : r[1]);
    }
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);   // lanes 0..31 <-> 32..63
    return (int)(((threadIdx.x >> 5) & 1) ? r[0] : r[1]);
}
template <int OFF>
__device__ __forceinline__ float lane_xor(float v) { return __int_as_float(lane_xor_i<OFF>(__float_as_int(v))); }

// sum over the 64 lanes of a wave: for off = 32,16,..,1: v = v + v(lane ^ off); every lane gets the total
__device__ __forceinline__ float wave_sum_butterfly(float v) {
    v = v + lane_xor<32>(v);
    v = v + lane_xor<16>(v);
    v = v + lane_xor<8>(v);
    v = v + lane_xor<4>(v);
    v = v + lane_xor<2>(v);
    v = v + lane_xor<1>(v);
    return v;
}

// the same within aligned groups of 16 lanes (off = 8,4,2,1)
__device__ __forceinline__ float group16_sum_butterfly(float v) {
    v = v + lane_xor<8>(v);
    v = v + lane_xor<4>(v);
    v = v + lane_xor<2>(v);
    v = v + lane_xor<1>(v);
    return v;
}

__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, lane_xor<8>(v));
    v = fmaxf(v, lane_xor<4>(v));
    v = fmaxf(v, lane_xor<2>(v));
    v = fmaxf(v, lane_xor<1>(v));
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, lane_xor<32>(v));
    v = fmaxf(v, lane_xor<16>(v));
    return group16_max(v);
}

// exp(x): 0 for x < -87 (and for NaN), x clamped to 88 above.  Branch-free: the argument is clamped into
// the polynomial's range and the out-of-range case is a select; 2^n is applied with ldexp, which gives the
// same bits as multiplying by the float 2^n for every n in [-126, 127] (checked on the device over dense
// ranges, results in the subnormal range included).
template <bool UPPER_CLAMP>
__device__ __forceinline__ float d_exp_impl(float x) {
    const bool in_range = x >= -87.0f;
    float xc = fmaxf(x, -87.0f);
    if (UPPER_CLAMP) xc = fminf(xc, 88.0f);
    const float n = rintf(xc * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693145751953125f, xc);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = __builtin_fmaf(p, r2, r) + 1.0f;
    const float v = ldexpf(y, (int)n);
    return in_range ? v : 0.0f;
}

__device__ __forceinline__ float d_exp(float x) { return d_exp_impl<true>(x); }
// for arguments known to be <= 0 (x - max): no upper clamp
__device__ __forceinline__ float d_exp_le0(float x) { return d_exp_impl<false>(x); }

__device__ __forceinline__ float d_log(float x) {
    uint32_t u = __float_as_uint(x);
    int e = (int)((u >> 23) & 0xffu) - 127;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = 0.111111111111f;
    p = __builtin_fmaf(p, z, 0.142857142857f);
    p = __builtin_fmaf(p, z, 0.2f);
    p = __builtin_fmaf(p, z, 0.333333333333f);
    p = p * z;
    float s2 = s + s;
    float r = __builtin_fmaf(s2, p, s2);
    float fe = (float)e;
    r = __builtin_fmaf(fe, 1.42860682030941723212e-6f, r);
    return __builtin_fmaf(fe, 0.693145751953125f, r);
}

__device__ __forceinline__ float d_sigmoid(float x) {
    if (x >= 0.0f) { float e = d_exp(-x); return 1.0f / (1.0f + e); }
    float e = d_exp(x);
    return e / (1.0f + e);
}

__device__ __forceinline__ float d_tanh(float x) {
    float a = fabsf(x);
    float t = d_exp(-2.0f * a);
    float r = (1.0f - t) / (1.0f + t);
    return copysignf(r, x);
}

// GRU gate update of one element (PyTorch gate order): r = sigmoid(i_r + h_r), z = sigmoid(i_z + h_z),
// n = tanh(i_n + r * h_n), h' = (1 - z) * n + z * h; the operation order is part of the canonical arithmetic.
__device__ __forceinline__ float gru_elem(float ir, float iz, float in, float hr, float hz, float hn, float h) {
    const float rg = d_sigmoid(ir + hr);
    const float zg = d_sigmoid(iz + hz);
    const float ng = d_tanh(__builtin_fmaf(rg, hn, in));
    return __builtin_fmaf(zg, h, (1.0f - zg) * ng);
}


