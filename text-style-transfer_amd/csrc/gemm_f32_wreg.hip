// fp32 products with K = 256 (d_model of the transformer-mt decoder) at a thousand rows and more: C = epilogue(bias + A * Wt).
// This is the vocabulary projection of those models' decode steps (tst_tfm_step_run); tst_gemm_mfma_f32 uses it for the layer
// products only under TST_GEMM_WREG=1 (measured equal end to end there).
//
// The tile kernels of gemm_f32.hip and gemm_f32_dma.hip read BOTH MFMA operands from LDS and stage the A panel once per 64-256
// columns of C.  Here the WEIGHTS stay in registers: a wave owns 32 columns of C for the whole kernel and keeps their 256 x 32
// slab of Wt as the 128 B-operands of its MFMA chain (128 VGPRs); a workgroup of 8 waves owns 256 columns and walks over 32-row
// tiles of A (with V = 32000: 125 column groups x 2 workgroups, ~245 tiles each, so the slab is loaded once per ~245 tiles).  Per
// tile the workgroup stages 32 whole rows of A (32 KB) once into LDS, and a wave reads its A operands with one ds_read_b128 per
// FOUR MFMAs: the row image is stored with the k of every group of eight permuted to (0, 2, 4, 6 | 1, 3, 5, 7), so that the sixteen
// bytes a lane reads are the operands of four consecutive k-steps (lane half h of step j takes k = 2 j + h).  Two tile buffers,
// one LDS-only barrier per tile; the loads of the next tile are in flight during the 128 MFMAs of the current one.
//
// Same arithmetic as every other fp32 product here: acc = bias[n]; for k ascending: acc = fma(A[m][k], Wt[k][n], acc),
// then relu, then + residual.  C may alias the residual (each element is read and then written by the same lane).
#include "tst_common.h"
#include <math.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int WR_K = 256;          // reduction length
constexpr int WR_BM = 32;          // rows of A per tile
constexpr int WR_BN = 256;         // columns of C per workgroup: 8 waves x 32
constexpr int WR_LDA = WR_K + 4;   // floats per LDS row: consecutive rows start 4 banks apart, so the 16-byte reads of 8 rows
                                   // (one 128-byte LDS access) fall into different banks

// RB = 32-row blocks per tile: with 2 a wave runs two independent MFMA chains over the same weights (half the barriers and
// tile edges per MFMA; 64-row tiles, 133 KB of LDS); the residual form keeps RB = 1 (its 16 extra registers per block).
// LN: the rows of A are normalised on their way into LDS (LayerNorm over the 256 elements of a row with gamma / beta): a wave
// stages whole rows with lane l holding the floats 4 l .. 4 l + 3, which is the layout of layernorm_reg_kernel, so the same
// operations in the same order give the same bits as the separate LayerNorm launch.
template <bool HAS_RES, bool RELU, bool NT, int RB, bool LN>
__global__ __launch_bounds__(512, 1) void gemm_f32_wreg_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                               int64_t ldw, const float* __restrict__ bias, const float* res,
                                                               int64_t ldr, float* C, int64_t ldc, int M, int N, int n_tiles,
                                                               int n_groups, int wg_per_group, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps) {
    static_assert(RB == 1 || (RB == 2 && !HAS_RES && !LN), "two row blocks only without a residual and without LayerNorm");
    constexpr int BM = WR_BM * RB;
    __shared__ __attribute__((aligned(16))) float As[2][BM * WR_LDA];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int group = blockIdx.x % n_groups, slot = blockIdx.x / n_groups;
    const int h = lane >> 5;
    const int col = group * WR_BN + wave * 32 + (lane & 31);
    const bool col_ok = col < N;                    // the last group of a width that is no multiple of 256 has idle columns
    const int lcol = col_ok ? col : N - 1;          // they load what the last column loads and store nothing

    // the wave's slab of the weights: w[j] = Wt[2 j + h][col], the B operand of k-step j (two 128-byte pieces of Wt per request)
    float w[WR_K / 2];
    {
        const float* wp = W + (int64_t)h * ldw + lcol;
#pragma unroll
        for (int j = 0; j < WR_K / 2; ++j) w[j] = wp[(int64_t)(2 * j) * ldw];
    }
    const float b = bias != nullptr ? bias[lcol] : 0.0f;

    // staging: wave v brings rows 4 RB v .. 4 RB v + 4 RB - 1 of a tile, lane l the floats 4 l .. 4 l + 3 of each (whole rows, 1 KB
    // per request)
    float4 st[4 * RB];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 4 * RB; ++i) {
            const int row = min(tile * BM + wave * 4 * RB + i, M - 1);
            st[i] = *reinterpret_cast<const float4*>(A + (int64_t)row * lda + 4 * lane);
        }
    };
    // lanes 2 g and 2 g + 1 hold k = 8 g .. 8 g + 7 of a row; they exchange two values each and write (0, 2, 4, 6) and (1, 3, 5, 7)
    float4 gg = make_float4(0.f, 0.f, 0.f, 0.f), bb = gg;
    if (LN) {
        gg = reinterpret_cast<const float4*>(gamma)[lane];
        bb = reinterpret_cast<const float4*>(beta)[lane];
    }
    auto write_tile = [&](int buf) {
        const bool odd = (lane & 1) != 0;
#pragma unroll
        for (int i = 0; i < 4 * RB; ++i) {
            float4 v = st[i];
            if (LN) {   // layernorm_reg_kernel<float, 1>, operation for operation
                float p = 0.0f;
                p = p + v.x; p = p + v.y; p = p + v.z; p = p + v.w;
                const float mean = wave_sum_butterfly(p) / 256.0f;
                p = 0.0f;
                float t;
                t = v.x - mean; p = __builtin_fmaf(t, t, p);
                t = v.y - mean; p = __builtin_fmaf(t, t, p);
                t = v.z - mean; p = __builtin_fmaf(t, t, p);
                t = v.w - mean; p = __builtin_fmaf(t, t, p);
                const float var = wave_sum_butterfly(p) / 256.0f;
                const float rstd = 1.0f / sqrtf(var + eps);
                v.x = __builtin_fmaf((v.x - mean) * rstd, gg.x, bb.x);
                v.y = __builtin_fmaf((v.y - mean) * rstd, gg.y, bb.y);
                v.z = __builtin_fmaf((v.z - mean) * rstd, gg.z, bb.z);
                v.w = __builtin_fmaf((v.w - mean) * rstd, gg.w, bb.w);
            }
            const float r0 = lane_xor<1>(odd ? v.x : v.y);
            const float r1 = lane_xor<1>(odd ? v.z : v.w);
            const float4 o = odd ? make_float4(r0, r1, v.y, v.w) : make_float4(v.x, v.z, r0, r1);
            *reinterpret_cast<float4*>(&As[buf][(wave * 4 * RB + i) * WR_LDA + 4 * lane]) = o;
        }
    };

    int tile = slot;
    if (tile < n_tiles) {
        load_tile(tile);
        write_tile(0);
    }
    __syncthreads();
    int buf = 0;
    for (; tile < n_tiles; tile += wg_per_group) {
        const int nxt = tile + wg_per_group;
        if (nxt < n_tiles) load_tile(nxt);
        const int row0 = tile * BM + 4 * h;     // accumulator register r of block rb holds row row0 + 32 rb + (r & 3) + 8 (r >> 2), column col
        float rv[16];
        if (HAS_RES) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = min(row0 + (r & 3) + 8 * (r >> 2), M - 1);
                rv[r] = res[(int64_t)row * ldr + lcol];
            }
        }
        f32x16 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = b;
        const float* ap = &As[buf][(lane & 31) * WR_LDA + 4 * h];
        // operands of the k-steps 4 q .. 4 q + 3 are requested two groups ahead; the scheduling barriers keep the requests where
        // they are written (left alone, the compiler moves every read down to just before its first use)
        float4 a0[RB], a1[RB], a2[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            a0[rb] = *reinterpret_cast<const float4*>(ap + rb * 32 * WR_LDA);
            a1[rb] = *reinterpret_cast<const float4*>(ap + rb * 32 * WR_LDA + 8);
            a2[rb] = a1[rb];
        }
#pragma unroll
        for (int q = 0; q < WR_K / 8; ++q) {
            if (q + 2 < WR_K / 8) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a2[rb] = *reinterpret_cast<const float4*>(ap + rb * 32 * WR_LDA + (q + 2) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[rb].x, w[4 * q + 0], acc[rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[rb].y, w[4 * q + 1], acc[rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[rb].z, w[4 * q + 2], acc[rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[rb].w, w[4 * q + 3], acc[rb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                a0[rb] = a1[rb];
                a1[rb] = a2[rb];
            }
        }
        // the next tile goes to LDS BEFORE this tile's results are stored: the wait for its loads (requested a whole tile ago)
        // must not have the stores in front of it, or every tile would end with a wait for their completion
        if (nxt < n_tiles) write_tile(buf ^ 1);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 32 * rb + (r & 3) + 8 * (r >> 2);
                float v = acc[rb][r];
                if (RELU) v = v > 0.0f ? v : 0.0f;
                if (HAS_RES) v = v + rv[r];
                if (row < M && col_ok) {
                    if (NT) __builtin_nontemporal_store(v, &C[(int64_t)row * ldc + col]);
                    else C[(int64_t)row * ldc + col] = v;
                }
            }
        }
        // a barrier over LDS only (__syncthreads() would also wait for the stores above to complete)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");     // no LDS read of the next tile may be moved in front of the barrier
        buf ^= 1;
    }
}

}  // namespace

// Shapes the kernel takes: K = 256, any width (the last group of 256 columns may be partial), up to 256 groups.
bool tst_gemm_wreg_ok(int64_t M, int64_t N, int64_t K) { return K == WR_K && N >= 1 && (N + WR_BN - 1) / WR_BN <= 256 && M >= 1; }

// nontemporal: the results leave the caches as they are written (logits that another kernel reads once, later)
static bool wreg_two_blocks() {     // TST_WREG_ROW_BLOCKS=1 in the environment (measurements): 32-row tiles for every launch
    static const bool v = [] { const char* e = getenv("TST_WREG_ROW_BLOCKS"); return !(e != nullptr && e[0] == '1'); }();
    return v;
}

void tst_launch_gemm_wreg(const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias, const float* res, int64_t ldr,
                          float* C, int64_t ldc, int64_t M, int64_t N, int act, int nontemporal, hipStream_t st) {
    const int n_groups = (int)((N + WR_BN - 1) / WR_BN);
    const bool nt = nontemporal && res == nullptr && act != 1;
    const int rb = (nt && M >= 4096 && wreg_two_blocks()) ? 2 : 1;
    const int n_tiles = (int)((M + WR_BM * rb - 1) / (WR_BM * rb));
    const int wg_per_group = n_tiles < 256 / n_groups ? n_tiles : 256 / n_groups;      // one workgroup per CU
    const dim3 grid((unsigned)(n_groups * wg_per_group));
#define TST_WREG_LAUNCH(HR, RL, NT, RB)                                                                                                 \
    hipLaunchKernelGGL((gemm_f32_wreg_kernel<HR, RL, NT, RB, false>), grid, dim3(512), 0, st, A, lda, Wt, ldw, bias, res, ldr, C, ldc, \
                       (int)M, (int)N, n_tiles, n_groups, wg_per_group, (const float*)nullptr, (const float*)nullptr, 0.0f)
    if (nt) {
        if (rb == 2) TST_WREG_LAUNCH(false, false, true, 2); else TST_WREG_LAUNCH(false, false, true, 1);
    } else if (res != nullptr) {
        if (act == 1) TST_WREG_LAUNCH(true, true, false, 1); else TST_WREG_LAUNCH(true, false, false, 1);
    } else {
        if (act == 1) TST_WREG_LAUNCH(false, true, false, 1); else TST_WREG_LAUNCH(false, false, false, 1);
    }
#undef TST_WREG_LAUNCH
}

// C = epilogue(bias + LayerNorm(x) * Wt): x [M][256] row-major and contiguous, gamma / beta [256]
void tst_launch_gemm_wreg_ln(const float* x, const float* gamma, const float* beta, float eps, const float* Wt, int64_t ldw,
                             const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, int act, hipStream_t st) {
    const int n_groups = (int)((N + WR_BN - 1) / WR_BN);
    const int n_tiles = (int)((M + WR_BM - 1) / WR_BM);
    const int wg_per_group = n_tiles < 256 / n_groups ? n_tiles : 256 / n_groups;
    const dim3 grid((unsigned)(n_groups * wg_per_group));
    if (act == 1)
        hipLaunchKernelGGL((gemm_f32_wreg_kernel<false, true, false, 1, true>), grid, dim3(512), 0, st, x, (int64_t)WR_K, Wt, ldw, bias,
                           (const float*)nullptr, (int64_t)0, C, ldc, (int)M, (int)N, n_tiles, n_groups, wg_per_group, gamma, beta, eps);
    else
        hipLaunchKernelGGL((gemm_f32_wreg_kernel<false, false, false, 1, true>), grid, dim3(512), 0, st, x, (int64_t)WR_K, Wt, ldw, bias,
                           (const float*)nullptr, (int64_t)0, C, ldc, (int)M, (int)N, n_tiles, n_groups, wg_per_group, gamma, beta, eps);
}
