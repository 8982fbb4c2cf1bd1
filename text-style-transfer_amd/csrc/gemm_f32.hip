// fp32 GEMM on the gfx950 matrix cores: C = epilogue(bias + A * Wt).
//
// v_mfma_f32_32x32x2_f32 accumulates exactly like a k-ordered fmaf chain (one rounding per term),
// so every output element is  acc = bias[n]; for k ascending: acc = fma(A[m][k], Wt[k][n], acc)
// whatever the tiling.  Tile: 128 x 128 x 16 per 256-thread workgroup; 4 waves as 2 (M) x 2 (N), each
// wave owns 64 x 64 = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator registers); small problems use
// 64 x 64 x 16 tiles (one MFMA tile per wave) so that every CU gets work.  A and Wt tiles are
// staged through registers into k-major LDS images ([k][m] and [k][n]) so that the one-float-per-lane
// MFMA operands are conflict-free ds_read_b32; global loads of tile t+1 are in flight while tile t is
// multiplied (three LDS buffers, one barrier per tile, taken in the middle of the tile's MFMAs).  Workgroups are renumbered so that each XCD
// walks a contiguous, m-grouped range of tiles and re-reads A / Wt panels from its own L2.
#include "tst_common.h"
#include <limits.h>
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16, GROUP_M = 16;

// WTM x WTN = MFMA tiles (32 x 32) per wave: the workgroup tile is (64 * WTM) x (64 * WTN).
// 2 x 4 (128 x 256: fewest LDS reads and staged bytes per MFMA) for wide outputs, 2 x 2 (128 x 128) for
// large problems, 1 x 1 (64 x 64) when the 128-tiling would leave CUs without work.
template <int WTM, int WTN, bool FULL_K, bool HAS_RES, bool RELU>
__global__ __launch_bounds__(256, (WTN == 4 ? 2 : 1)) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ W, int64_t ldw,
                                                       const float* __restrict__ bias, const float* res, int64_t ldr,
                                                       float* C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr int BM = 64 * WTM, BN = 64 * WTN, LDA = BM + 4, LDB = BN + 4;
    constexpr int NLA = WTM, NLB = WTN;  // float4 loads per thread: BM * BK / 4 / 256 and BN * BK / 4 / 256
    __shared__ __attribute__((aligned(16))) float As[3][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[3][BK][LDB];

    // workgroup id -> logical tile index: consecutive ids go to different XCDs, so give every XCD a
    // contiguous range (bijective for any grid size), then walk tiles in groups of GROUP_M rows.
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
    const int lb = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int per_group = GROUP_M * tiles_n;
    const int first_m = (lb / per_group) * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (lb % per_group) % gsz;
    const int tn = (lb % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // Staging registers for the next tile.  A: float4 along k of row ar[j]; Wt: float4 along n of k-row bk[j].
    // Loads are unconditional from clamped (always valid) addresses and out-of-range data is zeroed
    // afterwards: a conditional load would be compiled to a select between addresses and lose the
    // 16-byte global_load.
    // named registers (arrays here end up in scratch memory); ra1 unused when NLA == 1, rb1..3 when NLB < 2, 4
    float4 ra0, ra1, rb0, rb1, rb2, rb3;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int akq = (t & 3) * 4;
    const int bnq = (t % (BN / 4)) * 4;
    const int ar0 = t >> 2, ar1 = (t + 256) >> 2;
    constexpr int BKS = 1024 / BN;  // k rows of the Wt tile covered by one round of 256 float4 loads
    const int bk0 = t / (BN / 4), bk1 = bk0 + BKS, bk2 = bk0 + 2 * BKS, bk3 = bk0 + 3 * BKS;
    const float* a_row0 = A + (int64_t)min(m0 + ar0, M - 1) * lda;
    const float* a_row1 = A + (int64_t)min(m0 + ar1, M - 1) * lda;
    const int bcol_g = min(n0 + bnq, (int)ldw - 4);
    ra1 = zero4; rb1 = zero4; rb2 = zero4; rb3 = zero4;

#define TST_LOAD_TILE(k0)                                                                                     \
    do {                                                                                                      \
        const int gk_ = (k0) + akq;                                                                           \
        const int gkc_ = min(gk_, K - 4);                                                                     \
        ra0 = *reinterpret_cast<const float4*>(a_row0 + gkc_);                                                \
        rb0 = *reinterpret_cast<const float4*>(W + (int64_t)min((k0) + bk0, K - 1) * ldw + bcol_g);           \
        if (NLA == 2) ra1 = *reinterpret_cast<const float4*>(a_row1 + gkc_);                                  \
        if (NLB >= 2) rb1 = *reinterpret_cast<const float4*>(W + (int64_t)min((k0) + bk1, K - 1) * ldw + bcol_g); \
        if (NLB == 4) {                                                                                       \
            rb2 = *reinterpret_cast<const float4*>(W + (int64_t)min((k0) + bk2, K - 1) * ldw + bcol_g);       \
            rb3 = *reinterpret_cast<const float4*>(W + (int64_t)min((k0) + bk3, K - 1) * ldw + bcol_g);       \
        }                                                                                                     \
        if (!FULL_K) {                                                                                        \
            if (gk_ >= K) { ra0 = zero4; ra1 = zero4; }                                                       \
            if ((k0) + bk0 >= K) rb0 = zero4;                                                                 \
            if ((k0) + bk1 >= K) rb1 = zero4;                                                                 \
            if ((k0) + bk2 >= K) rb2 = zero4;                                                                 \
            if ((k0) + bk3 >= K) rb3 = zero4;                                                                 \
        }                                                                                                     \
    } while (0)
#define TST_STORE_TILE(buf)                                                                                   \
    do {                                                                                                      \
        As[buf][akq + 0][ar0] = ra0.x; As[buf][akq + 1][ar0] = ra0.y;                                         \
        As[buf][akq + 2][ar0] = ra0.z; As[buf][akq + 3][ar0] = ra0.w;                                         \
        *reinterpret_cast<float4*>(&Bs[buf][bk0][bnq]) = rb0;                                                 \
        if (NLA == 2) {                                                                                       \
            As[buf][akq + 0][ar1] = ra1.x; As[buf][akq + 1][ar1] = ra1.y;                                     \
            As[buf][akq + 2][ar1] = ra1.z; As[buf][akq + 3][ar1] = ra1.w;                                     \
        }                                                                                                     \
        if (NLB >= 2) *reinterpret_cast<float4*>(&Bs[buf][bk1][bnq]) = rb1;                                   \
        if (NLB == 4) {                                                                                       \
            *reinterpret_cast<float4*>(&Bs[buf][bk2][bnq]) = rb2;                                             \
            *reinterpret_cast<float4*>(&Bs[buf][bk3][bnq]) = rb3;                                             \
        }                                                                                                     \
    } while (0)

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        const int col = n0 + wn * 32 * WTN + j * 32 + (lane & 31);
        const float b = (bias != nullptr && col < N) ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = b;
    }

    // Three LDS buffers; per k-tile: first half of the MFMAs, then the staged registers of tile t+1 go to
    // LDS, the loads of tile t+2 are issued and the workgroup barrier is taken IN THE MIDDLE of the tile,
    // then the second half.  A wave so reaches the barrier with MFMAs in flight and has half a tile of
    // independent MFMAs right after it: the matrix pipe does not drain around the barrier.  The buffer
    // written in tile t was last read in tile t-2, and every wave has passed the barrier of tile t-1 since.
    const int nkt = (K + BK - 1) / BK;
    TST_LOAD_TILE(0);
    TST_STORE_TILE(0);
    if (nkt > 1) TST_LOAD_TILE(BK);
    __syncthreads();
    const int arow = wm * 32 * WTM + (lane & 31), bcol = wn * 32 * WTN + (lane & 31), khalf = lane >> 5;
    // The MFMA operands (one float of A and of Wt per lane and 32 x 32 tile) are read from LDS one k-step
    // ahead of the MFMAs that use them, also across the mid-tile barrier and into the next tile's buffer
    // (complete since that barrier), so the LDS latency is never exposed.
    float a_cur[WTM], b_cur[WTN], a_nxt[WTM], b_nxt[WTN];
#define TST_READ_OPERANDS(av, bv, bufi, kk)                                                                   \
    _Pragma("unroll") for (int i = 0; i < WTM; ++i) av[i] = As[bufi][(kk) + khalf][arow + 32 * i];            \
    _Pragma("unroll") for (int j = 0; j < WTN; ++j) bv[j] = Bs[bufi][(kk) + khalf][bcol + 32 * j];
    int buf = 0;
    TST_READ_OPERANDS(a_cur, b_cur, buf, 0)
    for (int kt = 0; kt < nkt; ++kt) {
        const int nbuf = buf == 2 ? 0 : buf + 1;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            if (kk + 2 < BK) {
                TST_READ_OPERANDS(a_nxt, b_nxt, buf, kk + 2)
            } else if (kt + 1 < nkt) {
                TST_READ_OPERANDS(a_nxt, b_nxt, nbuf, 0)
            }
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < WTM; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
            for (int j = 0; j < WTN; ++j) b_cur[j] = b_nxt[j];
            if (kk == BK / 2 - 2) {
                if (kt + 1 < nkt) {
                    TST_STORE_TILE(nbuf);
                    if (kt + 2 < nkt) TST_LOAD_TILE((kt + 2) * BK);
                }
                __syncthreads();
            }
        }
        buf = nbuf;
    }
#undef TST_READ_OPERANDS

    // epilogue: lane owns column col of 16 rows per 32 x 32 tile (rows (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
    // tiles that lie fully inside the matrix take the unguarded path
    const bool full_tile = m0 + BM <= M && n0 + BN <= N;
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            const int col = n0 + wn * 32 * WTN + j * 32 + (lane & 31);
            const int row0 = m0 + wm * 32 * WTM + i * 32 + 4 * (lane >> 5);
            float rv[16];
            if (full_tile) {
                float* cp = C + (int64_t)row0 * ldc + col;
                if (HAS_RES) {
                    const float* rp = res + (int64_t)row0 * ldr + col;
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = rp[(int64_t)((r & 3) + 8 * (r >> 2)) * ldr];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r];
                    if (RELU) v = v > 0.0f ? v : 0.0f;
                    if (HAS_RES) v = v + rv[r];
                    cp[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = v;
                }
            } else {
                const bool col_ok = col < N;
                if (HAS_RES) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + (r & 3) + 8 * (r >> 2);
                        rv[r] = (col_ok && row < M) ? res[(int64_t)row * ldr + col] : 0.0f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][r];
                    if (RELU) v = v > 0.0f ? v : 0.0f;
                    if (HAS_RES) v = v + rv[r];
                    if (col_ok && row < M) C[(int64_t)row * ldc + col] = v;
                }
            }
        }
}

#undef TST_LOAD_TILE
#undef TST_STORE_TILE

template <int WTM, int WTN>
void launch_gemm(const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias, const float* res, int64_t ldr,
                 float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t st) {
    constexpr int BTM = 64 * WTM, BTN = 64 * WTN;
    const int64_t tiles_m = (M + BTM - 1) / BTM, tiles_n = (N + BTN - 1) / BTN;
    const dim3 grid((unsigned)(tiles_m * tiles_n));
#define TST_GEMM_LAUNCH(FK, HR, RL)                                                                                    \
    hipLaunchKernelGGL((gemm_f32_kernel<WTM, WTN, FK, HR, RL>), grid, dim3(256), 0, st, A, lda, Wt, ldw, bias, res, ldr, C, ldc, \
                       (int)M, (int)N, (int)K, (int)tiles_m, (int)tiles_n)
    const int variant = (K % BK == 0 ? 4 : 0) | (res != nullptr ? 2 : 0) | (act == 1 ? 1 : 0);
    switch (variant) {
        case 0: TST_GEMM_LAUNCH(false, false, false); break;
        case 1: TST_GEMM_LAUNCH(false, false, true); break;
        case 2: TST_GEMM_LAUNCH(false, true, false); break;
        case 3: TST_GEMM_LAUNCH(false, true, true); break;
        case 4: TST_GEMM_LAUNCH(true, false, false); break;
        case 5: TST_GEMM_LAUNCH(true, false, true); break;
        case 6: TST_GEMM_LAUNCH(true, true, false); break;
        default: TST_GEMM_LAUNCH(true, true, true); break;
    }
#undef TST_GEMM_LAUNCH
}

// ---- 64 x 64 tiles with 32 k per tile ----------------------------------------------------------------------------------
// The 64 x 64 x 16 tile above asks for 64 bytes of every A row per k-tile: half a cache line per request, and the second
// half comes with the next tile, from the L2 again when four workgroups share the CU's 16 KB L1.  With 32 k per tile the
// 8 lanes of a row ask for one whole line, and there are half as many barriers.  Same staging scheme (registers, k-major
// LDS images, three buffers, the barrier in the middle of a tile), same fma chain per element; K % 32 == 0.
// LDA = 66: the k-major A image is written by 8 lane groups (k) x 8 rows per wave, 4 * 66 * i mod 32 = 8 i spreads them
// over the banks two deep (free for ds_write_b32).
template <bool HAS_RES, bool RELU>
__global__ __launch_bounds__(256) void gemm_f32_k32_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                           int64_t ldw, const float* __restrict__ bias, const float* res, int64_t ldr,
                                                           float* C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr int BM = 64, BN = 64, KT = 32, LDA = 66, LDB = 68;
    __shared__ __attribute__((aligned(16))) float As[3][KT][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[3][KT][LDB];

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
    const int lb = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int per_group = GROUP_M * tiles_n;
    const int first_m = (lb / per_group) * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (lb % per_group) % gsz;
    const int tn = (lb % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // A: thread -> 16 bytes at k = 4 (t & 7) of rows (t >> 3) and (t >> 3) + 32; Wt: 16 bytes at n = 4 (t & 15) of k-rows (t >> 4), + 16
    const int akq = (t & 7) * 4, ar = t >> 3;
    const int bnq = (t & 15) * 4, bk = t >> 4;
    const float* a_row0 = A + (int64_t)min(m0 + ar, M - 1) * lda + akq;
    const float* a_row1 = A + (int64_t)min(m0 + ar + 32, M - 1) * lda + akq;
    const float* w_p0 = W + (int64_t)bk * ldw + min(n0 + bnq, (int)ldw - 4);
    const float* w_p1 = w_p0 + 16 * ldw;
    float4 ra0, ra1, rb0, rb1;
    auto load_tile = [&](int k0) {
        ra0 = *reinterpret_cast<const float4*>(a_row0 + k0);
        ra1 = *reinterpret_cast<const float4*>(a_row1 + k0);
        rb0 = *reinterpret_cast<const float4*>(w_p0 + (int64_t)k0 * ldw);
        rb1 = *reinterpret_cast<const float4*>(w_p1 + (int64_t)k0 * ldw);
    };
    auto store_tile = [&](int buf) {
        As[buf][akq + 0][ar] = ra0.x; As[buf][akq + 1][ar] = ra0.y; As[buf][akq + 2][ar] = ra0.z; As[buf][akq + 3][ar] = ra0.w;
        As[buf][akq + 0][ar + 32] = ra1.x; As[buf][akq + 1][ar + 32] = ra1.y; As[buf][akq + 2][ar + 32] = ra1.z; As[buf][akq + 3][ar + 32] = ra1.w;
        *reinterpret_cast<float4*>(&Bs[buf][bk][bnq]) = rb0;
        *reinterpret_cast<float4*>(&Bs[buf][bk + 16][bnq]) = rb1;
    };

    f32x16 acc;
    {
        const int col = n0 + wn * 32 + (lane & 31);
        const float b = (bias != nullptr && col < N) ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b;
    }

    const int nkt = K / KT;
    load_tile(0);
    store_tile(0);
    if (nkt > 1) load_tile(KT);
    __syncthreads();
    const int arow = wm * 32 + (lane & 31), bcol = wn * 32 + (lane & 31), khalf = lane >> 5;
    float a_cur = As[0][khalf][arow], b_cur = Bs[0][khalf][bcol], a_nxt = 0.f, b_nxt = 0.f;
    int buf = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const int nbuf = buf == 2 ? 0 : buf + 1;
#pragma unroll
        for (int kk = 0; kk < KT; kk += 2) {
            if (kk + 2 < KT) {
                a_nxt = As[buf][kk + 2 + khalf][arow];
                b_nxt = Bs[buf][kk + 2 + khalf][bcol];
            } else if (kt + 1 < nkt) {
                a_nxt = As[nbuf][khalf][arow];
                b_nxt = Bs[nbuf][khalf][bcol];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur, acc, 0, 0, 0);
            a_cur = a_nxt;
            b_cur = b_nxt;
            if (kk == KT / 2 - 2) {
                if (kt + 1 < nkt) {
                    store_tile(nbuf);
                    if (kt + 2 < nkt) load_tile((kt + 2) * KT);
                }
                __syncthreads();
            }
        }
        buf = nbuf;
    }

    const bool full_tile = m0 + BM <= M && n0 + BN <= N;
    const int col = n0 + wn * 32 + (lane & 31);
    const int row0 = m0 + wm * 32 + 4 * (lane >> 5);
    float rv[16];
    if (full_tile) {
        float* cp = C + (int64_t)row0 * ldc + col;
        if (HAS_RES) {
            const float* rp = res + (int64_t)row0 * ldr + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = rp[(int64_t)((r & 3) + 8 * (r >> 2)) * ldr];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
            if (RELU) v = v > 0.0f ? v : 0.0f;
            if (HAS_RES) v = v + rv[r];
            cp[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = v;
        }
    } else {
        const bool col_ok = col < N;
        if (HAS_RES) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2);
                rv[r] = (col_ok && row < M) ? res[(int64_t)row * ldr + col] : 0.0f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            float v = acc[r];
            if (RELU) v = v > 0.0f ? v : 0.0f;
            if (HAS_RES) v = v + rv[r];
            if (col_ok && row < M) C[(int64_t)row * ldc + col] = v;
        }
    }
}

void launch_gemm_k32(const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias, const float* res, int64_t ldr,
                     float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t st) {
    const int64_t tiles_m = (M + 63) / 64, tiles_n = (N + 63) / 64;
    const dim3 grid((unsigned)(tiles_m * tiles_n));
#define TST_K32_LAUNCH(HR, RL)                                                                                         \
    hipLaunchKernelGGL((gemm_f32_k32_kernel<HR, RL>), grid, dim3(256), 0, st, A, lda, Wt, ldw, bias, res, ldr, C, ldc, (int)M, (int)N, \
                       (int)K, (int)tiles_m, (int)tiles_n)
    if (res != nullptr) {
        if (act == 1) TST_K32_LAUNCH(true, true); else TST_K32_LAUNCH(true, false);
    } else {
        if (act == 1) TST_K32_LAUNCH(false, true); else TST_K32_LAUNCH(false, false);
    }
#undef TST_K32_LAUNCH
}

int tst_ln_wreg_mode() {
    static const int v = [] { const char* e = getenv("TST_LN_WREG"); return (e != nullptr && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 0; }();
    return v;
}

int64_t tst_ln_wreg_min_rows() {
    static const int64_t v = [] { const char* e = getenv("TST_LN_WREG_MIN_ROWS"); return (e != nullptr && atoll(e) > 0) ? (int64_t)atoll(e) : (int64_t)3072; }();
    return v;
}

bool tst_gemm_wreg_on() {
    static const bool v = [] { const char* e = getenv("TST_GEMM_WREG"); return e != nullptr && e[0] == '1'; }();
    return v;
}

bool tst_gemm_no_k32() {
    static const bool v = [] { const char* e = getenv("TST_GEMM_NO_K32"); return e != nullptr && e[0] == '1'; }();
    return v;
}

// ---- small problems -------------------------------------------------------------------------------------------------
// With few output elements the 64 x 64 tiling leaves most of the 1024 SIMDs idle and every tile is one dependent chain of
// K / 2 MFMAs of 64 cycles.  Here one WAVE owns a 16 x (16 * NT) strip of C and uses v_mfma_f32_16x16x4_f32 (same
// k-ordered fmaf semantics: the four k of an instruction are accumulated in ascending order): 16 times as many independent
// chains per output area, each K / 4 instructions of 32 cycles.  No LDS: operands go from global memory (L2 resident at
// these sizes) to registers, KC k-values at a time, one chunk ahead of the MFMAs that use them.
//   A: lane (r = lane & 15, kq = lane >> 4) needs A[m0 + r][4 s + kq] for MFMA step s.  It loads the float4
//      A[m0 + r][16 c + 4 kq ..+3]; a 4 x 4 transpose over (16-lane row, register) with v_permlane32_swap and
//      v_permlane16_swap then leaves A[m0 + r][16 c + 4 j + kq] in register j.
//   W: the NT MFMA tiles of a strip interleave their columns: tile j owns columns n0 + NT * r + j, so the lane's
//      operands of all tiles for one k are ONE load of NT consecutive floats of Wt, and its results are NT consecutive
//      floats of C.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT> struct SkVec;
template <> struct SkVec<1> { typedef float type; };
template <> struct SkVec<2> { typedef float2 type; };
template <> struct SkVec<4> { typedef float4 type; };

template <int NT> __device__ __forceinline__ float sk_get(const typename SkVec<NT>::type& v, int j);
template <> __device__ __forceinline__ float sk_get<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float sk_get<2>(const float2& v, int j) { return j == 0 ? v.x : v.y; }
template <> __device__ __forceinline__ float sk_get<4>(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

template <int NT, bool HAS_RES, bool RELU>
__global__ __launch_bounds__(64) void gemm_f32_skinny_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                             int64_t ldw, const float* __restrict__ bias, const float* res,
                                                             int64_t ldr, float* C, int64_t ldc, int M, int N, int K, int tiles_n,
                                                             int vec_store) {
    typedef typename SkVec<NT>::type wvec;
    constexpr int KC = NT == 4 ? 64 : 128, STEPS = KC / 4;
    const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * 16, n0 = tn * 16 * NT;
    const int col0 = n0 + NT * r;                                   // first of the lane's NT columns
    const float* a_row = A + (int64_t)min(m0 + r, M - 1) * lda;     // rows beyond M: a valid row, never stored
    const float* w_col = W + min(col0, (int)ldw - NT);              // columns beyond ldw: valid ones, never stored

    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const float b = (bias != nullptr && col0 + j < N) ? bias[col0 + j] : 0.0f;
        acc[j] = f32x4{b, b, b, b};
    }
    float a_cur[STEPS], a_nxt[STEPS];
    wvec b_cur[STEPS], b_nxt[STEPS];

    // full chunk: A as float4 per 16 k (transposed in sk_mfma), W as one wvec per step
    auto load_full = [&](float* av, wvec* bv, int k0) {
#pragma unroll
        for (int c = 0; c < STEPS / 4; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(a_row + k0 + 16 * c + 4 * kq);
            av[4 * c + 0] = v.x; av[4 * c + 1] = v.y; av[4 * c + 2] = v.z; av[4 * c + 3] = v.w;
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) bv[s] = *reinterpret_cast<const wvec*>(w_col + (int64_t)(k0 + 4 * s + kq) * ldw);
    };
    auto mfma_full = [&](float* av, wvec* bv) {
#pragma unroll
        for (int c = 0; c < STEPS / 4; ++c) {
            unsigned x0 = __float_as_uint(av[4 * c]), x1 = __float_as_uint(av[4 * c + 1]), x2 = __float_as_uint(av[4 * c + 2]),
                     x3 = __float_as_uint(av[4 * c + 3]);
            const auto p02 = __builtin_amdgcn_permlane32_swap(x0, x2, false, false);
            const auto p13 = __builtin_amdgcn_permlane32_swap(x1, x3, false, false);
            const auto q01 = __builtin_amdgcn_permlane16_swap(p02[0], p13[0], false, false);
            const auto q23 = __builtin_amdgcn_permlane16_swap(p02[1], p13[1], false, false);
            av[4 * c] = __uint_as_float(q01[0]); av[4 * c + 1] = __uint_as_float(q01[1]);
            av[4 * c + 2] = __uint_as_float(q23[0]); av[4 * c + 3] = __uint_as_float(q23[1]);
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], sk_get<NT>(bv[s], j), acc[j], 0, 0, 0);
    };
    // residual values of the lane's results, requested before the k loop (scalar path of the epilogue: one-column strips)
    float rres[4];
    if (HAS_RES && NT == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = min(m0 + 4 * kq + i, M - 1);
            rres[i] = res[(int64_t)row * ldr + min(col0, N - 1)];
        }
    }
    const int k_full = K / KC * KC;
    if (NT == 1 && k_full >= 4 * KC) {
        // long inner dimension (one-column strips: the wave's registers allow it): chunks two ahead of the MFMAs that use them,
        // the 32 MFMAs of a chunk are shorter than a load from the L2; same ascending order of k
        float a_nx2[STEPS];
        wvec b_nx2[STEPS];
        load_full(a_cur, b_cur, 0);
        load_full(a_nxt, b_nxt, KC);
        for (int k0 = 0; k0 < k_full; k0 += 3 * KC) {
            if (k0 + 2 * KC < k_full) load_full(a_nx2, b_nx2, k0 + 2 * KC);
            mfma_full(a_cur, b_cur);
            if (k0 + KC < k_full) {
                if (k0 + 3 * KC < k_full) load_full(a_cur, b_cur, k0 + 3 * KC);
                mfma_full(a_nxt, b_nxt);
            }
            if (k0 + 2 * KC < k_full) {
                if (k0 + 4 * KC < k_full) load_full(a_nxt, b_nxt, k0 + 4 * KC);
                mfma_full(a_nx2, b_nx2);
            }
        }
    } else {
        if (k_full > 0) load_full(a_cur, b_cur, 0);
        for (int k0 = 0; k0 < k_full; k0 += 2 * KC) {
            if (k0 + KC < k_full) load_full(a_nxt, b_nxt, k0 + KC);
            mfma_full(a_cur, b_cur);
            if (k0 + KC < k_full) {
                if (k0 + 2 * KC < k_full) load_full(a_cur, b_cur, k0 + 2 * KC);
                mfma_full(a_nxt, b_nxt);
            }
        }
    }
    // the last K % KC values, four at a time: one dword of A per lane and step
    for (int k = k_full; k < K; k += 4) {
        const float a = a_row[k + kq];
        const wvec b = *reinterpret_cast<const wvec*>(w_col + (int64_t)(k + kq) * ldw);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sk_get<NT>(b, j), acc[j], 0, 0, 0);
    }

    // lane holds rows m0 + 4 kq + i (i = 0..3) of columns col0 .. col0 + NT - 1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + 4 * kq + i;
        if (row >= M) continue;
        float v[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            v[j] = acc[j][i];
            if (RELU) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
        }
        float* cp = C + (int64_t)row * ldc + col0;
        if (NT > 1 && vec_store && col0 + NT <= N) {
            if (HAS_RES) {
                const wvec rv = *reinterpret_cast<const wvec*>(res + (int64_t)row * ldr + col0);
#pragma unroll
                for (int j = 0; j < NT; ++j) v[j] = v[j] + sk_get<NT>(rv, j);
            }
            if (NT == 2) *reinterpret_cast<float2*>(cp) = make_float2(v[0], v[NT - 1]);
            if (NT == 4) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1 % NT], v[2 % NT], v[3 % NT]);
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (col0 + j < N) {
                    float o = v[j];
                    if (HAS_RES) o = o + (NT == 1 ? rres[i] : res[(int64_t)row * ldr + col0 + j]);
                    cp[j] = o;
                }
            }
        }
    }
}

template <int NT>
void launch_skinny(const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias, const float* res, int64_t ldr,
                   float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t st) {
    const int64_t tiles_m = (M + 15) / 16, tiles_n = (N + 16 * NT - 1) / (16 * NT);
    const dim3 grid((unsigned)(tiles_m * tiles_n));
    // NT consecutive results per lane go out as one store when C (and res) rows keep them aligned
    const int vec_store = ldc % NT == 0 && (uintptr_t)C % (4 * NT) == 0 &&
                          (res == nullptr || (ldr % NT == 0 && (uintptr_t)res % (4 * NT) == 0));
#define TST_SK_LAUNCH(HR, RL)                                                                                               \
    hipLaunchKernelGGL((gemm_f32_skinny_kernel<NT, HR, RL>), grid, dim3(64), 0, st, A, lda, Wt, ldw, bias, res, ldr, C, ldc, \
                       (int)M, (int)N, (int)K, (int)tiles_n, vec_store)
    switch ((res != nullptr ? 2 : 0) | (act == 1 ? 1 : 0)) {
        case 0: TST_SK_LAUNCH(false, false); break;
        case 1: TST_SK_LAUNCH(false, true); break;
        case 2: TST_SK_LAUNCH(true, false); break;
        default: TST_SK_LAUNCH(true, true); break;
    }
#undef TST_SK_LAUNCH
}

// LayerNorm folded into the small-problem kernel: C = epilogue(LayerNorm(x) * Wt + bias) for K = 256 or 512 (the row
// length).  A wave holds its 16 rows of x completely in registers (K / 16 float4 per lane), computes the statistics in
// exactly the order of layernorm_kernel (64 interleaved partial sums, partial = float4 chunk % 64, xor-butterfly 32..1;
// lane (r, kq) owns chunks 4 c + kq, so the butterfly levels 32, 16, 8, 4 pair registers of the lane and levels 2, 1 pair
// lanes 32 and 16 apart), normalises in place and multiplies.  Every wave of a row strip repeats the statistics: a few
// hundred VALU instructions against a kernel launch and a round trip of the normalised rows through memory.
template <int NT, int KCH, bool RELU>
__global__ __launch_bounds__(64) void gemm_f32_skinny_ln_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, const float* __restrict__ W,
                                                                int64_t ldw, const float* __restrict__ bias, float* C, int64_t ldc,
                                                                int M, int N, int tiles_n, int vec_store) {
    typedef typename SkVec<NT>::type wvec;
    constexpr int K = 16 * KCH;
    constexpr int BC = NT == 4 ? 4 : 8;            // 16-k chunks of Wt loaded ahead (64 or 128 k)
    const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * 16, n0 = tn * 16 * NT;
    const int col0 = n0 + NT * r;
    const float* x_row = X + (int64_t)min(m0 + r, M - 1) * ldx;
    const float* w_col = W + min(col0, (int)ldw - NT);

    wvec b_cur[4 * BC], b_nxt[4 * BC];
    auto load_b = [&](wvec* bv, int c0) {
#pragma unroll
        for (int s = 0; s < 4 * BC; ++s) bv[s] = *reinterpret_cast<const wvec*>(w_col + (int64_t)(16 * c0 + 4 * s + kq) * ldw);
    };
    load_b(b_cur, 0);                              // in flight during the normalisation

    float4 a[KCH];                                 // chunk c: elements 16 c + 4 kq .. + 3 of the lane's row
#pragma unroll
    for (int c = 0; c < KCH; ++c) a[c] = *reinterpret_cast<const float4*>(x_row + 16 * c + 4 * kq);
    // 64 partials per row: partial (4 c + kq) % 64 = 4 (c % 16) + kq collects chunks c, c + 16, ... in ascending order
    auto row_total = [&](float (&p)[16]) {
#pragma unroll
        for (int c = 0; c < 8; ++c) p[c] = p[c] + p[c + 8];      // butterfly level 32 (chunk ^ 32 <-> c ^ 8)
#pragma unroll
        for (int c = 0; c < 4; ++c) p[c] = p[c] + p[c + 4];      // 16
#pragma unroll
        for (int c = 0; c < 2; ++c) p[c] = p[c] + p[c + 2];      // 8
        float v = p[0] + p[1];                                   // 4
        v = v + lane_xor<32>(v);                                 // 2 (kq ^ 2)
        v = v + lane_xor<16>(v);                                 // 1 (kq ^ 1)
        return v;
    };
    float p[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float q = 0.0f;
#pragma unroll
        for (int c = j; c < KCH; c += 16) { q = q + a[c].x; q = q + a[c].y; q = q + a[c].z; q = q + a[c].w; }
        p[j] = q;
    }
    const float dlen = (float)K;
    const float mean = row_total(p) / dlen;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float q = 0.0f;
#pragma unroll
        for (int c = j; c < KCH; c += 16) {
            float t;
            t = a[c].x - mean; q = __builtin_fmaf(t, t, q);
            t = a[c].y - mean; q = __builtin_fmaf(t, t, q);
            t = a[c].z - mean; q = __builtin_fmaf(t, t, q);
            t = a[c].w - mean; q = __builtin_fmaf(t, t, q);
        }
        p[j] = q;
    }
    const float var = row_total(p) / dlen;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        const float4 gg = *reinterpret_cast<const float4*>(gamma + 16 * c + 4 * kq);
        const float4 bb = *reinterpret_cast<const float4*>(beta + 16 * c + 4 * kq);
        a[c].x = __builtin_fmaf((a[c].x - mean) * rstd, gg.x, bb.x);
        a[c].y = __builtin_fmaf((a[c].y - mean) * rstd, gg.y, bb.y);
        a[c].z = __builtin_fmaf((a[c].z - mean) * rstd, gg.z, bb.z);
        a[c].w = __builtin_fmaf((a[c].w - mean) * rstd, gg.w, bb.w);
    }

    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const float b = (bias != nullptr && col0 + j < N) ? bias[col0 + j] : 0.0f;
        acc[j] = f32x4{b, b, b, b};
    }
    auto mfma_chunks = [&](const float4* ac, wvec* bv) {   // BC chunks of 16 k: transpose (see gemm_f32_skinny_kernel), multiply
#pragma unroll
        for (int c = 0; c < BC; ++c) {
            const unsigned x0 = __float_as_uint(ac[c].x), x1 = __float_as_uint(ac[c].y), x2 = __float_as_uint(ac[c].z),
                           x3 = __float_as_uint(ac[c].w);
            const auto p02 = __builtin_amdgcn_permlane32_swap(x0, x2, false, false);
            const auto p13 = __builtin_amdgcn_permlane32_swap(x1, x3, false, false);
            const auto q01 = __builtin_amdgcn_permlane16_swap(p02[0], p13[0], false, false);
            const auto q23 = __builtin_amdgcn_permlane16_swap(p02[1], p13[1], false, false);
            const float av[4] = {__uint_as_float(q01[0]), __uint_as_float(q01[1]), __uint_as_float(q23[0]), __uint_as_float(q23[1])};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], sk_get<NT>(bv[4 * c + s], j), acc[j], 0, 0, 0);
        }
    };
    static_assert(KCH % (2 * BC) == 0, "row length must be a multiple of two Wt chunks");
#pragma unroll
    for (int c0 = 0; c0 < KCH; c0 += 2 * BC) {
        load_b(b_nxt, c0 + BC);
        mfma_chunks(&a[c0], b_cur);
        if (c0 + 2 * BC < KCH) load_b(b_cur, c0 + 2 * BC);
        mfma_chunks(&a[c0 + BC], b_nxt);
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + 4 * kq + i;
        if (row >= M) continue;
        float v[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            v[j] = acc[j][i];
            if (RELU) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
        }
        float* cp = C + (int64_t)row * ldc + col0;
        if (NT > 1 && vec_store && col0 + NT <= N) {
            if (NT == 2) *reinterpret_cast<float2*>(cp) = make_float2(v[0], v[NT - 1]);
            if (NT == 4) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1 % NT], v[2 % NT], v[3 % NT]);
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j)
                if (col0 + j < N) cp[j] = v[j];
        }
    }
}

template <int NT>
void launch_skinny_ln(const float* X, int64_t ldx, const float* gamma, const float* beta, float eps, const float* Wt, int64_t ldw,
                      const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, hipStream_t st) {
    const int64_t tiles_m = (M + 15) / 16, tiles_n = (N + 16 * NT - 1) / (16 * NT);
    const dim3 grid((unsigned)(tiles_m * tiles_n));
    const int vec_store = ldc % NT == 0 && (uintptr_t)C % (4 * NT) == 0;
#define TST_SKLN_LAUNCH(KCH_, RL)                                                                                                  \
    hipLaunchKernelGGL((gemm_f32_skinny_ln_kernel<NT, KCH_, RL>), grid, dim3(64), 0, st, X, ldx, gamma, beta, eps, Wt, ldw, bias, C, \
                       ldc, (int)M, (int)N, (int)tiles_n, vec_store)
    if (K == 256) { if (act == 1) TST_SKLN_LAUNCH(16, true); else TST_SKLN_LAUNCH(16, false); }
    else          { if (act == 1) TST_SKLN_LAUNCH(32, true); else TST_SKLN_LAUNCH(32, false); }
#undef TST_SKLN_LAUNCH
}

// ---- greedy decoding of small batches: final LayerNorm, vocabulary projection and arg-max in one kernel -------------------
// keys[m][slot] = max of key(logit[m][n], n) over the columns n of the tiles that use the slot, logit = bias[n] + sum_k
// LayerNorm(x)[m][k] * Wt[k][n]: neither the normalised rows nor the logits travel through memory.  Tiles of 64 x 64 x 32 with
// k-major LDS images in three buffers and one barrier per k-tile, as gemm_f32_k32_kernel, with these differences:
//  - every wave loads 16 of the tile's rows COMPLETELY into registers (lane (r, kq) holds the float4 chunks 4 c + kq of row r, as
//    gemm_f32_skinny_ln_kernel), computes their LayerNorm statistics in exactly the order of layernorm_kernel (64 interleaved
//    partial sums folded in the butterfly's order) and normalises them in place: the A tiles of the k loop are written to LDS
//    from these registers, so x is read once per workgroup and the k loop loads only Wt;
//  - the product is computed transposed (operands swapped: the same fma chain per element), so that a lane owns a ROW and
//    its 16 accumulator registers are 16 columns of it, ascending;
//  - the epilogue keeps the first largest logit of the lane's columns (v > best, as argmax_rows_kernel), merges the two halves
//    of the wave and, through LDS, the two waves that share the rows, and raises keys[row][tile column % TST_ARGMAX_SLOTS] with
//    one 64-bit atomic max per row and workgroup.  The slots spread the atomics of a row over 16 addresses.
// key = order-preserving bits of the value (with -0 read as +0) in the upper word, ~column in the lower: the largest key is the
// largest value at its smallest column.  NaN and -inf never form a key, so a row without any finite or +inf logit keeps key 0,
// which decodes to token 0, as tst_argmax_rows_f32 gives.  keys must be 0 on entry.
__device__ __forceinline__ unsigned long long argmax_key(float v, int col) {
    if (v == 0.0f) v = 0.0f;   // -0 -> +0: the two compare equal in argmax_rows_kernel
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (uint32_t)(~(uint32_t)col);
}

template <int KCH>
__global__ __launch_bounds__(256, (KCH <= 16 ? 3 : 2)) void ln_vocab_argmax_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, const float* __restrict__ W,
                                                                 int64_t ldw, const float* __restrict__ bias, unsigned long long* keys,
                                                                 int M, int N, int tiles_m, int tiles_n) {
    constexpr int K = 16 * KCH, BM = 64, BN = 64, KT = 32, LDA = 66, LDB = 68, NKT = K / KT;
    __shared__ __attribute__((aligned(16))) float As[3][KT][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[3][KT][LDB];
    __shared__ unsigned long long key_s[BM];

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
    const int lb = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int per_group = GROUP_M * tiles_n;
    const int first_m = (lb / per_group) * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (lb % per_group) % gsz;
    const int tn = (lb % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // Wt tile: 16 bytes at n = 4 (t & 15) of k-rows (t >> 4), + 16; the first two tiles are requested before anything else
    const int bnq = (t & 15) * 4, bk = t >> 4;
    const float* w_p0 = W + (int64_t)bk * ldw + min(n0 + bnq, (int)ldw - 4);
    const float* w_p1 = w_p0 + 16 * ldw;
    float4 rb0 = *reinterpret_cast<const float4*>(w_p0);
    float4 rb1 = *reinterpret_cast<const float4*>(w_p1);
    const int kq0 = lane >> 4;
    const float4 gf0 = *reinterpret_cast<const float4*>(gamma + 4 * kq0), gf1 = *reinterpret_cast<const float4*>(gamma + 16 + 4 * kq0);
    const float4 tf0 = *reinterpret_cast<const float4*>(beta + 4 * kq0), tf1 = *reinterpret_cast<const float4*>(beta + 16 + 4 * kq0);

    // rows m0 + 16 wave + r of x: lane (r, kq) holds the chunks 4 c + kq (elements 16 c + 4 kq .. + 3) of its row
    const int r = lane & 15, kq = lane >> 4;
    const int srow = 16 * wave + r;
    float4 a[KCH];
    float mean, rstd;
    {
        const float* x_row = X + (int64_t)min(m0 + srow, M - 1) * K;
#pragma unroll
        for (int c = 0; c < KCH; ++c) a[c] = *reinterpret_cast<const float4*>(x_row + 16 * c + 4 * kq);
        auto row_total = [&](float (&p)[16]) {
#pragma unroll
            for (int c = 0; c < 8; ++c) p[c] = p[c] + p[c + 8];      // butterfly level 32
#pragma unroll
            for (int c = 0; c < 4; ++c) p[c] = p[c] + p[c + 4];      // 16
#pragma unroll
            for (int c = 0; c < 2; ++c) p[c] = p[c] + p[c + 2];      // 8
            float v = p[0] + p[1];                                   // 4
            v = v + lane_xor<32>(v);                                 // 2
            v = v + lane_xor<16>(v);                                 // 1
            return v;
        };
        float p[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int c = j; c < KCH; c += 16) { s = s + a[c].x; s = s + a[c].y; s = s + a[c].z; s = s + a[c].w; }
            p[j] = s;
        }
        const float dlen = (float)K;
        mean = row_total(p) / dlen;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int c = j; c < KCH; c += 16) {
                float u;
                u = a[c].x - mean; s = __builtin_fmaf(u, u, s);
                u = a[c].y - mean; s = __builtin_fmaf(u, u, s);
                u = a[c].z - mean; s = __builtin_fmaf(u, u, s);
                u = a[c].w - mean; s = __builtin_fmaf(u, u, s);
            }
            p[j] = s;
        }
        const float var = row_total(p) / dlen;
        rstd = 1.0f / sqrtf(var + eps);
    }
    // k-tile kt of A is the chunks 2 kt and 2 kt + 1 of every row: k = 4 kq + i and 16 + 4 kq + i of the tile, normalised on
    // their way into LDS with the tile's gamma / beta (requested a tile ahead, with the Wt tile)
    float4 g0, g1, t0, t1;
    auto store_a = [&](int buf, const float4& c0, const float4& c1) {
        As[buf][4 * kq + 0][srow] = __builtin_fmaf((c0.x - mean) * rstd, g0.x, t0.x);
        As[buf][4 * kq + 1][srow] = __builtin_fmaf((c0.y - mean) * rstd, g0.y, t0.y);
        As[buf][4 * kq + 2][srow] = __builtin_fmaf((c0.z - mean) * rstd, g0.z, t0.z);
        As[buf][4 * kq + 3][srow] = __builtin_fmaf((c0.w - mean) * rstd, g0.w, t0.w);
        As[buf][16 + 4 * kq + 0][srow] = __builtin_fmaf((c1.x - mean) * rstd, g1.x, t1.x);
        As[buf][16 + 4 * kq + 1][srow] = __builtin_fmaf((c1.y - mean) * rstd, g1.y, t1.y);
        As[buf][16 + 4 * kq + 2][srow] = __builtin_fmaf((c1.z - mean) * rstd, g1.z, t1.z);
        As[buf][16 + 4 * kq + 3][srow] = __builtin_fmaf((c1.w - mean) * rstd, g1.w, t1.w);
    };
    auto store_b = [&](int buf) {
        *reinterpret_cast<float4*>(&Bs[buf][bk][bnq]) = rb0;
        *reinterpret_cast<float4*>(&Bs[buf][bk + 16][bnq]) = rb1;
    };
    auto load_gb = [&](int k0) {
        g0 = *reinterpret_cast<const float4*>(gamma + k0 + 4 * kq);
        g1 = *reinterpret_cast<const float4*>(gamma + k0 + 16 + 4 * kq);
        t0 = *reinterpret_cast<const float4*>(beta + k0 + 4 * kq);
        t1 = *reinterpret_cast<const float4*>(beta + k0 + 16 + 4 * kq);
    };
    auto load_b = [&](int k0) {
        rb0 = *reinterpret_cast<const float4*>(w_p0 + (int64_t)k0 * ldw);
        rb1 = *reinterpret_cast<const float4*>(w_p1 + (int64_t)k0 * ldw);
        load_gb(k0);
    };

    // lane owns row (lane & 31) of the wave's 32 x 32 block; register j is column (j & 3) + 8 (j >> 2) + 4 (lane >> 5)
    const int cbase = n0 + wn * 32 + 4 * (lane >> 5);
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int col = cbase + (j & 3) + 8 * (j >> 2);
        acc[j] = (bias != nullptr && col < N) ? bias[col] : 0.0f;
    }

    g0 = gf0; g1 = gf1; t0 = tf0; t1 = tf1;
    store_a(0, a[0], a[1]);
    store_b(0);
    load_b(KT);
    __syncthreads();
    const int arow = wm * 32 + (lane & 31), bcol = wn * 32 + (lane & 31), khalf = lane >> 5;
    float a_cur = As[0][khalf][arow], b_cur = Bs[0][khalf][bcol], a_nxt = 0.f, b_nxt = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int buf = kt % 3, nbuf = (kt + 1) % 3;
#pragma unroll
        for (int kk = 0; kk < KT; kk += 2) {
            if (kk + 2 < KT) {
                a_nxt = As[buf][kk + 2 + khalf][arow];
                b_nxt = Bs[buf][kk + 2 + khalf][bcol];
            } else if (kt + 1 < NKT) {
                a_nxt = As[nbuf][khalf][arow];
                b_nxt = Bs[nbuf][khalf][bcol];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b_cur, a_cur, acc, 0, 0, 0);
            a_cur = a_nxt;
            b_cur = b_nxt;
            if (kk == KT / 2 - 2) {
                if (kt + 1 < NKT) {
                    store_a(nbuf, a[(2 * kt + 2) % KCH], a[(2 * kt + 3) % KCH]);
                    store_b(nbuf);
                    if (kt + 2 < NKT) load_b((kt + 2) * KT);
                }
                __syncthreads();
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // the loop is unrolled (a[] is indexed by kt): keep the tiles' instructions apart
    }

    float bv = -INFINITY;
    int bi = INT_MAX;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int col = cbase + (j & 3) + 8 * (j >> 2);
        if (col < N && acc[j] > bv) { bv = acc[j]; bi = col; }
    }
    {
        const float ov = lane_xor<32>(bv);
        const int oi = lane_xor_i<32>(bi);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    // the two waves of a row block: the right one hands its keys over through LDS (its columns are the larger ones, so on
    // equal values the left wave's key is the larger, as it must be)
    unsigned long long key = bi != INT_MAX ? argmax_key(bv, bi) : 0ull;
    if (wn == 1 && lane < 32) key_s[wm * 32 + lane] = key;
    __syncthreads();
    if (wn == 0 && lane < 32) {
        const unsigned long long other = key_s[wm * 32 + lane];
        if (other > key) key = other;
        const int row = m0 + wm * 32 + lane;
        if (row < M && key != 0ull) {
            unsigned long long* kp = keys + (int64_t)row * TST_ARGMAX_SLOTS + (tn % TST_ARGMAX_SLOTS);
            // unconditionally: reading the key first, to skip atomics that cannot win, made the kernel 6 - 12 us slower (a load
            // round trip at the end of every workgroup) while the atomics themselves cost under 1 us (measured)
            atomicMax(kp, key);
        }
    }
}

int check_gemm_args(const char* who, const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* res, int64_t ldr,
                    float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act) {
    TST_REQUIRE(A && Wt && C, TST_ERR_ARG, "%s: null pointer", who);
    TST_REQUIRE(M >= 0 && N > 0 && K >= 4 && (act == 0 || act == 1), TST_ERR_ARG, "%s: bad sizes M=%lld N=%lld K=%lld", who, (long long)M,
                (long long)N, (long long)K);
    TST_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= K && ldw >= N && ldc >= N, TST_ERR_ARG,
                "%s: K, lda, ldw must be multiples of 4 and lda >= K, ldw >= N, ldc >= N", who);
    TST_REQUIRE(tst_aligned16(A) && tst_aligned16(Wt), TST_ERR_ARG, "%s: A and Wt must be 16-byte aligned", who);
    TST_REQUIRE(res == nullptr || ldr >= N, TST_ERR_ARG, "%s: ldr < N", who);
    TST_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), TST_ERR_UNSUPPORTED, "%s: size", who);
    return TST_OK;
}

}  // namespace

extern "C" int tst_gemm_skinny_f32(const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias,
                                   const float* res, int64_t ldr, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                   int act, int strip, void* stream) {
    TST_REQUIRE(tst_dims_ok({M, N, K}), TST_ERR_UNSUPPORTED, "tst_gemm_skinny_f32: a size exceeds 2^31 - 1");
    const int rc = check_gemm_args("tst_gemm_skinny_f32", A, lda, Wt, ldw, res, ldr, C, ldc, M, N, K, act);
    if (rc != TST_OK) return rc;
    TST_REQUIRE(strip == 1 || strip == 2 || strip == 4, TST_ERR_ARG, "tst_gemm_skinny_f32: strip must be 1, 2 or 4");
    if (M == 0) return TST_OK;
    TST_REQUIRE(((M + 15) / 16) * ((N + 15) / 16) < (1ll << 31), TST_ERR_UNSUPPORTED, "tst_gemm_skinny_f32: grid too large");
    if (strip == 1) launch_skinny<1>(A, lda, Wt, ldw, bias, res, ldr, C, ldc, M, N, K, act, (hipStream_t)stream);
    else if (strip == 2) launch_skinny<2>(A, lda, Wt, ldw, bias, res, ldr, C, ldc, M, N, K, act, (hipStream_t)stream);
    else launch_skinny<4>(A, lda, Wt, ldw, bias, res, ldr, C, ldc, M, N, K, act, (hipStream_t)stream);
    return tst_after_launch("tst_gemm_skinny_f32");
}

extern "C" int tst_ln_gemm_f32(const float* x, const float* gamma, const float* beta, float eps, float* h, const float* Wt,
                               int64_t ldw, const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act,
                               void* stream) {
    TST_REQUIRE(tst_dims_ok({M, N, K}), TST_ERR_UNSUPPORTED, "tst_ln_gemm_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(x && gamma && beta && h, TST_ERR_ARG, "tst_ln_gemm_f32: null pointer");
    const int rc = check_gemm_args("tst_ln_gemm_f32", x, K, Wt, ldw, nullptr, 0, C, ldc, M, N, K, act);
    if (rc != TST_OK) return rc;
    TST_REQUIRE(tst_aligned16(gamma) && tst_aligned16(beta) && tst_aligned16(h), TST_ERR_ARG, "tst_ln_gemm_f32: alignment");
    if (M == 0) return TST_OK;
    const int64_t tiles16 = ((M + 15) / 16) * ((N + 15) / 16);
    if ((K == 256 || K == 512) && tiles16 <= 2048) {   // small: LayerNorm inside the one-wave-per-strip kernel
        const hipStream_t st = (hipStream_t)stream;
        if (tiles16 <= 768) launch_skinny_ln<1>(x, K, gamma, beta, eps, Wt, ldw, bias, C, ldc, M, N, K, act, st);
        else launch_skinny_ln<2>(x, K, gamma, beta, eps, Wt, ldw, bias, C, ldc, M, N, K, act, st);
        return tst_after_launch("tst_ln_gemm_f32");
    }
    // large, K = 256, only under TST_LN_WREG in the environment (1: outputs of at least 512 columns, 2: all; from
    // TST_LN_WREG_MIN_ROWS rows, default 3072): LayerNorm inside the staging of the kernel that keeps the weights in registers.
    // Same bits; measured equal or slower inside the decode loop than the LayerNorm launch + tile kernels, hence off.
    if (K == 256 && tst_ln_wreg_mode() > 0 && tst_gemm_wreg_ok(M, N, K) && M >= tst_ln_wreg_min_rows() && (N >= 512 || tst_ln_wreg_mode() == 2)) {
        tst_launch_gemm_wreg_ln(x, gamma, beta, eps, Wt, ldw, bias, C, ldc, M, N, act, (hipStream_t)stream);
        return tst_after_launch("tst_ln_gemm_f32");
    }
    const int rc2 = tst_layernorm_f32(x, gamma, beta, eps, h, M, K, stream);
    if (rc2 != TST_OK) return rc2;
    return tst_gemm_mfma_f32(h, K, Wt, ldw, bias, nullptr, 0, C, ldc, M, N, K, act, stream);
}

// shapes for which the fused kernel replaces LayerNorm + tst_gemm_mfma_f32 + tst_argmax_rows_f32: rows of 256 or 512 floats and
// a product small enough that tst_gemm_mfma_f32 would itself run 64 x 64 tiles (or strips)
extern "C" int tst_vocab_argmax_ok(int64_t R, int64_t V, int64_t d) {
    return (d == 256 || d == 512) && R >= 1 && V >= 1 && V < (1ll << 30) && ((R + 127) / 128) * ((V + 127) / 128) < 640;
}

extern "C" int tst_ln_vocab_argmax_f32(const float* x, const float* gamma, const float* beta, float eps, const float* Wt, int64_t ldw,
                                       const float* bias, uint64_t* keys, int64_t R, int64_t V, int64_t d, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, V, d}), TST_ERR_UNSUPPORTED, "tst_ln_vocab_argmax_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(x && gamma && beta && Wt && keys, TST_ERR_ARG, "tst_ln_vocab_argmax_f32: null pointer");   // keys: [R][TST_ARGMAX_SLOTS]
    TST_REQUIRE(R >= 0 && V > 0 && ldw % 4 == 0 && ldw >= V && ldw >= 4, TST_ERR_ARG, "tst_ln_vocab_argmax_f32: bad sizes (ldw %% 4 == 0, ldw >= V)");
    TST_REQUIRE(tst_aligned16(x) && tst_aligned16(gamma) && tst_aligned16(beta) && tst_aligned16(Wt) && tst_aligned16(keys),
                TST_ERR_ARG, "tst_ln_vocab_argmax_f32: alignment");
    if (R == 0) return TST_OK;
    TST_REQUIRE(tst_vocab_argmax_ok(R, V, d), TST_ERR_UNSUPPORTED,
                "tst_ln_vocab_argmax_f32: rows of 256 or 512 floats and ceil(R / 128) * ceil(V / 128) < 640 (see tst_vocab_argmax_ok)");
    const int64_t tiles_m = (R + 63) / 64, tiles_n = (V + 63) / 64;
    const dim3 grid((unsigned)(tiles_m * tiles_n));
    if (d == 256)
        hipLaunchKernelGGL((ln_vocab_argmax_kernel<16>), grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, Wt, ldw, bias,
                           (unsigned long long*)keys, (int)R, (int)V, (int)tiles_m, (int)tiles_n);
    else
        hipLaunchKernelGGL((ln_vocab_argmax_kernel<32>), grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, Wt, ldw, bias,
                           (unsigned long long*)keys, (int)R, (int)V, (int)tiles_m, (int)tiles_n);
    return tst_after_launch("tst_ln_vocab_argmax_f32");
}

extern "C" int tst_gemm_mfma_f32(const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias,
                                 const float* res, int64_t ldr, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                 int act, void* stream) {
    TST_REQUIRE(tst_dims_ok({M, N, K}), TST_ERR_UNSUPPORTED, "tst_gemm_mfma_f32: a size exceeds 2^31 - 1");
    const int rc = check_gemm_args("tst_gemm_mfma_f32", A, lda, Wt, ldw, res, ldr, C, ldc, M, N, K, act);
    if (rc != TST_OK) return rc;
    if (M == 0) return TST_OK;
    TST_REQUIRE(((M + 63) / 64) * ((N + 63) / 64) < (1ll << 31), TST_ERR_UNSUPPORTED, "tst_gemm_mfma_f32: grid too large");
    // Few output elements: one wave per strip (measured crossovers on the decode shapes, in 16 x 16 tiles of C).
    // Otherwise 128 x 128 tiles unless they would give fewer than 2.5 workgroups per CU (256 CUs; measured crossover): then 64 x 64.
    const int64_t tiles16 = ((M + 15) / 16) * ((N + 15) / 16);
    const int64_t big_tiles = ((M + 127) / 128) * ((N + 127) / 128);
    if (tiles16 <= 768)
        launch_skinny<1>(A, lda, Wt, ldw, bias, res, ldr, C, ldc, M, N, K, act, (hipStream_t)stream);
    else if (tiles16 <= 2048)
        launch_skinny<2>(A, lda, Wt, ldw, bias, res, ldr, C, ldc, M, N, K, act, (hipStream_t)stream);
    else if (tiles16 <= 3072)
        launch_skinny<4>(A, lda, Wt, ldw, bias, res, ldr, C, ldc, M, N, K, act, (hipStream_t)stream);
    else if (tst_gemm_wreg_on() && tst_gemm_wreg_ok(M, N, K))
        tst_launch_gemm_wreg(A, lda, Wt, ldw, bias, res, ldr, C, ldc, M, N, act, 0, (hipStream_t)stream);
    else if (big_tiles >= 4096 && N >= 2048)
        launch_gemm<2, 4>(A, lda, Wt, ldw, bias, res, ldr, C, ldc, M, N, K, act, (hipStream_t)stream);
    else if (big_tiles >= 640)
        launch_gemm<2, 2>(A, lda, Wt, ldw, bias, res, ldr, C, ldc, M, N, K, act, (hipStream_t)stream);
    else if (K % 32 == 0 && !tst_gemm_no_k32())
        launch_gemm_k32(A, lda, Wt, ldw, bias, res, ldr, C, ldc, M, N, K, act, (hipStream_t)stream);
    else
        launch_gemm<1, 1>(A, lda, Wt, ldw, bias, res, ldr, C, ldc, M, N, K, act, (hipStream_t)stream);
    return tst_after_launch("tst_gemm_mfma_f32");
}
