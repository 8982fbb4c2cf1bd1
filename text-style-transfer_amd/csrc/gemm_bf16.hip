// bf16 GEMM on the gfx950 matrix cores: C = epilogue(bias + A * W^T), fp32 accumulation.
//
// A [M][K] and W [N][K] are bf16 with K contiguous (an nn.Linear weight as it is stored), so both MFMA
// operands of v_mfma_f32_32x32x16_bf16 — lane (r = lane & 31, h = lane >> 5) holds 8 consecutive k of row r —
// are single 16-byte reads.  Tile 128 x 128 x 64 per 256-thread workgroup, 4 waves as 2 (M) x 2 (N), each
// wave 64 x 64 = 2 x 2 MFMA tiles.  Tiles are staged global -> registers -> LDS (rows padded from 128 to
// 144 bytes, which makes the 16-lane groups of ds_read_b128 hit 16 different 16-byte bank slots), two LDS
// buffers, the loads of tile t+1 in flight while tile t is multiplied, one barrier per tile.  The epilogue
// goes through LDS (the tile buffers are free by then) so that every lane stores 16 contiguous bytes:
// the accumulator layout (one column per lane) would otherwise give 2- or 4-byte scattered stores.
// Workgroups are renumbered so that each XCD walks a contiguous, m-grouped range of tiles.
#include "tst_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BM = 128, BN = 128, BK = 64, GROUP_M = 16;
constexpr int ROWB = BK * 2 + 16;              // bytes per LDS tile row (128 data + 16 pad)
constexpr int TILE_BYTES = BM * ROWB;          // one operand tile in LDS: 18432 bytes
constexpr int CS_LD = 68;                      // floats per row of a wave's 64 x 64 epilogue staging tile

union Chunk {
    uint4 u;
    bf16x8 v;
};

template <bool OUT_BF16, bool HAS_RES, bool RELU>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const __bf16* __restrict__ A, int64_t lda,
                                                        const __bf16* __restrict__ W, int64_t ldw,
                                                        const float* __restrict__ bias, const float* res, int64_t ldr,
                                                        void* Cout, int64_t ldc, int M, int N, int K, int tiles_m,
                                                        int tiles_n) {
    // [buffer][operand][row][ROWB]; reused as 4 x [64][CS_LD] floats by the epilogue (69632 <= 73728 bytes)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE_BYTES];

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

    // staging: thread owns 16-byte chunk (t & 7) of rows (t >> 3) + 32 j, j = 0..3, of both operand tiles
    const int sc = t & 7, sr = t >> 3;
    const __bf16* a_ptr[4];
    const __bf16* w_ptr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a_ptr[j] = A + (int64_t)min(m0 + sr + 32 * j, M - 1) * lda;
        w_ptr[j] = W + (int64_t)min(n0 + sr + 32 * j, N - 1) * ldw;
    }
    uint4 ra[4], rb[4];
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    auto load_tile = [&](int k0) {
        const int gk = k0 + sc * 8;
        const int gkc = min(gk, K - 8);        // always a valid address; data beyond K is zeroed below
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = *reinterpret_cast<const uint4*>(a_ptr[j] + gkc);
            rb[j] = *reinterpret_cast<const uint4*>(w_ptr[j] + gkc);
        }
        if (gk >= K) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { ra[j] = zero; rb[j] = zero; }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* as = smem + (buf * 2 + 0) * TILE_BYTES + sr * ROWB + sc * 16;
        unsigned char* bs = smem + (buf * 2 + 1) * TILE_BYTES + sr * ROWB + sc * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<uint4*>(as + 32 * j * ROWB) = ra[j];
            *reinterpret_cast<uint4*>(bs + 32 * j * ROWB) = rb[j];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        const float b = (bias != nullptr && col < N) ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = b;
    }

    const int nkt = (K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    // fragment addresses: row (wave tile + lane & 31), 16 bytes at k-step * 32 + (lane >> 5) * 16
    const int frag_off = (lane & 31) * ROWB + (lane >> 5) * 16;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tile((kt + 1) * BK);
        const unsigned char* as = smem + (buf * 2 + 0) * TILE_BYTES + wm * 64 * ROWB + frag_off;
        const unsigned char* bs = smem + (buf * 2 + 1) * TILE_BYTES + wn * 64 * ROWB + frag_off;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            Chunk fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i].u = *reinterpret_cast<const uint4*>(as + i * 32 * ROWB + ks * 32);
                fb[i].u = *reinterpret_cast<const uint4*>(bs + i * 32 * ROWB + ks * 32);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].v, fb[j].v, acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) store_tile(buf ^ 1);   // last read in tile kt - 1; every wave has passed that barrier
        __syncthreads();
    }

    // epilogue through LDS: the wave writes its 64 x 64 fp32 tile (one column per lane), then reads it back
    // by rows, 8 consecutive columns per lane, and stores 16 (bf16) or 2 x 16 (fp32) contiguous bytes
    float* cs = reinterpret_cast<float*>(smem) + wave * 64 * CS_LD;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][j][r];
                if (RELU) v = v > 0.0f ? v : 0.0f;
                cs[row * CS_LD + j * 32 + (lane & 31)] = v;
            }
    // a wave only reads what it wrote itself: no workgroup barrier needed, only the LDS counter wait the
    // compiler inserts
    const int c8 = (lane & 7) * 8;
    const int gcol = n0 + wn * 64 + c8;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int lrow = p * 8 + (lane >> 3);
        const int grow = m0 + wm * 64 + lrow;
        if (grow >= M || gcol >= N) continue;
        float v[8];
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(cs + lrow * CS_LD + c8);
        *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(cs + lrow * CS_LD + c8 + 4);
        const bool full = gcol + 8 <= N;
        if (HAS_RES) {
            const float* rp = res + (int64_t)grow * ldr + gcol;
            if (full) {
                const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
            } else {
                for (int e = 0; e < 8; ++e)
                    if (gcol + e < N) v[e] += rp[e];
            }
        }
        if (OUT_BF16) {
            __bf16* cp = reinterpret_cast<__bf16*>(Cout) + (int64_t)grow * ldc + gcol;
            if (full) {
                Chunk o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.v[e] = (__bf16)v[e];
                *reinterpret_cast<uint4*>(cp) = o.u;
            } else {
                for (int e = 0; e < 8; ++e)
                    if (gcol + e < N) cp[e] = (__bf16)v[e];
            }
        } else {
            float* cp = reinterpret_cast<float*>(Cout) + (int64_t)grow * ldc + gcol;
            if (full) {
                *reinterpret_cast<float4*>(cp) = *reinterpret_cast<const float4*>(v);
                *reinterpret_cast<float4*>(cp + 4) = *reinterpret_cast<const float4*>(v + 4);
            } else {
                for (int e = 0; e < 8; ++e)
                    if (gcol + e < N) cp[e] = v[e];
            }
        }
    }
}

}  // namespace

extern "C" int tst_gemm_mfma_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                  const float* res, int64_t ldr, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                  int act, int out_bf16, void* stream) {
    TST_REQUIRE(A && W && C, TST_ERR_ARG, "tst_gemm_mfma_bf16: null pointer");
    TST_REQUIRE(M >= 0 && N > 0 && K >= 8 && (act == 0 || act == 1), TST_ERR_ARG, "tst_gemm_mfma_bf16: bad sizes M=%lld N=%lld K=%lld",
                (long long)M, (long long)N, (long long)K);
    TST_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldc % (out_bf16 ? 8 : 4) == 0 && lda >= K && ldw >= K && ldc >= N,
                TST_ERR_ARG, "tst_gemm_mfma_bf16: K, lda, ldw must be multiples of 8, ldc of 8 (bf16 out) or 4, lda >= K, ldw >= K, ldc >= N");
    TST_REQUIRE(tst_aligned16(A) && tst_aligned16(W) && tst_aligned16(C), TST_ERR_ARG, "tst_gemm_mfma_bf16: A, W, C must be 16-byte aligned");
    TST_REQUIRE(res == nullptr || (ldr >= N && ldr % 4 == 0 && tst_aligned16(res)), TST_ERR_ARG,
                "tst_gemm_mfma_bf16: res needs ldr >= N, ldr %% 4 == 0 and 16-byte alignment");
    TST_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), TST_ERR_UNSUPPORTED, "tst_gemm_mfma_bf16: size");
    if (M == 0) return TST_OK;
    const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    TST_REQUIRE(tiles_m * tiles_n < (1ll << 31), TST_ERR_UNSUPPORTED, "tst_gemm_mfma_bf16: grid too large");
    const dim3 grid((unsigned)(tiles_m * tiles_n));
    hipStream_t st = (hipStream_t)stream;
#define TST_BF16_LAUNCH(OB, HR, RL)                                                                                 \
    hipLaunchKernelGGL((gemm_bf16_kernel<OB, HR, RL>), grid, dim3(256), 0, st, (const __bf16*)A, lda, (const __bf16*)W, \
                       ldw, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, (int)tiles_m, (int)tiles_n)
    const int variant = (out_bf16 ? 4 : 0) | (res != nullptr ? 2 : 0) | (act == 1 ? 1 : 0);
    switch (variant) {
        case 0: TST_BF16_LAUNCH(false, false, false); break;
        case 1: TST_BF16_LAUNCH(false, false, true); break;
        case 2: TST_BF16_LAUNCH(false, true, false); break;
        case 3: TST_BF16_LAUNCH(false, true, true); break;
        case 4: TST_BF16_LAUNCH(true, false, false); break;
        case 5: TST_BF16_LAUNCH(true, false, true); break;
        case 6: TST_BF16_LAUNCH(true, true, false); break;
        default: TST_BF16_LAUNCH(true, true, true); break;
    }
#undef TST_BF16_LAUNCH
    return tst_after_launch("tst_gemm_mfma_bf16");
}
