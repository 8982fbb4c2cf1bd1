// A row tile's chain through a pre-norm transformer layer as ONE kernel (fp32, d_model = 256):
//
//   x[m][:] = x[m][:] + (A[m][:] * W1 + b1)            the output projection of an attention block, or the second
//                                                      feed-forward product, added to the residual stream
//   y[m][:] = LayerNorm(x[m][:])                       the LayerNorm in front of the next block
//   C[m][:] = act(y[m][:] * W2 + b2)                   that block's first product (cross-q, ff1, the next layer's q|k|v),
//                                                      or, after the last layer, y is stored k-major for the vocabulary GEMM
//
// A workgroup owns 64 whole rows, so the LayerNorm statistics stay inside it and neither x nor y makes a round trip
// through memory between the three operations (three launches and two passes over [rows][256] otherwise).
// Every element is produced by the operation sequence of the separate entry points (tst_gemm_mfma_f32 with a residual,
// tst_layernorm_f32, tst_gemm_mfma_f32): the same k-ordered fma chain from the bias on v_mfma_f32_32x32x2_f32, the same
// 64 interleaved partial sums and xor-butterfly order in the statistics, so results are bit-identical.
//
// Data flow: the A operand of a product is a 64 x 256 panel held k-major in LDS (Ak[k][m]) and shared by the 8 waves; the
// first product's panels are transposed into it through registers (the next panel is requested while the current one is
// multiplied), the second product's panel is written there by the LayerNorm phase.  A wave owns 64 rows x 32 columns of a
// 256-column block, and the weights of ITS 32 columns stream through a ring of its own: NSW stages of 8 k x 32 columns
// (1 KiB) filled by LDS-DMA (global_load_lds), which the wave itself waits for with counted vmcnt: the k loops contain NO
// workgroup barrier, the waves run decoupled and cover each other's waits.  The first product is computed transposed
// (operands swapped, same fma chain), so that a lane owns two rows and holds 16 elements of each: residual add, statistics
// and normalisation happen in the accumulator registers, and the partial sums of the eight waves meet through LDS.  All
// LDS accesses and the panel loads are inline assembly: the compiler would order every LDS access and every wait for a
// load it knows of behind ALL pending LDS-DMA.
#include "tst_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

namespace {

constexpr int D = 256, BM = 64, LDA = 66, KS = 8, NSW = 7;
constexpr int AK_B = D * LDA * 4;                   // 66 KiB: the k-major A panel (rows padded to 66: two-deep bank spread
                                                    // for the transposing writes)
constexpr int RING_OFF = AK_B, WRING_B = NSW * 1024;
constexpr int P_OFF = RING_OFF + 8 * WRING_B, P_B = 8 * 8 * 64 * 4;
constexpr int SMEM_B = P_OFF + 2 * P_B;             // 154 KiB
constexpr int NSTORE = 32;                          // global stores of a lane per 256-column chunk of C
constexpr int NPRE = 8;                             // global loads of a lane per A panel
static_assert(RING_OFF % 1024 == 0, "the DMA stages are 1 KiB aligned");
static_assert(NSW - 2 + NSTORE < 64, "vmcnt is a 6-bit counter");
static_assert(SMEM_B <= 160 * 1024, "LDS of a CU");

__device__ __forceinline__ void lds_w32(uint32_t addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// measurement builds (tools/probe/make_chain_ablations.py): CHAIN_ABL bit 0 = no counted DMA waits in the k loops, bit 1 = no
// DMA in the k loops, bit 2 = no operand reads in the k loops, bit 3 = no LDS waits in the k loops.  Results are wrong then.
#ifndef CHAIN_ABL
#define CHAIN_ABL 0
#endif
__device__ __forceinline__ void loop_lds_wait() {
    if (!(CHAIN_ABL & 8)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int OFF>
__device__ __forceinline__ void gload16(f32x4& v, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "n"(OFF) : "memory");
}

// NCH: 256-column chunks of the second product (0: none, the normalised rows go out k-major instead).
template <int NCH, bool RELU2>
__global__ __launch_bounds__(512) void chain_kernel(const float* __restrict__ A, int64_t lda, int K1, const float* __restrict__ W1,
                                                    int64_t ldw1, const float* __restrict__ b1, float* x,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                    const float* __restrict__ W2, int64_t ldw2, const float* __restrict__ b2,
                                                    float* __restrict__ C, int64_t ldc, float* __restrict__ yT, int64_t ldt, int M) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_B];
    const int t = threadIdx.x, lane = t & 63, l = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m0 = blockIdx.x * BM;
    const bool full = m0 + BM <= M;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // ---- panel staging: thread -> row (t >> 3), 16-byte chunks (t & 7) + 8 i of the row's 256 k (8 lanes = one 128-byte line)
    const int prow = t >> 3, pc8 = t & 7;
    const float* a_row = A + (int64_t)min(m0 + prow, M - 1) * lda + 4 * pc8;
    f32x4 pa[8];
    auto load_panel = [&](int k0) {
        const float* p = a_row + k0;
        gload16<0>(pa[0], p); gload16<128>(pa[1], p); gload16<256>(pa[2], p); gload16<384>(pa[3], p);
        gload16<512>(pa[4], p); gload16<640>(pa[5], p); gload16<768>(pa[6], p); gload16<896>(pa[7], p);
    };
    auto write_panel = [&]() {
        const uint32_t base = lds_base + ((4 * pc8) * LDA + prow) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t ad = base + i * (32 * LDA * 4);
            lds_w32(ad, pa[i][0]);
            lds_w32(ad + LDA * 4, pa[i][1]);
            lds_w32(ad + 2 * LDA * 4, pa[i][2]);
            lds_w32(ad + 3 * LDA * 4, pa[i][3]);
        }
    };

    // ---- the wave's weight ring: a stage = 8 weight rows (k) x the wave's 32 columns; lane -> k = lane >> 3, 16 bytes (lane & 7)
    auto issue = [&](const float* src, int slot) {
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(smem + RING_OFF + w * WRING_B + slot * 1024), 16, 0, 0);
    };

    // ---- operands of MFMA step s (k = 2 s + h) of a stage: the two 32-row blocks of the panel, the wave's 32 columns
    const uint32_t a_off = lds_base + (h * LDA + l) * 4;
    const uint32_t b_off = lds_base + RING_OFF + w * WRING_B + (h * 32 + l) * 4;
    float fa[2][2], fb[2];
    bool in_loop = false;   // (measurement builds)
    auto read_step = [&](int set, uint32_t as, uint32_t bs) {
        if ((CHAIN_ABL & 4) && in_loop) return;
        asm volatile("ds_read_b32 %0, %1" : "=v"(fa[set][0]) : "v"(as));
        asm volatile("ds_read_b32 %0, %1" : "=v"(fa[set][1]) : "v"(as + 128));
        asm volatile("ds_read_b32 %0, %1" : "=v"(fb[set]) : "v"(bs));
    };

    f32x16 acc[2];

    // ====================== first product, transposed: lane owns rows m0 + l and m0 + 32 + l ======================
    // acc[i][4 g + e] = element (row m0 + 32 i + l, column cb + 8 g + e)
    const int cb = w * 32 + 4 * h;
    load_panel(0);
    f32x4 b4[4];
    if (b1 != nullptr) {
        const float* p = b1 + cb;
        gload16<0>(b4[0], p); gload16<32>(b4[1], p); gload16<64>(b4[2], p); gload16<96>(b4[3], p);
    }
    const int G1 = K1 / KS;                          // >= 32
    const float* wsrc1 = W1 + (int64_t)(lane >> 3) * ldw1 + w * 32 + 4 * (lane & 7);
#pragma unroll
    for (int p = 0; p < NSW - 1; ++p) issue(wsrc1 + (int64_t)(p * KS) * ldw1, p);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSW - 1) : "memory");    // the panel and the bias (older than the NSW - 1 stages) are in registers
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float b = b1 != nullptr ? b4[g][e] : 0.0f;
            acc[0][4 * g + e] = b;
            acc[1][4 * g + e] = b;
        }
    write_panel();
    lds_wait();
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSW - 2) : "memory");    // stage 0 landed
    {
        int st = 0;
        read_step(0, a_off, b_off);
        if (CHAIN_ABL & 4) read_step(1, a_off, b_off);
        in_loop = true;
        for (int g = 0; g < G1; ++g) {
            const int kp = g & 31;
            const int st_next = st + 1 == NSW ? 0 : st + 1;
            if (kp == 0 && g > 0) {
                // next panel: every wave is past its last read of the old one; the new one was requested 32 stages ago
                __builtin_amdgcn_s_barrier();
                write_panel();
                lds_wait();
                __builtin_amdgcn_s_barrier();
                read_step(0, a_off, b_off + st * 1024);
            }
            const bool pre = g + 32 < G1;            // (kp == 0): this panel is followed by another one
            if (kp == 0 && pre) load_panel((g + 32) * KS);
            const uint32_t as = a_off + kp * (KS * LDA * 4), bs = b_off + st * 1024;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                loop_lds_wait();
                if (s < 3) read_step((s + 1) & 1, as + (s + 1) * (2 * LDA * 4), bs + (s + 1) * 256);
                else if (kp != 31) read_step(0, as + KS * LDA * 4, b_off + st_next * 1024);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[s & 1], fa[s & 1][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[s & 1], fa[s & 1][1], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);   // the MFMAs stay in front of the next step's wait
                if (s == 1) {
                    // stage g + NSW - 1 goes into the slot of stage g - 1; then stage g + 1 must have landed: NSW - 2 younger
                    // stages may stay in flight, and the panel loads while they are younger than stage g + 1
                    if (g + NSW - 1 < G1) {
                        if (!(CHAIN_ABL & 2)) issue(wsrc1 + (int64_t)((g + NSW - 1) * KS) * ldw1, st == 0 ? NSW - 1 : st - 1);
                        if (CHAIN_ABL & 1) {}
                        else if (pre && kp < NSW - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSW - 2 + NPRE) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSW - 2) : "memory");
                    } else if (!(CHAIN_ABL & 1)) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            st = st_next;
        }
    }

    // the wave's ring is its own: the second product's first stages are requested now and land during the LayerNorm phase
    constexpr int G2 = NCH * 32;
    const float* wsrc2 = NCH > 0 ? W2 + (int64_t)(lane >> 3) * ldw2 + w * 32 + 4 * (lane & 7) : nullptr;
    if (NCH > 0) {
#pragma unroll
        for (int p = 0; p < NSW - 1; ++p) issue(wsrc2 + (int64_t)(p * KS) * ldw2, p);
    }

    // ---- residual add, x out, LayerNorm statistics in the order of layernorm_kernel
    float y[2][16];
    float4 gg[4], bb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        gg[g] = *reinterpret_cast<const float4*>(gamma + cb + 8 * g);
        bb[g] = *reinterpret_cast<const float4*>(beta + cb + 8 * g);
    }
    // bias of every chunk of the second product: there a lane owns column (chunk * 256 + w * 32 + l)
    float b2v[NCH > 0 ? NCH : 1];
#pragma unroll
    for (int c = 0; c < NCH; ++c) b2v[c] = b2 != nullptr ? b2[c * 256 + w * 32 + l] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + 32 * i + l;
        float* xr = x + (int64_t)min(row, M - 1) * D + cb;
        float4 r4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) r4[g] = *reinterpret_cast<const float4*>(xr + 8 * g);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            y[i][4 * g] = acc[i][4 * g] + r4[g].x;
            y[i][4 * g + 1] = acc[i][4 * g + 1] + r4[g].y;
            y[i][4 * g + 2] = acc[i][4 * g + 2] + r4[g].z;
            y[i][4 * g + 3] = acc[i][4 * g + 3] + r4[g].w;
        }
        if (row < M) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(xr + 8 * g) = make_float4(y[i][4 * g], y[i][4 * g + 1], y[i][4 * g + 2], y[i][4 * g + 3]);
        }
    }
    // A row's 64 partial sums (one per 16-byte chunk, chunk p = w * 8 + 2 g + h) are combined as the xor-butterfly
    // 32, 16, 8, 4, 2, 1 does: levels 32, 16 and 8 pair the waves (through LDS), 4 and 2 pair g, 1 pairs the wave halves.
    // P[pass][wave][4 i + g][lane]
    auto row_totals = [&](const float (&q)[8], int which, float (&tot)[2]) {
        const uint32_t pb = lds_base + P_OFF + which * P_B + lane * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) lds_w32(pb + (w * 8 + j) * 256, q[j]);
        lds_wait();
        __builtin_amdgcn_s_barrier();
        float o[8][8];                               // o[k][j]: the partial of wave w ^ k
#pragma unroll
        for (int k = 1; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_read_b32 %0, %1" : "=v"(o[k][j]) : "v"(pb + ((w ^ k) * 8 + j) * 256));
        lds_wait();
        float s3[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float s2a = (q[j] + o[4][j]) + (o[2][j] + o[6][j]);        // levels 32 (w ^ 4) and 16 (w ^ 2)
            const float s2b = (o[1][j] + o[5][j]) + (o[3][j] + o[7][j]);     // the same for wave w ^ 1
            s3[j] = s2a + s2b;                                               // level 8 (w ^ 1)
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float s4a = s3[4 * i] + s3[4 * i + 2], s4b = s3[4 * i + 1] + s3[4 * i + 3];   // level 4: g ^ 2
            float v = s4a + s4b;                                                                // level 2: g ^ 1
            v = v + lane_xor<32>(v);                                                            // level 1: h
            tot[i] = v;
        }
    };
    float q[8], tot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float p = 0.0f;
            p = p + y[i][4 * g]; p = p + y[i][4 * g + 1]; p = p + y[i][4 * g + 2]; p = p + y[i][4 * g + 3];
            q[4 * i + g] = p;
        }
    row_totals(q, 0, tot);
    const float mean[2] = {tot[0] / 256.0f, tot[1] / 256.0f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float p = 0.0f, d;
            d = y[i][4 * g] - mean[i]; p = __builtin_fmaf(d, d, p);
            d = y[i][4 * g + 1] - mean[i]; p = __builtin_fmaf(d, d, p);
            d = y[i][4 * g + 2] - mean[i]; p = __builtin_fmaf(d, d, p);
            d = y[i][4 * g + 3] - mean[i]; p = __builtin_fmaf(d, d, p);
            q[4 * i + g] = p;
        }
    row_totals(q, 1, tot);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float var = tot[i] / 256.0f;
        const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            y[i][4 * g] = __builtin_fmaf((y[i][4 * g] - mean[i]) * rstd, gg[g].x, bb[g].x);
            y[i][4 * g + 1] = __builtin_fmaf((y[i][4 * g + 1] - mean[i]) * rstd, gg[g].y, bb[g].y);
            y[i][4 * g + 2] = __builtin_fmaf((y[i][4 * g + 2] - mean[i]) * rstd, gg[g].z, bb[g].z);
            y[i][4 * g + 3] = __builtin_fmaf((y[i][4 * g + 3] - mean[i]) * rstd, gg[g].w, bb[g].w);
        }
    }

    if (NCH == 0) {
        // the normalised rows k-major: yT[column][row], 32 lanes = 128 contiguous bytes
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + 32 * i + l;
            if (row < M) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yT[(int64_t)(cb + 8 * (r >> 2) + (r & 3)) * ldt + row] = y[i][r];
            }
        }
        return;
    }

    // ================================ second product: the normalised rows are its A panel ================================
    // (every wave is past the first product's panel: the barriers of the statistics)
    {
        const uint32_t base = lds_base + (cb * LDA + l) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) lds_w32(base + (8 * (r >> 2) + (r & 3)) * (LDA * 4) + i * 128, y[i][r]);
    }
    lds_wait();
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSW - 2) : "memory");   // stage 0 landed (everything issued after the stages is younger: stricter)
    __builtin_amdgcn_sched_barrier(0);
    {
        int st = 0;
        in_loop = false;
        read_step(0, a_off, b_off);
        if (CHAIN_ABL & 4) read_step(1, a_off, b_off);
        in_loop = true;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = b2v[c];
            for (int kp = 0; kp < 32; ++kp) {
                const int g = c * 32 + kp;
                const int st_next = st + 1 == NSW ? 0 : st + 1;
                const uint32_t as = a_off + kp * (KS * LDA * 4), bs = b_off + st * 1024;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    loop_lds_wait();
                    if (s < 3) read_step((s + 1) & 1, as + (s + 1) * (2 * LDA * 4), bs + (s + 1) * 256);
                    else if (g + 1 < G2) read_step(0, kp == 31 ? a_off : as + KS * LDA * 4, b_off + st_next * 1024);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][0], fb[s & 1], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][1], fb[s & 1], acc[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s == 1) {
                        // as in the first product; the stores of the previous chunk's epilogue may stay in flight while they
                        // are younger than stage g + 1 (a partial tile issues fewer of them: strict count)
                        if (g + NSW - 1 < G2) {
                            const int gn = g + NSW - 1;
                            if (!(CHAIN_ABL & 2)) issue(wsrc2 + (int64_t)((gn & 31) * KS) * ldw2 + (gn >> 5) * 256, st == 0 ? NSW - 1 : st - 1);
                            if (CHAIN_ABL & 1) {}
                            else if (c > 0 && full && kp < NSW - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSW - 2 + NSTORE) : "memory");
                            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSW - 2) : "memory");
                        } else if (!(CHAIN_ABL & 1)) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                st = st_next;
            }
            // epilogue of chunk c: lane owns column col of rows r0 + (r & 3) + 8 (r >> 2) of each 32-row block
            const int col = c * 256 + w * 32 + l;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r0 = m0 + 32 * i + 4 * h;
                float* cp = C + (int64_t)r0 * ldc + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][r];
                    if (RELU2) v = v > 0.0f ? v : 0.0f;
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (full || r0 + ro < M) cp[(int64_t)ro * ldc] = v;
                }
            }
        }
    }
}

}  // namespace

extern "C" int tst_chain_ok(int64_t R, int64_t d, int64_t K1, int64_t N2) {
    return (R > 0 && R < (1ll << 31) - BM && d == D && K1 >= D && K1 % D == 0 && K1 < (1ll << 24) && (N2 == 0 || N2 == 256 || N2 == 768 || N2 == 1024))
               ? 1
               : 0;
}

extern "C" int tst_chain_f32(const float* A, int64_t lda, int64_t K1, const float* W1t, int64_t ldw1, const float* b1, float* x,
                             const float* gamma, const float* beta, float eps, const float* W2t, int64_t ldw2, const float* b2, int act2,
                             float* C, int64_t ldc, int64_t N2, float* yT, int64_t ldt, int64_t R, int64_t d, void* stream) {
    TST_REQUIRE(tst_dims_ok({K1, N2, R, d}), TST_ERR_UNSUPPORTED, "tst_chain_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(A && W1t && x && gamma && beta, TST_ERR_ARG, "tst_chain_f32: null pointer");
    TST_REQUIRE(R >= 0 && (act2 == 0 || act2 == 1), TST_ERR_ARG, "tst_chain_f32: bad sizes");
    if (R == 0) return TST_OK;
    TST_REQUIRE(tst_chain_ok(R, d, K1, N2), TST_ERR_UNSUPPORTED,
                "tst_chain_f32: needs d == 256, K1 a multiple of 256 and N2 in {0, 256, 768, 1024} (got d=%lld K1=%lld N2=%lld)", (long long)d,
                (long long)K1, (long long)N2);
    TST_REQUIRE(lda >= K1 && lda % 4 == 0 && ldw1 >= D && ldw1 % 4 == 0, TST_ERR_ARG, "tst_chain_f32: lda >= K1, ldw1 >= 256, both multiples of 4");
    TST_REQUIRE(tst_aligned16(A) && tst_aligned16(W1t) && tst_aligned16(x) && tst_aligned16(gamma) && tst_aligned16(beta) && tst_aligned16(b1),
                TST_ERR_ARG, "tst_chain_f32: A, W1t, b1, x, gamma, beta must be 16-byte aligned");
    if (N2 > 0) {
        TST_REQUIRE(W2t && C, TST_ERR_ARG, "tst_chain_f32: null W2t or C");
        TST_REQUIRE(ldw2 >= N2 && ldw2 % 4 == 0 && ldc >= N2 && tst_aligned16(W2t), TST_ERR_ARG,
                    "tst_chain_f32: ldw2 >= N2, ldw2 %% 4 == 0, ldc >= N2, W2t 16-byte aligned");
    } else {
        TST_REQUIRE(yT != nullptr && ldt >= R, TST_ERR_ARG, "tst_chain_f32: without a second product yT is required, ldt >= R");
    }
    const dim3 grid((unsigned)((R + BM - 1) / BM)), block(512);
    hipStream_t st = (hipStream_t)stream;
#define TST_CHAIN_LAUNCH(NCH, RL)                                                                                                        \
    hipLaunchKernelGGL((chain_kernel<NCH, RL>), grid, block, 0, st, A, lda, (int)K1, W1t, ldw1, b1, x, gamma, beta, eps, W2t, ldw2, b2, C, ldc, \
                       yT, ldt, (int)R)
    if (N2 == 0) TST_CHAIN_LAUNCH(0, false);
    else if (N2 == 256) { if (act2) TST_CHAIN_LAUNCH(1, true); else TST_CHAIN_LAUNCH(1, false); }
    else if (N2 == 768) { if (act2) TST_CHAIN_LAUNCH(3, true); else TST_CHAIN_LAUNCH(3, false); }
    else { if (act2) TST_CHAIN_LAUNCH(4, true); else TST_CHAIN_LAUNCH(4, false); }
#undef TST_CHAIN_LAUNCH
    return tst_after_launch("tst_chain_f32");
}
