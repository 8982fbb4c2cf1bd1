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

// ---------------------------------------------------------------------------------------------------------
// Fast path, K % 32 == 0: tiles of 32 k staged by LDS-DMA (global_load_lds, 16 bytes per lane), no staging
// registers and no LDS store instructions.  One wave-instruction fills 16 tile rows of 64 bytes in lane order,
// so the LDS image is plain [row][4 slots of 16 bytes]; bank conflicts of the fragment reads are avoided by a
// permutation on the SOURCE side: slot s of row r holds k-chunk s ^ ((r >> 2) & 3), and the ds_read_b128 of
// chunk c of row r goes to slot c ^ ((r >> 2) & 3) (the 16 lanes of a read group then touch 16 different
// 16-byte bank slots).  NST stages in a ring, NST - 1 tiles in flight: the DMA of tile t + NST - 1 is issued
// when tile t starts, a counted wait (all but the NST - 2 youngest tiles) plus one barrier per tile publishes
// tile t + 1.  The fragment reads are inline assembly with explicit waits: the compiler orders every LDS read
// it knows of behind ALL pending LDS-DMA, which would stall each tile on the DMA of the following ones.
// The product is computed transposed, D = W_tile * A_tile^T: a lane then owns ONE output row (m = lane & 31)
// and its 16 accumulator registers are 4 groups of 4 consecutive columns: bias and residual are 16-byte
// accesses, fp32 results 16-byte stores; bf16 results are packed to 8 bytes per group and pass through a
// wave-private LDS image so that 8 lanes store one 128-byte row segment.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int DK = 32;   // k per tile of the DMA kernel

// LayerNorm of the rows the tile has just produced, for the whole-row tile (128 x 512) of a residual product: see LN below
struct LnOut {
    const float* gamma = nullptr;
    const float* beta = nullptr;
    float eps = 0.0f;
    __bf16* h = nullptr;
    int64_t ldh = 0;
};

template <int WVM, int WVN, int WTM, int WTN, int NST, bool OUT_BF16, bool HAS_RES, bool RELU, bool LN = false>
__global__ __launch_bounds__(64 * WVM * WVN) void gemm_bf16_dma_kernel(const __bf16* __restrict__ A, int64_t lda,
                                                            const __bf16* __restrict__ W, int64_t ldw,
                                                            const float* __restrict__ bias, const float* res, int64_t ldr,
                                                            void* Cout, int64_t ldc, int M, int N, int K, int tiles_m,
                                                            int tiles_n, const LnOut ln = LnOut{}) {
    // workgroup tile: WVM x WVN waves, each with WTM x WTN MFMA tiles of 32 x 32
    constexpr int NWV = WVM * WVN, TM = 32 * WTM * WVM, TN = 32 * WTN * WVN;
    constexpr int STAGE = (TM + TN) * 64;                // bytes of one stage: A rows then W rows, 64 bytes each
    constexpr int OUT_LD = 16 + 64 * WTN;                // bytes per row of a wave's bf16 output image (+16: bank spread)
    constexpr int IMG = 64 * OUT_LD;                     // the image holds 64 rows (two 32-row blocks) at a time
    constexpr int SMEM = NST * STAGE > NWV * IMG ? NST * STAGE : NWV * IMG;
    static_assert(TM % (16 * NWV) == 0 && TN % (16 * NWV) == 0 && WTM % 2 == 0, "tile shape");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
    const int lb = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    // m-group of the tile order: the 32 tiles an XCD works on at a time are GM x (32 / GM); with 256-row tiles 8 x 4 keeps
    // both operand panels (2 MB + 1 MB at K = 512) in the XCD's 4 MB L2 next to the lines being written
    constexpr int GM = TM >= 256 ? 8 : GROUP_M;
    const int per_group = GM * tiles_n;
    const int first_m = (lb / per_group) * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (lb % per_group) % gsz;
    const int tn = (lb % per_group) / gsz;
    const int m0 = tm * TM, n0 = tn * TN;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WVN, wn = wave % WVN;

    // DMA sources: wave w fills the 16-row blocks w, w + NWV, ... of the A rows and of the W rows; lane l of a
    // block: row l >> 2, LDS slot l & 3, which takes k-chunk (l & 3) ^ ((row >> 2) & 3).  Rows beyond the matrix
    // are clamped (their products are never stored).
    constexpr int NA = TM / (16 * NWV), NW = TN / (16 * NWV);   // wave-instructions per wave and tile
    const int srow = lane >> 2;
    const __bf16* a_src[NA];
    const __bf16* w_src[NW];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int row = (wave + NWV * j) * 16 + srow;
        a_src[j] = A + (int64_t)min(m0 + row, M - 1) * lda + (((lane & 3) ^ ((row >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int row = (wave + NWV * j) * 16 + srow;
        w_src[j] = W + (int64_t)min(n0 + row, N - 1) * ldw + (((lane & 3) ^ ((row >> 2) & 3)) * 8);
    }
    auto stage = [&](int st, int k0) {
        unsigned char* base = smem + st * STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(a_src[j] + k0), (lds_void_t*)(base + j * (1024 * NWV)), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < NW; ++j)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(w_src[j] + k0), (lds_void_t*)(base + TM * 64 + j * (1024 * NWV)), 16, 0, 0);
    };

    // accumulators, transposed: acc[i][j][4 g + e] = C[m = 32 i + (lane & 31)][n = 32 j + 8 g + 4 (lane >> 5) + e]
    f32x16 acc[WTM][WTN];
    const int h = lane >> 5;
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        const int nb = n0 + wn * 32 * WTN + j * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias != nullptr) {
                if (nb + 8 * g + 3 < N) {
                    b4 = *reinterpret_cast<const float4*>(bias + nb + 8 * g);
                } else {
                    b4.x = bias[min(nb + 8 * g, N - 1)]; b4.y = bias[min(nb + 8 * g + 1, N - 1)];
                    b4.z = bias[min(nb + 8 * g + 2, N - 1)]; b4.w = bias[min(nb + 8 * g + 3, N - 1)];
                }
            }
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                acc[i][j][4 * g] = b4.x; acc[i][j][4 * g + 1] = b4.y; acc[i][j][4 * g + 2] = b4.z; acc[i][j][4 * g + 3] = b4.w;
            }
        }
    }

    const int nkt = K / DK;
    // fragment read: row (lane & 31) of a 32-row block, chunk 2 ks + h -> slot (2 ks + h) ^ x, x = ((lane & 31) >> 2) & 3
    const int x = ((lane & 31) >> 2) & 3;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t a_off = lds_base + wm * 32 * WTM * 64 + (lane & 31) * 64;
    const uint32_t w_off = lds_base + TM * 64 + wn * 32 * WTN * 64 + (lane & 31) * 64;
    union Frag { u32x4 u; bf16x8 v; };
    Frag fa[2][WTM], fw[2][WTN];   // fragments of the two k-steps of a tile
    auto read_step = [&](int st, int ks) {
        const uint32_t as = a_off + st * STAGE, ws = w_off + st * STAGE;
        const uint32_t slot = ((2 * ks + h) ^ x) * 16;
#pragma unroll
        for (int i = 0; i < WTM; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[ks][i].u) : "v"(as + i * 2048 + slot));
#pragma unroll
        for (int j = 0; j < WTN; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(fw[ks][j].u) : "v"(ws + j * 2048 + slot));
    };
    auto mfma_step = [&](int ks) {
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ks][j].v, fa[ks][i].v, acc[i][j], 0, 0, 0);
    };
    auto lds_wait = [&]() {   // the scheduling barrier keeps the register-only MFMAs below the wait
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // An LDS-DMA is complete for OTHER waves only after the issuing wave has waited for it (vector-memory counter)
    // and both have passed a barrier.  NDMA instructions per wave and tile; tiles complete in issue order.
    // Schedule of tile kt (stage kt % NST):  wait for the k-step-0 fragments (read during the previous tile) | issue
    // the k-step-1 reads | MFMAs of k-step 0 | wait: tile kt + 1 landed, own reads done | BARRIER (the MFMAs just
    // issued keep the matrix pipe busy) | stage kt % NST is free: DMA of tile kt + NST into it | issue the k-step-0
    // reads of tile kt + 1 | MFMAs of k-step 1.
    constexpr int NDMA = NA + NW;
#pragma unroll
    for (int p = 0; p < NST; ++p)
        if (p < nkt) stage(p, p * DK);
    if (nkt >= NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NST - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_step(0, 0);
    int st = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const int st_next = st + 1 == NST ? 0 : st + 1;
        lds_wait();
        read_step(st, 1);
        mfma_step(0);
        if (kt + NST - 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NST - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_wait();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NST < nkt) stage(st, (kt + NST) * DK);
        if (kt + 1 < nkt) read_step(st_next, 0);
        mfma_step(1);
        st = st_next;
    }
    __builtin_amdgcn_s_barrier();   // the epilogue reuses the stages: every wave must be past its last fragment read

    // epilogue: per 32 x 32 block and register group g the lane has 4 consecutive columns of its row;
    // two 32-row blocks (64 rows) at a time
#pragma unroll
    for (int i0 = 0; i0 < WTM; i0 += 2) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = i0 + ii;
            const int m = m0 + wm * 32 * WTM + i * 32 + (lane & 31);
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                const int nb = n0 + wn * 32 * WTN + j * 32 + 4 * h;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[r] = acc[i][j][r];
                    if (RELU) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
                }
                const bool row_ok = m < M;
                const bool full = n0 + wn * 32 * WTN + j * 32 + 32 <= N;   // wave-uniform
                if (HAS_RES && row_ok) {
                    const float* rp = res + (int64_t)m * ldr + nb;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (full) {
                            const float4 r4 = *reinterpret_cast<const float4*>(rp + 8 * g);
                            v[4 * g] += r4.x; v[4 * g + 1] += r4.y; v[4 * g + 2] += r4.z; v[4 * g + 3] += r4.w;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (nb + 8 * g + e < N) v[4 * g + e] += rp[8 * g + e];
                        }
                    }
                }
                if (LN) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = v[r];
                }
                if (OUT_BF16) {
                    // pack each group of 4 columns to 8 bytes and put it into the wave's LDS image [64 rows][32 WTN columns]
                    unsigned char* img = smem + wave * IMG + (ii * 32 + (lane & 31)) * OUT_LD + (j * 32 + 4 * h) * 2;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                        union { bf16x4 b; uint2 u; } cv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) cv.b[e] = (__bf16)v[4 * g + e];
                        *reinterpret_cast<uint2*>(img + 16 * g) = cv.u;
                    }
                } else if (row_ok) {
                    float* cp = reinterpret_cast<float*>(Cout) + (int64_t)m * ldc + nb;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (full) {
                            *reinterpret_cast<float4*>(cp + 8 * g) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (nb + 8 * g + e < N) cp[8 * g + e] = v[4 * g + e];
                        }
                    }
                }
            }
        }
        if (OUT_BF16) {
            // the wave reads its image back by rows: LPR lanes x 16 bytes = the 32 WTN columns of a row
            // (a wave only reads what it wrote itself: the compiler's LDS counter waits are all that is needed)
            constexpr int LPR = 4 * WTN, RPP = 64 / LPR;    // lanes per row, rows per pass
            const unsigned char* img = smem + wave * IMG;
            const int c8 = (lane % LPR) * 8;
            const int gcol = n0 + wn * 32 * WTN + c8;
#pragma unroll
            for (int p = 0; p < 64 / RPP; ++p) {
                const int lrow = p * RPP + lane / LPR;
                const int grow = m0 + wm * 32 * WTM + i0 * 32 + lrow;
                if (grow >= M || gcol >= N) continue;
                const uint4 o = *reinterpret_cast<const uint4*>(img + lrow * OUT_LD + c8 * 2);
                __bf16* cp = reinterpret_cast<__bf16*>(Cout) + (int64_t)grow * ldc + gcol;
                if (gcol + 8 <= N) {
                    // nontemporal: a wide bf16 result (the logits) is read again only after it has left the caches, and
                    // written back through the L2 it pushes the operand panels out (measured +20 % on the projection)
                    u32x4 ov;
                    ov[0] = o.x; ov[1] = o.y; ov[2] = o.z; ov[3] = o.w;
                    __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(cp));
                } else {
                    Chunk oc;
                    oc.u = o;
                    for (int e = 0; e < 8; ++e)
                        if (gcol + e < N) cp[e] = oc.v[e];
                }
            }
        }
    }
    // LN: the tile holds whole rows of N = 512 (x = acc + res is in the accumulators): h = LayerNorm(x) in bf16, with the
    // statistics formed in exactly the order of layernorm_kernel, whose lane p owns the 16-byte chunks p and p + 64 of a row and
    // sums chunk p's four elements, then chunk p + 64's, before the xor-butterfly 32 .. 1 over the 64 partial sums.  Here a lane
    // owns rows (i) and chunk c = wn * 32 + j * 8 + 2 g + h: waves wn = 0, 1 start the partial sums (PA), waves 2, 3 continue
    // them with their chunks (PB), then every wave combines the two halves (level 32), j (16, 8), g (4, 2) and the wave halves (1).
    if constexpr (LN) {
        static_assert(WVM == 2 && WVN == 4 && WTM == 2 && WTN == 4 && HAS_RES && !OUT_BF16 && !RELU, "LN: the 128 x 512 residual tile");
        float* PA = reinterpret_cast<float*>(smem);                     // [wm][wn & 1][i][j * 4 + g][lane]
        float* PB = PA + 2 * 2 * 2 * 16 * 64;
        const int pmine = ((wm * 2 + (wn & 1)) * 2) * 16 * 64 + lane;
        float mean[2] = {0.0f, 0.0f}, rstd[2] = {0.0f, 0.0f};
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            float q[2][16];
            auto extend = [&]() {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float p = q[i][j * 4 + g];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (pass == 0) {
                                    p = p + acc[i][j][4 * g + e];
                                } else {
                                    const float t = acc[i][j][4 * g + e] - mean[i];
                                    p = __builtin_fmaf(t, t, p);
                                }
                            }
                            q[i][j * 4 + g] = p;
                        }
            };
            if (wn < 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int c = 0; c < 16; ++c) q[i][c] = 0.0f;
                extend();
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int c = 0; c < 16; ++c) PA[pmine + (i * 16 + c) * 64] = q[i][c];
            }
            __syncthreads();
            if (wn >= 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int c = 0; c < 16; ++c) q[i][c] = PA[pmine + (i * 16 + c) * 64];
                extend();
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int c = 0; c < 16; ++c) PB[pmine + (i * 16 + c) * 64] = q[i][c];
            }
            __syncthreads();
            const int p0 = ((wm * 2 + 0) * 2) * 16 * 64 + lane, p1 = ((wm * 2 + 1) * 2) * 16 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float s1[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) s1[c] = PB[p0 + (i * 16 + c) * 64] + PB[p1 + (i * 16 + c) * 64];   // level 32
                float s2[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) s2[c] = s1[c] + s1[c + 8];                                            // level 16: j ^ 2
                float s3[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) s3[g] = s2[g] + s2[4 + g];                                            // level 8: j ^ 1
                const float s4a = s3[0] + s3[2], s4b = s3[1] + s3[3];                                             // level 4: g ^ 2
                float v = s4a + s4b;                                                                              // level 2: g ^ 1
                v = v + lane_xor<32>(v);                                                                          // level 1: h
                if (pass == 0) mean[i] = v / 512.0f;
                else rstd[i] = 1.0f / sqrtf(v / 512.0f + ln.eps);
            }
        }
        const int h2 = lane >> 5;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = wn * 128 + j * 32 + 8 * g + 4 * h2;
                const float4 gg = *reinterpret_cast<const float4*>(ln.gamma + col), bb = *reinterpret_cast<const float4*>(ln.beta + col);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = m0 + wm * 64 + i * 32 + (lane & 31);
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    union { bf16x4 b; uint2 u; } cv;
                    cv.b[0] = (__bf16)__builtin_fmaf((acc[i][j][4 * g] - mean[i]) * rstd[i], gg.x, bb.x);
                    cv.b[1] = (__bf16)__builtin_fmaf((acc[i][j][4 * g + 1] - mean[i]) * rstd[i], gg.y, bb.y);
                    cv.b[2] = (__bf16)__builtin_fmaf((acc[i][j][4 * g + 2] - mean[i]) * rstd[i], gg.z, bb.z);
                    cv.b[3] = (__bf16)__builtin_fmaf((acc[i][j][4 * g + 3] - mean[i]) * rstd[i], gg.w, bb.w);
                    if (m < M) *reinterpret_cast<uint2*>(ln.h + (int64_t)m * ln.ldh + col) = cv.u;
                }
            }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Persistent form of the 256 x 256 tile for bf16 results without a residual (vocabulary projection, q|k|v, ff1):
// one workgroup per CU walks the tiles slot, slot + 32, ... of its XCD's contiguous range.  When the last k-step of
// a tile has been read, the first NST stages of the NEXT tile (and its 1 KiB of bias, also by LDS-DMA) are requested
// before the epilogue starts, so the epilogue runs under that latency, and the epilogue's stores drain under the next
// tile's k loop: they are nontemporal (4 GB of logits per launch would otherwise push the operand panels out of the
// L2) and nothing waits for them before k-tile NST - 1 of the next tile.  Vector-memory operations of a wave complete
// in issue order, so the counted waits below count the stores that were issued between the DMA they wait for and now;
// after a partial tile (fewer stores than NSTORE) the stricter count without stores is used.
// The bf16 image of the epilogue is a region of its own behind the stages (32 rows per wave at a time), written
// and read with inline assembly: LDS accesses the compiler knows of would be ordered behind every pending DMA.
// STATS (vocabulary projection feeding a beam search): for every row and every slice of 64 columns the epilogue also emits
// the slice's largest logit and the sum of exp(logit - largest), both taken from the bf16-ROUNDED values that are stored, so
// that the selection pass needs the row's 2 * ceil(N / 64) partial values and the few slices that can hold a top-k element
// instead of the whole row.  A wave's 64 columns of a row sit 32 and 32 in the two lanes (h = 0, 1) of that row: one exchange
// across the wave halves per reduction.  Layout of `part`: [row block of 32][slice][max of rows 0..31 | sum of rows 0..31]
// floats, so that a wave stores two contiguous 128-byte pieces per slice (one instruction, lanes h = 0: max, h = 1: sum).
template <int NST, bool RELU, bool STATS>
__global__ __launch_bounds__(512) void gemm_bf16_persist_kernel(const __bf16* __restrict__ A, int64_t lda,
                                                                const __bf16* __restrict__ W, int64_t ldw,
                                                                const float* __restrict__ bias, __bf16* __restrict__ Cout,
                                                                int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n,
                                                                float* __restrict__ part, int n_slices) {
    constexpr int WVN = 4, WTM = 4, WTN = 2, NWV = 8, TM = 256, TN = 256, GM = 8;
    constexpr int STAGE = (TM + TN) * 64;
    constexpr int OUT_LD = 16 + 64 * WTN;                // bytes per row of a wave's output image
    constexpr int IMG = 32 * OUT_LD;                     // 32 rows at a time
    constexpr int BIAS_OFF = NST * STAGE, IMG_OFF = BIAS_OFF + NWV * 1024;
    constexpr int NA = TM / (16 * NWV), NW = TN / (16 * NWV), NDMA = NA + NW;
    constexpr int NSTORE = WTM * (STATS ? 5 : 4);        // global stores of a wave per full tile
    static_assert(NDMA * (NST - 1) + NSTORE < 64, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[IMG_OFF + NWV * IMG];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WVN, wn = wave % WVN;
    const int h = lane >> 5, srow = lane >> 2;
    const int nkt = K / DK;                              // >= NST (checked by the launcher)

    // this workgroup's tiles: XCD (bid & 7) owns a contiguous range of the m-grouped tile order, slot bid >> 3 of
    // gridDim.x / 8 takes every (gridDim.x / 8)-th tile of it
    const int nwg = tiles_m * tiles_n, bid = blockIdx.x, slots = gridDim.x >> 3;
    const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
    const int lb_begin = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    const int lb_end = lb_begin + q + (xcd < rem ? 1 : 0);
    int lb = lb_begin + (bid >> 3);
    if (lb >= lb_end) return;                            // workgroup-uniform

    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    int m0, n0;
    const __bf16* a_src[NA];
    const __bf16* w_src[NW];
    auto locate = [&](int l) {
        const int per_group = GM * tiles_n;
        const int first_m = (l / per_group) * GM;
        const int gsz = min(tiles_m - first_m, GM);
        m0 = (first_m + (l % per_group) % gsz) * TM;
        n0 = ((l % per_group) / gsz) * TN;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int row = (wave + NWV * j) * 16 + srow;
            a_src[j] = A + (int64_t)min(m0 + row, M - 1) * lda + (((lane & 3) ^ ((row >> 2) & 3)) * 8);
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int row = (wave + NWV * j) * 16 + srow;
            w_src[j] = W + (int64_t)min(n0 + row, N - 1) * ldw + (((lane & 3) ^ ((row >> 2) & 3)) * 8);
        }
    };
    auto stage = [&](int st, int k0) {
        unsigned char* base = smem + st * STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(a_src[j] + k0), (lds_void_t*)(base + j * (1024 * NWV)), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < NW; ++j)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(w_src[j] + k0), (lds_void_t*)(base + TM * 64 + j * (1024 * NWV)), 16, 0, 0);
    };
    // every wave fetches the 256 bias values of the tile into a slot of its own (one more DMA, the oldest of the group)
    auto request_tile = [&]() {
        if (bias != nullptr)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(bias + min(n0 + 4 * lane, N - 4)), (lds_void_t*)(smem + BIAS_OFF + wave * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < NST; ++p) stage(p, p * DK);
    };

    const int x = ((lane & 31) >> 2) & 3;
    const uint32_t a_off = lds_base + wm * 32 * WTM * 64 + (lane & 31) * 64;
    const uint32_t w_off = lds_base + TM * 64 + wn * 32 * WTN * 64 + (lane & 31) * 64;
    union Frag { u32x4 u; bf16x8 v; };
    Frag fa[2][WTM], fw[2][WTN];
    f32x16 acc[WTM][WTN];
    auto read_step = [&](int st, int ks) {
        const uint32_t as = a_off + st * STAGE, ws = w_off + st * STAGE;
        const uint32_t slot = ((2 * ks + h) ^ x) * 16;
#pragma unroll
        for (int i = 0; i < WTM; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[ks][i].u) : "v"(as + i * 2048 + slot));
#pragma unroll
        for (int j = 0; j < WTN; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(fw[ks][j].u) : "v"(ws + j * 2048 + slot));
    };
    auto mfma_step = [&](int ks) {
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ks][j].v, fa[ks][i].v, acc[i][j], 0, 0, 0);
    };
    auto lds_wait = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    locate(lb);
    request_tile();
    bool stores_behind = false;     // NSTORE stores were issued after the DMA group the next waits refer to
    while (true) {
        // stage 0 (and the bias, which is older) landed
        if (stores_behind) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NST - 1) + NSTORE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NST - 1)) : "memory");
        // accumulators from the wave's own bias slot: acc[i][j][4 g + e] belongs to column 32 j + 8 g + 4 h + e of the wave's 64
        if (bias != nullptr) {
            const uint32_t bo = lds_base + BIAS_OFF + wave * 1024 + (wn * 64 + 4 * h) * 4;
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                u32x4 bq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1" : "=v"(bq[g]) : "v"(bo + (j * 32 + 8 * g) * 4));
                lds_wait();
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float b = __uint_as_float(bq[g][e]);
#pragma unroll
                        for (int i = 0; i < WTM; ++i) acc[i][j][4 * g + e] = b;
                    }
            }
        } else {
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        }
        __builtin_amdgcn_s_barrier();
        read_step(0, 0);
        int st = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            const int st_next = st + 1 == NST ? 0 : st + 1;
            lds_wait();
            read_step(st, 1);
            mfma_step(0);
            if (kt + NST - 1 < nkt) {
                if (stores_behind && kt < NST - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NST - 2) + NSTORE) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NST - 2)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            lds_wait();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (kt + NST < nkt) stage(st, (kt + NST) * DK);
            if (kt + 1 < nkt) read_step(st_next, 0);
            mfma_step(1);
            st = st_next;
        }
        __builtin_amdgcn_s_barrier();   // every wave is past its last fragment read: the stages are free

        const int tm0 = m0, tn0 = n0;
        lb += slots;
        const bool more = lb < lb_end;
        if (more) {
            locate(lb);
            request_tile();
        }

        // epilogue of tile (tm0, tn0): per 32-row block the lane packs its 2 x 16 results to bf16 into the wave's image
        // (row = lane & 31), then 8 lanes read one 128-byte row segment back and store it
        const bool full = tm0 + TM <= M && tn0 + TN <= N;
        const uint32_t img = lds_base + IMG_OFF + wave * IMG;
        const uint32_t wr = img + (lane & 31) * OUT_LD + 8 * h;
        const uint32_t rd = img + (lane >> 3) * OUT_LD + (lane & 7) * 16;
        const int gcol = tn0 + wn * 64 + (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
            float wv[STATS ? 16 * WTN : 1];              // the lane's 32 values of its row as stored (bf16, widened)
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < WTN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    union { bf16x4 b; uint2 u; } cv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][4 * g + e];
                        if (RELU) v = v > 0.0f ? v : 0.0f;
                        cv.b[e] = (__bf16)v;
                        if (STATS) {
                            float w = (float)cv.b[e];
                            if (!full && tn0 + wn * 64 + 32 * j + 8 * g + 4 * h + e >= N) w = -INFINITY;   // not a column of C
                            wv[16 * j + 4 * g + e] = w;
                            mx = fmaxf(mx, w);
                        }
                    }
                    asm volatile("ds_write_b64 %0, %1" ::"v"(wr + j * 64 + 16 * g), "v"(cv.u) : "memory");
                }
            u32x4 o[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) asm volatile("ds_read_b128 %0, %1" : "=v"(o[p]) : "v"(rd + p * 8 * OUT_LD) : "memory");
            if (STATS) {
                mx = fmaxf(mx, lane_xor<32>(mx));
                float sum = 0.0f;
#pragma unroll
                for (int q = 0; q < 16 * WTN; ++q) sum += __builtin_amdgcn_exp2f((wv[q] - mx) * 1.44269504088896341f);
                sum += lane_xor<32>(sum);
                const int blk = (tm0 + wm * 128 + i * 32) >> 5, slice = (tn0 >> 6) + wn;
                float* dst = part + ((int64_t)blk * n_slices + slice) * 64 + lane;     // lanes 0..31: max, 32..63: sum
                const float val = h ? sum : mx;
                if (full) __builtin_nontemporal_store(val, dst);
                else if (blk * 32 < M && slice < n_slices) __builtin_nontemporal_store(val, dst);
            }
            lds_wait();
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int grow = tm0 + wm * 128 + i * 32 + p * 8 + (lane >> 3);
                __bf16* cp = Cout + (int64_t)grow * ldc + gcol;
                if (full) {
                    __builtin_nontemporal_store(o[p], reinterpret_cast<u32x4*>(cp));
                } else if (grow < M && gcol < N) {
                    if (gcol + 8 <= N) {
                        __builtin_nontemporal_store(o[p], reinterpret_cast<u32x4*>(cp));
                    } else {
                        Frag oc;
                        oc.u = o[p];
                        for (int e = 0; e < 8; ++e)
                            if (gcol + e < N) cp[e] = oc.v[e];
                    }
                }
            }
        }
        if (!more) break;
        stores_behind = full;
    }
}

// TST_BF16_NO_PERSIST=1 in the environment selects the one-tile-per-workgroup kernel (measurements)
bool tst_bf16_no_persist() {
    static const bool v = [] { const char* e = getenv("TST_BF16_NO_PERSIST"); return e != nullptr && e[0] == '1'; }();
    return v;
}

// TST_BF16_TILE=0|1|2|3 forces the 128 x 128, 256 x 128, 256 x 256 or 128 x 512 (whole rows of a 512-column result) tile (measurements)
int tst_bf16_forced_shape() {
    static const int v = [] { const char* e = getenv("TST_BF16_TILE"); return (e != nullptr && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : -1; }();
    return v;
}

}  // namespace


extern "C" int tst_gemm_mfma_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                  const float* res, int64_t ldr, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                  int act, int out_bf16, void* stream) {
    TST_REQUIRE(tst_dims_ok({M, N, K}), TST_ERR_UNSUPPORTED, "tst_gemm_mfma_bf16: a size exceeds 2^31 - 1");
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
    hipStream_t st = (hipStream_t)stream;
    const int variant = (out_bf16 ? 4 : 0) | (res != nullptr ? 2 : 0) | (act == 1 ? 1 : 0);
    if (K % DK == 0 && tst_aligned16(bias)) {
        // LDS-DMA kernel: 256 x 256 tiles (8 waves) when they give every CU a workgroup, else 256 x 128 (8 waves of 64 x 64) under
        // the same condition, else 128 x 128 (4 waves).  The larger the tile, the fewer operand bytes pass through the CU's
        // vector-memory path per MFMA (64 (TM + TN) / (TM TN) of its capacity: 0.5, 0.75, 1.0).
        const int64_t big_tiles = ((M + 255) / 256) * ((N + 255) / 256), mid_tiles = ((M + 255) / 256) * ((N + 127) / 128);
        // measured on the language model's shapes (K = 512 / 2048, 8 k - 64 k rows): a wide bf16 result (the vocabulary) is best
        // on the persistent 256 x 256 kernel; q|k|v and ff1 (N <= 2048) on 256 x 128 at every row count (more tiles than CUs, two
        // workgroups per CU: 4-10 % over 256 x 256); the fp32 + residual products on 256 x 256 from 192 tiles on
        const bool plain_bf16 = out_bf16 && res == nullptr;
        int shape = plain_bf16 ? (N >= 4096 && big_tiles >= 256 ? 2 : mid_tiles >= 128 ? 1 : 0)
                               : (big_tiles >= 192 ? 2 : mid_tiles >= 128 ? 1 : 0);
        if (const int forced = tst_bf16_forced_shape(); forced >= 0) shape = forced;
        const bool big = shape == 2;
        const int64_t tm = shape == 3 ? (M + 127) / 128 : shape >= 1 ? (M + 255) / 256 : (M + 127) / 128;
        const int64_t tn = shape == 3 ? (N + 511) / 512 : shape == 2 ? (N + 255) / 256 : (N + 127) / 128;
        TST_REQUIRE(tm * tn < (1ll << 31), TST_ERR_UNSUPPORTED, "tst_gemm_mfma_bf16: grid too large");
        const dim3 g2((unsigned)(tm * tn));
        // (N % 4: the persistent kernel fetches the bias in aligned groups of 4)
        if (big && out_bf16 && res == nullptr && K / DK >= 3 && N % 4 == 0 && !tst_bf16_no_persist()) {
            // one workgroup per CU (the kernel's LDS allows no second one), a multiple of 8 so that every XCD has the same slots
            const dim3 gp(256);
            if (act == 1)
                hipLaunchKernelGGL((gemm_bf16_persist_kernel<3, true, false>), gp, dim3(512), 0, st, (const __bf16*)A, lda, (const __bf16*)W, ldw,
                                   bias, (__bf16*)C, ldc, (int)M, (int)N, (int)K, (int)tm, (int)tn, nullptr, 0);
            else
                hipLaunchKernelGGL((gemm_bf16_persist_kernel<3, false, false>), gp, dim3(512), 0, st, (const __bf16*)A, lda, (const __bf16*)W, ldw,
                                   bias, (__bf16*)C, ldc, (int)M, (int)N, (int)K, (int)tm, (int)tn, nullptr, 0);
            return tst_after_launch("tst_gemm_mfma_bf16");
        }
#define TST_BF16_DMA(OB, HR, RL)                                                                                      \
    do {                                                                                                              \
        if (shape == 3)                                                                                               \
            hipLaunchKernelGGL((gemm_bf16_dma_kernel<2, 4, 2, 4, 3, OB, HR, RL>), g2, dim3(512), 0, st, (const __bf16*)A, lda, \
                               (const __bf16*)W, ldw, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, (int)tm, (int)tn); \
        else if (shape == 2)                                                                                          \
            hipLaunchKernelGGL((gemm_bf16_dma_kernel<2, 4, 4, 2, 3, OB, HR, RL>), g2, dim3(512), 0, st, (const __bf16*)A, lda, \
                               (const __bf16*)W, ldw, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, (int)tm, (int)tn); \
        else if (shape == 1)                                                                                          \
            hipLaunchKernelGGL((gemm_bf16_dma_kernel<4, 2, 2, 2, 3, OB, HR, RL>), g2, dim3(512), 0, st, (const __bf16*)A, lda, \
                               (const __bf16*)W, ldw, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, (int)tm, (int)tn); \
        else                                                                                                          \
            hipLaunchKernelGGL((gemm_bf16_dma_kernel<2, 2, 2, 2, 4, OB, HR, RL>), g2, dim3(256), 0, st, (const __bf16*)A, lda, \
                               (const __bf16*)W, ldw, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, (int)tm, (int)tn); \
    } while (0)
        switch (variant) {
            case 0: TST_BF16_DMA(false, false, false); break;
            case 1: TST_BF16_DMA(false, false, true); break;
            case 2: TST_BF16_DMA(false, true, false); break;
            case 3: TST_BF16_DMA(false, true, true); break;
            case 4: TST_BF16_DMA(true, false, false); break;
            case 5: TST_BF16_DMA(true, false, true); break;
            case 6: TST_BF16_DMA(true, true, false); break;
            default: TST_BF16_DMA(true, true, true); break;
        }
#undef TST_BF16_DMA
        return tst_after_launch("tst_gemm_mfma_bf16");
    }
    const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    TST_REQUIRE(tiles_m * tiles_n < (1ll << 31), TST_ERR_UNSUPPORTED, "tst_gemm_mfma_bf16: grid too large");
    const dim3 grid((unsigned)(tiles_m * tiles_n));
#define TST_BF16_LAUNCH(OB, HR, RL)                                                                                 \
    hipLaunchKernelGGL((gemm_bf16_kernel<OB, HR, RL>), grid, dim3(256), 0, st, (const __bf16*)A, lda, (const __bf16*)W, \
                       ldw, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, (int)tiles_m, (int)tiles_n)
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


extern "C" int tst_gemm_res_ln_ok(int64_t M, int64_t N, int64_t K) {
    return (M > 0 && M < (1ll << 31) - 128 && N == 512 && K >= 96 && K % DK == 0 && K < (1ll << 31)) ? 1 : 0;
}

extern "C" int tst_gemm_res_ln_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* x, const float* gamma,
                                    const float* beta, float eps, void* h, int64_t ldh, int64_t M, int64_t N, int64_t K, void* stream) {
    TST_REQUIRE(tst_dims_ok({M, N, K}), TST_ERR_UNSUPPORTED, "tst_gemm_res_ln_bf16: a size exceeds 2^31 - 1");
    TST_REQUIRE(A && W && x && gamma && beta && h, TST_ERR_ARG, "tst_gemm_res_ln_bf16: null pointer");
    TST_REQUIRE(M >= 0, TST_ERR_ARG, "tst_gemm_res_ln_bf16: bad sizes");
    if (M == 0) return TST_OK;
    TST_REQUIRE(tst_gemm_res_ln_ok(M, N, K), TST_ERR_UNSUPPORTED, "tst_gemm_res_ln_bf16: needs N == 512 and K a multiple of 32, at least 96 (got N=%lld K=%lld)",
                (long long)N, (long long)K);
    TST_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldh % 4 == 0 && lda >= K && ldw >= K && ldh >= N, TST_ERR_ARG,
                "tst_gemm_res_ln_bf16: lda, ldw multiples of 8, ldh of 4; lda >= K, ldw >= K, ldh >= N");
    TST_REQUIRE(tst_aligned16(A) && tst_aligned16(W) && tst_aligned16(x) && tst_aligned16(bias) && tst_aligned16(gamma) && tst_aligned16(beta) &&
                    ((uintptr_t)h & 7u) == 0, TST_ERR_ARG, "tst_gemm_res_ln_bf16: A, W, x, bias, gamma, beta 16-byte aligned, h 8-byte aligned");
    LnOut ln;
    ln.gamma = gamma; ln.beta = beta; ln.eps = eps; ln.h = (__bf16*)h; ln.ldh = ldh;
    const int64_t tm = (M + 127) / 128;
    hipLaunchKernelGGL((gemm_bf16_dma_kernel<2, 4, 2, 4, 3, false, true, false, true>), dim3((unsigned)tm), dim3(512), 0, (hipStream_t)stream,
                       (const __bf16*)A, lda, (const __bf16*)W, ldw, bias, x, N, x, N, (int)M, (int)N, (int)K, (int)tm, 1, ln);
    return tst_after_launch("tst_gemm_res_ln_bf16");
}

// The shapes for which the vocabulary projection can emit its per-slice statistics: those that tst_gemm_mfma_bf16 runs on the
// persistent 256 x 256 kernel (a bf16 result of at least 4096 columns and 256 tiles, K a multiple of 32 and at least 96)
// and whose rows have at most 512 slices of 64 columns (what tst_topk_from_stats_bf16 holds per row).
extern "C" int tst_vocab_stats_ok(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N < 4096 || N > 32768 || N % 8 != 0 || K % DK != 0 || K / DK < 3 || M >= (1ll << 31) || N >= (1ll << 31) || K >= (1ll << 31)) return 0;
    return ((M + 255) / 256) * ((N + 255) / 256) >= 256 ? 1 : 0;
}

extern "C" int tst_vocabproj_stats_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                                        int64_t M, int64_t N, int64_t K, float* part, int64_t n_slices, void* stream) {
    TST_REQUIRE(tst_dims_ok({M, N, K, n_slices}), TST_ERR_UNSUPPORTED, "tst_vocabproj_stats_bf16: a size exceeds 2^31 - 1");
    TST_REQUIRE(A && W && C && part, TST_ERR_ARG, "tst_vocabproj_stats_bf16: null pointer");
    TST_REQUIRE(tst_vocab_stats_ok(M, N, K), TST_ERR_UNSUPPORTED,
                "tst_vocabproj_stats_bf16: M=%lld N=%lld K=%lld is not a shape of the persistent kernel (tst_vocab_stats_ok)", (long long)M,
                (long long)N, (long long)K);
    TST_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && lda >= K && ldw >= K && ldc >= N, TST_ERR_ARG,
                "tst_vocabproj_stats_bf16: lda, ldw, ldc must be multiples of 8, lda >= K, ldw >= K, ldc >= N");
    TST_REQUIRE(tst_aligned16(A) && tst_aligned16(W) && tst_aligned16(C) && tst_aligned16(bias) && tst_aligned16(part), TST_ERR_ARG,
                "tst_vocabproj_stats_bf16: A, W, C, bias, part must be 16-byte aligned");
    TST_REQUIRE(n_slices == (N + 63) / 64, TST_ERR_ARG, "tst_vocabproj_stats_bf16: n_slices must be ceil(N / 64)");
    const int64_t tm = (M + 255) / 256, tn = (N + 255) / 256;
    hipLaunchKernelGGL((gemm_bf16_persist_kernel<3, false, true>), dim3(256), dim3(512), 0, (hipStream_t)stream, (const __bf16*)A, lda,
                       (const __bf16*)W, ldw, bias, (__bf16*)C, ldc, (int)M, (int)N, (int)K, (int)tm, (int)tn, part, (int)n_slices);
    return tst_after_launch("tst_vocabproj_stats_bf16");
}
