// fp32 GEMM for the vocabulary projection with BOTH operands staged by LDS-DMA: C = bias + A * Wt, A given k-major.
//
// gemm_f32_kernel stages its tiles through registers (global_load -> VGPR -> ds_write) because the row-major activations
// have to be transposed into the k-major LDS image the one-float-per-lane MFMA operands want.  When the producer of A
// writes it k-major (AT[k][m]: tst_layernorm_t_f32), a k-tile of either operand is a set of contiguous row pieces, and
// global_load_lds (16 bytes per lane, 1 KiB per wave-instruction, lane-linear in LDS) lands it in exactly that image:
// no staging registers, no LDS store instructions, and the tile loads of the next stages stay in flight across barriers.
// Every output element is still acc = bias[n]; for k ascending: acc = fma(A[m][k], Wt[k][n], acc) on
// v_mfma_f32_32x32x2_f32, so results are bit-identical to gemm_f32_kernel.
// Tile 128 x 256 x BK per 256-thread workgroup (4 waves as 2 x 2, 64 x 128 per wave), NST stages in a ring.  Operand
// reads are inline assembly with explicit waits: the compiler would order every LDS read behind ALL pending LDS-DMA.
#include "tst_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

namespace {

constexpr int TN = 256, GROUP_M = 16;

// TM x 256 x BK tile per 256-thread workgroup, 4 waves as WVM x WVN, each with WTM x WTN MFMA tiles of 32 x 32.  Built
// for 128 rows (2 x 2 waves, 2 x 4 tiles per wave): the vocabulary projection.  For the layer products (256 .. 1024
// columns, K = 256 / 1024) 64- and 32-row tiles of this kernel, with residual / relu epilogues and k-major outputs, were
// measured against gemm_f32_kernel at 4096 .. 32768 rows and were slower or equal in every case (those products are
// short-K and close to the memory roofline: a row moves 3 KB per 131 kFLOP): not instantiated.
// OUT_T: the product is computed transposed (operands swapped: the same fma chain per element), a lane then owns a ROW
// and its 16 registers are 16 columns, and C (and res) are K-MAJOR, C[n][m]: 32 lanes store 128 contiguous bytes.
template <int TM, int WVM, int WVN, int BK, int NST, bool HAS_RES, bool RELU, bool OUT_T, bool NT_STORE = false>
__global__ __launch_bounds__(64 * WVM * WVN, (WVM * WVN == 4 ? 2 : 1)) void gemm_f32_dma_kernel(const float* __restrict__ AT, int64_t ldat, const float* __restrict__ W,
                                                              int64_t ldw, const float* __restrict__ bias, const float* res, int64_t ldr,
                                                              float* C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr int WTM = TM / WVM / 32, WTN = TN / WVN / 32;
    constexpr int A_BYTES = BK * TM * 4, B_BYTES = BK * TN * 4, STAGE = A_BYTES + B_BYTES;
    constexpr int NWV = WVM * WVN, PIECE = 1024 * NWV;                          // a round of 1 KiB pieces, one per wave
    constexpr int NA = A_BYTES / PIECE, NB = B_BYTES / PIECE, NDMA = NA + NB;   // wave-instructions per wave and tile
    constexpr int LPR = TM / 4, RPP = 64 / LPR;                               // lanes per A row, A rows per 1 KiB piece
    static_assert((NWV == 4 || NWV == 8) && A_BYTES % PIECE == 0 && B_BYTES % PIECE == 0 && NST >= 3 && WTM >= 1 && WTN >= 1, "tile shape");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
    const int lb = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int per_group = GROUP_M * tiles_n;
    const int first_m = (lb / per_group) * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (lb % per_group) % gsz;
    const int tn = (lb % per_group) / gsz;
    const int m0 = tm * TM, n0 = tn * TN;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WVN, wn = wave % WVN;

    // DMA sources.  A piece (1 KiB) is RPP k-rows of the A tile (TM floats each): lane l -> k = RPP p + l / LPR,
    // m = 4 (l % LPR); a piece of the Wt tile is one k-row (256 floats): lane l -> n = 4 l.  Wave w moves pieces w, w + NWV, ...
    // The rows of AT are ldat >= tiles_m * TM floats long, so the tile's columns always exist (columns beyond M hold
    // stale values whose products are never stored); columns of Wt beyond ldw are clamped.
    const float* a_src = AT + (int64_t)(RPP * wave + lane / LPR) * ldat + m0 + 4 * (lane % LPR);
    const float* w_src = W + (int64_t)wave * ldw + min(n0 + 4 * lane, (int)ldw - 4);
    auto stage = [&](int st, int k0) {
        unsigned char* base = smem + st * STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(a_src + (int64_t)(k0 + NWV * RPP * j) * ldat), (lds_void_t*)(base + j * PIECE), 16, 0,
                                             0);
#pragma unroll
        for (int j = 0; j < NB; ++j)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(w_src + (int64_t)(k0 + NWV * j) * ldw), (lds_void_t*)(base + A_BYTES + j * PIECE), 16, 0,
                                             0);
    };

    // accumulators.  !OUT_T: lane owns column (lane & 31) of tile j, registers are 16 rows.  OUT_T: lane owns row
    // (lane & 31) of tile i, registers are 16 columns c(r) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of tile j.
    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        const int nb = n0 + wn * 32 * WTN + j * 32;
        if (OUT_T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = nb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float b = (bias != nullptr && col < N) ? bias[col] : 0.0f;
#pragma unroll
                for (int i = 0; i < WTM; ++i) acc[i][j][r] = b;
            }
        } else {
            const int col = nb + (lane & 31);
            const float b = (bias != nullptr && col < N) ? bias[col] : 0.0f;
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = b;
        }
    }

    // operand of MFMA step s (k = 2 s + khalf) of a stage: A image [k][TM], Wt image [k][256]
    const int khalf = lane >> 5;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t a_off = lds_base + (khalf * TM + wm * 32 * WTM + (lane & 31)) * 4;
    const uint32_t b_off = lds_base + A_BYTES + (khalf * TN + wn * 32 * WTN + (lane & 31)) * 4;
    float fa[2][WTM], fb[2][WTN];   // two operand sets: the one being multiplied and the one being read
    auto read_step = [&](int set, uint32_t as, uint32_t bs) {
#pragma unroll
        for (int i = 0; i < WTM; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(fa[set][i]) : "v"(as + i * 128));
#pragma unroll
        for (int j = 0; j < WTN; ++j) asm volatile("ds_read_b32 %0, %1" : "=v"(fb[set][j]) : "v"(bs + j * 128));
    };
    auto lds_wait = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfma_step = [&](int set) {
        // raised priority over the step's MFMAs: measured +1 % on the vocabulary projection here (the same pair around the bf16
        // kernels' MFMA steps costs them 2-3 %)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
                acc[i][j] = OUT_T ? __builtin_amdgcn_mfma_f32_32x32x2f32(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // Ring of NST stages; tiles kt + 1 .. kt + NST - 2 are in flight while tile kt is multiplied.  In the middle of tile kt:
    // wait until tile kt + 1 has landed (all but the NST - 3 youngest tiles), barrier (it also tells that every wave is past
    // tile kt - 1), then the DMA of tile kt + NST - 1 goes into the stage tile kt - 1 used.  An LDS-DMA is complete for other
    // waves only after the issuing wave has waited for it and both have passed a barrier.
    constexpr int STEPS = BK / 2;
    const int nkt = K / BK;
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
        if (p < nkt) stage(p, p * BK);
    if (nkt >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NST - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_step(0, a_off, b_off);
    int st = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const int st_next = st + 1 == NST ? 0 : st + 1;
        const uint32_t as = a_off + st * STAGE, bs = b_off + st * STAGE;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            lds_wait();                                          // operands of step s are in registers
            if (s + 1 < STEPS) read_step((s + 1) & 1, as + (s + 1) * (2 * TM * 4), bs + (s + 1) * (2 * TN * 4));
            else if (kt + 1 < nkt) read_step((s + 1) & 1, a_off + st_next * STAGE, b_off + st_next * STAGE);
            // Left to the scheduler, the reads end up between the LAST MFMAs of the step.  Pinned in front of the step's MFMAs
            // or one behind each of its first MFMAs, the 128-row tile lost 3-5 % (measured); the scheduler's placement stays.
            mfma_step(s & 1);
            if (s == STEPS / 2 - 1) {
                if (kt + NST - 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NST - 3)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (kt + NST - 1 < nkt) stage(st == 0 ? NST - 1 : st - 1, (kt + NST - 1) * BK);
            }
        }
        st = st_next;
    }

    const bool full_tile = m0 + TM <= M && n0 + TN <= N;
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            if (OUT_T) {
                // lane owns row `row`; register r is column c(r): C[col][row], 32 lanes = 128 contiguous bytes
                const int row = m0 + wm * 32 * WTM + i * 32 + (lane & 31);
                const int col0 = n0 + wn * 32 * WTN + j * 32 + 4 * (lane >> 5);
                float rv[16];
                if (HAS_RES) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int col = col0 + (r & 3) + 8 * (r >> 2);
                        rv[r] = (full_tile || (col < N && row < M)) ? res[(int64_t)col * ldr + row] : 0.0f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = col0 + (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][r];
                    if (RELU) v = v > 0.0f ? v : 0.0f;
                    if (HAS_RES) v = v + rv[r];
                    if (full_tile || (col < N && row < M)) C[(int64_t)col * ldc + row] = v;
                }
            } else {
                // lane owns column col of 16 rows per 32 x 32 tile (rows (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))
                const int col = n0 + wn * 32 * WTN + j * 32 + (lane & 31);
                const int row0 = m0 + wm * 32 * WTM + i * 32 + 4 * (lane >> 5);
                float rv[16];
                if (HAS_RES) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + (r & 3) + 8 * (r >> 2);
                        rv[r] = (full_tile || (col < N && row < M)) ? res[(int64_t)row * ldr + col] : 0.0f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][r];
                    if (RELU) v = v > 0.0f ? v : 0.0f;
                    if (HAS_RES) v = v + rv[r];
                    if (full_tile || (col < N && row < M)) {
                        if (NT_STORE) __builtin_nontemporal_store(v, &C[(int64_t)row * ldc + col]);
                        else C[(int64_t)row * ldc + col] = v;
                    }
                }
            }
        }
}

// Persistent form of the 128 x 256 tile (plain product, nontemporal stores): at most two workgroups per CU walk the tiles
// slot, slot + slots, ... of their XCD's contiguous range, and the ring of stages runs on ACROSS tiles: while the last NST - 1
// stages of a tile are multiplied, the stages requested are the first ones of the next tile, so a tile starts with its operands
// in LDS and its bias in registers (requested a tile ahead), and the 128 stores per lane of a tile's epilogue drain under the
// next tile's k loop.  Vector-memory operations of a wave complete in issue order and vmcnt saturates at 63: while the stage
// waited for is older than a full tile's stores, "all but the 63 youngest operations" covers it; after a partial tile (fewer
// stores) the strict count is used.  Same fma chain per element as gemm_f32_dma_kernel.
template <int BK, int NST>
__global__ __launch_bounds__(256, 2) void gemm_f32_dma_persist_kernel(const float* __restrict__ AT, int64_t ldat, const float* __restrict__ W,
                                                                      int64_t ldw, const float* __restrict__ bias, float* C, int64_t ldc, int M,
                                                                      int N, int K, int tiles_m, int tiles_n) {
    constexpr int TM = 128, WVN = 2, WTM = 2, WTN = 4;
    constexpr int A_BYTES = BK * TM * 4, B_BYTES = BK * TN * 4, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = A_BYTES / 4096, NB = B_BYTES / 4096, NDMA = NA + NB;
    constexpr int LPR = TM / 4, RPP = 64 / LPR;
    constexpr int NBIAS = WTN;                       // bias loads of a lane per tile
    static_assert(NDMA * (NST - 3) + NBIAS < 63 && NST >= 4, "counted waits");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WVN, wn = wave % WVN;
    const int nkt = K / BK;

    // this workgroup's tiles: XCD (bid & 7) owns a contiguous range of the m-grouped tile order, slot bid >> 3 takes every
    // (gridDim.x / 8)-th tile of it
    const int nwg = tiles_m * tiles_n, bid = blockIdx.x, slots = gridDim.x >> 3;
    const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
    const int lb_begin = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    const int lb_end = lb_begin + q + (xcd < rem ? 1 : 0);
    const int lb0 = lb_begin + (bid >> 3);
    if (lb0 >= lb_end) return;                       // workgroup-uniform
    const int n_tiles = (lb_end - lb0 + slots - 1) / slots;
    const int n_stages = n_tiles * nkt;

    auto tile_of = [&](int l, int& m0, int& n0) {
        const int per_group = GROUP_M * tiles_n;
        const int first_m = (l / per_group) * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        m0 = (first_m + (l % per_group) % gsz) * TM;
        n0 = ((l % per_group) / gsz) * TN;
    };

    // ---- request side: the stream of stages (tile by tile, k ascending) is requested NST - 1 stages ahead of its use
    int lb_r = lb0, kt_r = 0, slot_r = 0;
    const float *a_src, *w_src;
    auto locate_r = [&]() {
        int m0, n0;
        tile_of(lb_r, m0, n0);
        a_src = AT + (int64_t)(RPP * wave + lane / LPR) * ldat + m0 + 4 * (lane % LPR);
        w_src = W + (int64_t)wave * ldw + min(n0 + 4 * lane, (int)ldw - 4);
    };
    auto request_stage = [&]() {
        unsigned char* base = smem + slot_r * STAGE + wave * 1024;
        const int k0 = kt_r * BK;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(a_src + (int64_t)(k0 + 4 * RPP * j) * ldat), (lds_void_t*)(base + j * 4096), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(w_src + (int64_t)(k0 + 4 * j) * ldw), (lds_void_t*)(base + A_BYTES + j * 4096), 16, 0, 0);
        slot_r = slot_r + 1 == NST ? 0 : slot_r + 1;
        if (++kt_r == nkt) {
            kt_r = 0;
            lb_r += slots;
            if (lb_r < lb_end) locate_r();
        }
    };
    // the bias of a tile, requested a tile ahead with loads the compiler does not track (it would wait for them, and for
    // every DMA in front of them, at once): lane owns columns n0 + wn * 128 + 32 j + (lane & 31)
    float nbias[NBIAS];
    auto request_bias = [&](int l) {
        int m0, n0;
        tile_of(l, m0, n0);
#pragma unroll
        for (int j = 0; j < NBIAS; ++j) {
            const float* p = bias + min(n0 + wn * 32 * WTN + j * 32 + (lane & 31), N - 1);
            asm volatile("global_load_dword %0, %1, off" : "=v"(nbias[j]) : "v"(p) : "memory");
        }
    };

    const int khalf = lane >> 5;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t a_off = lds_base + (khalf * TM + wm * 32 * WTM + (lane & 31)) * 4;
    const uint32_t b_off = lds_base + A_BYTES + (khalf * TN + wn * 32 * WTN + (lane & 31)) * 4;
    float fa[2][WTM], fb[2][WTN];
    auto read_step = [&](int set, uint32_t as, uint32_t bs) {
#pragma unroll
        for (int i = 0; i < WTM; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(fa[set][i]) : "v"(as + i * 128));
#pragma unroll
        for (int j = 0; j < WTN; ++j) asm volatile("ds_read_b32 %0, %1" : "=v"(fb[set][j]) : "v"(bs + j * 128));
    };
    auto lds_wait = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 acc[WTM][WTN];

    locate_r();
    if (bias != nullptr) request_bias(lb0);
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
        if (p < n_stages) request_stage();
    if (n_stages >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NST - 2)) : "memory");   // stage 0 and the bias
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_step(0, a_off, b_off);

    constexpr int STEPS = BK / 2;
    int st = 0, g = 0;                               // slot and index (in this workgroup's stream) of the stage being multiplied
    bool stores_behind = false;                      // a full tile's stores were issued in front of this tile
    for (int lb = lb0; lb < lb_end; lb += slots) {
        int m0, n0;
        tile_of(lb, m0, n0);
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            const float b = bias != nullptr ? nbias[j] : 0.0f;
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = b;
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool bias_behind = bias != nullptr && lb + slots < lb_end;
        if (bias_behind) request_bias(lb + slots);
        for (int kt = 0; kt < nkt; ++kt, ++g) {
            const int st_next = st + 1 == NST ? 0 : st + 1;
            const uint32_t as = a_off + st * STAGE, bs = b_off + st * STAGE;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                lds_wait();
                if (s + 1 < STEPS) read_step((s + 1) & 1, as + (s + 1) * (2 * TM * 4), bs + (s + 1) * (2 * TN * 4));
                else if (g + 1 < n_stages) read_step((s + 1) & 1, a_off + st_next * STAGE, b_off + st_next * STAGE);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][i], fb[s & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                if (s == STEPS / 2 - 1) {
                    // stage g + 1 must have landed: NST - 3 younger stages may be in flight, plus what was issued behind it at
                    // the start of this tile (the previous tile's stores, the next tile's bias loads)
                    if (g + NST - 2 < n_stages) {
                        if (kt <= NST - 3 && stores_behind) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
                        else if (kt <= NST - 3 && bias_behind) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NST - 3) + NBIAS) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NST - 3)) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + NST - 1 < n_stages) request_stage();
                }
            }
            st = st_next;
        }
        // epilogue: lane owns column col of 16 rows per 32 x 32 block (rows (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))
        const bool full_tile = m0 + TM <= M && n0 + TN <= N;
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                const int col = n0 + wn * 32 * WTN + j * 32 + (lane & 31);
                const int row0 = m0 + wm * 32 * WTM + i * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2);
                    if (full_tile || (col < N && row < M)) __builtin_nontemporal_store(acc[i][j][r], &C[(int64_t)row * ldc + col]);
                }
            }
        stores_behind = full_tile;
    }
}

// y^T = LayerNorm(x)^T: the statistics and the affine map of layernorm_kernel (64 interleaved partial sums, xor-butterfly;
// one wave per row), the normalised rows of a 64-row block collected in LDS and written k-major, yT[k][r], 64 consecutive
// rows (256 bytes) per wave-instruction.
__global__ __launch_bounds__(256) void layernorm_t_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                          float eps, float* __restrict__ yT, int64_t ldt, int64_t R, int d) {
    extern __shared__ float tile[];   // [64][d + 1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d4 = d >> 2, ldl = d + 1;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    for (int rr = wave; rr < 64; rr += 4) {
        const int64_t r = r0 + rr;
        if (r >= R) break;
        const float4* xr = reinterpret_cast<const float4*>(x) + r * d4;
        float p = 0.0f;
        for (int c = lane; c < d4; c += 64) {
            const float4 v = xr[c];
            p = p + v.x; p = p + v.y; p = p + v.z; p = p + v.w;
        }
        const float dd = (float)d;
        const float mean = wave_sum_butterfly(p) / dd;
        p = 0.0f;
        for (int c = lane; c < d4; c += 64) {
            const float4 v = xr[c];
            float t;
            t = v.x - mean; p = __builtin_fmaf(t, t, p);
            t = v.y - mean; p = __builtin_fmaf(t, t, p);
            t = v.z - mean; p = __builtin_fmaf(t, t, p);
            t = v.w - mean; p = __builtin_fmaf(t, t, p);
        }
        const float var = wave_sum_butterfly(p) / dd;
        const float rstd = 1.0f / sqrtf(var + eps);
        for (int c = lane; c < d4; c += 64) {
            const float4 v = xr[c];
            const float4 gg = reinterpret_cast<const float4*>(g)[c];
            const float4 bb = reinterpret_cast<const float4*>(b)[c];
            float* tp = tile + rr * ldl + 4 * c;
            tp[0] = __builtin_fmaf((v.x - mean) * rstd, gg.x, bb.x);
            tp[1] = __builtin_fmaf((v.y - mean) * rstd, gg.y, bb.y);
            tp[2] = __builtin_fmaf((v.z - mean) * rstd, gg.z, bb.z);
            tp[3] = __builtin_fmaf((v.w - mean) * rstd, gg.w, bb.w);
        }
    }
    __syncthreads();
    const int64_t r = r0 + lane;
    if (r < R)
        for (int k = wave; k < d; k += 4) yT[(int64_t)k * ldt + r] = tile[lane * ldl + k];
}

}  // namespace

extern "C" int tst_layernorm_t_f32(const float* x, const float* gamma, const float* beta, float eps, float* yT, int64_t ldt, int64_t R,
                                   int64_t d, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, d}), TST_ERR_UNSUPPORTED, "tst_layernorm_t_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(x && gamma && beta && yT, TST_ERR_ARG, "tst_layernorm_t_f32: null pointer");
    TST_REQUIRE(R >= 0 && d >= 4 && d % 4 == 0 && ldt >= R, TST_ERR_ARG, "tst_layernorm_t_f32: bad sizes (d %% 4 == 0, ldt >= R)");
    TST_REQUIRE(d <= 512, TST_ERR_UNSUPPORTED, "tst_layernorm_t_f32: rows of at most 512 elements (a 64-row block is staged in LDS)");
    TST_REQUIRE(tst_aligned16(x) && tst_aligned16(gamma) && tst_aligned16(beta), TST_ERR_ARG, "tst_layernorm_t_f32: alignment");
    if (R == 0) return TST_OK;
    hipLaunchKernelGGL(layernorm_t_kernel, dim3((unsigned)((R + 63) / 64)), dim3(256), (size_t)(64 * (d + 1) * 4), (hipStream_t)stream, x,
                       gamma, beta, eps, yT, ldt, R, (int)d);
    return tst_after_launch("tst_layernorm_t_f32");
}

namespace {

// 256 x 256 tiles on 8 waves (4 x 2, each 64 x 128 as in the 128-row tile): half the staged bytes per MFMA, one workgroup per CU
template <int BK, int NST>
void launch_plain256(const float* AT, int64_t ldat, const float* Wt, int64_t ldw, const float* bias, float* C, int64_t ldc, int64_t M,
                     int64_t N, int64_t K, hipStream_t st) {
    const int64_t tiles_m = (M + 255) / 256, tiles_n = (N + TN - 1) / TN;
    hipLaunchKernelGGL((gemm_f32_dma_kernel<256, 4, 2, BK, NST, false, false, false, true>), dim3((unsigned)(tiles_m * tiles_n)), dim3(512), 0, st,
                       AT, ldat, Wt, ldw, bias, nullptr, 0, C, ldc, (int)M, (int)N, (int)K, (int)tiles_m, (int)tiles_n);
}

template <int BK, int NST, bool NT_STORE = false>
void launch_plain128(const float* AT, int64_t ldat, const float* Wt, int64_t ldw, const float* bias, float* C, int64_t ldc, int64_t M,
                     int64_t N, int64_t K, hipStream_t st) {
    const int64_t tiles_m = (M + 127) / 128, tiles_n = (N + TN - 1) / TN;
    hipLaunchKernelGGL((gemm_f32_dma_kernel<128, 2, 2, BK, NST, false, false, false, NT_STORE>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, st,
                       AT, ldat, Wt, ldw, bias, nullptr, 0, C, ldc, (int)M, (int)N, (int)K, (int)tiles_m, (int)tiles_n);
}

}  // namespace

extern "C" int tst_gemm_dma_f32(const float* AT, int64_t ldat, const float* Wt, int64_t ldw, const float* bias, float* C, int64_t ldc,
                                int64_t M, int64_t N, int64_t K, int variant, void* stream) {
    TST_REQUIRE(tst_dims_ok({M, N, K}), TST_ERR_UNSUPPORTED, "tst_gemm_dma_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(AT && Wt && C, TST_ERR_ARG, "tst_gemm_dma_f32: null pointer");
    TST_REQUIRE(M >= 0 && N > 0 && K >= 16 && K % 16 == 0, TST_ERR_ARG, "tst_gemm_dma_f32: bad sizes (K must be a multiple of 16)");
    const int64_t tiles_m = (M + 127) / 128, tiles_n = (N + TN - 1) / TN;
    TST_REQUIRE(ldat % 4 == 0 && ldat >= tiles_m * 128 && ldw % 4 == 0 && ldw >= N && ldw >= 4 && ldc >= N, TST_ERR_ARG,
                "tst_gemm_dma_f32: ldat must be a multiple of 4 and at least M rounded up to 128; ldw %% 4 == 0, ldw >= N, ldc >= N");
    TST_REQUIRE(tst_aligned16(AT) && tst_aligned16(Wt), TST_ERR_ARG, "tst_gemm_dma_f32: AT and Wt must be 16-byte aligned");
    TST_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31) && tiles_m * tiles_n < (1ll << 31), TST_ERR_UNSUPPORTED,
                "tst_gemm_dma_f32: size");
    if (M == 0) return TST_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case 1: launch_plain128<8, 4>(AT, ldat, Wt, ldw, bias, C, ldc, M, N, K, st); break;
        case 0: launch_plain128<16, 3>(AT, ldat, Wt, ldw, bias, C, ldc, M, N, K, st); break;
        case 3: launch_plain128<8, 6, true>(AT, ldat, Wt, ldw, bias, C, ldc, M, N, K, st); break;
        case 5:     // 256-row tiles: AT must have its rows padded to a multiple of 256 columns
            TST_REQUIRE(ldat >= (M + 255) / 256 * 256, TST_ERR_ARG, "tst_gemm_dma_f32: variant 5 needs ldat >= M rounded up to 256");
            launch_plain256<8, 6>(AT, ldat, Wt, ldw, bias, C, ldc, M, N, K, st);
            break;
        case 4: {   // persistent: two workgroups per CU, a multiple of 8 so that every XCD has the same slots
            const int64_t nwg = tiles_m * tiles_n;
            const unsigned grid = (unsigned)(nwg >= 512 ? 512 : (nwg + 7) / 8 * 8);
            hipLaunchKernelGGL((gemm_f32_dma_persist_kernel<8, 6>), dim3(grid), dim3(256), 0, st, AT, ldat, Wt, ldw, bias, C, ldc, (int)M, (int)N,
                               (int)K, (int)tiles_m, (int)tiles_n);
            break;
        }
        default: launch_plain128<8, 6>(AT, ldat, Wt, ldw, bias, C, ldc, M, N, K, st); break;
    }
    return tst_after_launch("tst_gemm_dma_f32");
}
