// GRU cell step as ONE kernel: the input product x * Wih, the recurrent product h * Whh and the gate update.
//
// For a tile of 128 rows x 64 hidden columns a workgroup accumulates the six gate pre-activations of every element
// (i_r, i_z, i_n over K = E from bias_ih; h_r, h_z, h_n over K = H from bias_hh) in separate MFMA accumulators, so each is
// the same k-ordered fma chain as the stand-alone GEMM (tst_gemm_mfma_f32) produces, and applies gru_elem to them in the
// epilogue: the [R][3H] arrays gi and gh never exist in memory (24 KB per row-step at H = 1024 against 8 KB of state).
// Tiling as gemm_f32_kernel: 4 waves as 2 (M) x 2 (N), v_mfma_f32_32x32x2_f32, k-major LDS images of the A tile and of
// the three gate column blocks of the weight tile, register-staged triple buffering with the barrier in the middle of a
// k-tile.  A wave owns 64 rows x 32 columns x 6 gates = 12 MFMA tiles (192 accumulator registers): two waves per SIMD.
#include "tst_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BN = 64, BK = 16, LDA = BM + 4, LDB = BN + 4, GROUP_M = 16;

struct GruSmem {
    float As[3][BK][LDA];
    float Bs[3][3][BK][LDB];
};

// acc[i][g] += A[rows of the tile][0..K) * W[0..K)[g * H + columns of the tile], g = 0..2; K % 16 == 0.
// Called once with (x, Wih) and once with (h, Whh); the caller separates the calls with a barrier.
template <bool OUT_T>
__device__ __forceinline__ void gate_products(const float* __restrict__ A, int64_t lda, const float* __restrict__ W, int64_t ldw,
                                              int K, int H, int M, int m0, int n0, f32x16 (&acc)[2][3], GruSmem& sm) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int akq = (t & 3) * 4, ar0 = t >> 2, ar1 = (t + 256) >> 2;
    const int bnq = (t & 15) * 4, bk = t >> 4;
    const float* a_row0 = A + (int64_t)min(m0 + ar0, M - 1) * lda;
    const float* a_row1 = A + (int64_t)min(m0 + ar1, M - 1) * lda;
    const float* w_col = W + n0 + bnq;
    float4 ra0, ra1, rb0, rb1, rb2;
    auto load_tile = [&](int k0) {
        ra0 = *reinterpret_cast<const float4*>(a_row0 + k0 + akq);
        ra1 = *reinterpret_cast<const float4*>(a_row1 + k0 + akq);
        const float* wp = w_col + (int64_t)(k0 + bk) * ldw;
        rb0 = *reinterpret_cast<const float4*>(wp);
        rb1 = *reinterpret_cast<const float4*>(wp + H);
        rb2 = *reinterpret_cast<const float4*>(wp + 2 * H);
    };
    auto store_tile = [&](int buf) {
        sm.As[buf][akq + 0][ar0] = ra0.x; sm.As[buf][akq + 1][ar0] = ra0.y; sm.As[buf][akq + 2][ar0] = ra0.z; sm.As[buf][akq + 3][ar0] = ra0.w;
        sm.As[buf][akq + 0][ar1] = ra1.x; sm.As[buf][akq + 1][ar1] = ra1.y; sm.As[buf][akq + 2][ar1] = ra1.z; sm.As[buf][akq + 3][ar1] = ra1.w;
        *reinterpret_cast<float4*>(&sm.Bs[buf][0][bk][bnq]) = rb0;
        *reinterpret_cast<float4*>(&sm.Bs[buf][1][bk][bnq]) = rb1;
        *reinterpret_cast<float4*>(&sm.Bs[buf][2][bk][bnq]) = rb2;
    };
    const int nkt = K / BK;
    load_tile(0);
    store_tile(0);
    if (nkt > 1) load_tile(BK);
    __syncthreads();
    const int arow = wm * 64 + (lane & 31), bcol = wn * 32 + (lane & 31), khalf = lane >> 5;
    float a_cur[2], b_cur[3], a_nxt[2], b_nxt[3];
    auto read_operands = [&](float (&av)[2], float (&bv)[3], int bufi, int kk) {
        av[0] = sm.As[bufi][kk + khalf][arow];
        av[1] = sm.As[bufi][kk + khalf][arow + 32];
        bv[0] = sm.Bs[bufi][0][kk + khalf][bcol];
        bv[1] = sm.Bs[bufi][1][kk + khalf][bcol];
        bv[2] = sm.Bs[bufi][2][kk + khalf][bcol];
    };
    int buf = 0;
    read_operands(a_cur, b_cur, 0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int nbuf = buf == 2 ? 0 : buf + 1;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            if (kk + 2 < BK) {
                read_operands(a_nxt, b_nxt, buf, kk + 2);
            } else if (kt + 1 < nkt) {
                read_operands(a_nxt, b_nxt, nbuf, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[i][g] = OUT_T ? __builtin_amdgcn_mfma_f32_32x32x2f32(b_cur[g], a_cur[i], acc[i][g], 0, 0, 0)
                                      : __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur[g], acc[i][g], 0, 0, 0);
            a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1];
            b_cur[0] = b_nxt[0]; b_cur[1] = b_nxt[1]; b_cur[2] = b_nxt[2];
            if (kk == BK / 2 - 2) {
                if (kt + 1 < nkt) {
                    store_tile(nbuf);
                    if (kt + 2 < nkt) load_tile((kt + 2) * BK);
                }
                __syncthreads();
            }
        }
        buf = nbuf;
    }
}

// TAB: the input product is not computed but READ: gi[row][:] = GiTab[tok[row]][:], a table with one row per vocabulary entry that
// holds bias_ih + embedding[v] * Wih (made once per model by tst_gemm_mfma_f32 over the embedding matrix: the same fma chain per
// element, so the same bits).  A decoder whose layer-0 input is a plain embedding lookup then runs half the matrix work per
// step.  The lane's table values are requested before the recurrent product and arrive under it (they take the registers that
// the gi accumulators would have).
// OUT_T: the products are computed transposed (operands swapped: the same fma chain per element), so that a lane owns a ROW
// of the tile and its 16 registers are 16 columns (4 groups of 4 consecutive ones): the new state goes out as 16-byte pieces
// of rows AND a second time k-major, HoutT[column][row] (32 lanes = 128 contiguous bytes), the operand layout of the LDS-DMA
// GEMM that multiplies it with the output projection.
template <bool OUT_T, bool TAB>
__global__ __launch_bounds__(256, 2) void gru_fused_kernel(const float* __restrict__ X, int64_t ldx, int E, const float* __restrict__ Hin,
                                                           int64_t ldh, int H, const float* __restrict__ Wih, int64_t ldwi,
                                                           const float* __restrict__ bih, const float* __restrict__ Whh, int64_t ldwh,
                                                           const float* __restrict__ bhh, float* __restrict__ Hout, int64_t ldo,
                                                           float* __restrict__ HoutT, int64_t ldt, int M, int tiles_m, int tiles_n,
                                                           const int32_t* __restrict__ tok, int V) {
    // TAB: X is the table [V][ldx >= 3 H], E is unused, Wih / bih are not read
    __shared__ __attribute__((aligned(16))) GruSmem sm;
    // workgroup id -> tile: every XCD walks a contiguous, m-grouped range of tiles (as gemm_f32_kernel)
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

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    f32x16 gi[2][3], gh[2][3];
    if (OUT_T) {
        // lane owns row (lane & 31) of each 32-row tile; register r is column cbase + (r & 3) + 8 (r >> 2)
        const int cbase = n0 + wn * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bi = (!TAB && bih != nullptr) ? *reinterpret_cast<const float4*>(bih + g * H + cbase + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 bh = bhh != nullptr ? *reinterpret_cast<const float4*>(bhh + g * H + cbase + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (!TAB) { gi[i][g][4 * q] = bi.x; gi[i][g][4 * q + 1] = bi.y; gi[i][g][4 * q + 2] = bi.z; gi[i][g][4 * q + 3] = bi.w; }
                    gh[i][g][4 * q] = bh.x; gh[i][g][4 * q + 1] = bh.y; gh[i][g][4 * q + 2] = bh.z; gh[i][g][4 * q + 3] = bh.w;
                }
            }
        if (TAB) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = min(m0 + wm * 64 + i * 32 + (lane & 31), M - 1);
                const int id = min(max(tok[row], 0), V - 1);
                const float* tp = X + (int64_t)id * ldx + cbase;
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4*>(tp + g * H + 8 * q);
                        gi[i][g][4 * q] = v.x; gi[i][g][4 * q + 1] = v.y; gi[i][g][4 * q + 2] = v.z; gi[i][g][4 * q + 3] = v.w;
                    }
            }
        } else {
            gate_products<true>(X, ldx, Wih, ldwi, E, H, M, m0, n0, gi, sm);
            __syncthreads();
        }
        gate_products<true>(Hin, ldh, Whh, ldwh, H, H, M, m0, n0, gh, sm);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + wm * 64 + i * 32 + (lane & 31);
            const float* hp = Hin + (int64_t)min(row, M - 1) * ldh + cbase;
            float4 hv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) hv[q] = *reinterpret_cast<const float4*>(hp + 8 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 o;
                o.x = gru_elem(gi[i][0][4 * q], gi[i][1][4 * q], gi[i][2][4 * q], gh[i][0][4 * q], gh[i][1][4 * q], gh[i][2][4 * q], hv[q].x);
                o.y = gru_elem(gi[i][0][4 * q + 1], gi[i][1][4 * q + 1], gi[i][2][4 * q + 1], gh[i][0][4 * q + 1], gh[i][1][4 * q + 1],
                               gh[i][2][4 * q + 1], hv[q].y);
                o.z = gru_elem(gi[i][0][4 * q + 2], gi[i][1][4 * q + 2], gi[i][2][4 * q + 2], gh[i][0][4 * q + 2], gh[i][1][4 * q + 2],
                               gh[i][2][4 * q + 2], hv[q].z);
                o.w = gru_elem(gi[i][0][4 * q + 3], gi[i][1][4 * q + 3], gi[i][2][4 * q + 3], gh[i][0][4 * q + 3], gh[i][1][4 * q + 3],
                               gh[i][2][4 * q + 3], hv[q].w);
                if (row < M) {
                    *reinterpret_cast<float4*>(Hout + (int64_t)row * ldo + cbase + 8 * q) = o;
                    if (HoutT != nullptr) {   // (always set without TAB)
                        float* tp = HoutT + (int64_t)(cbase + 8 * q) * ldt + row;
                        tp[0] = o.x; tp[ldt] = o.y; tp[2 * ldt] = o.z; tp[3 * ldt] = o.w;
                    }
                }
            }
        }
        return;
    }
    const int col = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float bi = bih != nullptr ? bih[g * H + col] : 0.0f;
        const float bh = bhh != nullptr ? bhh[g * H + col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { gi[i][g][r] = bi; gh[i][g][r] = bh; }
    }
    static_assert(OUT_T || !TAB, "the table form exists in the row-per-lane layout only (16-byte table reads)");
    gate_products<false>(X, ldx, Wih, ldwi, E, H, M, m0, n0, gi, sm);
    __syncthreads();      // every wave has read its last operands before the images are overwritten
    gate_products<false>(Hin, ldh, Whh, ldwh, H, H, M, m0, n0, gh, sm);

    // lane owns column col of rows row0 + (r & 3) + 8 * (r >> 2) of each 32-row tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row0 = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
        float hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            hv[r] = Hin[(int64_t)min(row, M - 1) * ldh + col];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            const float o = gru_elem(gi[i][0][r], gi[i][1][r], gi[i][2][r], gh[i][0][r], gh[i][1][r], gh[i][2][r], hv[r]);
            if (row < M) Hout[(int64_t)row * ldo + col] = o;
        }
    }
}

}  // namespace

extern "C" int tst_gru_cell_fused_ok(int64_t E, int64_t H) { return (E >= BK && E % BK == 0 && H >= BN && H % BN == 0) ? 1 : 0; }

namespace {

int gru_cell_fused_impl(const char* who, const float* x, int64_t ldx, const float* h, int64_t ldh, const float* wih, int64_t ld_wih,
                        const float* bih, const float* whh, int64_t ld_whh, const float* bhh, float* h_out, int64_t ldo, float* h_out_t,
                        int64_t ldt, int64_t R, int64_t E, int64_t H, void* stream) {
    TST_REQUIRE(x && h && wih && whh && h_out, TST_ERR_ARG, "%s: null pointer", who);
    TST_REQUIRE(R >= 0 && R < (1ll << 31) && E > 0 && H > 0, TST_ERR_ARG, "%s: bad sizes", who);
    TST_REQUIRE(tst_gru_cell_fused_ok(E, H), TST_ERR_UNSUPPORTED,
                "%s: needs E %% 16 == 0 and H %% 64 == 0 (got E=%lld, H=%lld); use the two products and tst_gru_cell_step_f32", who,
                (long long)E, (long long)H);
    TST_REQUIRE(ldx >= E && ldh >= H && ldo >= H && ld_wih >= 3 * H && ld_whh >= 3 * H && ldx % 4 == 0 && ldh % 4 == 0 && ld_wih % 4 == 0 &&
                    ld_whh % 4 == 0, TST_ERR_ARG, "%s: leading dimensions (multiples of 4, at least the row length)", who);
    TST_REQUIRE(tst_aligned16(x) && tst_aligned16(h) && tst_aligned16(wih) && tst_aligned16(whh), TST_ERR_ARG,
                "%s: x, h, wih, whh must be 16-byte aligned", who);
    // every workgroup reads whole rows of h while others write h_out: the two must not overlap
    const char *hb = (const char*)h, *ob = (const char*)h_out;
    const int64_t hbytes = (R > 0 ? ((R - 1) * ldh + H) : 0) * 4, obytes = (R > 0 ? ((R - 1) * ldo + H) : 0) * 4;
    TST_REQUIRE(ob + obytes <= hb || hb + hbytes <= ob, TST_ERR_ARG, "%s: h_out must not overlap h", who);
    if (h_out_t != nullptr) {
        TST_REQUIRE(ldt >= R && ldo % 4 == 0 && tst_aligned16(h_out) && tst_aligned16(bih) && tst_aligned16(bhh), TST_ERR_ARG,
                    "%s: the k-major copy needs ldt >= R, ldo %% 4 == 0 and 16-byte aligned h_out, bih, bhh", who);
        const char* tb = (const char*)h_out_t;
        const int64_t tbytes = (R > 0 ? ((H - 1) * ldt + R) : 0) * 4;
        TST_REQUIRE(tb + tbytes <= hb || hb + hbytes <= tb, TST_ERR_ARG, "%s: h_out_t must not overlap h", who);
    }
    if (R == 0) return TST_OK;
    const int64_t tiles_m = (R + BM - 1) / BM, tiles_n = H / BN;
    TST_REQUIRE(tiles_m * tiles_n < (1ll << 31), TST_ERR_UNSUPPORTED, "%s: grid too large", who);
    if (h_out_t != nullptr)
        hipLaunchKernelGGL((gru_fused_kernel<true, false>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, (hipStream_t)stream, x, ldx, (int)E, h,
                           ldh, (int)H, wih, ld_wih, bih, whh, ld_whh, bhh, h_out, ldo, h_out_t, ldt, (int)R, (int)tiles_m, (int)tiles_n,
                           (const int32_t*)nullptr, 0);
    else
        hipLaunchKernelGGL((gru_fused_kernel<false, false>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, (hipStream_t)stream, x, ldx, (int)E, h,
                           ldh, (int)H, wih, ld_wih, bih, whh, ld_whh, bhh, h_out, ldo, (float*)nullptr, (int64_t)0, (int)R, (int)tiles_m,
                           (int)tiles_n, (const int32_t*)nullptr, 0);
    return tst_after_launch(who);
}

}  // namespace

extern "C" int tst_gru_cell_fused_tab_f32(const float* gi_tab, int64_t ld_tab, const int32_t* tok, int64_t V, const float* h, int64_t ldh,
                                          const float* whh, int64_t ld_whh, const float* bhh, float* h_out, int64_t ldo, float* h_out_t,
                                          int64_t ldt, int64_t R, int64_t H, void* stream) {
    TST_REQUIRE(tst_dims_ok({V, R, H}), TST_ERR_UNSUPPORTED, "tst_gru_cell_fused_tab_f32: a size exceeds 2^31 - 1");
    const char* who = "tst_gru_cell_fused_tab_f32";
    TST_REQUIRE(gi_tab && tok && h && whh && h_out, TST_ERR_ARG, "%s: null pointer", who);
    TST_REQUIRE(R >= 0 && R < (1ll << 31) && H > 0 && V > 0 && V < (1ll << 31), TST_ERR_ARG, "%s: bad sizes", who);
    TST_REQUIRE(H >= BN && H % BN == 0, TST_ERR_UNSUPPORTED, "%s: needs H %% 64 == 0 (got %lld)", who, (long long)H);
    TST_REQUIRE(ld_tab >= 3 * H && ldh >= H && ldo >= H && ld_whh >= 3 * H && ld_tab % 4 == 0 && ldh % 4 == 0 && ld_whh % 4 == 0, TST_ERR_ARG,
                "%s: leading dimensions (multiples of 4, at least the row length)", who);
    TST_REQUIRE(tst_aligned16(gi_tab) && tst_aligned16(h) && tst_aligned16(whh), TST_ERR_ARG, "%s: gi_tab, h, whh must be 16-byte aligned", who);
    const char *hb = (const char*)h, *ob = (const char*)h_out;
    const int64_t hbytes = (R > 0 ? ((R - 1) * ldh + H) : 0) * 4, obytes = (R > 0 ? ((R - 1) * ldo + H) : 0) * 4;
    TST_REQUIRE(ob + obytes <= hb || hb + hbytes <= ob, TST_ERR_ARG, "%s: h_out must not overlap h", who);
    TST_REQUIRE(ldo % 4 == 0 && tst_aligned16(h_out) && tst_aligned16(bhh), TST_ERR_ARG, "%s: ldo %% 4 == 0 and 16-byte aligned h_out, bhh", who);
    if (h_out_t != nullptr) {
        TST_REQUIRE(ldt >= R, TST_ERR_ARG, "%s: the k-major copy needs ldt >= R", who);
        const char* tb = (const char*)h_out_t;
        const int64_t tbytes = (R > 0 ? ((H - 1) * ldt + R) : 0) * 4;
        TST_REQUIRE(tb + tbytes <= hb || hb + hbytes <= tb, TST_ERR_ARG, "%s: h_out_t must not overlap h", who);
    }
    if (R == 0) return TST_OK;
    const int64_t tiles_m = (R + BM - 1) / BM, tiles_n = H / BN;
    TST_REQUIRE(tiles_m * tiles_n < (1ll << 31), TST_ERR_UNSUPPORTED, "%s: grid too large", who);
    // the row-per-lane layout whether or not the k-major copy is wanted (h_out_t may be NULL)
    hipLaunchKernelGGL((gru_fused_kernel<true, true>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, (hipStream_t)stream, gi_tab, ld_tab,
                       0, h, ldh, (int)H, (const float*)nullptr, (int64_t)0, (const float*)nullptr, whh, ld_whh, bhh, h_out, ldo, h_out_t, ldt,
                       (int)R, (int)tiles_m, (int)tiles_n, tok, (int)V);
    return tst_after_launch(who);
}

extern "C" int tst_gru_cell_fused_f32(const float* x, int64_t ldx, const float* h, int64_t ldh, const float* wih, int64_t ld_wih,
                                      const float* bih, const float* whh, int64_t ld_whh, const float* bhh, float* h_out,
                                      int64_t ldo, int64_t R, int64_t E, int64_t H, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, E, H}), TST_ERR_UNSUPPORTED, "tst_gru_cell_fused_f32: a size exceeds 2^31 - 1");
    return gru_cell_fused_impl("tst_gru_cell_fused_f32", x, ldx, h, ldh, wih, ld_wih, bih, whh, ld_whh, bhh, h_out, ldo, nullptr, 0, R, E, H,
                               stream);
}

extern "C" int tst_gru_cell_fused_t_f32(const float* x, int64_t ldx, const float* h, int64_t ldh, const float* wih, int64_t ld_wih,
                                        const float* bih, const float* whh, int64_t ld_whh, const float* bhh, float* h_out,
                                        int64_t ldo, float* h_out_t, int64_t ldt, int64_t R, int64_t E, int64_t H, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, E, H}), TST_ERR_UNSUPPORTED, "tst_gru_cell_fused_t_f32: a size exceeds 2^31 - 1");
    TST_REQUIRE(h_out_t != nullptr, TST_ERR_ARG, "tst_gru_cell_fused_t_f32: null h_out_t");
    return gru_cell_fused_impl("tst_gru_cell_fused_t_f32", x, ldx, h, ldh, wih, ld_wih, bih, whh, ld_whh, bhh, h_out, ldo, h_out_t, ldt, R,
                               E, H, stream);
}
