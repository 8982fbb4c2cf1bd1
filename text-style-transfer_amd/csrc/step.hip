// Host-side executors of a decode step: the launch sequence of a whole model step (transformer stack, GRU stack) and of
// the selection / retirement tail, each issued from native code in one call, so that small batches are not bound by
// per-launch interpreter overhead.  Contains no kernels; every launch goes through the entry points of this library.
#include "tst_common.h"

namespace {

struct StepCtx {
    const tst_tfm_step* m;
    int64_t R;
    void* stream;
    bool bf16;

    int layernorm(const float* x, const float* g, const float* b, void* y) const {
        return bf16 ? tst_layernorm_f32_bf16(x, g, b, m->eps, y, R, m->d, stream)
                    : tst_layernorm_f32(x, g, b, m->eps, (float*)y, R, m->d, stream);
    }
    // out = epilogue(LayerNorm(x) * w + bias) with the output in the activation type; fp32: one entry point (one kernel
    // when the problem is small), bf16: LayerNorm into h, then the product
    int ln_gemm(const float* g, const float* b, const void* w, int64_t ldw, const float* bias, void* out, int64_t ldo, int64_t N,
                int act) const {
        if (!bf16)
            return tst_ln_gemm_f32(m->x, g, b, m->eps, (float*)m->h, (const float*)w, ldw, bias, (float*)out, ldo, R, N, m->d, act,
                                   stream);
        const int rc = layernorm(m->x, g, b, m->h);
        if (rc != TST_OK) return rc;
        return gemm(m->h, m->d, w, ldw, bias, nullptr, 0, out, ldo, N, m->d, act, true);
    }
    // out = epilogue(a * w + bias); out_lo: the output has the activation type (bf16 when the model is), else fp32
    int gemm(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const float* res, int64_t ldr, void* out,
             int64_t ldo, int64_t N, int64_t K, int act, bool out_lo) const {
        return bf16 ? tst_gemm_mfma_bf16(a, lda, w, ldw, bias, res, ldr, out, ldo, R, N, K, act, out_lo ? 1 : 0, stream)
                    : tst_gemm_mfma_f32((const float*)a, lda, (const float*)w, ldw, bias, res, ldr, (float*)out, ldo, R, N, K, act,
                                        stream);
    }
};

// TST_TFM_CHAIN=1 in the environment (measurements): large fp32 steps of a d_model = 256 decoder with cross-attention run the
// residual product -> LayerNorm -> next product chains as one kernel each (tst_chain_f32).  Same results; measured slower
// than the separate launches at most row counts (a 64-row tile per CU: rounds of 16384 rows), hence off by default.
// The fp32 vocabulary projection of a large step.  d_model = 256: the kernel that holds the weights in registers
// (gemm_f32_wreg.hip; row-major final LayerNorm; 124-126 TFLOP/s inside the decode loop) from TST_WREG_MIN_ROWS rows on.  Other
// widths: the final LayerNorm written k-major and both operands staged by LDS-DMA (tst_gemm_dma_f32, variant 3; 118-120 TFLOP/s
// at d_model 256) from TST_DMA_MIN_ROWS rows on.  TST_VOCAB_VARIANT=<n> in the environment (measurements): 0..4 = that
// tst_gemm_dma_f32 variant for every width, 5 = the default.
int tst_vocab_variant() {
    static const int v = [] { const char* e = getenv("TST_VOCAB_VARIANT"); return (e != nullptr && e[0] >= '0' && e[0] <= '5') ? e[0] - '0' : 5; }();
    return v;
}

// TST_BF16_LN_FUSE=<rows> in the environment (measurements): bf16 steps of a model_size = 512 decoder without cross-attention with
// at least that many rows run every residual product together with the LayerNorm behind it (tst_gemm_res_ln_bf16).  0 / unset: never.
int64_t tst_bf16_ln_fuse_rows() {
    static const int64_t v = [] { const char* e = getenv("TST_BF16_LN_FUSE"); return e != nullptr ? (int64_t)atoll(e) : 0; }();
    return v;
}

bool tst_tfm_chain() {
    static const bool v = [] { const char* e = getenv("TST_TFM_CHAIN"); return e != nullptr && e[0] == '1'; }();
    return v;
}

}  // namespace

#define TST_TRY(expr)               \
    do {                            \
        const int rc_ = (expr);     \
        if (rc_ != TST_OK) return rc_; \
    } while (0)

extern "C" int tst_tfm_step_run(const tst_tfm_step* m, const int32_t* tok, const int32_t* anc, const int32_t* sent, int64_t R,
                                int64_t t, int phases, void* stream) {
    TST_REQUIRE(tst_dims_ok({R, t}), TST_ERR_UNSUPPORTED, "tst_tfm_step_run: a size exceeds 2^31 - 1");
    TST_REQUIRE(m != nullptr && m->layers != nullptr, TST_ERR_ARG, "tst_tfm_step_run: null plan");
    TST_REQUIRE(m->n_layers >= 1 && m->d > 0 && m->heads > 0 && m->d_ff > 0 && m->V > 0, TST_ERR_ARG, "tst_tfm_step_run: bad sizes");
    TST_REQUIRE(R >= 0 && R <= m->slab_rows, TST_ERR_ARG, "tst_tfm_step_run: %lld rows, buffers hold %lld", (long long)R,
                (long long)m->slab_rows);
    TST_REQUIRE(t >= 0 && t < m->n_pos, TST_ERR_ARG, "tst_tfm_step_run: position %lld outside 0..%lld", (long long)t,
                (long long)m->n_pos - 1);
    TST_REQUIRE((phases & ~7) == 0 && phases != 0, TST_ERR_ARG, "tst_tfm_step_run: bad phases %d", phases);
    TST_REQUIRE(m->x && m->h && m->a && m->ff && m->logits, TST_ERR_ARG, "tst_tfm_step_run: null work buffer");
    if (R == 0) return TST_OK;
    const int64_t d = m->d;
    const StepCtx c{m, R, stream, m->bf16 != 0};
    const int64_t esz = c.bf16 ? 2 : 4;
    // large fp32 steps: the final LayerNorm writes its rows k-major and the vocabulary projection stages both operands by LDS-DMA
    // greedy decoding of a small fp32 batch: the step ends in the arg-max keys of its logits (tst_ln_vocab_argmax_f32)
    const bool keys_out = m->argmax_keys != nullptr;
    TST_REQUIRE(!keys_out || (!c.bf16 && phases == (TST_STEP_EMBED | TST_STEP_LAYERS | TST_STEP_VOCAB) && tst_vocab_argmax_ok(R, m->V, d)),
                TST_ERR_UNSUPPORTED, "tst_tfm_step_run: argmax_keys needs fp32, all three phases and a shape tst_vocab_argmax_ok accepts");
    const bool wreg_vocab = !keys_out && !c.bf16 && tst_vocab_variant() == 5 && !tst_tfm_chain() && R >= TST_WREG_MIN_ROWS &&
                            tst_gemm_wreg_ok(R, m->V, d);
    const bool dma_vocab = !wreg_vocab && !keys_out && !c.bf16 && m->ht != nullptr && R >= TST_DMA_MIN_ROWS && d % 16 == 0 && d <= 512;
    TST_REQUIRE(!dma_vocab || (m->ld_ht % 4 == 0 && m->ld_ht >= (R + 127) / 128 * 128), TST_ERR_ARG,
                "tst_tfm_step_run: ld_ht must be a multiple of 4 and at least the rows rounded up to 128");

    if (phases & TST_STEP_EMBED) {
        TST_REQUIRE(tok && m->emb && m->pe, TST_ERR_ARG, "tst_tfm_step_run: null embedding input");
        if (keys_out) {
            TST_REQUIRE(m->z == nullptr || sent != nullptr, TST_ERR_ARG, "tst_tfm_step_run: z without sent");
            TST_TRY(tst_embed_posadd_clear_f32(tok, m->emb, m->pe + t * d, m->scale, m->z, sent, m->x, m->argmax_keys, R, d, m->V, stream));
        } else if (m->z != nullptr) {   // embedding, positional row and the sentence's conditioning vector in one pass
            TST_REQUIRE(sent != nullptr, TST_ERR_ARG, "tst_tfm_step_run: z without sent");
            TST_TRY(tst_embed_posadd_rows_f32(tok, m->emb, m->pe + t * d, nullptr, m->scale, m->z, sent, m->x, R, d, m->V, stream));
        } else {
            TST_TRY(tst_embed_posadd_f32(tok, m->emb, m->pe + t * d, nullptr, m->scale, m->x, R, d, m->V, stream));
        }
    }
    if (phases & TST_STEP_LAYERS) {
        TST_REQUIRE(anc != nullptr, TST_ERR_ARG, "tst_tfm_step_run: null ancestor table");
        bool chain = dma_vocab && tst_tfm_chain() && R >= 8192 && tst_chain_ok(R, d, d, d) && tst_chain_ok(R, d, d, m->d_ff) &&
                     tst_chain_ok(R, d, m->d_ff, 3 * d) && tst_chain_ok(R, d, m->d_ff, 0);
        for (int li = 0; chain && li < m->n_layers; ++li) chain = m->layers[li].cwq != nullptr;
        for (int li = 0; chain && li < m->n_layers; ++li) {
            const tst_tfm_layer& L = m->layers[li];
            TST_REQUIRE(L.cache && L.wqkv && L.wo && L.w1 && L.w2 && L.kv_cross && L.cwo && m->q && sent, TST_ERR_ARG,
                        "tst_tfm_step_run: layer %d incomplete", li);
            float* slab = (float*)L.cache + t * m->slab_rows * 3 * d;
            float* xf = m->x;
            if (li == 0) TST_TRY(c.ln_gemm(L.ln_self_g, L.ln_self_b, L.wqkv, L.ld_wqkv, L.bqkv, slab, 3 * d, 3 * d, 0));
            TST_TRY(tst_attn_self_step_f32((const float*)L.cache, m->slab_rows, anc, m->anc_ld, t, (float*)m->a, R, d, m->heads, stream));
            TST_TRY(tst_chain_f32((const float*)m->a, d, d, (const float*)L.wo, L.ld_wo, L.bo, xf, L.ln_cross_g, L.ln_cross_b, m->eps,
                                  (const float*)L.cwq, L.ld_cwq, L.cbq, 0, (float*)m->q, d, d, nullptr, 0, R, d, stream));
            if (m->group > 1 && R % m->group == 0)
                TST_TRY(tst_attn_cross_group_f32((const float*)m->q, (const float*)L.kv_cross, m->mask, sent, m->kvrow, m->S, (float*)m->a, R,
                                                 d, m->heads, m->group, stream));
            else if (m->kvrow != nullptr)
                TST_TRY(tst_attn_cross_packed_f32((const float*)m->q, (const float*)L.kv_cross, m->mask, sent, m->kvrow, m->S, (float*)m->a, R,
                                                  d, m->heads, stream));
            else
                TST_TRY(tst_attn_cross_step_f32((const float*)m->q, (const float*)L.kv_cross, m->mask, sent, m->S, (float*)m->a, R, d,
                                                m->heads, stream));
            TST_TRY(tst_chain_f32((const float*)m->a, d, d, (const float*)L.cwo, L.ld_cwo, L.cbo, xf, L.ln_ff_g, L.ln_ff_b, m->eps,
                                  (const float*)L.w1, L.ld_w1, L.b1, 1, (float*)m->ff, m->d_ff, m->d_ff, nullptr, 0, R, d, stream));
            if (li + 1 < m->n_layers) {
                const tst_tfm_layer& Ln = m->layers[li + 1];
                TST_REQUIRE(Ln.cache && Ln.wqkv, TST_ERR_ARG, "tst_tfm_step_run: layer %d incomplete", li + 1);
                float* slab_n = (float*)Ln.cache + t * m->slab_rows * 3 * d;
                TST_TRY(tst_chain_f32((const float*)m->ff, m->d_ff, m->d_ff, (const float*)L.w2, L.ld_w2, L.b2, xf, Ln.ln_self_g, Ln.ln_self_b,
                                      m->eps, (const float*)Ln.wqkv, Ln.ld_wqkv, Ln.bqkv, 0, slab_n, 3 * d, 3 * d, nullptr, 0, R, d, stream));
            } else {
                TST_TRY(tst_chain_f32((const float*)m->ff, m->d_ff, m->d_ff, (const float*)L.w2, L.ld_w2, L.b2, xf, m->ln_g, m->ln_b, m->eps,
                                      nullptr, 0, nullptr, 0, nullptr, 0, 0, m->ht, m->ld_ht, R, d, stream));
            }
        }
        bool ln_fuse = c.bf16 && tst_bf16_ln_fuse_rows() > 0 && R >= tst_bf16_ln_fuse_rows() && tst_gemm_res_ln_ok(R, d, d) &&
                       tst_gemm_res_ln_ok(R, d, m->d_ff);
        for (int li = 0; ln_fuse && li < m->n_layers; ++li) ln_fuse = m->layers[li].cwq == nullptr;
        for (int li = 0; ln_fuse && li < m->n_layers; ++li) {
            // h always holds LayerNorm(x) for the next product: written by the previous residual product, or (layer 0) by a launch
            const tst_tfm_layer& L = m->layers[li];
            TST_REQUIRE(L.cache && L.wqkv && L.wo && L.w1 && L.w2, TST_ERR_ARG, "tst_tfm_step_run: layer %d incomplete", li);
            char* slab = (char*)L.cache + t * m->slab_rows * 3 * d * esz;
            if (li == 0) TST_TRY(c.layernorm(m->x, L.ln_self_g, L.ln_self_b, m->h));
            TST_TRY(c.gemm(m->h, d, L.wqkv, L.ld_wqkv, L.bqkv, nullptr, 0, slab, 3 * d, 3 * d, d, 0, true));
            TST_TRY(tst_attn_self_step_bf16(L.cache, m->slab_rows, anc, m->anc_ld, t, m->a, R, d, m->heads, stream));
            TST_TRY(tst_gemm_res_ln_bf16(m->a, d, L.wo, L.ld_wo, L.bo, m->x, L.ln_ff_g, L.ln_ff_b, m->eps, m->h, d, R, d, d, stream));
            TST_TRY(c.gemm(m->h, d, L.w1, L.ld_w1, L.b1, nullptr, 0, m->ff, m->d_ff, m->d_ff, d, 1, true));
            const bool last = li + 1 == m->n_layers;
            TST_TRY(tst_gemm_res_ln_bf16(m->ff, m->d_ff, L.w2, L.ld_w2, L.b2, m->x, last ? m->ln_g : m->layers[li + 1].ln_self_g,
                                         last ? m->ln_b : m->layers[li + 1].ln_self_b, m->eps, m->h, d, R, d, m->d_ff, stream));
        }
        for (int li = 0; !chain && !ln_fuse && li < m->n_layers; ++li) {
            const tst_tfm_layer& L = m->layers[li];
            TST_REQUIRE(L.cache && L.wqkv && L.wo && L.w1 && L.w2, TST_ERR_ARG, "tst_tfm_step_run: layer %d incomplete", li);
            char* slab = (char*)L.cache + t * m->slab_rows * 3 * d * esz;   // the rows of position t
            TST_TRY(c.ln_gemm(L.ln_self_g, L.ln_self_b, L.wqkv, L.ld_wqkv, L.bqkv, slab, 3 * d, 3 * d, 0));
            TST_TRY(c.bf16 ? tst_attn_self_step_bf16(L.cache, m->slab_rows, anc, m->anc_ld, t, m->a, R, d, m->heads, stream)
                           : tst_attn_self_step_f32((const float*)L.cache, m->slab_rows, anc, m->anc_ld, t, (float*)m->a, R, d,
                                                    m->heads, stream));
            TST_TRY(c.gemm(m->a, d, L.wo, L.ld_wo, L.bo, m->x, d, m->x, d, d, d, 0, false));
            if (L.cwq != nullptr) {
                TST_REQUIRE(L.kv_cross && L.cwo && m->q && sent, TST_ERR_ARG, "tst_tfm_step_run: layer %d cross block incomplete", li);
                TST_TRY(c.ln_gemm(L.ln_cross_g, L.ln_cross_b, L.cwq, L.ld_cwq, L.cbq, m->q, d, d, 0));
                // the caller says that the rows are the beams of a beam search, m->group adjacent rows per sentence
                const bool grouped = !c.bf16 && m->group > 1 && R % m->group == 0;
                if (grouped) {
                    TST_TRY(tst_attn_cross_group_f32((const float*)m->q, (const float*)L.kv_cross, m->mask, sent, m->kvrow, m->S,
                                                     (float*)m->a, R, d, m->heads, m->group, stream));
                } else if (m->kvrow != nullptr) {
                    TST_REQUIRE(!c.bf16 && m->mask != nullptr, TST_ERR_ARG, "tst_tfm_step_run: a packed k|v needs fp32 and a mask");
                    TST_TRY(tst_attn_cross_packed_f32((const float*)m->q, (const float*)L.kv_cross, m->mask, sent, m->kvrow, m->S,
                                                      (float*)m->a, R, d, m->heads, stream));
                } else {
                    TST_TRY(c.bf16 ? tst_attn_cross_step_bf16(m->q, L.kv_cross, m->mask, sent, m->S, m->a, R, d, m->heads, stream)
                                   : tst_attn_cross_step_f32((const float*)m->q, (const float*)L.kv_cross, m->mask, sent, m->S,
                                                             (float*)m->a, R, d, m->heads, stream));
                }
                TST_TRY(c.gemm(m->a, d, L.cwo, L.ld_cwo, L.cbo, m->x, d, m->x, d, d, d, 0, false));
            }
            TST_TRY(c.ln_gemm(L.ln_ff_g, L.ln_ff_b, L.w1, L.ld_w1, L.b1, m->ff, m->d_ff, m->d_ff, 1));
            TST_TRY(c.gemm(m->ff, m->d_ff, L.w2, L.ld_w2, L.b2, m->x, d, m->x, d, d, m->d_ff, 0, false));
        }
        if (chain || ln_fuse || keys_out) {}   // the last chain wrote the final LayerNorm k-major / the last residual product wrote it
                                               // into h / the arg-max kernel normalises the rows itself
        else if (dma_vocab) TST_TRY(tst_layernorm_t_f32(m->x, m->ln_g, m->ln_b, m->eps, m->ht, m->ld_ht, R, d, stream));
        else TST_TRY(c.layernorm(m->x, m->ln_g, m->ln_b, m->h));
    }
    if (phases & TST_STEP_VOCAB) {
        TST_REQUIRE(m->wout != nullptr, TST_ERR_ARG, "tst_tfm_step_run: null output projection");
        if (keys_out)
            TST_TRY(tst_ln_vocab_argmax_f32(m->x, m->ln_g, m->ln_b, m->eps, (const float*)m->wout, m->ld_wout, m->bout, m->argmax_keys, R, m->V,
                                            d, stream));
        else if (wreg_vocab) {
            TST_REQUIRE(m->ld_wout >= m->V && m->ld_logits >= m->V && tst_aligned16(m->h), TST_ERR_ARG,
                        "tst_tfm_step_run: ld_wout and ld_logits must be at least V, h 16-byte aligned");
            tst_launch_gemm_wreg((const float*)m->h, d, (const float*)m->wout, m->ld_wout, m->bout, nullptr, 0, (float*)m->logits, m->ld_logits, R,
                                 m->V, 0, 1, (hipStream_t)stream);
            TST_TRY(tst_after_launch("tst_tfm_step_run: vocabulary projection"));
        } else if (dma_vocab)
            TST_TRY(tst_gemm_dma_f32(m->ht, m->ld_ht, (const float*)m->wout, m->ld_wout, m->bout, (float*)m->logits, m->ld_logits, R, m->V, d,
                                     tst_vocab_variant() == 5 ? 3 : tst_vocab_variant(), stream));   // nontemporal stores: the logits leave the caches before they are read
        else if (c.bf16 && m->part != nullptr && tst_vocab_stats_ok(R, m->V, d))
            // the selection pass of a beam search reads these statistics instead of the rows (tst_select_step_run)
            TST_TRY(tst_vocabproj_stats_bf16(m->h, d, m->wout, m->ld_wout, m->bout, m->logits, m->ld_logits, R, m->V, d, m->part,
                                             m->n_slices, stream));
        else
            TST_TRY(c.gemm(m->h, d, m->wout, m->ld_wout, m->bout, nullptr, 0, m->logits, m->ld_logits, m->V, d, 0, true));
    }
    return TST_OK;
}

extern "C" int tst_gru_step_run(const tst_gru_step* m, const int32_t* tok, const int32_t* sent, int64_t R, int phases, void* stream) {
    TST_REQUIRE(tst_dims_ok({R}), TST_ERR_UNSUPPORTED, "tst_gru_step_run: a size exceeds 2^31 - 1");
    TST_REQUIRE(m != nullptr && m->layers != nullptr, TST_ERR_ARG, "tst_gru_step_run: null plan");
    TST_REQUIRE(m->n_layers >= 1 && m->E > 0 && m->H > 0 && m->V > 0, TST_ERR_ARG, "tst_gru_step_run: bad sizes");
    TST_REQUIRE(R >= 0 && R <= m->rows, TST_ERR_ARG, "tst_gru_step_run: %lld rows, buffers hold %lld", (long long)R, (long long)m->rows);
    TST_REQUIRE((phases & ~15) == 0 && (phases & 7) != 0, TST_ERR_ARG, "tst_gru_step_run: bad phases %d", phases);
    const bool fused = (phases & TST_STEP_FUSED_CELL) != 0;
    TST_REQUIRE(!fused || tst_gru_cell_fused_ok(m->E, m->H), TST_ERR_UNSUPPORTED,
                "tst_gru_step_run: TST_STEP_FUSED_CELL needs E %% 16 == 0 and H %% 64 == 0");
    TST_REQUIRE(m->x && m->logits && (fused || (m->gi && m->gh)), TST_ERR_ARG, "tst_gru_step_run: null work buffer");
    TST_REQUIRE(!fused || m->gi_tab == nullptr || m->H % 64 == 0, TST_ERR_UNSUPPORTED, "tst_gru_step_run: gi_tab with TST_STEP_FUSED_CELL needs H %% 64 == 0");
    if (R == 0) return TST_OK;
    // layer 0 reads its input pre-activations from the per-token table instead of multiplying the embeddings (which are then
    // not needed at all): token inputs, no per-sentence vector added to them
    const bool use_tab = m->gi_tab != nullptr && (phases & TST_STEP_EMBED) && (phases & TST_STEP_LAYERS) && m->extra == nullptr;
    TST_REQUIRE(!use_tab || (tok != nullptr && m->ld_gi_tab >= 3 * (int64_t)m->H), TST_ERR_ARG, "tst_gru_step_run: gi_tab needs tok and ld_gi_tab >= 3 H");
    if ((phases & TST_STEP_EMBED) && !use_tab) {
        TST_REQUIRE(tok && m->emb, TST_ERR_ARG, "tst_gru_step_run: null embedding input");
        TST_TRY(tst_embed_posadd_f32(tok, m->emb, nullptr, nullptr, 1.0f, m->x, R, m->E, m->V, stream));
    }
    const tst_gru_layer& top = m->layers[m->n_layers - 1];
    const float* top_h = top.h;   // the state the output projection reads
    // large fused steps with a wide vocabulary: the top cell also writes its state k-major and the projection stages both
    // operands by LDS-DMA
    const bool dma_vocab = fused && m->ht != nullptr && (phases & TST_STEP_LAYERS) && (phases & TST_STEP_VOCAB) && R >= TST_DMA_MIN_ROWS &&
                           m->V >= 2048 && m->H % 16 == 0;
    TST_REQUIRE(!dma_vocab || (m->ld_ht % 4 == 0 && m->ld_ht >= (R + 127) / 128 * 128), TST_ERR_ARG,
                "tst_gru_step_run: ld_ht must be a multiple of 4 and at least the rows rounded up to 128");
    if (phases & TST_STEP_LAYERS) {
        if (m->extra != nullptr) {
            TST_REQUIRE(sent != nullptr, TST_ERR_ARG, "tst_gru_step_run: extra without sent");
            TST_TRY(tst_add_rows_f32(m->x, m->extra, sent, R, m->E, stream));
        }
        const float* x = m->x;
        int64_t in = m->E;
        for (int l = 0; l < m->n_layers; ++l) {
            const tst_gru_layer& L = m->layers[l];
            TST_REQUIRE(L.wih && L.whh && L.h, TST_ERR_ARG, "tst_gru_step_run: layer %d incomplete", l);
            if (fused) {
                TST_REQUIRE(L.h_alt != nullptr, TST_ERR_ARG, "tst_gru_step_run: layer %d has no h_alt", l);
                if (use_tab && l == 0) {
                    const bool kmajor = dma_vocab && m->n_layers == 1;
                    TST_TRY(tst_gru_cell_fused_tab_f32(m->gi_tab, m->ld_gi_tab, tok, m->V, L.h, m->H, L.whh, L.ld_whh, L.bhh, L.h_alt, m->H,
                                                       kmajor ? m->ht : nullptr, kmajor ? m->ld_ht : 0, R, m->H, stream));
                } else if (dma_vocab && l == m->n_layers - 1)
                    TST_TRY(tst_gru_cell_fused_t_f32(x, in, L.h, m->H, L.wih, L.ld_wih, L.bih, L.whh, L.ld_whh, L.bhh, L.h_alt, m->H, m->ht,
                                                     m->ld_ht, R, in, m->H, stream));
                else
                    TST_TRY(tst_gru_cell_fused_f32(x, in, L.h, m->H, L.wih, L.ld_wih, L.bih, L.whh, L.ld_whh, L.bhh, L.h_alt, m->H, R, in,
                                                   m->H, stream));
                x = L.h_alt;
                in = m->H;
                continue;
            }
            if (!(use_tab && l == 0))
                TST_TRY(tst_gemm_mfma_f32(x, in, L.wih, L.ld_wih, L.bih, nullptr, 0, m->gi, 3 * m->H, R, 3 * m->H, in, 0, stream));
            TST_TRY(tst_gemm_mfma_f32(L.h, m->H, L.whh, L.ld_whh, L.bhh, nullptr, 0, m->gh, 3 * m->H, R, 3 * m->H, m->H, 0, stream));
            if (use_tab && l == 0) TST_TRY(tst_gru_cell_step_tab_f32(m->gi_tab, m->ld_gi_tab, tok, m->V, m->gh, L.h, L.h, R, m->H, stream));
            else TST_TRY(tst_gru_cell_step_f32(m->gi, m->gh, L.h, L.h, R, m->H, stream));
            x = L.h;
            in = m->H;
        }
    }
    if (fused && (phases & TST_STEP_LAYERS)) top_h = top.h_alt;
    if (phases & TST_STEP_VOCAB) {
        TST_REQUIRE(m->wout != nullptr, TST_ERR_ARG, "tst_gru_step_run: null output projection");
        if (dma_vocab)
            TST_TRY(tst_gemm_dma_f32(m->ht, m->ld_ht, m->wout, m->ld_wout, m->bout, m->logits, m->ld_logits, R, m->V, m->H, 3, stream));
        else
            TST_TRY(tst_gemm_mfma_f32(top_h, m->H, m->wout, m->ld_wout, m->bout, nullptr, 0, m->logits, m->ld_logits, R, m->V, m->H, 0,
                                      stream));
    }
    return TST_OK;
}

extern "C" int tst_gru_reorder_run(const tst_gru_step* m, const int32_t* row_parent, int64_t R, void* stream) {
    TST_REQUIRE(tst_dims_ok({R}), TST_ERR_UNSUPPORTED, "tst_gru_reorder_run: a size exceeds 2^31 - 1");
    TST_REQUIRE(m != nullptr && m->layers != nullptr && row_parent != nullptr, TST_ERR_ARG, "tst_gru_reorder_run: null pointer");
    TST_REQUIRE(R >= 0 && R <= m->rows, TST_ERR_ARG, "tst_gru_reorder_run: %lld rows, buffers hold %lld", (long long)R,
                (long long)m->rows);
    for (int l = 0; l < m->n_layers; ++l) {
        TST_REQUIRE(m->layers[l].h && m->layers[l].h_alt, TST_ERR_ARG, "tst_gru_reorder_run: layer %d incomplete", l);
        TST_TRY(tst_gather_rows_f32(m->layers[l].h, row_parent, m->layers[l].h_alt, R, m->H, stream));
    }
    return TST_OK;
}

extern "C" int tst_select_step_run(const tst_select_step* s, const tst_rows* cur, const tst_rows* nxt, int64_t n_sent, int64_t k_in,
                                   int64_t step, uint32_t sample_step, int last_step, void* stream) {
    TST_REQUIRE(tst_dims_ok({n_sent, k_in, step}), TST_ERR_UNSUPPORTED, "tst_select_step_run: a size exceeds 2^31 - 1");
    TST_REQUIRE(s != nullptr && cur != nullptr && nxt != nullptr, TST_ERR_ARG, "tst_select_step_run: null pointer");
    TST_REQUIRE(s->k >= 1 && s->k <= 8 && (k_in == 1 || k_in == s->k) && n_sent >= 0, TST_ERR_ARG, "tst_select_step_run: bad sizes");
    const int64_t R = n_sent * k_in;
    if (s->k == 1) {
        TST_REQUIRE(s->am_keys == nullptr || !s->sample, TST_ERR_ARG, "tst_select_step_run: am_keys and sampling cannot be combined");
        if (s->am_keys != nullptr) {
            // the step left arg-max keys instead of logits
        } else if (s->sample) {
            TST_TRY(s->bf16_logits ? tst_sample_rows_bf16(s->logits, s->ld_logits, R, s->V, s->temperature, s->seed, sample_step,
                                                          cur->sent, s->am, stream)
                                   : tst_sample_rows_f32((const float*)s->logits, s->ld_logits, R, s->V, s->temperature, s->seed,
                                                         sample_step, cur->sent, s->am, stream));
        } else {
            TST_TRY(s->bf16_logits ? tst_argmax_rows_bf16(s->logits, s->ld_logits, R, s->V, s->am, stream)
                                   : tst_argmax_rows_f32((const float*)s->logits, s->ld_logits, R, s->V, s->am, stream));
        }
        return tst_greedy_retire_compact_bounded(s->am_keys != nullptr ? nullptr : s->am, cur, nxt, s->sel_parent, s->sel_tok, s->sel_score,
                                                 s->sel_fin, s->sel_nemit, s->lp, n_sent, step, s->T, s->eos, last_step, s->out_tok,
                                                 s->out_len, s->out_score, s->scratch_alive_pos, s->row_parent, s->n_alive, s->n_sent_dev,
                                                 s->am_keys, s->n_alive_host, stream);
    } else {
        if (s->bf16_logits && s->part != nullptr && tst_vocab_stats_ok(R, s->V, s->stats_k))
            TST_TRY(tst_topk_from_stats_bf16(s->logits, s->ld_logits, s->part, s->n_slices, R, s->V, s->k, s->top_val, s->top_idx, s->lse,
                                             stream));
        else
            TST_TRY(s->bf16_logits ? tst_logsoftmax_topk_bf16(s->logits, s->ld_logits, R, s->V, s->k, s->top_val, s->top_idx, s->lse, stream)
                                   : tst_logsoftmax_topk_f32((const float*)s->logits, s->ld_logits, R, s->V, s->k, s->top_val, s->top_idx,
                                                             s->lse, stream));
        TST_TRY(tst_beam_merge(cur, s->top_val, s->top_idx, s->lse, n_sent, s->k, k_in, s->eos, s->sel_parent, s->sel_tok, s->sel_score,
                               s->sel_fin, s->sel_nemit, stream));
    }
    return tst_retire_compact_bounded(cur, nxt, s->sel_parent, s->sel_tok, s->sel_score, s->sel_fin, s->sel_nemit, s->lp, n_sent, s->k, step,
                                      s->T, s->eos, last_step, s->top_ends, s->out_tok, s->out_len, s->out_score, s->scratch_alive_pos,
                                      s->row_parent, s->n_alive, s->n_sent_dev, s->n_alive_host, stream);
}
