/* C ABI of libtst_hip.so: the gfx950 kernels behind the batched decode loop.
 *
 * The reference has no native code and no FFI: its decode loop (/root/reference/src/inference.py:11-181)
 * calls a PyTorch model once per generated token and picks tokens with torch ops.  Each entry point
 * below replaces the torch op(s) named in its comment; the Python package binds them with ctypes.
 *
 * Conventions: plain C; every pointer is a device pointer owned by the caller unless marked "host";
 * no allocation, no synchronisation and no global mutable state inside (one thread per stream is
 * safe); sizes are int64_t; the last argument is the hipStream_t to launch on (passed as void*);
 * return value 0 = launched, negative = error (see TST_ERR_*), never an exception.
 * Row counts, widths, vocabulary sizes and other counts may be at most 2^31 - 1 (TST_ERR_UNSUPPORTED beyond: the
 * kernels count rows and columns in 32-bit integers); leading dimensions and whole buffers may be larger, every
 * offset is formed in 64 bits (batches of 280 000 rows x 32000 logits are part of the tests).
 * The _f32 entry points compute in fp32 with a fixed operation order ("canonical arithmetic"): a GEMM output
 * is acc = bias[n]; for k ascending: acc = fma(a[m][k], w[k][n], acc); row reductions use interleaved
 * partial sums combined by an xor-butterfly; exp/log/sigmoid/tanh are fixed polynomials.
 * The _bf16 entry points take bf16 storage (passed as void*: 2-byte elements, the upper half of the fp32
 * bit pattern, rounded to nearest even) and do their arithmetic in fp32: sums of bf16 products on the
 * matrix cores (v_mfma_f32_32x32x16_bf16) accumulate in fp32; results are rounded to bf16 once, on store.
 */
#ifndef TST_HIP_H
#define TST_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TST_OK 0
#define TST_ERR_ARG (-1)         /* null pointer, bad size or alignment */
#define TST_ERR_LAUNCH (-2)      /* the HIP runtime refused the launch */
#define TST_ERR_UNSUPPORTED (-3) /* shape outside what the kernel was built for */

/* ABI version of this header (major * 100 + minor): 101 = 100 + tst_gemm_skinny_f32, tst_attn_cross_packed_f32,
 * tst_tfm_step_run, tst_gru_step_run, tst_gru_reorder_run, tst_select_step_run,
 * tst_attn_cross_group_f32, tst_ln_gemm_f32,
 * tst_greedy_retire_compact; 102 = 101 + tst_retire_compact_noalloc;
 * 103 = 102 + tst_gru_cell_fused_f32, tst_gru_cell_fused_ok, TST_STEP_FUSED_CELL;
 * 104 = 103 + tst_gemm_dma_f32, tst_layernorm_t_f32, tst_tfm_step.ht / ld_ht, tst_attn_cross_mfma_f32,
 * tst_select_step.n_sent_dev, tst_pack_sources, tst_scatter_rows_f32;
 * 105 = 104 + tst_vocab_stats_ok, tst_vocabproj_stats_bf16, tst_topk_from_stats_bf16, tst_tfm_step.part / n_slices,
 * tst_select_step.part / n_slices / stats_k, tst_gru_cell_fused_t_f32, tst_gru_step.ht / ld_ht;
 * 106 = 105 + tst_chain_ok, tst_chain_f32, tst_gemm_res_ln_ok, tst_gemm_res_ln_bf16,
 * tst_embed_posadd_rows_f32 (additions only);
 * 107 = 106 + tst_vocab_argmax_ok, tst_ln_vocab_argmax_f32, tst_tfm_step.argmax_keys, tst_select_step.am_keys /
 * n_alive_host (the structs grew at their ends);
 * 108 = 107 + tst_gru_cell_fused_tab_f32, tst_gru_cell_step_tab_f32, tst_gru_step.gi_tab / ld_gi_tab;
 * 109 = 108 + TST_WREG_MIN_ROWS (no new entry point: tst_tfm_step_run picks another kernel for the vocabulary projection at d = 256;
 * counts beyond 2^31 - 1 are refused by every entry point). */
int tst_version(void);
/* Message of the last error on the calling thread (host pointer, valid until the next failing call). */
const char* tst_last_error(void);

/* x[r][:] = emb[tok[r]][:] * scale + pe[pos[r]][:]   (pos NULL: the single row pe[0..d); pe NULL: no add)
 * replaces nn.Embedding lookup * sqrt(d_model) + positional rows at the first line of a model forward
 * called from /root/reference/src/inference.py:132.  d % 4 == 0. */
int tst_embed_posadd_f32(const int32_t* tok, const float* emb, const float* pe, const int32_t* pos, float scale, float* x,
                         int64_t R, int64_t d, int64_t V, void* stream);

/* The same followed by x[r][:] += z[zidx[r]][:] (tst_add_rows_f32) in ONE pass over x: the embedding of a z-conditioned
 * language model, lm(z, x) (/root/reference/src/trainers/lm_trainer.py:72-74).  Same additions in the same order. */
int tst_embed_posadd_rows_f32(const int32_t* tok, const float* emb, const float* pe, const int32_t* pos, float scale,
                              const float* z, const int32_t* zidx, float* x, int64_t R, int64_t d, int64_t V, void* stream);

/* x[r][:] = x[r][:] * scale + pe[:]   (pe NULL: no add), in place.  The tail of an embedding that was computed
 * as a product, token distributions [R][V] times the embedding matrix [V][d] (tst_gemm_mfma_f32), which is
 * what a model does with the distributions that /root/reference/src/inference.py:104-105 feeds back under
 * gumbel=True.  d % 4 == 0. */
int tst_scale_posadd_f32(float* x, float scale, const float* pe, int64_t R, int64_t d, void* stream);

/* y = LayerNorm(x) over the last dimension (biased variance), rows of d floats, d % 4 == 0. */
int tst_layernorm_f32(const float* x, const float* gamma, const float* beta, float eps, float* y, int64_t R, int64_t d,
                      void* stream);

/* C[M][N] = epilogue(bias[n] + sum_k A[m][k] * Wt[k][n]) on the fp32 matrix cores
 * (v_mfma_f32_32x32x2_f32), epilogue = relu if act == 1, then + res[m][n] if res != NULL (res may alias C).
 * Wt is the nn.Linear weight transposed to [K][ldw]; replaces nn.Linear / the GRU gate products / the
 * vocabulary projection.  Requirements: K % 4 == 0, lda % 4 == 0, ldw % 4 == 0, ldw >= N,
 * 16-byte aligned A and Wt. */
int tst_gemm_mfma_f32(const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias, const float* res,
                      int64_t ldr, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, void* stream);

/* C[M][N] = bias[n] + sum_k A[m][k] * Wt[k][n] with A given K-MAJOR, AT[k][m] (tst_layernorm_t_f32 writes it so): both
 * operand tiles are then contiguous row pieces and are staged by LDS-DMA (global_load_lds) instead of through registers.
 * Same fma chain per element as tst_gemm_mfma_f32, hence bit-identical results.  For the vocabulary projection of large
 * batches (nn.Linear d_model -> V of the model called at /root/reference/src/inference.py:132).
 * K % 16 == 0; ldat a multiple of 4 and >= M rounded up to a multiple of 128 (columns of AT beyond M may hold anything);
 * ldw % 4 == 0, ldw >= N; AT and Wt 16-byte aligned.  variant: 0 = tiles of 16 k in a ring of 3, 1 = 8 k x 4, 2 = 8 k x 6,
 * 3 = 8 k x 6 with nontemporal stores of C, 4 = 3 as a persistent kernel (two workgroups per CU walk the tiles, the ring of
 * stages runs on across tiles), 5 = 256 x 256 tiles on 8 waves (needs ldat >= M rounded up to 256) (tuning; same results). */
int tst_gemm_dma_f32(const float* AT, int64_t ldat, const float* Wt, int64_t ldw, const float* bias, float* C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, int variant, void* stream);

/* yT[k][r] = LayerNorm(x)[r][k]: tst_layernorm_f32 with the result stored k-major (ldt >= R floats per k), the operand
 * layout of tst_gemm_dma_f32.  Same statistics and affine map, element for element.  d % 4 == 0, d <= 512. */
int tst_layernorm_t_f32(const float* x, const float* gamma, const float* beta, float eps, float* yT, int64_t ldt, int64_t R,
                        int64_t d, void* stream);

/* The same product by the kernel that tst_gemm_mfma_f32 chooses for small M * N: one wave per 16 x (16 * strip) strip of
 * C (strip = 1, 2 or 4) on v_mfma_f32_16x16x4_f32.  Same arguments, same requirements, bit-identical results (every
 * element is the same k-ordered fma chain); exported so that tests and tuning can reach it at any size. */
int tst_gemm_skinny_f32(const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias, const float* res,
                        int64_t ldr, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, int strip, void* stream);

/* C[M][N] = epilogue(bias[n] + sum_k LayerNorm(x)[m][k] * Wt[k][n]): tst_layernorm_f32 over rows of K floats followed by
 * tst_gemm_mfma_f32 (act as there, no residual), with identical results.  For small problems (K = 256 or 512 and at
 * most 2048 tiles of 16 x 16 in C) it is ONE kernel that never stores the normalised rows; otherwise the two entry
 * points run one after the other and h [M][K] (caller's scratch, always required, contents unspecified afterwards)
 * receives the normalised rows.  x is [M][K] contiguous. */
int tst_ln_gemm_f32(const float* x, const float* gamma, const float* beta, float eps, float* h, const float* Wt, int64_t ldw,
                    const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, void* stream);

/* Greedy decoding of a small batch: keys[m][s] = max of key(logit[m][n], n) over some of the columns n (every column counts
 * for exactly one slot s of its row), with
 *     logit[m][n] = bias[n] + sum_k LayerNorm(x)[m][k] * Wt[k][n]
 * i.e. tst_layernorm_f32, tst_gemm_mfma_f32 and tst_argmax_rows_f32 as ONE kernel that stores neither the normalised rows
 * nor the logits (torch.argmax of the model output at /root/reference/src/inference.py:132-145 under sampling_type "max").
 * key(v, n) = (order-preserving bits of v, -0 read as +0) << 32 | ~n, raised with 64-bit atomic max operations, so keys
 * [R][TST_ARGMAX_SLOTS] MUST BE ZERO ON ENTRY.  With key = the largest of a row's TST_ARGMAX_SLOTS keys (unsigned), the
 * selected token of the row is (key == 0) ? 0 : ~(uint32_t)key: the first largest logit, NaN and -inf never selected, token 0
 * for a row without any other value: exactly what tst_argmax_rows_f32 returns on the logits of the two separate calls (same
 * fma chains, same LayerNorm order).  x is [R][d] contiguous, d = 256 or 512; tst_vocab_argmax_ok tells whether a shape is
 * accepted (products small enough for 64 x 64 tiles: ceil(R / 128) * ceil(V / 128) < 640).
 * ldw % 4 == 0, ldw >= V; 16-byte aligned x, gamma, beta, Wt; bias may be NULL. */
#define TST_ARGMAX_SLOTS 16
int tst_vocab_argmax_ok(int64_t R, int64_t V, int64_t d);
int tst_ln_vocab_argmax_f32(const float* x, const float* gamma, const float* beta, float eps, const float* Wt, int64_t ldw,
                            const float* bias, uint64_t* keys, int64_t R, int64_t V, int64_t d, void* stream);

/* A residual product, the LayerNorm behind it and the product behind that as ONE kernel per tile of 64 whole rows
 * (d_model = 256):
 *     x[m][:] += A[m][:] * W1t + b1                      (tst_gemm_mfma_f32 with res == C == x; K1 inner elements)
 *     y[m][:]  = LayerNorm(x[m][:])                      (tst_layernorm_f32)
 *     C[m][:]  = act2(y[m][:] * W2t + b2)                (tst_gemm_mfma_f32, N2 columns; act2 = 1: relu)
 * or, with N2 == 0, yT[k][m] = y[m][k] (tst_layernorm_t_f32: the k-major operand of tst_gemm_dma_f32) instead of the last line.
 * Neither x nor y travels through memory between the three operations.  Results are bit-identical to the three separate
 * calls (same fma chains, same order of the LayerNorm partial sums).  These are the chains of a pre-norm decoder layer:
 * attention output projection -> LayerNorm -> cross-attention q projection / first feed-forward product, second feed-forward
 * product -> LayerNorm -> next layer's q|k|v projection or the final LayerNorm, in the model called at
 * /root/reference/src/inference.py:132.  Requirements (tst_chain_ok): d == 256, K1 a multiple of 256, N2 in {0, 256, 768,
 * 1024}; x is [R][256] contiguous; lda, ldw1, ldw2 multiples of 4; 16-byte aligned A, W1t, W2t, b1, x, gamma, beta; ldt >= R. */
int tst_chain_ok(int64_t R, int64_t d, int64_t K1, int64_t N2);
int tst_chain_f32(const float* A, int64_t lda, int64_t K1, const float* W1t, int64_t ldw1, const float* b1, float* x,
                  const float* gamma, const float* beta, float eps, const float* W2t, int64_t ldw2, const float* b2, int act2,
                  float* C, int64_t ldc, int64_t N2, float* yT, int64_t ldt, int64_t R, int64_t d, void* stream);

/* y = LayerNorm(x) as above with the result rounded to bf16 (the operand of a following bf16 GEMM). */
int tst_layernorm_f32_bf16(const float* x, const float* gamma, const float* beta, float eps, void* y, int64_t R, int64_t d,
                           void* stream);

/* x[r][:] += z[idx[r]][:], rows of d floats (d % 4 == 0): adds the conditioning vector z of a row's sentence
 * to its embedding (the z argument of the reference's language models, lm(z, x),
 * /root/reference/src/trainers/lm_trainer.py:72-74). */
int tst_add_rows_f32(float* x, const float* z, const int32_t* idx, int64_t R, int64_t d, void* stream);

/* C[M][N] = epilogue(bias[n] + sum_k A[m][k] * W[n][k]) on the bf16 matrix cores, fp32 accumulation.
 * A [M][lda] and W [N][ldw] are bf16 with k contiguous (W is the nn.Linear weight as stored, [out][in]).
 * epilogue = relu if act == 1, then + res[m][n] (fp32, may alias an fp32 C) if res != NULL; C is bf16 if
 * out_bf16 != 0, else fp32.  Requirements: K, lda, ldw multiples of 8; ldc a multiple of 8 (bf16) or 4
 * (fp32); ldr a multiple of 4; 16-byte aligned A, W, C, res.
 * bf16 results are written with nontemporal stores (a wide result such as the logits is read again only after it has
 * left the caches); large bf16 products without a residual run as one persistent workgroup per CU.  Tile shape and
 * kernel form are chosen from M, N, K and never change the value of an element. */
int tst_gemm_mfma_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* res,
                       int64_t ldr, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, int out_bf16,
                       void* stream);

/* x[M][512] += A * W^T + bias (tst_gemm_mfma_bf16 with res == C == x, fp32) and h = LayerNorm(x) rounded to bf16
 * (tst_layernorm_f32_bf16) as ONE kernel on whole-row tiles of 128 x 512: the LayerNorm behind a residual product of a
 * model_size = 512 bf16 decoder (the model called at /root/reference/src/trainers/lm_trainer.py:72-74,86) without its launch and
 * without reading x back.  Bit-identical to the two calls.  N must be 512, K a multiple of 32 and >= 96 (tst_gemm_res_ln_ok);
 * x is [M][512] contiguous; alignment as for tst_gemm_mfma_bf16. */
int tst_gemm_res_ln_ok(int64_t M, int64_t N, int64_t K);
int tst_gemm_res_ln_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* x, const float* gamma,
                         const float* beta, float eps, void* h, int64_t ldh, int64_t M, int64_t N, int64_t K, void* stream);

/* One decoder self-attention step with an incremental KV cache.
 * cache: [n_pos][slab_rows][3*d] fp32, the fused q|k|v projection output of every past step; the rows of
 * position `pos` (the current step) must already be written.  Row r attends to positions p = 0..pos,
 * reading k/v of position p from cache row anc[r*anc_ld + p] (p < pos) or row r (p == pos).
 * out[r][d] = softmax(q k^T / sqrt(d/heads)) v per head.  replaces the decoder self-attention of the
 * model called at /root/reference/src/inference.py:132 (whole-prefix recompute there). */
int tst_attn_self_step_f32(const float* cache, int64_t slab_rows, const int32_t* anc, int64_t anc_ld, int64_t pos,
                           float* out, int64_t R, int64_t d, int64_t heads, void* stream);

/* One decoder cross-attention step.  q [R][d]; kv [n_sent][S][2*d] (k | v of the encoder outputs);
 * mask [n_sent][S] bytes (0 = padding, may be NULL); row r uses sentence sent[r]. */
int tst_attn_cross_step_f32(const float* q, const float* kv, const uint8_t* mask, const int32_t* sent, int64_t S,
                            float* out, int64_t R, int64_t d, int64_t heads, void* stream);

/* tst_attn_cross_step_f32 over a PACKED k|v: kv [n_packed][2*d] holds only the unmasked source positions, and
 * kvrow [n_sent][S] gives the row of kv for position p of sentence s (read only where mask is non-zero).  Scores,
 * softmax and value sum are those of the padded form (positions keep their numbers), so results are identical; what
 * changes is that no k|v row has to exist for a padding position.  Used by the encoder pass
 * (/root/reference/src/trainers/transformer_mt_trainer.py:91), which then computes nothing for padding. */
int tst_attn_cross_packed_f32(const float* q, const float* kv, const uint8_t* mask, const int32_t* sent, const int32_t* kvrow,
                              int64_t S, float* out, int64_t R, int64_t d, int64_t heads, void* stream);

/* Cross-attention step for beam search: rows g*group .. g*group + group - 1 are the hypotheses of one sentence and share
 * its keys and values, which are then loaded once per group instead of once per row (group = 2, 4 or 8; any other value,
 * R % group != 0 or an unsupported head size run the row-per-wave kernel).  A group whose rows do not name the same
 * sentence is computed row by row, so the grouping is a hint, never a requirement.  kvrow NULL: kv is
 * [n_sent][S][2*d]; else kv is packed as for tst_attn_cross_packed_f32.  Results are identical to those entry points. */
int tst_attn_cross_group_f32(const float* q, const float* kv, const uint8_t* mask, const int32_t* sent, const int32_t* kvrow,
                             int64_t S, float* out, int64_t R, int64_t d, int64_t heads, int64_t group, void* stream);

/* tst_attn_cross_group_f32 with the beams of a sentence as the rows of matrix-core products: per (sentence, head) the
 * scores are Q [beams x dh] K^T and the outputs W [beams x S] V on v_mfma_f32_4x4x1_16B_f32, the key tile staged once in
 * LDS.  The score dot products are single fma chains over the head's elements (the other kernels sum 16 partials), so
 * results agree with tst_attn_cross_group_f32 to the last bits only; softmax and value sums are the same.  Needs
 * R % group == 0, group 2..8, S <= 64, head size 4, 8, 16, 32, 64 or 128 (TST_ERR_UNSUPPORTED otherwise).  Measured slower
 * than tst_attn_cross_group_f32 at the benchmark sizes and therefore not used by tst_tfm_step_run. */
int tst_attn_cross_mfma_f32(const float* q, const float* kv, const uint8_t* mask, const int32_t* sent, const int32_t* kvrow,
                            int64_t S, float* out, int64_t R, int64_t d, int64_t heads, int64_t group, void* stream);

/* The two attention steps with bf16 cache / q / k|v / output (same layouts, 2-byte elements); scores,
 * softmax and the value sum are computed in fp32 in the same order as the fp32 kernels. */
int tst_attn_self_step_bf16(const void* cache, int64_t slab_rows, const int32_t* anc, int64_t anc_ld, int64_t pos,
                            void* out, int64_t R, int64_t d, int64_t heads, void* stream);
int tst_attn_cross_step_bf16(const void* q, const void* kv, const uint8_t* mask, const int32_t* sent, int64_t S,
                             void* out, int64_t R, int64_t d, int64_t heads, void* stream);

/* GRU cell update (PyTorch gate order r, z, n); gi = x W_ih^T + b_ih and gh = h W_hh^T + b_hh are [R][3H]:
 * r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) * n + z * h.
 * replaces the elementwise part of nn.GRU.  h_out may alias h.  H % 4 == 0. */
int tst_gru_cell_step_f32(const float* gi, const float* gh, const float* h, float* h_out, int64_t R, int64_t H,
                          void* stream);

/* The whole GRU cell step as ONE kernel: h_out = cell(x Wih + bih, h Whh + bhh, h) with the gate pre-activations kept in
 * matrix-core accumulators (six per element: i_r, i_z, i_n, h_r, h_z, h_n, each the k-ordered fma chain of
 * tst_gemm_mfma_f32) and tst_gru_cell_step_f32's update applied in the epilogue; gi and gh [R][3H] are never stored.
 * Results are bit-identical to the three-call form.  replaces nn.GRU's step of the decoder called at
 * /root/reference/src/inference.py:132 (model: /root/reference/src/trainers/cyclegan_trainer.py:64).
 * x [R][ldx >= E], h [R][ldh >= H], wih [E][ld_wih >= 3H], whh [H][ld_whh >= 3H] (nn.GRU weights transposed, columns
 * r | z | n), bih / bhh [3H] or NULL, h_out [R][ldo >= H] must NOT overlap h.  Needs E % 16 == 0 and H % 64 == 0
 * (tst_gru_cell_fused_ok tells; TST_ERR_UNSUPPORTED otherwise), leading dimensions multiples of 4, 16-byte aligned inputs. */
int tst_gru_cell_fused_ok(int64_t E, int64_t H);
int tst_gru_cell_fused_f32(const float* x, int64_t ldx, const float* h, int64_t ldh, const float* wih, int64_t ld_wih,
                           const float* bih, const float* whh, int64_t ld_whh, const float* bhh, float* h_out, int64_t ldo,
                           int64_t R, int64_t E, int64_t H, void* stream);

/* The GRU cell step of a layer whose input is a plain embedding lookup, with the input product taken from a table instead of
 * computed: gi_tab [V][ld_tab >= 3H] holds bias_ih + embedding[v] * Wih for every vocabulary entry v (the caller makes it once
 * per model with tst_gemm_mfma_f32 over the embedding matrix: the same fma chain per element as the product inside
 * tst_gru_cell_fused_f32 / the first product of the three-call form, hence the same bits), and row r uses row tok[r] of it (ids
 * clamped to 0 .. V-1 as in tst_embed_posadd_f32).  Replaces nn.Embedding + the input half of nn.GRU for one time step
 * (/root/reference/src/inference.py:170-177 steps such a decoder); half the matrix work of the cell.
 * tst_gru_cell_fused_tab_f32: the recurrent product and the update in one kernel (H % 64 == 0), h_out must not overlap h;
 * h_out_t as in tst_gru_cell_fused_t_f32, or NULL.  tst_gru_cell_step_tab_f32: the update alone, gh [R][3H] given (h_out may
 * alias h).  Same results as tst_gru_cell_fused_f32 / tst_gru_cell_step_f32 on x = embedding[tok]. */
int tst_gru_cell_fused_tab_f32(const float* gi_tab, int64_t ld_tab, const int32_t* tok, int64_t V, const float* h, int64_t ldh,
                               const float* whh, int64_t ld_whh, const float* bhh, float* h_out, int64_t ldo, float* h_out_t,
                               int64_t ldt, int64_t R, int64_t H, void* stream);
int tst_gru_cell_step_tab_f32(const float* gi_tab, int64_t ld_tab, const int32_t* tok, int64_t V, const float* gh, const float* h,
                              float* h_out, int64_t R, int64_t H, void* stream);

/* tst_gru_cell_fused_f32 that ALSO stores the new state k-major, h_out_t[j][r] = h_out[r][j] (ldt >= R floats per hidden unit):
 * the A operand layout of tst_gemm_dma_f32, so that the output projection H -> V of a large batch stages both operands by
 * LDS-DMA.  Same values as tst_gru_cell_fused_f32, bit for bit (the products are computed transposed: the same fma chain
 * per element).  Additionally ldo % 4 == 0, h_out, bih, bhh 16-byte aligned, h_out_t must not overlap h. */
int tst_gru_cell_fused_t_f32(const float* x, int64_t ldx, const float* h, int64_t ldh, const float* wih, int64_t ld_wih,
                             const float* bih, const float* whh, int64_t ld_whh, const float* bhh, float* h_out, int64_t ldo,
                             float* h_out_t, int64_t ldt, int64_t R, int64_t E, int64_t H, void* stream);

/* dst[r][:] = src[idx[r]][:], rows of `width` floats (width % 4 == 0); carried GRU state on reorder. */
int tst_gather_rows_f32(const float* src, const int32_t* idx, float* dst, int64_t R, int64_t width, void* stream);

/* dst[idx[r]][:] = src[r][:], rows of `width` floats (width % 4 == 0), idx without repeats: puts the packed encoder outputs
 * back into the padded [B][S][d] tensor that /root/reference/src/trainers/transformer_mt_trainer.py:91 hands to the loop. */
int tst_scatter_rows_f32(const float* src, const int32_t* idx, float* dst, int64_t R, int64_t width, void* stream);

/* Packing of a right- or otherwise padded source batch src [B][S] (int64 token ids, `pad` = padding id), so that the encoder
 * computes only real positions: mask [B][S] (1 = token), kvrow [B][S] = packed row of a token (-1 at padding), and for
 * every packed row (sentence after sentence, positions ascending) its token, position, sentence and padded row b * S + p
 * (arrays of B * S entries; the first n_live are written).  work: 2 * B int32 of scratch; n_live: device int32, the number
 * of tokens.  Replaces the mask / nonzero / cumsum / index arithmetic around the encoder call of the reference's
 * validation loop.  Three launches. */
int tst_pack_sources(const int64_t* src, int64_t pad, int64_t B, int64_t S, uint8_t* mask, int32_t* kvrow, int32_t* tok,
                     int32_t* pos, int32_t* sent, int32_t* padded_row, int32_t* work, int32_t* n_live, void* stream);

/* tok[r] = index of the first maximal element of logits[r][0..V): torch.max(dim=-1)[1] at
 * /root/reference/src/inference.py:145. */
int tst_argmax_rows_f32(const float* logits, int64_t ld, int64_t R, int64_t V, int32_t* tok, void* stream);

/* out[r][:] = softmax((softmax(logits[r][:]) + G) / temperature), G = -log(-log(u[r][:] + 1e-20) + 1e-20):
 * F.softmax(next_tokens_dists, dim=1) followed by gumbel_softmax_sample(.., temperature) at
 * /root/reference/src/inference.py:137-138 (/root/reference/src/utils/gumbel.py:11-21).  u holds the uniform
 * draws in [0, 1) that the reference takes from torch.rand; they are an input so that a caller reproduces the
 * reference's noise stream.  out must not alias logits or u.  temperature > 0. */
int tst_gumbel_softmax_f32(const float* logits, int64_t ld, int64_t R, int64_t V, const float* u, int64_t ldu,
                           float temperature, float* out, int64_t ldo, void* stream);

/* out[r][v] = log(sum_i weights[i] * softmax(logits_i[r][:])[v]), logits_i = logits + i * model_stride, i < n_models <= 8:
 * WeightedLMEnsemble.weight at /root/reference/src/prod/throwaway_models.py:32-36 (softmax of every member's logits,
 * weighted sum, log).  weights is a device array of n_models floats.  out may be one of the members' buffers only if
 * it is the LAST one (a row is read completely before it is written, but members are read in ascending order). */
int tst_mix_logprobs_f32(const float* logits, int64_t model_stride, int64_t ld, int64_t n_models, const float* weights,
                         int64_t R, int64_t V, float* out, int64_t ldo, void* stream);

/* tok[r] ~ Categorical(logits = logits[r] / temperature): Categorical(...).sample() at
 * /root/reference/src/inference.py:147, drawn by the Gumbel-max trick, argmax_v (x_v / temperature - log(-log u_v)),
 * with u_v in (0,1) from a counter-based hash of (seed, step, sent[r], v), so a sentence's draws do not depend on
 * which other rows are in the batch. */
int tst_sample_rows_f32(const float* logits, int64_t ld, int64_t R, int64_t V, float temperature, uint32_t seed,
                        uint32_t step, const int32_t* sent, int32_t* tok, void* stream);

/* Per row: lse[r] = log sum exp(logits[r]) and the k largest logits (value descending, index ascending),
 * 1 <= k <= 8, k <= V.  Feeds tst_beam_merge (the reference has no beam search; build-defined). */
int tst_logsoftmax_topk_f32(const float* logits, int64_t ld, int64_t R, int64_t V, int64_t k, float* top_val,
                            int32_t* top_idx, float* lse, void* stream);

/* The three selection kernels on bf16 logits (ld in elements; rows 16-byte aligned when ld % 8 == 0). */
int tst_argmax_rows_bf16(const void* logits, int64_t ld, int64_t R, int64_t V, int32_t* tok, void* stream);
int tst_sample_rows_bf16(const void* logits, int64_t ld, int64_t R, int64_t V, float temperature, uint32_t seed,
                         uint32_t step, const int32_t* sent, int32_t* tok, void* stream);
int tst_logsoftmax_topk_bf16(const void* logits, int64_t ld, int64_t R, int64_t V, int64_t k, float* top_val,
                             int32_t* top_idx, float* lse, void* stream);

/* The vocabulary projection of a beam-search step with the first half of the selection pass in its epilogue: C as
 * tst_gemm_mfma_bf16 (bf16 result, bias, no activation, no residual), and for every row and every slice of 64 columns of it
 * the slice's largest STORED (bf16-rounded) logit and the sum of exp(logit - largest) over the slice.  Together they replace
 * nn.Linear d_model -> V followed by the first pass of log_softmax / topk over the [rows, V] logits
 * (/root/reference/src/inference.py:132-145 picks tokens from those logits; the reference has no beam search).
 * part: [ceil(M / 32)][n_slices][64] floats, n_slices = ceil(N / 64): block b, slice s holds the maxima of rows 32 b .. 32 b + 31
 * in its first 32 floats and their sums in the last 32 (entries of rows >= M are unspecified).
 * Only for the shapes tst_vocab_stats_ok accepts (those the persistent 256 x 256 kernel serves: 4096 <= N <= 32768, N % 8 == 0,
 * K % 32 == 0, K >= 96, at least 256 tiles of 256 x 256); TST_ERR_UNSUPPORTED otherwise.  C is bit-identical to
 * tst_gemm_mfma_bf16's. */
int tst_vocab_stats_ok(int64_t M, int64_t N, int64_t K);
int tst_vocabproj_stats_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                             int64_t M, int64_t N, int64_t K, float* part, int64_t n_slices, void* stream);

/* tst_logsoftmax_topk_bf16 from those statistics: lse[r] = M + log(sum_s sum_s * exp(max_s - M)), M = max_s max_s; the k
 * largest logits are found in the slices whose maximum reaches the k-th largest slice maximum (those slices are re-read from
 * `logits`, everything else of the row is not).  top_val / top_idx equal tst_logsoftmax_topk_bf16's exactly (a selection);
 * lse agrees with it to fp32 rounding (another summation order).  V % 8 == 0, ld % 8 == 0, n_slices = ceil(V / 64) <= 512. */
int tst_topk_from_stats_bf16(const void* logits, int64_t ld, const float* part, int64_t n_slices, int64_t R, int64_t V,
                             int64_t k, float* top_val, int32_t* top_idx, float* lse, void* stream);

/* Decode-loop state.  All arrays are device memory owned by the caller; "cur" arrays are read, "nxt"
 * arrays are written (ping-pong).  Rows of one sentence are contiguous (k rows per sentence).
 *   hist  [rows][T+1] int32  tokens so far, BOS first          anc [rows][T] int32  cache row per position
 *   sent  [rows] int32       original sentence of the row      tok [rows] int32     last token
 *   score [rows] fp32        sum of log-probabilities          fin [rows] int32     1 = hypothesis ended
 *   nemit [rows] int32       emitted tokens, EOS included */
typedef struct tst_rows {
    int32_t* hist;
    int32_t* anc;
    int32_t* sent;
    int32_t* tok;
    float* score;
    int32_t* fin;
    int32_t* nemit;
} tst_rows;

/* Beam merge, one wave per sentence.  The sentence has k_in input rows (rows a*k_in .. a*k_in+k_in-1 of
 * `cur`, top_val, top_idx [row][k] and lse) and gets k output slots (a*k .. a*k+k-1 of sel_*): every live input
 * beam offers its k best tokens with value score + (top_val - lse), a finished beam offers itself unchanged;
 * the k best offers (ties: lower beam, then lower rank) become the new beams in that order.  At the first
 * step all beams of a sentence would be identical, so the caller runs one row per sentence (k_in = 1);
 * afterwards k_in = k.  Writes per output slot: parent (input row index), tok, score, fin, nemit. */
int tst_beam_merge(const tst_rows* cur, const float* top_val, const int32_t* top_idx, const float* lse, int64_t n_sent,
                   int64_t k, int64_t k_in, int32_t eos, int32_t* sel_parent, int32_t* sel_tok, float* sel_score,
                   int32_t* sel_fin, int32_t* sel_nemit, void* stream);

/* Greedy counterpart of tst_beam_merge (k = 1): parent = row, tok = argmax token, fin = (tok == eos). */
int tst_greedy_select(const int32_t* argmax_tok, int64_t n_rows, int32_t eos, int32_t* sel_parent, int32_t* sel_tok,
                      float* sel_score, int32_t* sel_fin, int32_t* sel_nemit, const tst_rows* cur, void* stream);

/* Retire finished sentences and compact the batch: /root/reference/src/inference.py:94-120 and :65-69.
 * A sentence ends when all its k new beams are finished, when last_step != 0, or - if top_ends != 0 - when
 * its top beam (slot 0, the highest score) is finished.  With a length penalty of 0 a finished top beam
 * can never be overtaken (scores only decrease, ties go to the lower slot), so top_ends = 1 then gives
 * the same outputs as waiting for all beams, in fewer steps.  The sentence's best beam
 * (score / lp[nemit], ties: lower beam; lp = length-penalty table of T + 2 floats) is written to
 * out_tok[sentence][0..T], out_len (BOS included, EOS stripped) and out_score.  Surviving sentences keep
 * their order; their rows are gathered from `cur` through sel_parent into `nxt`, the new token is
 * appended to hist and the parent row index to anc (skipped when cur->anc is NULL); row_parent[new row]
 * = old row, for callers that carry further per-row state.  scratch_alive_pos: 2 * n_sent int32.
 * n_alive (device, 1 int32) receives the number of surviving sentences.  Three launches on `stream` (one up to 160 rows). */
int tst_retire_compact(const tst_rows* cur, const tst_rows* nxt, const int32_t* sel_parent, const int32_t* sel_tok,
                       const float* sel_score, const int32_t* sel_fin, const int32_t* sel_nemit, const float* lp,
                       int64_t n_sent, int64_t k, int64_t step, int64_t T, int32_t eos, int last_step, int top_ends,
                       int32_t* out_tok, int32_t* out_len, float* out_score, int32_t* scratch_alive_pos,
                       int32_t* row_parent, int32_t* n_alive, void* stream);

/* tst_retire_compact for callers that cannot allocate device memory and have no workspace to pass (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments
 * and results as tst_retire_compact without scratch_alive_pos: the 2 * n_sent int32 of alive flags and prefix sums live in an
 * array inside the library's code object (512 KiB, held on every GPU the library runs on until the process ends).
 * n_sent <= TST_NOALLOC_MAX_SENT, else TST_ERR_UNSUPPORTED.  Calls no hipMalloc / hipFree; the first call on a device
 * resolves that device's copy of the array (hipGetSymbolAddress, the device must be current as for any launch): that
 * lookup may load the code object or synchronise, so make ONE warm-up call per device outside any stream capture (any
 * valid arguments, n_sent = 0 included); every later call on that device only launches kernels.  All calls
 * on one GPU share the array: calls that overlap there (different streams or threads) race; calls on one stream do not. */
#define TST_NOALLOC_MAX_SENT 65536
int tst_retire_compact_noalloc(const tst_rows* cur, const tst_rows* nxt, const int32_t* sel_parent, const int32_t* sel_tok,
                               const float* sel_score, const int32_t* sel_fin, const int32_t* sel_nemit, const float* lp,
                               int64_t n_sent, int64_t k, int64_t step, int64_t T, int32_t eos, int last_step, int top_ends,
                               int32_t* out_tok, int32_t* out_len, float* out_score, int32_t* row_parent, int32_t* n_alive,
                               void* stream);

/* tst_greedy_select followed by tst_retire_compact with k = 1 (top_ends does not apply), same arguments and results; up to
 * 160 sentences the two are ONE launch of one workgroup (tst_retire_compact alone also takes that route up to 160 rows). */
int tst_greedy_retire_compact(const int32_t* argmax_tok, const tst_rows* cur, const tst_rows* nxt, int32_t* sel_parent,
                              int32_t* sel_tok, float* sel_score, int32_t* sel_fin, int32_t* sel_nemit, const float* lp,
                              int64_t n_sent, int64_t step, int64_t T, int32_t eos, int last_step, int32_t* out_tok,
                              int32_t* out_len, float* out_score, int32_t* scratch_alive_pos, int32_t* row_parent,
                              int32_t* n_alive, void* stream);

/* One decode step of a pre-norm transformer decoder stack, launched from native code: the sequence of
 * entry points above that evaluates the model called at /root/reference/src/inference.py:132 for ONE new position
 * (embedding, per layer: LayerNorm, q|k|v projection into the cache, self-attention, output projection + residual,
 * [LayerNorm, q projection, cross-attention, output projection + residual,] LayerNorm, feed-forward + residual;
 * final LayerNorm; vocabulary projection).  It adds no arithmetic of its own: results are those of the single
 * calls.  The structs are HOST memory and only read during the call; every pointer inside is device memory.
 * With bf16 != 0 the weights (void*) are bf16 [out][in] as for tst_gemm_mfma_bf16 and h, q, a, ff, the caches,
 * kv_cross and logits are bf16; x, biases, LayerNorm parameters, emb, pe and z are always fp32.
 * With bf16 == 0 the weights are fp32 [in][ld] as for tst_gemm_mfma_f32. */
typedef struct tst_tfm_layer {
    const float *ln_self_g, *ln_self_b;   /* LayerNorm before self-attention */
    const void* wqkv;  const float* bqkv;  int64_t ld_wqkv;   /* fused q|k|v projection, N = 3d */
    const void* wo;    const float* bo;    int64_t ld_wo;
    const float *ln_cross_g, *ln_cross_b; /* cross-attention block; cwq == NULL: the layer has none */
    const void* cwq;   const float* cbq;   int64_t ld_cwq;
    const void* cwo;   const float* cbo;   int64_t ld_cwo;
    const float *ln_ff_g, *ln_ff_b;
    const void* w1;    const float* b1;    int64_t ld_w1;     /* d -> d_ff, relu */
    const void* w2;    const float* b2;    int64_t ld_w2;     /* d_ff -> d */
    void* cache;            /* [n_pos][slab_rows][3d] self-attention q|k|v of every position */
    const void* kv_cross;   /* [n_sent][S][2d] k|v of the encoder outputs (cross block only) */
} tst_tfm_layer;

typedef struct tst_tfm_step {
    int32_t n_layers, d, heads, d_ff, V, bf16;
    int32_t group;          /* adjacent rows per sentence at THIS step (hint for tst_attn_cross_group_f32), or 0 / 1;
                               the caller may change it between calls */
    int32_t reserved;
    int64_t n_pos, slab_rows, S, anc_ld;
    const float* emb;       /* [V][d] */
    const float* pe;        /* [n_pos][d] positional rows */
    float scale;            /* embedding scale */
    float eps;              /* LayerNorm epsilon */
    const float* z;         /* [n_sent][d] added to the embedding of a sentence's rows, or NULL */
    const uint8_t* mask;    /* [n_sent][S] source padding mask of the cross-attention, or NULL */
    const int32_t* kvrow;   /* NULL: kv_cross is [n_sent][S][2d]; else kv_cross is packed and this is its
                               [n_sent][S] row map (tst_attn_cross_packed_f32; fp32 only, needs mask) */
    const float *ln_g, *ln_b;             /* final LayerNorm */
    const void* wout;  const float* bout;  int64_t ld_wout;
    float* x;               /* [rows][d] residual stream */
    void *h, *q, *a;        /* [rows][d] work buffers */
    void* ff;               /* [rows][d_ff] */
    void* logits;           /* [rows][ld_logits] output */
    int64_t ld_logits;
    const tst_tfm_layer* layers;          /* host array of n_layers */
    float* ht;              /* fp32 only, optional: [d][ld_ht] k-major copy target of the final LayerNorm.  When set and the
                               step has at least TST_DMA_MIN_ROWS rows, the final LayerNorm writes here (tst_layernorm_t_f32)
                               and the vocabulary projection is tst_gemm_dma_f32; h is then not written.  Same logits.
                               Not used at d = 256, where steps of at least TST_WREG_MIN_ROWS rows project from h with the
                               weights held in registers (same logits again, faster; a host need not provide ht there). */
    int64_t ld_ht;          /* >= slab_rows rounded up to a multiple of 128 */
    float* part;            /* bf16 only, optional: [ceil(slab_rows / 32)][n_slices][64] statistics of the logits.  When set and
                               tst_vocab_stats_ok(rows, V, d), the vocabulary projection is tst_vocabproj_stats_bf16 and
                               tst_select_step_run (given the same buffer) reads them instead of the whole rows.  Same logits. */
    int64_t n_slices;       /* ceil(V / 64) */
    uint64_t* argmax_keys;  /* fp32 only, optional (the caller may change it between calls): [slab_rows][TST_ARGMAX_SLOTS].  When set, a call with
                               all three phases on a shape tst_vocab_argmax_ok(rows, V, d) accepts ends in
                               tst_ln_vocab_argmax_f32 instead of the final LayerNorm and the vocabulary projection: the
                               embedding kernel clears the keys of rows 0..rows-1, logits and h are NOT written.  Any other call with it
                               set fails with TST_ERR_UNSUPPORTED. */
} tst_tfm_step;
#define TST_DMA_MIN_ROWS 2048
#define TST_WREG_MIN_ROWS 1024

#define TST_STEP_EMBED 1   /* x = emb[tok] * scale + pe[t] (+ z[sent]); without it x must hold the embeddings */
#define TST_STEP_LAYERS 2  /* the layer stack and the final LayerNorm (into h) */
#define TST_STEP_VOCAB 4   /* logits = h * wout + bout */
#define TST_STEP_FUSED_CELL 8 /* tst_gru_step_run only: every layer's cell is ONE kernel (tst_gru_cell_fused_f32) that reads the
                                 state from h and writes the new state to h_alt; the caller swaps h and h_alt of every layer
                                 in its struct after the call (as after tst_gru_reorder_run).  Same results. */
/* Runs the phases selected in `phases` for rows 0..R-1 at position t (0 <= t < n_pos): tok [R] token ids,
 * anc [R][anc_ld] cache rows of earlier positions, sent [R] sentence of each row.  Stops at the first failing
 * launch and returns its code.  A caller that times the vocabulary projection calls it twice (phases 3, then 4). */
int tst_tfm_step_run(const tst_tfm_step* m, const int32_t* tok, const int32_t* anc, const int32_t* sent, int64_t R,
                     int64_t t, int phases, void* stream);

/* One decode step of a GRU decoder / language model (embedding -> n_layers GRU cells with carried state -> linear to
 * logits), launched from native code like tst_tfm_step_run: x = emb[tok] (+ extra[sent]); per layer gi = x Wih + bih,
 * gh = h Whh + bhh, h = cell(gi, gh, h), x = h; logits = x Wout + bout.  Host structs, device pointers, fp32. */
typedef struct tst_gru_layer {
    const float* wih;  const float* bih;  int64_t ld_wih;   /* [in][3H] */
    const float* whh;  const float* bhh;  int64_t ld_whh;   /* [H][3H] */
    float* h;          /* [rows][H] carried state, updated in place */
    float* h_alt;      /* [rows][H] target of tst_gru_reorder_run */
} tst_gru_layer;

typedef struct tst_gru_step {
    int32_t n_layers, E, H, V;
    int64_t rows;           /* capacity of the work buffers */
    const float* emb;       /* [V][E] */
    const float* extra;     /* [n_sent][E] added to the embedding of a sentence's rows, or NULL */
    const float* wout;  const float* bout;  int64_t ld_wout;
    float *x, *gi, *gh;     /* [rows][E], [rows][3H], [rows][3H] */
    float* logits;  int64_t ld_logits;
    const tst_gru_layer* layers;   /* host array of n_layers */
    float* ht;              /* optional: [H][ld_ht] k-major copy of the top layer's new state.  When set, a step with
                               TST_STEP_FUSED_CELL of at least TST_DMA_MIN_ROWS rows and V >= 2048 writes it
                               (tst_gru_cell_fused_t_f32) and its output projection is tst_gemm_dma_f32.  Same logits. */
    int64_t ld_ht;          /* a multiple of 4, >= rows rounded up to a multiple of 128 */
    const float* gi_tab;    /* optional: [V][ld_gi_tab >= 3H] = bih + emb[v] * Wih of layer 0 for every token v.  When set, a call
                               with TST_STEP_EMBED and TST_STEP_LAYERS of a plan without `extra` does not compute the embeddings
                               (x is NOT written) nor layer 0's input product: its cell reads gi_tab[tok[r]]
                               (tst_gru_cell_fused_tab_f32 / tst_gru_cell_step_tab_f32).  Same logits. */
    int64_t ld_gi_tab;
} tst_gru_step;

/* phases as for tst_tfm_step_run (TST_STEP_EMBED | TST_STEP_LAYERS | TST_STEP_VOCAB). */
int tst_gru_step_run(const tst_gru_step* m, const int32_t* tok, const int32_t* sent, int64_t R, int phases, void* stream);
/* h_alt[r] = h[row_parent[r]] for every layer (rows re-gathered after compaction / a beam reorder); the caller then
 * swaps h and h_alt in its struct. */
int tst_gru_reorder_run(const tst_gru_step* m, const int32_t* row_parent, int64_t R, void* stream);

/* The tail of a decode step in one call: token selection on the logits, then retirement and compaction.
 *   k == 1: tst_argmax_rows (sample == 0) or tst_sample_rows (sample != 0), tst_greedy_select
 *   k  > 1: tst_logsoftmax_topk, tst_beam_merge
 *   then tst_retire_compact.  Arguments as for those entry points; the struct holds what stays the same over the steps
 *   of one decode call. */
typedef struct tst_select_step {
    int32_t k, eos, V, bf16_logits;
    const void* logits;  int64_t ld_logits;
    int32_t* am;                                   /* [rows] selected token (k == 1) */
    float* top_val;  int32_t* top_idx;  float* lse;   /* [rows][k], [rows][k], [rows] (k > 1) */
    int32_t* sel_parent;  int32_t* sel_tok;  float* sel_score;  int32_t* sel_fin;  int32_t* sel_nemit;
    const float* lp;  int64_t T;  int32_t top_ends;  int32_t sample;
    float temperature;  uint32_t seed;
    int32_t* out_tok;  int32_t* out_len;  float* out_score;  int32_t* scratch_alive_pos;  int32_t* row_parent;
    int32_t* n_alive;
    const int32_t* n_sent_dev;   /* NULL, or device int32: the TRUE number of sentences entering the step, when the n_sent
                                    argument of tst_select_step_run is only an upper bound (a caller that launches a step
                                    before it has read the previous step's n_alive).  Sentences beyond it are ignored; the rows
                                    the caller stepped for them are wasted work, nothing else.  Must not alias n_alive. */
    const float* part;           /* NULL, or the statistics tst_tfm_step_run leaves beside bf16 logits (tst_tfm_step.part): for
                                    k > 1 and a step of a shape tst_vocab_stats_ok(rows, V, stats_k) accepts, the selection is
                                    tst_topk_from_stats_bf16 instead of tst_logsoftmax_topk_bf16 */
    int64_t n_slices;            /* ceil(V / 64) */
    int64_t stats_k;             /* inner dimension of the vocabulary projection that wrote `part` (d_model) */
    const uint64_t* am_keys;     /* NULL, or (k == 1, sample == 0) the keys tst_tfm_step_run left in tst_tfm_step.argmax_keys: the
                                    tokens are decoded from them, logits are not read and am is not written */
    int32_t* n_alive_host;       /* NULL, or a device-visible address of page-locked HOST memory that receives the same count as
                                    n_alive, stored by the kernel that computes it (a caller that reads the count one step late
                                    saves the device-to-host copy it would otherwise queue behind every step) */
} tst_select_step;

int tst_select_step_run(const tst_select_step* s, const tst_rows* cur, const tst_rows* nxt, int64_t n_sent, int64_t k_in,
                        int64_t step, uint32_t sample_step, int last_step, void* stream);

#ifdef __cplusplus
}
#endif
#endif
