/* TEST INFRASTRUCTURE: canonical-arithmetic CPU oracle for the decode hot path.
 *
 * Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load the library
 * built from this file; the product never does.
 *
 * What it restates: the per-step arithmetic behind the decode loop of
 * /root/reference/src/inference.py:122-149 (model forward -> last-position logits -> argmax) for the
 * model contracts used at its call sites (/root/reference/src/trainers/transformer_mt_trainer.py:91-98,
 * /root/reference/src/trainers/cyclegan_trainer.py:329-332).  The model sources themselves are an absent
 * submodule (/root/reference/.gitmodules:1-3), so the arithmetic follows the specification in
 * oracle/torch_ref.py, against which this file is pinned (tests/test_oracle_golden.py).
 *
 * "Canonical arithmetic": every fp32 operation and every summation order is fixed, so that a second
 * implementation that follows the same order (the HIP kernels) can be compared bit for bit.
 *   - dot products of a GEMM: acc = bias[n] (or 0); for k ascending: acc = fmaf(a[m][k], w[k][n], acc)
 *   - exp / log / sigmoid / tanh: the polynomial forms below (only +, *, fma, /, rint, bit operations)
 *   - row sums over NL interleaved partial sums followed by an xor-butterfly (see sum_butterfly)
 * Build: gcc -O3 -fPIC -shared -ffp-contract=off -mavx2 -mfma -fopenmp (no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ scalar functions */
static inline float as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

float orc_exp(float x) {
    if (!(x >= -87.0f)) return 0.0f; /* also maps -inf and NaN to 0 */
    if (x > 88.0f) x = 88.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    return y * as_float((uint32_t)((int)n + 127) << 23);
}

/* natural log for finite x > 0 (normal range) */
float orc_log(float x) {
    uint32_t u = as_uint(x);
    int e = (int)((u >> 23) & 0xffu) - 127;
    float m = as_float((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = 0.111111111111f;
    p = fmaf(p, z, 0.142857142857f);
    p = fmaf(p, z, 0.2f);
    p = fmaf(p, z, 0.333333333333f);
    p = p * z;
    float s2 = s + s;
    float r = fmaf(s2, p, s2);
    float fe = (float)e;
    r = fmaf(fe, 1.42860682030941723212e-6f, r);
    return fmaf(fe, 0.693145751953125f, r);
}

float orc_sigmoid(float x) {
    if (x >= 0.0f) { float e = orc_exp(-x); return 1.0f / (1.0f + e); }
    float e = orc_exp(x);
    return e / (1.0f + e);
}

float orc_tanh(float x) {
    float a = fabsf(x);
    float t = orc_exp(-2.0f * a);
    float r = (1.0f - t) / (1.0f + t);
    return copysignf(r, x);
}

void orc_map(int which, const float* x, float* y, int64_t n) {
    for (int64_t i = 0; i < n; ++i)
        y[i] = which == 0 ? orc_exp(x[i]) : which == 1 ? orc_log(x[i]) : which == 2 ? orc_sigmoid(x[i]) : orc_tanh(x[i]);
}

/* partial sums p[0..nl) -> total; nl a power of two <= 256.
 * Within each aligned group of 64 partials: for off = 32,16,...,1 every p[l] becomes p[l] + p[l ^ off]
 * (offsets >= the group size are skipped); groups of 64 are then added left to right. */
static float sum_butterfly(float* p, int nl) {
    int g = nl < 64 ? nl : 64;
    float tmp[256];
    for (int off = g >> 1; off >= 1; off >>= 1) {
        for (int l = 0; l < nl; ++l) tmp[l] = p[l] + p[l ^ off];
        memcpy(p, tmp, sizeof(float) * (size_t)nl);
    }
    float t = p[0];
    for (int q = 64; q < nl; q += 64) t = t + p[q];
    return t;
}

/* ------------------------------------------------------------------ GEMM */
/* C[m][n] = epilogue( bias[n] + sum_k A[m][k] * Wt[k][n] ), k ascending, one fma per term.
 * epilogue: relu if act == 1; then + res[m][n] if res != NULL. */
void orc_gemm(const float* A, int64_t lda, const float* Wt, int64_t ldw, const float* bias, const float* res,
              int64_t ldr, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int act) {
    enum { NB = 64 };
    int64_t nchunks = (N + NB - 1) / NB;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t c = 0; c < nchunks; ++c) {
        int64_t n0 = c * NB;
        int nb = (int)((N - n0) < NB ? (N - n0) : NB);
        for (int64_t m = 0; m < M; ++m) {
            float acc[NB];
            for (int j = 0; j < NB; ++j) acc[j] = (bias && j < nb) ? bias[n0 + j] : 0.0f;
            const float* a = A + m * lda;
            if (nb == NB) {
                for (int64_t k = 0; k < K; ++k) {
                    float av = a[k];
                    const float* w = Wt + k * ldw + n0;
                    for (int j = 0; j < NB; ++j) acc[j] = fmaf(av, w[j], acc[j]);
                }
            } else {
                for (int64_t k = 0; k < K; ++k) {
                    float av = a[k];
                    const float* w = Wt + k * ldw + n0;
                    for (int j = 0; j < nb; ++j) acc[j] = fmaf(av, w[j], acc[j]);
                }
            }
            for (int j = 0; j < nb; ++j) {
                float v = acc[j];
                if (act == 1) v = v > 0.0f ? v : 0.0f;
                if (res) v = v + res[m * ldr + n0 + j];
                C[m * ldc + n0 + j] = v;
            }
        }
    }
}

/* ------------------------------------------------------------------ embedding + position */
/* out[r][c] = emb[tok[r]][c] * scale + pe[c] (two roundings) */
void orc_embed_posadd(const int32_t* tok, const float* emb, const float* pe, float scale, float* out, int64_t R,
                      int64_t d) {
    for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < d; ++c) {
            float v = emb[(int64_t)tok[r] * d + c] * scale;
            out[r * d + c] = pe ? v + pe[c] : v;
        }
}

/* ------------------------------------------------------------------ LayerNorm */
/* 64 interleaved partial sums (element c -> partial (c / 4) % 64, c ascending) + butterfly; biased variance. */
void orc_layernorm(const float* x, const float* g, const float* b, float eps, float* y, int64_t R, int64_t d) {
#pragma omp parallel for
    for (int64_t r = 0; r < R; ++r) {
        const float* xr = x + r * d;
        float p[64];
        for (int l = 0; l < 64; ++l) p[l] = 0.0f;
        for (int64_t c = 0; c < d; ++c) p[(c >> 2) & 63] = p[(c >> 2) & 63] + xr[c];
        float mean = sum_butterfly(p, 64) / (float)d;
        for (int l = 0; l < 64; ++l) p[l] = 0.0f;
        for (int64_t c = 0; c < d; ++c) { float dv = xr[c] - mean; p[(c >> 2) & 63] = fmaf(dv, dv, p[(c >> 2) & 63]); }
        float var = sum_butterfly(p, 64) / (float)d;
        float rstd = 1.0f / sqrtf(var + eps);
        for (int64_t c = 0; c < d; ++c) y[r * d + c] = fmaf((xr[c] - mean) * rstd, g[c], b[c]);
    }
}

/* ------------------------------------------------------------------ attention, one query per row */
/* dot product over dh elements with 16 partial sums: element e -> partial (e / 4) % 16, e ascending,
 * fma accumulation from 0; butterfly over the 16 partials. */
static float dot16(const float* a, const float* b, int64_t dh) {
    float p[16];
    for (int l = 0; l < 16; ++l) p[l] = 0.0f;
    for (int64_t e = 0; e < dh; ++e) { int l = (int)((e >> 2) & 15); p[l] = fmaf(a[e], b[e], p[l]); }
    return sum_butterfly(p, 16);
}

/* the same dot product as ONE chain: acc = 0; for e ascending: acc = fma(a[e], b[e], acc) (the order of a matrix-core
 * product: the cross-attention of beam groups computes the scores of a sentence's beams as Q K^T on the MFMA units) */
static float dot_chain(const float* a, const float* b, int64_t dh) {
    float acc = 0.0f;
    for (int64_t e = 0; e < dh; ++e) acc = fmaf(a[e], b[e], acc);
    return acc;
}

/* q, out: [R][heads*dh].  K, V: [n_entries][n_stride][heads*dh]; row r attends to entry kv_row[r],
 * positions 0..n-1; mask (optional) [n_entries][n_stride] bytes, 0 = position excluded.
 * scores s_p = dot16(q, K_p) * scale (chain != 0: dot_chain); w_p = exp(s_p - max) / sum (64 interleaved partials + butterfly);
 * out[c] = sum_p fma(w_p, V_p[c], acc), p ascending. */
void orc_attention(const float* q, const float* K, const float* V, const int32_t* kv_row, const uint8_t* mask,
                   int64_t R, int64_t n, int64_t n_stride, int64_t heads, int64_t dh, float scale, float* out, int chain) {
    int64_t d = heads * dh;
#pragma omp parallel for
    for (int64_t r = 0; r < R; ++r) {
        int64_t ent = kv_row ? kv_row[r] : r;
        float* s = (float*)malloc(sizeof(float) * (size_t)n);
        for (int64_t h = 0; h < heads; ++h) {
            const float* qh = q + r * d + h * dh;
            float mx = -INFINITY;
            for (int64_t p = 0; p < n; ++p) {
                if (mask && !mask[ent * n_stride + p]) { s[p] = -INFINITY; continue; }
                const float* kp = K + (ent * n_stride + p) * d + h * dh;
                s[p] = (chain ? dot_chain(qh, kp, dh) : dot16(qh, kp, dh)) * scale;
                if (s[p] > mx) mx = s[p];
            }
            float part[64];
            for (int l = 0; l < 64; ++l) part[l] = 0.0f;
            for (int64_t p = 0; p < n; ++p) { s[p] = orc_exp(s[p] - mx); part[p & 63] = part[p & 63] + s[p]; }
            float sum = sum_butterfly(part, 64);
            for (int64_t p = 0; p < n; ++p) s[p] = s[p] / sum;
            for (int64_t c = 0; c < dh; ++c) {
                float acc = 0.0f;
                for (int64_t p = 0; p < n; ++p)
                    acc = fmaf(s[p], V[(ent * n_stride + p) * d + h * dh + c], acc);
                out[r * d + h * dh + c] = acc;
            }
        }
        free(s);
    }
}

/* ------------------------------------------------------------------ GRU cell (PyTorch gate order r,z,n) */
/* gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh, both [R][3H] */
void orc_gru_cell(const float* gi, const float* gh, const float* h, float* hout, int64_t R, int64_t H) {
    for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < H; ++c) {
            const float* a = gi + r * 3 * H;
            const float* b = gh + r * 3 * H;
            float rg = orc_sigmoid(a[c] + b[c]);
            float zg = orc_sigmoid(a[H + c] + b[H + c]);
            float ng = orc_tanh(fmaf(rg, b[2 * H + c], a[2 * H + c]));
            hout[r * H + c] = fmaf(zg, h[r * H + c], (1.0f - zg) * ng);
        }
}

/* ------------------------------------------------------------------ selection over the vocabulary */
/* first maximal index, the tie rule of torch.max used at /root/reference/src/inference.py:145 */
void orc_argmax_rows(const float* x, int64_t ld, int64_t R, int64_t V, int32_t* out) {
    for (int64_t r = 0; r < R; ++r) {
        const float* xr = x + r * ld;
        int64_t best = 0;
        for (int64_t v = 1; v < V; ++v) if (xr[v] > xr[best]) best = v;
        out[r] = (int32_t)best;
    }
}

/* per row: lse = log sum exp(x), computed in one pass by 256 interleaved partial states (m_l, s_l).
 * Chunk c = elements 4c..4c+3 (elements beyond V count as -inf).  Partial l consumes its chunks
 * l, l + 256, l + 512, ... in groups of four: with the group's 16 values x_0..x_15 in ascending index order
 *   m' = max(m, x_0..x_15);  s = s * exp(m - m');  s += exp(x_0 - m'); ...; s += exp(x_15 - m');  m = m'.
 * Then M = max_l m_l, t_l = s_l * exp(m_l - M), total = butterfly sum of the t_l (64-groups, then groups
 * left to right), lse = M + log(total).
 * And the k largest logits (value desc, index asc). */
void orc_logsoftmax_topk(const float* x, int64_t ld, int64_t R, int64_t V, int k, float* top_val, int32_t* top_idx,
                         float* lse) {
#pragma omp parallel for
    for (int64_t r = 0; r < R; ++r) {
        const float* xr = x + r * ld;
        float pm[256], ps[256];
        for (int l = 0; l < 256; ++l) { pm[l] = -INFINITY; ps[l] = 0.0f; }
        int64_t nch = (V + 3) / 4;
        for (int64_t c0 = 0; c0 < nch; c0 += 1024) {          /* one group per partial and iteration */
            for (int l = 0; l < 256 && c0 + l < nch; ++l) {
                float e[16];
                for (int u = 0; u < 4; ++u)
                    for (int q = 0; q < 4; ++q) {
                        int64_t idx = 4 * (c0 + l + 256 * u) + q;
                        e[4 * u + q] = idx < V ? xr[idx] : -INFINITY;
                    }
                float mn = pm[l];
                for (int i = 0; i < 16; ++i) mn = fmaxf(mn, e[i]);
                float sacc = ps[l] * orc_exp(pm[l] - mn);
                for (int i = 0; i < 16; ++i) sacc = sacc + orc_exp(e[i] - mn);
                ps[l] = sacc;
                pm[l] = mn;
            }
        }
        float M = pm[0];
        for (int l = 1; l < 256; ++l) if (pm[l] > M) M = pm[l];
        float p[256];
        for (int l = 0; l < 256; ++l) p[l] = ps[l] * orc_exp(pm[l] - M);
        lse[r] = M + orc_log(sum_butterfly(p, 256));
        for (int j = 0; j < k; ++j) { top_val[r * k + j] = -INFINITY; top_idx[r * k + j] = -1; }
        for (int64_t v = 0; v < V; ++v) {
            float xv = xr[v];
            int j = k;
            while (j > 0 && xv > top_val[r * k + j - 1]) --j; /* strict: equal values keep the earlier index first */
            if (j < k) {
                for (int i = k - 1; i > j; --i) { top_val[r * k + i] = top_val[r * k + i - 1]; top_idx[r * k + i] = top_idx[r * k + i - 1]; }
                top_val[r * k + j] = xv; top_idx[r * k + j] = (int32_t)v;
            }
        }
    }
}

/* ------------------------------------------------------------------ sampling (Gumbel-max) */
/* Restates Categorical(logits = x / temperature).sample() of /root/reference/src/inference.py:147 as
 * argmax_v (x_v / temperature + g_v), g_v = -log(-log(u_v)), with u_v in (0,1) from a counter-based integer
 * hash of (seed, step, sentence, v): the draw for a sentence does not depend on the other rows of the batch.
 * First maximal index wins ties. */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

float orc_gumbel(uint32_t seed, uint32_t step, uint32_t sent, uint32_t v) {
    uint32_t h = mix32(v ^ mix32(sent * 0x9E3779B9u + mix32(step * 0x85EBCA6Bu + seed)));
    float u = (float)(h >> 9) * 1.1920928955078125e-07f + 5.9604644775390625e-08f; /* ((h >> 9) + 0.5) / 2^23: exact in fp32, inside (0, 1) */
    return -orc_log(-orc_log(u));
}

void orc_sample_rows(const float* x, int64_t ld, int64_t R, int64_t V, float temperature, uint32_t seed, uint32_t step,
                     const int32_t* sent, int32_t* out) {
#pragma omp parallel for
    for (int64_t r = 0; r < R; ++r) {
        const float* xr = x + r * ld;
        int64_t best = 0;
        float bv = -INFINITY;
        for (int64_t v = 0; v < V; ++v) {
            float key = xr[v] / temperature + orc_gumbel(seed, step, (uint32_t)sent[r], (uint32_t)v);
            if (key > bv) { bv = key; best = v; }
        }
        out[r] = (int32_t)best;
    }
}
