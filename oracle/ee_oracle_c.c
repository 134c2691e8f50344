/* TEST INFRASTRUCTURE - NOT PRODUCT CODE.
 *
 * C / OpenMP restatement (float32) of the reference's early-exit evaluation path, the "build's own CPU restatement (C++/OpenMP, fp32)" that
 * SURVEY.md section 8d asks to time beside torch-CPU.  Same algorithm and the same line references as oracle/ee_oracle.py (EE/... =
 * /root/reference/EE/..., HF: = transformers models/layoutlmv3/modeling_layoutlmv3.py at 5.15.0); pinned to the numpy oracle and through it to
 * the golden vectors of the composed reference by tests/test_oracle_golden.py.  Only tests/ and bench.py's cpu_baseline leg load it
 * (oracle/ee_oracle_c.py); the product never does.
 *
 *   full depth, every exit evaluated (LayoutLMv3EEForSequenceClassification.forward EE/models/LayoutLMv3.py:696-749, 871-896 ->
 *   LayoutLMv3ModelEE.forward :375-665 -> LayoutLMv3EncoderEE.forward :151-305); the policy runs on the returned store in numpy.
 *
 * Layout: LayoutLMv3 only (the image-only DiT variant stays on the numpy oracle).  The relative-position bias is NOT materialised per layer
 * (the reference builds (B, heads, S, S) once, HF:415-457): per (document, head) a row of scores is biased from the three bucket LUTs.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t H, L, heads, I, vocab, max_pos, pad_id, max_2d, cs, ss, bins1, maxd1, bins2, maxd2, R, P, C, K, T;
    float eps;
} eec_cfg;

/* weights in the order oracle/ee_oracle_c.py assembles them */
enum { W_WORD = 0, W_TYPE, W_POS, W_X, W_Y, W_HT, W_WD, W_EMB_G, W_EMB_B, W_PATCH_W, W_PATCH_B, W_CLS, W_POS_EMBED, W_NORM_G, W_NORM_B,
       W_LN_G, W_LN_B, W_REL1, W_RELX, W_RELY, W_GLOBAL_COUNT };
enum { LW_Q_W = 0, LW_Q_B, LW_K_W, LW_K_B, LW_V_W, LW_V_B, LW_AO_W, LW_AO_B, LW_AO_G, LW_AO_BETA, LW_F1_W, LW_F1_B, LW_F2_W, LW_F2_B, LW_F_G,
       LW_F_BETA, LW_COUNT };
typedef struct { const float *dense_w, *dense_b, *out_w, *out_b; int32_t out_dim; } eec_head;

/* y[M][N] = x[M][K] W[N][K]^T + b  (nn.Linear).  Four rows share every W row; the k loop is an explicit SIMD reduction. */
static void linear(const float* x, int M, int K, const float* W, const float* b, int N, float* y) {
#pragma omp parallel for schedule(static)
    for (int m0 = 0; m0 < M; m0 += 4) {
        const int mr = M - m0 < 4 ? M - m0 : 4;
        const float* a0 = x + (size_t)m0 * K;
        const float* a1 = x + (size_t)(m0 + (mr > 1 ? 1 : 0)) * K;
        const float* a2 = x + (size_t)(m0 + (mr > 2 ? 2 : 0)) * K;
        const float* a3 = x + (size_t)(m0 + (mr > 3 ? 3 : 0)) * K;
        for (int n = 0; n < N; ++n) {
            const float* w = W + (size_t)n * K;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma omp simd reduction(+ : s0, s1, s2, s3)
            for (int k = 0; k < K; ++k) {
                const float wk = w[k];
                s0 += a0[k] * wk; s1 += a1[k] * wk; s2 += a2[k] * wk; s3 += a3[k] * wk;
            }
            const float bn = b ? b[n] : 0.f;
            y[(size_t)m0 * N + n] = s0 + bn;
            if (mr > 1) y[(size_t)(m0 + 1) * N + n] = s1 + bn;
            if (mr > 2) y[(size_t)(m0 + 2) * N + n] = s2 + bn;
            if (mr > 3) y[(size_t)(m0 + 3) * N + n] = s3 + bn;
        }
    }
}

/* torch.nn.LayerNorm on rows: biased variance, two passes */
static void layer_norm_rows(float* x, int M, int H, const float* g, const float* b, float eps) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        float* r = x + (size_t)m * H;
        float s = 0.f;
#pragma omp simd reduction(+ : s)
        for (int i = 0; i < H; ++i) s += r[i];
        const float mean = s / (float)H;
        float v = 0.f;
#pragma omp simd reduction(+ : v)
        for (int i = 0; i < H; ++i) { const float d = r[i] - mean; v += d * d; }
        const float rstd = 1.0f / sqrtf(v / (float)H + eps);
        for (int i = 0; i < H; ++i) r[i] = (r[i] - mean) * rstd * g[i] + b[i];
    }
}

static void exit_head(const eec_head* h, const float* x, int B, int H, float* hid, float* out) {       /* EE/models/LayoutLMv3.py:86-93, HF:799-823 */
    const float* in = x;
    if (h->dense_w) {
        linear(x, B, H, h->dense_w, h->dense_b, H, hid);
        for (size_t i = 0; i < (size_t)B * H; ++i) hid[i] = tanhf(hid[i]);
        in = hid;
    }
    linear(in, B, H, h->out_w, h->out_b, h->out_dim, out);
}

static void mean_rows(const float* x, int B, int S, int H, float* out) {          /* mean over ALL S positions of every document */
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < H; ++i) {
            float s = 0.f;
            for (int t = 0; t < S; ++t) s += x[((size_t)b * S + t) * H + i];
            out[(size_t)b * H + i] = s / (float)S;
        }
}

/* exit_kinds[j]: -1 vision_avg, -2 text_avg, -3 text_visual_concat, l > 0 encoder layer l (ascending, embedding exits first: the reference's
 * evaluation order).  heads[j] the exit's head, heads[n_exits] the final classifier.  gate != 0: the store holds classifier(gate input)
 * (EE/models/LayoutLMv3.py:764-792).  lut1 / lut2: relative_position_bucket over delta in [-1023, 1023] (index delta + 1023) for the 1D and the
 * 2D bias, made by oracle/ee_oracle.py's bucket_lut (pinned to HF's own function by tests/golden/bucket_lut.npz; torch evaluates the log branch
 * in float32 and truncates, which a C logf need not reproduce on the bucket edges).  store (n_exits + 1, B, K) float64 = what the harness keeps
 * (EE/utils.py:160-193).  hidden_cls (L + 1, B, H) float32 or NULL. */
int eec_forward(const eec_cfg* c, const float* const* gw, const float* const* lw, const eec_head* heads, int32_t n_exits, const int32_t* exit_kinds,
                int32_t gate, const uint8_t* lut1, const uint8_t* lut2, int32_t B, const int64_t* input_ids, const int64_t* attention_mask, const int64_t* bbox, const float* pixel_values,
                double* store, float* hidden_cls) {
    const int H = c->H, T = c->T, G = c->R / c->P, NP = G * G, Pv = NP + 1, S = T + Pv, nh = c->heads, d = H / nh, K = c->K, I = c->I;
    const size_t rows = (size_t)B * S;
    float* x = malloc(rows * H * 4);
    float* y = malloc(rows * H * 4);
    float* q = malloc(rows * H * 4);
    float* k = malloc(rows * H * 4);
    float* v = malloc(rows * H * 4);
    float* ctx = malloc(rows * H * 4);
    float* h1 = malloc(rows * I * 4);
    float* pooled = malloc((size_t)B * H * 4);
    float* hid = malloc((size_t)B * H * 4);
    float* lg = malloc((size_t)B * (K > 2 ? K : 2) * 4);
    int* pos = malloc(rows * sizeof(int));
    int* bx = malloc(rows * sizeof(int));
    int* by = malloc(rows * sizeof(int));
    unsigned char* valid = malloc(rows);
    if (!x || !y || !q || !k || !v || !ctx || !h1 || !pooled || !hid || !lg || !pos || !bx || !by || !valid) return 1;
    int e = 0;
    const eec_head* cls_head = &heads[n_exits];
#define RUN_EXIT(inp)                                                                                      \
    do {                                                                                                   \
        const eec_head* hh = gate ? cls_head : &heads[e];                                                  \
        exit_head(hh, (inp), B, H, hid, lg);                                                               \
        for (int b_ = 0; b_ < B; ++b_)                                                                     \
            for (int k_ = 0; k_ < K; ++k_) store[((size_t)e * B + b_) * K + k_] = (double)lg[(size_t)b_ * hh->out_dim + k_]; \
        ++e;                                                                                               \
    } while (0)

    /* ---- A1 image embeddings: Conv2d(k = s = P) as a GEMM over (c, ky, kx); cls token; + pos_embed; LayerNorm eps 1e-6 (HF:71-83, 603-618) ---- */
    {
        const int Kp = c->C * c->P * c->P;
        float* patches = h1;             /* (B * NP, Kp) */
#pragma omp parallel for schedule(static)
        for (int bp = 0; bp < B * NP; ++bp) {
            const int b = bp / NP, p = bp % NP, gy = p / G, gx = p % G;
            for (int ch = 0; ch < c->C; ++ch)
                for (int py = 0; py < c->P; ++py)
                    for (int px = 0; px < c->P; ++px)
                        patches[(size_t)bp * Kp + (ch * c->P + py) * c->P + px] =
                            pixel_values[(((size_t)b * c->C + ch) * c->R + gy * c->P + py) * c->R + gx * c->P + px];
        }
        linear(patches, B * NP, Kp, gw[W_PATCH_W], gw[W_PATCH_B], H, ctx);
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < Pv; ++t) {
                float* r = x + ((size_t)b * S + T + t) * H;
                const float* src = t == 0 ? gw[W_CLS] : ctx + ((size_t)b * NP + t - 1) * H;
                for (int i = 0; i < H; ++i) r[i] = src[i] + gw[W_POS_EMBED][(size_t)t * H + i];
            }
        for (int b = 0; b < B; ++b) layer_norm_rows(x + ((size_t)b * S + T) * H, Pv, H, gw[W_NORM_G], gw[W_NORM_B], 1e-6f);
    }
    if (e < n_exits && exit_kinds[e] == -1) {                                      /* vision_avg, EE/models/LayoutLMv3.py:465-483 */
        for (int b = 0; b < B; ++b) mean_rows(x + ((size_t)b * S + T) * H, 1, Pv, H, pooled + (size_t)b * H);
        RUN_EXIT(pooled);
    }
    /* ---- A2 text embeddings (HF:160-199, 112-146): word + type[0] + position[cumsum] + six spatial slices; LayerNorm ---- */
    for (int b = 0; b < B; ++b) {
        int64_t run = 0;
        for (int t = 0; t < T; ++t) {
            const int64_t id = input_ids[(size_t)b * T + t];
            const int m = id != c->pad_id;
            run += m;
            const int64_t pid = run * m + c->pad_id;
            const int64_t* bb = bbox + ((size_t)b * T + t) * 4;
            int64_t hgt = bb[3] - bb[1], wid = bb[2] - bb[0];
            const int64_t hi = c->max_2d - 1;
            hgt = hgt < 0 ? 0 : hgt > hi ? hi : hgt;
            wid = wid < 0 ? 0 : wid > hi ? hi : wid;
            float* r = x + ((size_t)b * S + t) * H;
            const float* wrow = gw[W_WORD] + (size_t)id * H;
            const float* prow = gw[W_POS] + (size_t)pid * H;
            for (int i = 0; i < H; ++i) r[i] = wrow[i] + gw[W_TYPE][i] + prow[i];
            const float* seg[6] = {gw[W_X] + (size_t)bb[0] * c->cs, gw[W_Y] + (size_t)bb[1] * c->cs, gw[W_X] + (size_t)bb[2] * c->cs,
                                   gw[W_Y] + (size_t)bb[3] * c->cs, gw[W_HT] + (size_t)hgt * c->ss, gw[W_WD] + (size_t)wid * c->ss};
            int o = 0;
            for (int sgi = 0; sgi < 6; ++sgi) {
                const int wdt = sgi < 4 ? c->cs : c->ss;
                for (int i = 0; i < wdt; ++i) r[o + i] += seg[sgi][i];
                o += wdt;
            }
        }
        layer_norm_rows(x + (size_t)b * S * H, T, H, gw[W_EMB_G], gw[W_EMB_B], c->eps);
    }
    if (e < n_exits && exit_kinds[e] == -2) {                                      /* text_avg (pads included), :519-534 */
        for (int b = 0; b < B; ++b) mean_rows(x + (size_t)b * S * H, 1, T, H, pooled + (size_t)b * H);
        RUN_EXIT(pooled);
    }
    /* ---- A3: cat(text, visual), LayerNorm (:549-566); positions / boxes / mask of the S rows ---- */
    layer_norm_rows(x, (int)rows, H, gw[W_LN_G], gw[W_LN_B], c->eps);
    if (e < n_exits && exit_kinds[e] == -3) {                                      /* text_visual_concat: mean over ALL positions, :581-605 */
        mean_rows(x, B, S, H, pooled);
        RUN_EXIT(pooled);
    }
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < S; ++t) {
            const size_t r = (size_t)b * S + t;
            if (t < T) {
                pos[r] = t;                                                        /* arange(T), NOT the pad-aware embedding positions (:559-563) */
                bx[r] = (int)bbox[((size_t)b * T + t) * 4 + 0];
                by[r] = (int)bbox[((size_t)b * T + t) * 4 + 3];                    /* the 2D bias buckets x0 and y1 (HF:433-434) */
                valid[r] = attention_mask ? attention_mask[(size_t)b * T + t] != 0 : 1;
            } else {
                const int p = t - T;
                pos[r] = p;
                if (p == 0) { bx[r] = 1; by[r] = 999; }                            /* cls box [1,1,999,999] (HF:575-596) */
                else {
                    const int gy = (p - 1) / G, gx = (p - 1) % G;
                    bx[r] = (1000 * gx) / G;
                    by[r] = (1000 * (gy + 1)) / G;
                }
                valid[r] = 1;
            }
        }
    if (hidden_cls)
        for (int b = 0; b < B; ++b) memcpy(hidden_cls + (size_t)b * H, x + (size_t)b * S * H, (size_t)H * 4);
    const float sd = sqrtf((float)d);
    const float fmin = -3.4028234663852886e38f;                                    /* finfo(float32).min, :622-624 */
    /* ---- encoder layers (HF:235-303, 343-368, 485-512) ---- */
    for (int l = 0; l < c->L; ++l) {
        const float* const* w = lw + (size_t)l * LW_COUNT;
        linear(x, (int)rows, H, w[LW_Q_W], w[LW_Q_B], H, q);
        linear(x, (int)rows, H, w[LW_K_W], w[LW_K_B], H, k);
        linear(x, (int)rows, H, w[LW_V_W], w[LW_V_B], H, v);
        /* attention: scores = (Q / sqrt d) K^T + (rel_pos + rel_2d_pos) / sqrt d + mask (HF:263-272); CogView softmax = stable softmax (HF:223-233) */
#pragma omp parallel
        {
            float* sc = malloc((size_t)S * 4);
#pragma omp for collapse(2) schedule(dynamic, 8)
            for (int bh = 0; bh < B * nh; ++bh)
                for (int i = 0; i < S; ++i) {
                    const int b = bh / nh, hd = bh % nh;
                    const size_t ri = (size_t)b * S + i;
                    const float* qi = q + ri * H + (size_t)hd * d;
                    const float* t1 = gw[W_REL1] + (size_t)hd * c->bins1;
                    const float* tx = gw[W_RELX] + (size_t)hd * c->bins2;
                    const float* ty = gw[W_RELY] + (size_t)hd * c->bins2;
                    float mx = -INFINITY;
                    for (int j = 0; j < S; ++j) {
                        const size_t rj = (size_t)b * S + j;
                        const float* kj = k + rj * H + (size_t)hd * d;
                        float s = 0.f;
#pragma omp simd reduction(+ : s)
                        for (int t = 0; t < d; ++t) s += (qi[t] / sd) * kj[t];
                        const float bias = t1[lut1[pos[rj] - pos[ri] + 1023]] + (tx[lut2[bx[rj] - bx[ri] + 1023]] + ty[lut2[by[rj] - by[ri] + 1023]]);
                        s = s + bias / sd;
                        s = s + (valid[rj] ? 0.f : fmin);
                        sc[j] = s;
                        mx = s > mx ? s : mx;
                    }
                    float sum = 0.f;
                    for (int j = 0; j < S; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
                    float* o = ctx + ri * H + (size_t)hd * d;
                    for (int t = 0; t < d; ++t) o[t] = 0.f;
                    for (int j = 0; j < S; ++j) {
                        const float p = sc[j] / sum;
                        const float* vj = v + ((size_t)b * S + j) * H + (size_t)hd * d;
#pragma omp simd
                        for (int t = 0; t < d; ++t) o[t] += p * vj[t];
                    }
                }
            free(sc);
        }
        linear(ctx, (int)rows, H, w[LW_AO_W], w[LW_AO_B], H, y);
        for (size_t i = 0; i < rows * H; ++i) y[i] += x[i];
        layer_norm_rows(y, (int)rows, H, w[LW_AO_G], w[LW_AO_BETA], c->eps);       /* HF:299-303 */
        linear(y, (int)rows, H, w[LW_F1_W], w[LW_F1_B], I, h1);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < rows * (size_t)I; ++i) h1[i] = 0.5f * h1[i] * (1.0f + erff(h1[i] * 0.70710678118654752440f));      /* GELU (erf) */
        linear(h1, (int)rows, I, w[LW_F2_W], w[LW_F2_B], H, x);
        for (size_t i = 0; i < rows * H; ++i) x[i] += y[i];
        layer_norm_rows(x, (int)rows, H, w[LW_F_G], w[LW_F_BETA], c->eps);          /* HF:508-512 */
        if (hidden_cls)
            for (int b = 0; b < B; ++b) memcpy(hidden_cls + ((size_t)(l + 1) * B + b) * H, x + (size_t)b * S * H, (size_t)H * 4);
        if (e < n_exits && exit_kinds[e] == l + 1) {                               /* CLS row -> exit head, :222-248 */
            for (int b = 0; b < B; ++b) memcpy(pooled + (size_t)b * H, x + (size_t)b * S * H, (size_t)H * 4);
            RUN_EXIT(pooled);
        }
    }
    for (int b = 0; b < B; ++b) memcpy(pooled + (size_t)b * H, x + (size_t)b * S * H, (size_t)H * 4);
    exit_head(cls_head, pooled, B, H, hid, lg);                                     /* final classifier, :730-731 */
    for (int b = 0; b < B; ++b)
        for (int kk = 0; kk < K; ++kk) store[((size_t)n_exits * B + b) * K + kk] = (double)lg[(size_t)b * K + kk];
#undef RUN_EXIT
    free(x); free(y); free(q); free(k); free(v); free(ctx); free(h1); free(pooled); free(hid); free(lg);
    free(pos); free(bx); free(by); free(valid);
    return e == n_exits ? 0 : 2;
}
