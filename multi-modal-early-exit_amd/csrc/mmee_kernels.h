// Parameter blocks + launchers of the row / exit kernels (prep_embed.hip, exit_ops.hip).
#pragma once
#include "mmee_common.h"

namespace mmee {

struct PrepArgs {
    const long long* input_ids;      // (B,T)
    const long long* attention_mask; // (B,T) or null
    const long long* bbox;           // (B,T,4)
    const long long* position_ids;   // (B,T) or null
    const long long* token_type_ids; // (B,T) or null (range check only; the embedding kernel reads them)
    int B, T, Pv, G;
    int pad_id, vocab, max_2d, max_pos, type_vocab;
    int dense_rows;
    // outputs
    int* text_dst;                   // (B,T) row index inside the document, -1 = dropped pad row
    int* emb_pos;                    // (B,T) position id for the position-embedding lookup
    int* ntext;                      // (B)   kept text rows
    int* doc_off;                    // (B+1) stage-0 dense offsets
    int* x_src;                      // (B)   physical offsets (= doc_off at stage 0)
    int* doc_orig;                   // (B)
    RowMeta* meta;
    StageCounts* counts;
    int* err_flag;                   // bit0 token id, bit1 bbox, bit2 position id, bit3 token_type id out of range
};

struct EmbedArgs {
    const long long* input_ids;
    const long long* token_type_ids; // or null
    const long long* bbox;
    const int* emb_pos;
    const int* text_dst;
    const int* ntext;
    const int* doc_off;
    int B, T, Pv, H, cs, ss, max_2d, vocab, type_vocab;
    const float *word, *type, *pos, *xtab, *ytab, *htab, *wtab;
    const float* inputs_embeds;      // (B,T,H) or null: read in place of word[input_ids] (ee_set_inputs_embeds)
    const float *ln1_g, *ln1_b;      // text: embeddings.LayerNorm; visual: layoutlmv3.norm
    float eps1;
    const float *ln2_g, *ln2_b;      // layoutlmv3.LayerNorm
    float eps2;
    const float *cls_token, *pos_embed, *vis_raw;
    float* X;
    void* Xs;                        // when given, the rows are written as split-f16 planes (x split_scale) here and X is not written
    float split_scale;
    int* err_flag;                   // Xs: kErrSplitOverflow
    float* text_part;                // (B, ceil(T/32), H) or null
    float* vis_part;                 // (B, ceil(Pv/32), H) or null
    float* cat_part;                 // (B, cat_chunks, H) or null; text chunks first, then visual chunks
    int cat_chunks;
};

struct HeadOutArgs {
    const float* in;                 // rows of length H
    int ld;
    const int* gather;               // optional: input row of active doc i is in[gather[i]]
    const float* W;                  // [Ko][H]
    const float* b;                  // [Ko]
    int H, Ko;
    const int* n_docs_ptr;
    float* out;                      // [n_docs][Ko]
};

// thresholds, then temperatures, of one ee_graph_launch: a kernel ARGUMENT of set_thresholds_kernel (no host buffer has to outlive the call)
struct ThrPack {
    double v[2 * (64 + 4)];          // 2 x (MMEE_MAX_ENCODER_EXITS + 3 embedding exits + final)
    int n;
};

struct DecideArgs {
    const float* pol_logits;         // [n_docs][K]   logits the policy sees (ramp: head logits; gate: classifier(gate input))
    const float* head_logits;        // [n_docs][Kh]  raw exit-head logits (== pol_logits for ramps); null for the final stage
    int K, Kh;
    double thr, temp;
    const double* thr_ptr;           // captured-graph forwards (ee_graph_capture): threshold / temperature of exit e at [exit_index] of device
    const double* temp_ptr;          // vectors refreshed in front of every replay; null: the by-value arguments above
    int criterion, is_final, no_exit, exit_index, B;
    // current stage
    const StageCounts* counts;
    const int* doc_orig;
    const int* doc_off;              // dense offsets (n_docs + 1)
    const int* x_phys;               // physical X offsets of the current stage's documents
    // next stage
    StageCounts* n_counts;
    int* n_doc_orig;
    int* n_doc_off;
    int* n_x_src;
    int* n_meta_src;
    // outputs (original document numbering)
    float* out_logits;               // (B,K)
    int* out_exit;                   // (B)
    float* out_conf;                 // (B)
    float* out_all_logits;           // (E+1,B,K)
    float* out_all_crit;             // (E+1,B)
    float* out_head_logits;          // (E,B,Kh)
    float* out_head_crit;            // (E,B)
};

// X-space CLS probe (xprobe.hip)
struct XProbeArgs {
    const char* xs;                  // split rows of X (LayerNorm output), H * 4 bytes per row, scaled by 1 / xs_inv
    float xs_inv;
    const int* x_phys;               // [n_docs] physical first (CLS) row of every active document
    const int* doc_off;              // [n_docs + 1] dense row offsets (lengths; context rows are written at doc_off[d])
    const int* doc_orig;             // [n_docs] original document id (pair-index slab)
    const StageCounts* counts;
    const float* qc;                 // [n_docs][H] Q of the CLS rows, already divided by sqrt(d)
    const float *wk, *bk, *bv;       // key projection and the biases (rows h * 64 + t of the fused weight), f32
    const float* wv_s;               // value projection as split-f16 rows (the fused Q | K | V weight of ee_finalize), scaled by 1 / wv_inv
    float wv_inv;
    float *u, *s0, *cvec;            // scratch: [max_docs][heads][H], [max_docs][heads][2] (q . b_k, plane scale of u), [max_docs][heads][H]
    int *order, *ticket;             // scratch: [max_docs] documents by falling length, [1] ticket counter of xprobe_attn_kernel
    void* ctx;                       // out: context rows (split planes scaled by ctx_scale), row doc_off[d]
    float ctx_scale;
    const unsigned* pair_idx;
    size_t idx_doc_stride;
    const float *w1, *wx, *wy;       // raw bucket tables [heads][bins]
    int bins1, bins2;
    float inv_sqrt_d;
    int H, heads;
    int* err_flag;
};
bool xprobe_supports(const XProbeArgs& a, int max_len);
void launch_xprobe(const XProbeArgs& a, int max_docs, int max_len, int num_cus, hipStream_t s);

struct ImageDesc {
    long long offset;                // byte offset of the image inside the packed uint8 buffer (HWC, or HW when c == 1)
    int h, w, c, pad;
};

void launch_preprocess_images(const unsigned char* images, const ImageDesc* desc, int B, int R, int KMAX, int max_h, int2* bounds,
                              int* kk, unsigned char* tmp, const float* lut, float* out, unsigned char* out_u8, hipStream_t s);
void launch_collate_pad(const long long* ids, const long long* boxes, const long long* offsets, int B, int T, long long pad_id,
                        long long* out_ids, long long* out_mask, long long* out_bbox, hipStream_t s);
void launch_prep(const PrepArgs& a, hipStream_t s);
void launch_prep_uniform(int B, int Pv, int* doc_off, int* x_src, int* doc_orig, RowMeta* meta, StageCounts* counts, hipStream_t s);
void launch_embed_text(const EmbedArgs& a, hipStream_t s);
void launch_embed_visual(const EmbedArgs& a, hipStream_t s);
void launch_pool_finish(const float* part, int chunks, int H, float count, float* pooled, int B, hipStream_t s);
void launch_ln_rows(const float* src, float* dst, const int* row_src, const int* n_rows_ptr, int max_rows, int H,
                    const float* g, const float* b, float eps, int num_cus, hipStream_t s, void* dst_split = nullptr,
                    float split_scale = 1.0f, int* err_flag = nullptr, int pre_parts = 0, size_t pre_stride = 0, const float* pre_bias = nullptr,
                    const void* pre_resid = nullptr, float pre_resid_inv = 0.f);
void launch_embed_beit(const float* patch, const float* cls, const float* pos, int B, int Pv, int H, float* X, hipStream_t s);
void launch_patch_mean(const float* X, int H, const int* x_phys, const int* doc_off, const int* n_docs_ptr, float* pooled,
                       int max_docs, hipStream_t s);
void launch_head_out(const HeadOutArgs& a, int max_docs, hipStream_t s);
void launch_decide(const DecideArgs& a, hipStream_t s);
void launch_pack_results(const float* logits, const int* exit_layer, const float* conf, int n, int K, int* rows, hipStream_t s);
void launch_unpack_results(const int* rows, int n, int K, float* logits, int* exit_layer, float* conf, hipStream_t s);
void launch_compact_rows(const StageCounts* n_counts, const int* n_doc_off, const int* n_x_src, const int* n_meta_src,
                         const RowMeta* meta_old, RowMeta* meta_new, int* row_src, int max_docs, int num_cus, hipStream_t s);
// out[doc_orig ? doc_orig[i] : i] = CLS row of active document i; split_inv != 0: X holds split-f16 rows scaled by 1 / split_inv
// every row of every document -> out[(d * (T + Pv) + position)][H] (ee_set_hidden_states_out); text_dst null: image-only, Pv rows per document
// attention_maps.hip: side kernels of `output_attentions` / `head_mask` (dump-all, whole layers; never on the hot path)
bool attention_probs_supports(int S);
void launch_attention_probs(const float* qkv, int ld, int split, float qkv_scale, const RowMeta* meta, const int* doc_off, const float* t1,
                            const float* tx, const float* ty, int n1, int c1, int n2, int c2, int H, int heads, int S, int B,
                            const float* head_scale, float* out, hipStream_t s);
void launch_head_scale_ctx(float* ctx, int ld, const int* n_rows_ptr, int max_rows, int H, const float* head_scale, int split, float scale,
                           int num_cus, int* err_flag, hipStream_t s);
void launch_rows_to_padded(const float* X, float split_inv, int H, int B, int T, int Pv, const int* text_dst, const int* ntext, const int* doc_off,
                           float* out, hipStream_t s);
void launch_gather_cls(const float* X, int H, const int* x_phys, const int* doc_orig, const int* n_docs_ptr,
                       float* out, int max_docs, hipStream_t s, float split_inv = 0.f);
void launch_policy_scan(const double* logits, int E1, int N, int K, const double* thr_dev, int* exits, double* pred,
                        double* conf, int* counts, hipStream_t s);
void launch_threshold_sweep(const double* conf, const unsigned char* correct, int E1, int N, const double* thr, int V,
                            double* acc, double* mean_exit, int* hist, hipStream_t s);
void launch_msp_table(const double* logits, const long long* refs, int E1, int N, int K, double* conf, unsigned char* correct,
                      hipStream_t s);
void launch_temperature_fit(const double* logits, const long long* labels, int E1, int N, int K, int max_iter, double* T_out,
                            double* nll_out, double* acc_out, double* conf_out, int* iters_out, hipStream_t s);
void launch_build_value_tables(const float* w1, const float* wx, const float* wy, const unsigned char* lut1,
                               const unsigned char* lut2, int heads, int bins1, int bins2, int n1, int n2, float inv_sqrt_d,
                               float* t1, float* tx, float* ty, hipStream_t s);

}  // namespace mmee
