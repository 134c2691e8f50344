/*
 * mmee.h — C-ABI of the MI355X-native early-exit document-classification path (libmmee_hip.so).
 *
 * The reference (Jordy-VL/multi-modal-early-exit) is pure Python and has no FFI layer; its boundary for this path is
 * two Python call surfaces.  Each entry point below names the reference interface it replaces:
 *
 *   ee_create / ee_load_tensor / ee_finalize   <- configs.build_model -> LayoutLMv3EEForSequenceClassification
 *                                                 .from_pretrained (EE/configs.py:389-411; model tree
 *                                                 EE/models/LayoutLMv3.py:308-356, 669-694).  Tensors are addressed by
 *                                                 their HF parameter names (EE/models/
 *                                                 EELayoutLM_exit_named_parameters-wotherexits.json).
 *   ee_forward                                  <- LayoutLMv3EEForSequenceClassification.forward
 *                                                 (EE/models/LayoutLMv3.py:696-749, 871-896) + the harness loop
 *                                                 utils.get_logits (EE/utils.py:169-193) + the early-exit decision of
 *                                                 Policy.max_confidence_global_thresholding_policy /
 *                                                 accuracy_calibration_heuristic (EE/policy.py:12-111) fused in,
 *                                                 per-exit temperature of EE/generic_scaling.py:54-61 applied first.
 *   ee_policy_scan                              <- Policy.* on a dumped (E+1, N, K) logits array (EE/policy.py:28-45,
 *                                                 87-104; called from EE/eval.py:87-98).
 *   ee_threshold_sweep                          <- thresh.opt1 / large_scale.opt0_2D vectorised exit-index search
 *                                                 (EE/thresh.py:184-215, EE/large_scale.py:68-84).
 *
 * Conventions: every function returns 0 on success, non-zero on error (message via ee_last_error).  All pointers
 * marked "dev" are device (HBM) pointers borrowed from the caller for the duration of the enqueued work; kernels are
 * enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream); no call synchronises with the host
 * unless its comment says so.  One handle per device; a handle is not thread-safe.
 */
#ifndef MMEE_H
#define MMEE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMEE_ABI_VERSION 3
#define MMEE_MAX_ENCODER_EXITS 64

/* embedding-level exits, in the order the reference evaluates them (EE/models/LayoutLMv3.py:465-605) */
enum { MMEE_EXIT_VISION_AVG = 0, MMEE_EXIT_TEXT_AVG = 1, MMEE_EXIT_TEXT_VISUAL_CONCAT = 2 };
/* encoder_layer_strategy (EE/models/EE_modules.py:167-172) */
enum { MMEE_STRATEGY_RAMP = 0, MMEE_STRATEGY_GATE = 1 };
/* inference_strategy (EE/models/EE_modules.py:116-146): max_confidence exits on crit > thr, entropy on crit < thr */
enum { MMEE_CRIT_MAX_CONFIDENCE = 0, MMEE_CRIT_ENTROPY = 1 };
/* model family: LayoutLMv3 (text + layout + image, the reference's EE model) or BEiT / DiT (image only; BASELINE configs[4],
 * the reference's "dit" branch EE/configs.py:429-449 — exit heads there are this build's extrapolation) */
enum { MMEE_ARCH_LAYOUTLMV3 = 0, MMEE_ARCH_BEIT = 1 };
/* arithmetic of the encoder GEMMs */
/* MMEE_PREC_F32: v_mfma_f32_32x32x2_f32 on f32 operands.  MMEE_PREC_F32_SPLIT: the four big Linear layers of every encoder
 * layer run on the f16 matrix cores with every f32 operand split into two f16 planes (hi + lo, 22 significant bits) and
 * three MFMA terms per product, f32 accumulation — measured at least as accurate as the f32 MFMA chain, same
 * 1e-4 / bit-exact parity bar; needs hidden_size and intermediate_size to be multiples of 256.  Range: the f16 planes hold
 * 16 x LayerNorm outputs, 16 x Q/K/V, 64 x attention context, 16 x GELU outputs and 2^e x weights (e chosen per tensor at
 * ee_finalize); values beyond +-60000 / scale (|activation| > 3750, |context| > 937) are clamped, far outside what
 * LayoutLMv3 / DiT checkpoints produce — use MMEE_PREC_F32 for a model that exceeds it.  MMEE_PREC_BF16 is reserved and
 * rejected: plain bf16 cannot meet the tolerance. */
enum { MMEE_PREC_F32 = 0, MMEE_PREC_BF16 = 1, MMEE_PREC_F32_SPLIT = 2 };
/* ee_load_tensor dtypes */
enum { MMEE_DT_F32 = 0, MMEE_DT_F16 = 1, MMEE_DT_BF16 = 2 };
/* ee_forward flags */
enum {
    MMEE_FLAG_DENSE_ROWS = 1,  /* keep all T text rows per document (pad rows computed, masked as keys) instead of the
                                  ragged layout that drops pad rows; the two layouts agree to ROUNDING (<= 2e-5 on
                                  logits: the attention sums a row's keys in tiles whose boundaries differ), not
                                  to the bit; exit indices are equal.  This is the A/B switch                    */
    MMEE_FLAG_NO_EXIT = 2,     /* dump-all mode: evaluate every exit for every document, nobody leaves early
                                  (the reference's own behaviour, EE/utils.py:63-71 "impossible thresholds")        */
    MMEE_FLAG_WHOLE_LAYERS = 4,/* run every encoder layer whole before its exit decision (what the reference does,
                                  EE/models/LayoutLMv3.py:757-768), never "probe first" (ee_last_layer_plan)          */
    MMEE_FLAG_PROBE_ALWAYS = 8,/* probe first at every layer that ends in a decision, whatever ee_set_probe_mask pinned.  With neither flag
                                  and no pinned mask this is also the DEFAULT (round 5): the schedule is a function of the call, never of
                                  timing or of earlier forwards.  Whole layers, probe-first and any pinned mix give identical results bit for
                                  bit without MMEE_FLAG_XPROBE; the flags only pin the schedule, for A/B runs and tests          */
    MMEE_FLAG_ONE_TERM = 32,   /* REPORTED low-precision mode, never a parity path (SURVEY 8d config 2 "bf16 throughput mode reports its measured
                                  deviation separately"): the layer GEMMs and the attention of an MMEE_PREC_F32_SPLIT LayoutLMv3 handle run ONE f16
                                  MFMA term per MAC (hi planes only, f32 accumulate) instead of three; CLS probes and exit heads keep three.  Logits
                                  leave the 1e-4 bar and exit indices may flip: bench.py reports rate, max |dlogit| and flip rate as `lowprec` */
    MMEE_FLAG_XPROBE = 16      /* probe-first layers take the CLS context in X space (csrc/xprobe.hip): score_j = (W_k^T q) . x_j + q . b_k,
                                  ctx = W_v (sum_j p_j x_j) + b_v.  No Q | K | V projection exists when the decision is taken: the layer's
                                  Q | K | V GEMM then runs for the documents that STAY only, and not at all in the last layer.  A
                                  re-association of the same arithmetic (~1e-6 on the CLS row): exit indices and the 1e-4 logit bar hold,
                                  bit-identity with MMEE_FLAG_WHOLE_LAYERS does not.  LayoutLMv3, MMEE_PREC_F32_SPLIT, no dump-all */
};

typedef struct ee_handle ee_handle;

typedef struct ee_config {
    int32_t abi_version;            /* MMEE_ABI_VERSION */
    /* HF LayoutLMv3Config fields the path reads */
    int32_t hidden_size, num_hidden_layers, num_attention_heads, intermediate_size;
    int32_t vocab_size, max_position_embeddings, type_vocab_size, pad_token_id;
    int32_t max_2d_position_embeddings, coordinate_size, shape_size;
    int32_t rel_pos_bins, max_rel_pos, rel_2d_pos_bins, max_rel_2d_pos;
    int32_t input_size, patch_size, num_channels, num_labels;
    float layer_norm_eps;
    /* ExitConfig (EE/models/EE_modules.py:175-195) */
    int32_t n_embedding_exits;                      /* 0..3 */
    int32_t embedding_exits[3];                     /* MMEE_EXIT_*, evaluation order */
    int32_t n_encoder_exits;
    int32_t encoder_exit_layers[MMEE_MAX_ENCODER_EXITS];   /* 1-based, ascending; head k <-> encoder.early_exits.k */
    int32_t exit_head_num_layers;                   /* 1 or 2 */
    int32_t strategy;                               /* MMEE_STRATEGY_* */
    int32_t criterion;                              /* MMEE_CRIT_* */
    /* workspace sizing */
    int32_t max_docs;                               /* largest B one ee_forward call may pass */
    int32_t max_text_len;                           /* largest T */
    int32_t precision;                              /* MMEE_PREC_* */
    /* model family (appended in ABI version 2) */
    int32_t arch;                                   /* MMEE_ARCH_* */
    int32_t use_abs_pos;                            /* BEiT: use_absolute_position_embeddings */
    int32_t layer_scale;                            /* BEiT: layer_scale_init_value > 0 (lambda_1 / lambda_2 present) */
    int32_t use_mean_pooling;                       /* BEiT: must be 1 (DiT); pooled = LayerNorm(mean of patch tokens) */
} ee_config;

int ee_create(const ee_config* cfg, ee_handle** out);
int ee_destroy(ee_handle* h);
const char* ee_last_error(const ee_handle* h);      /* h may be NULL: error of the last failed ee_create */

/* Copy one parameter into the handle (the library owns its weight memory and may re-layout it).  `name` is the HF
 * parameter name; `data` is a host pointer (is_device = 0) or a device pointer (is_device = 1) to a C-contiguous
 * tensor of `dtype`; shape is checked against the config.  Unknown names are an error. */
int ee_load_tensor(ee_handle* h, const char* name, const void* data, const int64_t* shape, int32_t ndim,
                   int32_t dtype, int32_t is_device);
/* Verify every parameter the config needs was loaded and build derived tables (relative-position value tables).
 * Synchronises with the device. */
int ee_finalize(ee_handle* h);
/* Number of parameters the config expects / name of the i-th one (for loaders and tests). */
int32_t ee_num_expected_tensors(const ee_handle* h);
const char* ee_expected_tensor_name(const ee_handle* h, int32_t i);

/*
 * One pass of the hot path over a batch of B documents with T text tokens each.
 *
 *   input_ids      dev int64 (B,T)        attention_mask  dev int64 (B,T) or NULL (= ones)
 *   bbox           dev int64 (B,T,4)      pixel_values    dev float (B,C,R,R)
 *   token_type_ids dev int64 (B,T) or NULL (= zeros)      position_ids dev int64 (B,T) or NULL (= pad-aware cumsum)
 *   thresholds     host double [E+1]: exit e leaves when sign(crit_e, thresholds[e]) (strict); entry E (final) unused
 *   temperatures   host double [E+1] or NULL: logits of exit e are divided by temperatures[e] before the criterion
 *                  and in every returned logit (calibrated logits, EE/eval.py:321-323)
 * outputs (any may be NULL except out_exit):
 *   out_logits     dev float  (B,K)   logits at the exit the document left through   ("predictions")
 *   out_exit       dev int32  (B,)    index into the exit list, E = final classifier  ("exits_store")
 *   out_conf       dev float  (B,)    criterion value at that exit
 *   out_all_logits dev float  (E+1,B,K)  every evaluated exit's policy logits (ramp: exit_states[j][0]; gate:
 *                  gated_logits[j]); rows of exits a document never reached are left untouched
 *   out_all_crit   dev float  (E+1,B)    criterion of every evaluated exit on the policy logits
 *   out_head_logits dev float (E,B,Kh)   raw exit-head logits (Kh = K for ramps, 2 for gates) = exit_states[j][0]
 *   out_head_crit  dev float  (E,B)      criterion on the raw head logits = exit_states[j][1]
 *   out_hidden_cls dev float  (L+1,B,H)  CLS row entering layer 0 and leaving every layer (debug / parity)
 */
int ee_forward(ee_handle* h, const int64_t* input_ids, const int64_t* attention_mask, const int64_t* bbox,
               const float* pixel_values, const int64_t* token_type_ids, const int64_t* position_ids,
               int32_t B, int32_t T, const double* thresholds, const double* temperatures, uint32_t flags,
               float* out_logits, int32_t* out_exit, float* out_conf, float* out_all_logits, float* out_all_crit,
               float* out_head_logits, float* out_head_crit, float* out_hidden_cls, void* stream);

/*
 * The same forward as a captured launch list (hipGraph), for callers that keep the reference's small batches (eval_batch_size = 1,
 * EE/configs.py:36; the loop EE/utils.py:169-193 issues one forward per document): ~185 launches per forward become one graph launch.
 *
 * ee_graph_capture takes EXACTLY the arguments of ee_forward.  It runs the call once eagerly on `stream` (argument validation, one-time kernel
 * set-up; the outputs hold that call's results), synchronises, then captures the same launch list and instantiates it; *graph_id names it.
 * The graph is bound to the POINTERS it was captured with -- inputs and outputs are static buffers the caller refills / reads between
 * replays -- and to (B, T, flags, which outputs were non-NULL) and to the handle's exit-layer schedule at capture time (ee_set_probe_mask).
 * Thresholds and temperatures are NOT baked in: the decide kernels read them from a device vector that every ee_graph_launch refreshes.
 * `stream` must be a created stream (the legacy null stream cannot be captured).  Not capturable: the one-shot side inputs / outputs
 * (ee_set_inputs_embeds, ee_set_hidden_states_out, ee_set_head_mask, ee_set_attentions_out) and an armed ee_profile.
 *
 * ee_graph_launch replays it on `stream` (any stream, the null stream included) with this launch's thresholds (host double [E+1]; may be NULL
 * for a graph captured with MMEE_FLAG_NO_EXIT) and temperatures (host double [E+1] or NULL = 1.0: a division by 1.0 is exact, so a graph
 * replayed without temperatures returns the bits of the eager call without them).  Same arithmetic, same launch order, same bits as
 * ee_forward on the same inputs (tests/test_gpu_round6.py).  Error reporting, ee_last_stage_counts, ee_last_flops and ee_last_layer_plan work
 * as after ee_forward.  Rows of out_all_* / out_head_* that a replay does not reach keep what the buffers held before (as ee_forward).
 * ee_graph_destroy releases the executable graph (ee_destroy releases all of them).
 */
int ee_graph_capture(ee_handle* h, const int64_t* input_ids, const int64_t* attention_mask, const int64_t* bbox,
                     const float* pixel_values, const int64_t* token_type_ids, const int64_t* position_ids,
                     int32_t B, int32_t T, const double* thresholds, const double* temperatures, uint32_t flags,
                     float* out_logits, int32_t* out_exit, float* out_conf, float* out_all_logits, float* out_all_crit,
                     float* out_head_logits, float* out_head_crit, float* out_hidden_cls, void* stream, int32_t* graph_id);
int ee_graph_launch(ee_handle* h, int32_t graph_id, const double* thresholds, const double* temperatures, void* stream);
int ee_graph_destroy(ee_handle* h, int32_t graph_id);

/* Per-stage statistics of the last ee_forward (synchronises with `stream`): active documents and packed rows entering
 * each of the E+1 exit stages.  n_stages_out receives E+1. */
int ee_last_stage_counts(ee_handle* h, int32_t* docs_out, int32_t* rows_out, int32_t cap, int32_t* n_stages_out,
                         void* stream);
/* FLOPs (2*M*N*K counting) the GEMM and attention kernels of the last ee_forward executed (synchronises). */
int ee_last_flops(ee_handle* h, double* gemm_flops, double* attn_flops, void* stream);
/* How the last ee_forward ran each encoder layer (synchronises).  A layer that ends in a decision (an exit head, or the final
 * classifier) is run "probe first" in split precision: Q|K|V for every row of the stage, then the layer's output for the CLS row of
 * every document only (the one row the head reads, EE/models/LayoutLMv3.py:757-768), the decision, and the rest of the layer for
 * the documents that stay.  Per layer l < cap: rows whose Q|K|V was projected, rows the attention / attention-out / FFN ran on
 * (0: none, the last layer), documents probed (0: the layer was run whole).  probe_flops: FLOPs of all the probes, which
 * ee_last_flops leaves out. */
int ee_last_layer_plan(ee_handle* h, int32_t* rows_qkv, int32_t* rows_main, int32_t* docs_probe, int32_t cap, double* probe_flops,
                       void* stream);

/* The exit criterion of every LATER ee_forward (MMEE_CRIT_*; ee_config.criterion is its initial value).  The reference's evaluation driver
 * overrides `model.config.exit_config["inference_strategy"]` after the model has been built (EE/utils.py:62-78); the Python mirror forwards that
 * write here so that those lines run unchanged. */
int ee_set_criterion(ee_handle* h, int32_t criterion);

/* Pin the exit-layer schedule.  DEFAULT (enabled == 0): every layer that ends in a decision is probed first.  Rounds 2-4 chose per layer
 * from the stage populations of "the handle's most recent FINISHED forward" -- a timing-dependent host decision, and under MMEE_FLAG_XPROBE
 * (whose probe is a re-association) the same inputs could return different low bits from run to run.  Round 5: the library never picks a
 * schedule by itself; the same call always issues the same launches and returns the same bits (the reference's policy is deterministic,
 * EE/policy.py:28-45).  enabled != 0: bit l of `mask` says whether encoder layer l (0-based) is probed first; layers without an exit ignore
 * their bit, the last layer is always probed, and MMEE_FLAG_WHOLE_LAYERS / MMEE_FLAG_PROBE_ALWAYS still override.  The mask stays until it is
 * changed: it is part of the handle's configuration, like the thresholds are part of the call. */
int ee_set_probe_mask(ee_handle* h, int32_t enabled, uint64_t mask);
/* The priced alternative to "probe everywhere": which exit layers are worth probing first, judged by a cost model from the
 * stage populations of the LAST ee_forward on this handle, which must have been a thresholded one (synchronises).  `flags`: the flags the
 * caller is going to run with (MMEE_FLAG_XPROBE changes the probe's price).  A pure function of those populations: the caller decides whether
 * to pin the result with ee_set_probe_mask (bench.py and EarlyExitEngine.pin_schedule() do, once, after a warm-up forward). */
int ee_suggest_probe_mask(ee_handle* h, uint32_t flags, uint64_t* mask_out, void* stream);
/* Shader-clock stamps: out_dev (dev uint64[MMEE_CLOCK_STAMP_WORDS], 16-byte aligned) receives one (s_memtime = shader clocks, s_memrealtime =
 * 100 MHz) pair PER CU, slot = XCC_ID * 256 + HW_ID[15:8], as seen by one-wave workgroups enqueued on `stream`; slots no workgroup reached stay 0.
 * Two stamps around a region give the clock the chip HELD over it: mean over the slots filled in both of d(out[2s]) / d(out[2s + 1]) x 0.1 GHz
 * (the counters of different CUs are not aligned with each other: only same-slot differences mean anything).  bench.py
 * `docs_per_sec_per_ghz`: the boxes of a pool hold different clocks under the same load. */
#define MMEE_CLOCK_STAMP_WORDS 4096
int ee_clock_stamp(uint64_t* out_dev, void* stream);

/* `inputs_embeds` of the reference signature (EE/models/LayoutLMv3.py:383, 414-417 -> LayoutLMv3TextEmbeddings.forward, HF:185-186:
 * "if inputs_embeds is None: inputs_embeds = self.word_embeddings(input_ids)").  embeds: dev float (B,T,H) of the NEXT ee_forward call, read
 * in place of the word-embedding rows; consumed by that call (pass it again for the next one), NULL clears it.  ee_forward still takes
 * input_ids -- validated and used for the default, padding-aware position ids; a caller without token ids passes pad-free dummies (any valid
 * id != pad_token_id) together with the sequential position_ids of HF:148-158 (pad_token_id + 1 + t), which is what the host mirror does. */
int ee_set_inputs_embeds(ee_handle* h, const float* embeds);

/* `output_hidden_states=True` of the reference signature (EE/models/LayoutLMv3.py:386, 396-400; the encoder collects the hidden state
 * entering every layer and the last layer's output, :164, 182-183, 284-285).  out: dev float (L+1, B, T+Pv, H) -- (L+1, B, Pv, H) for the
 * image-only model -- filled by the NEXT ee_forward, which must carry MMEE_FLAG_NO_EXIT | MMEE_FLAG_WHOLE_LAYERS (nobody leaves, every layer
 * runs on every document, as in the reference's forward); consumed by that call, NULL clears it.  Positions the attention mask drops hold
 * the values the reference computes for them under MMEE_FLAG_DENSE_ROWS and zeros in the ragged layout (their rows do not exist there).
 * In MMEE_PREC_F32_SPLIT the values are the hi + lo planes the next layer actually reads (22 significant bits). */
int ee_set_hidden_states_out(ee_handle* h, float* out);
/* `head_mask` of the reference signature (EE/models/LayoutLMv3.py:382, 631-641: get_head_mask -> one factor per layer and head, applied as
 * `attention_probs = attention_probs * head_mask` in LayoutLMv3SelfAttention.forward of transformers 4.26).  mask: dev float (L, heads) of the NEXT ee_forward,
 * which must carry MMEE_FLAG_NO_EXIT | MMEE_FLAG_WHOLE_LAYERS (what `model.forward` runs); consumed by that call, NULL clears it.  LayoutLMv3 only.
 * Applied by a side kernel to the context rows behind the fused attention kernel (probs * m @ V == m * (probs @ V)): the hot path is untouched. */
int ee_set_head_mask(ee_handle* h, const float* mask);
/* `output_attentions=True` (EE/models/LayoutLMv3.py:157, 219-220, 301: one (B, heads, S, S) tensor of attention probabilities per layer, after the head
 * mask).  out: dev float (L, B, heads, S, S), S = T + patches + 1, filled by the NEXT ee_forward, which must carry MMEE_FLAG_NO_EXIT |
 * MMEE_FLAG_WHOLE_LAYERS | MMEE_FLAG_DENSE_ROWS (every position, masked keys with probability 0, as the reference computes them); S <= 1280.
 * 24 MB per document and layer at S = 709: a debugging / analysis output recomputed by a side kernel, never materialised on the hot path. */
int ee_set_attentions_out(ee_handle* h, float* out);

/*
 * The policy on a dumped logits array.  logits dev double (E1,N,K); thresholds host double [E1]
 * (global threshold: repeat it).  exits dev int32 (N,), predictions dev double (N,K), confidence dev double (N,) or
 * NULL, counts dev int32 [E1] or NULL (documents per exit).  Strict '>' on float64 max-softmax, last exit fallback.
 */
int ee_policy_scan(const double* logits, int32_t E1, int32_t N, int32_t K, const double* thresholds,
                   int32_t* exits, double* predictions, double* confidence, int32_t* counts, void* stream);

/*
 * The row of the north star's ONE all-gather, for hosts that call RCCL themselves (the Python host does the same with tensor views, dist.py):
 * per document K + 2 int32 words = [logits (K) as their float32 bit patterns | exit_layer | confidence bit pattern].  ee_pack_results builds
 * rows (dev int32 (n, K+2)) from the three ee_forward outputs; after `ncclAllGather(rows, all_rows, n * (K + 2), ncclInt32, comm, stream)`
 * (shards padded to one size, rank r's local row i = document r + i * world) ee_unpack_results splits rows back (any output may be NULL).
 * Integer words are never flushed, canonicalised or rounded on the way.
 */
int ee_pack_results(const float* logits, const int32_t* exit_layer, const float* confidence, int32_t n, int32_t K,
                    int32_t* rows, void* stream);
int ee_unpack_results(const int32_t* rows, int32_t n, int32_t K, float* logits, int32_t* exit_layer, float* confidence,
                      void* stream);

/*
 * Many threshold vectors at once over a confidence table (EE/thresh.py:184-215 / EE/large_scale.py:42-84 semantics:
 * exit(v,n) = argmax_e(conf[e,n] >= thr[v,e]) = first exit whose confidence reaches its threshold, 0 when none does;
 * accuracy = mean(correct[exit(v,n), n]), average exit = mean(exit(v,n)), EE/large_scale.py:87-96).
 * conf dev double (E1,N) (float64 like the reference's CSF table), correct dev uint8 (E1,N), thr dev double (V,E1).
 * Outputs dev: acc double (V,), mean_exit double (V,), exit_hist int32 (V,E1) or NULL.
 */
int ee_threshold_sweep(const double* conf, const uint8_t* correct, int32_t E1, int32_t N, const double* thr, int32_t V,
                       double* acc, double* mean_exit, int32_t* exit_hist, void* stream);
/* The confidence table of the sweep: conf[e,n] = max softmax (float64) of logits[e,n,:] (CSF "msp", EE/thresh.py:55-57),
 * correct[e,n] = (argmax_k logits[e,n,k] == references[n]).  logits dev double (E1,N,K); references dev int64 (N,) or NULL
 * with correct NULL. */
int ee_msp_table(const double* logits, const int64_t* references, int32_t E1, int32_t N, int32_t K, double* conf,
                 uint8_t* correct, void* stream);

/*
 * Per-exit temperature fit on the device (TemperatureScaler.set_temperature, EE/generic_scaling.py:64-111, as driven per
 * exit by calibrate(), EE/eval.py:313-337): for every exit e, T[e] = argmin_T mean NLL(softmax(logits[e] / T), labels),
 * starting from T = 1, by Newton iterations on 1/T (the objective is convex in 1/T).  logits dev double (E1,N,K),
 * labels dev int64 (N,) with values in [0,K).  Outputs dev (E1,): temperature; optional nll / accuracy /
 * avg_confidence of the scaled logits and the iteration count.  The ECE that calibrate() also records comes from a
 * remote metric (evaluate.load("jordyvl/ece"), EE/metrics.py:479-498) that is not available offline and is not built.
 */
int ee_temperature_fit(const double* logits, const int64_t* labels, int32_t E1, int32_t N, int32_t K, int32_t max_iter,
                       double* temperature, double* nll, double* accuracy, double* avg_confidence, int32_t* iterations,
                       void* stream);

/*
 * Device-side input feed (replaces the host image processor + collator in front of the model, EE/data/RVL_CDIP.py:246-262
 * and EE/utils.py:93-98, 173).
 *
 * ee_preprocess_images: B raw page images -> pixel_values (B,3,R,R) float32.  `images` dev uint8: the images packed back to
 *   back, each HWC RGB (c = 3) or HW greyscale (c = 1, replicated to 3 channels like Image.convert("RGB")); `desc` dev array
 *   of B records {int64 offset; int32 h, w, c, pad}.  Resize = Pillow Image.resize(BILINEAR) bit for bit, then HF rescale
 *   (1/255) and normalise (mean = std = 0.5).  in/out size ratio must be <= 31.  `workspace` dev scratch of
 *   ee_preprocess_workspace_bytes(B, R, max_h) bytes (max_h = largest image height).  resized_u8 (B,R,R,3) optional.
 * ee_collate_pad: ragged token streams (ids dev int64 [total], boxes dev int64 [total,4], offsets dev int64 [B+1]) ->
 *   max_length tensors (B,T): input_ids padded with pad_id, attention_mask, bbox padded with zeros; longer inputs truncated.
 */
int ee_preprocess_images(const uint8_t* images, const void* desc, int32_t B, int32_t R, int32_t max_h, void* workspace,
                         size_t workspace_bytes, float* pixel_values, uint8_t* resized_u8, void* stream);
size_t ee_preprocess_workspace_bytes(int32_t B, int32_t R, int32_t max_h);
int ee_collate_pad(const int64_t* ids, const int64_t* boxes, const int64_t* offsets, int32_t B, int32_t T, int64_t pad_id,
                   int64_t* out_ids, int64_t* out_mask, int64_t* out_bbox, void* stream);

/* Per-kernel timing of subsequent ee_forward calls with HIP events recorded on the launch stream (adds two event
 * records per launch; keep it off in timed runs).  ee_profile(h, 1) arms it and clears old records; every ee_forward
 * replaces the records.  ee_profile_read synchronises the device and returns, for kernel role idx = 0,1,..., the
 * role name as "role|hip_kernel_symbol(s)", the summed milliseconds and the number of timed launches of the last
 * ee_forward; it returns 2 when idx is past the last role. */
int ee_profile(ee_handle* h, int32_t enable);
int ee_profile_read(ee_handle* h, int32_t idx, char* name_out, int32_t name_cap, double* total_ms, int32_t* launches);

/* Micro-benchmark / unit-test hook: ONE launch of the path's GEMM kernel, Cout[M,N] = epi(A[M,K] W[N,K]^T + bias (+ resid)),
 * epi 0 = bias, 1 = GELU(erf), 2 = + residual, 3 = tanh; all pointers dev float; N % 128 == 0, K % 32 == 0.
 * wgs_per_cu sizes the persistent grid (0 = default).  row_src (dev int32 [M] or NULL) gathers the A (and residual) rows as
 * the layer after an exit stage does.  clk_probe (dev, 2 x grid uint64, or NULL): per workgroup
 * {shader cycles, 100 MHz real-time ticks} spent in the kernel, i.e. the clock the chip held (diagnostic). */
int ee_debug_gemm(const float* A, const float* W, const float* bias, const float* resid, float* Cout, int32_t M, int32_t N,
                  int32_t K, int32_t epi, int32_t wgs_per_cu, const int32_t* row_src, uint64_t* clk_probe, void* stream);

/* Unit-test / micro-benchmark hook of the split-precision GEMM (MMEE_PREC_F32_SPLIT): the f32 inputs A [rows_A, K] and
 * W [N, K] are converted to split-f16 rows with the given power-of-two scales, then `iters` launches of the kernel compute
 * Cout = epi(A[row_src ? row_src[r] : r] W^T + bias (+ resid)).  out_split != 0: Cout receives split-f16 rows (64-byte
 * groups [hi 16 f16 | lo 16 f16]) scaled by out_scale instead of f32.  ms_out (host float[2], may be NULL): [0] = average
 * milliseconds per launch, [1] = shader clock in GHz when a diagnostic bit is set in epi (bits 4..: timing diagnostics).  N % 256 == 0, K % 16 == 0. */
int ee_debug_gemm_split(const float* A, const float* W, const float* bias, const float* resid, float* Cout, int32_t M, int32_t N,
                        int32_t K, int32_t epi, int32_t out_split, float a_scale, float w_scale, float out_scale,
                        const int32_t* row_src, int32_t rows_A, int32_t iters, float* ms_out, void* stream);

/* Diagnostic of the two-heads-per-item attention kernel (attention_pair.hip): with MMEE_ATTN_STAMPS=1 in the environment the
 * launches run a build with in-kernel s_memtime stamps; this call synchronises, copies the eight phase sums (shader cycles summed over
 * waves: 0 wait for the tile's LDS-DMA, 1 DMA issue, 2 bias gathers, 3 Q K^T MFMAs, 4 / 5 softmax + P V of head A / B, 6 item prologue,
 * 7 barrier) to out8 and clears them.  Timing shares only; never part of the path. */
int ee_debug_attn_stamps(uint64_t* out8);

/* Host-only helper (no GPU needed): the relative_position_bucket LUT (HF modeling_layoutlmv3.py:392-413) over
 * delta in [-max_delta, max_delta]; out_host has 2*max_delta+1 entries, index = delta + max_delta.  Exposed so the LUT
 * the kernels use can be pinned against the HF-generated golden table. */
int ee_bucket_lut(int32_t num_buckets, int32_t max_distance, int32_t max_delta, uint8_t* out_host);

#ifdef __cplusplus
}
#endif
#endif /* MMEE_H */
