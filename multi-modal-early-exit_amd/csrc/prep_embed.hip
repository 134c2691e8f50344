// Document preparation, text / visual embedding rows, mean-pool exits' inputs and the row LayerNorm.
//
//   doc_prep / doc_scan / row_meta : the packed ("ragged") row layout.  A document contributes its text rows up to the LAST position
//       the attention mask keeps (row 0 = CLS always; a masked position before that one stays a row, masked as a key) followed by its
//       197 visual rows.  Trailing pad rows never enter the encoder:
//       they are masked as keys (EE/models/LayoutLMv3.py:622-624) and no encoder-level exit reads them
//       (:226 takes hidden[:,0,:]), so dropping them changes no observable output.  MMEE_FLAG_DENSE_ROWS keeps them.
//   embed_text   : A2 = LayoutLMv3TextEmbeddings.forward (HF:160-199, spatial concat HF:112-136) + the model-level
//                  LayerNorm of A3 (EE/models/LayoutLMv3.py:565); also the column sums for text_avg (:519-520) and
//                  text_visual_concat (:581-582), which DO include pad rows.
//   embed_visual : A1 tail = forward_image after the patch GEMM (EE/models/LayoutLMv3.py:358-373): prepend cls_token,
//                  + pos_embed, LayerNorm eps 1e-6, then the model-level LayerNorm; column sums for vision_avg (:466).
//   ln_rows      : LayerNorm of LayoutLMv3SelfOutput / LayoutLMv3Output (HF:299-303, 508-512) on packed rows.
// All of these are HBM-bound row kernels: one wave per row, 16-byte vector loads, two-pass LayerNorm in registers.
#include "mmee_common.h"
#include "mmee_kernels.h"

namespace mmee {

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void doc_prep_kernel(PrepArgs a) {
    __shared__ int s_kept[256], s_np[256];
    const int b = blockIdx.x, tid = threadIdx.x, T = a.T;
    const int ept = (T + 255) / 256;
    const int j0 = tid * ept;
    int kept_cnt = 0, np_cnt = 0, bad = 0;
    // Round 6: the kept text rows of a document are a PREFIX of its tokens -- everything up to the last position the mask keeps (row 0 = CLS
    // always).  A masked position INSIDE that prefix (a hole in the mask; the tokenizer never makes one, but the signature allows it) stays a
    // row, masked as a key exactly as under MMEE_FLAG_DENSE_ROWS, so that row j of a document is token j and the 1-D relative position of a
    // key tile is "tile base + lane constant" (attention_idx.hip, IDX16).  Trailing pad rows are dropped as before.
    int last = 0;
    for (int e = 0; e < ept; ++e) {
        const int j = j0 + e;
        if (j < T && a.attention_mask && a.attention_mask[(size_t)b * T + j] != 0) last = j;
    }
    if (!a.attention_mask) last = T - 1;
    s_kept[tid] = last;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_kept[tid] = s_kept[tid] > s_kept[tid + o] ? s_kept[tid] : s_kept[tid + o];
        __syncthreads();
    }
    const int last_kept = s_kept[0];
    __syncthreads();
    for (int e = 0; e < ept; ++e) {
        const int j = j0 + e;
        if (j < T) {
            const long long id = a.input_ids[(size_t)b * T + j];
            kept_cnt += (a.dense_rows || j <= last_kept) ? 1 : 0;
            np_cnt += (id != a.pad_id) ? 1 : 0;
            bad |= (id < 0 || id >= a.vocab) ? 1 : 0;
            if (a.token_type_ids) {
                const long long tt = a.token_type_ids[(size_t)b * T + j];
                bad |= (tt < 0 || tt >= a.type_vocab) ? 8 : 0;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const long long v = a.bbox[((size_t)b * T + j) * 4 + c];
                bad |= (v < 0 || v >= a.max_2d) ? 2 : 0;
            }
        }
    }
    s_kept[tid] = kept_cnt;
    s_np[tid] = np_cnt;
    __syncthreads();
    // exclusive scan over 256 partial sums (tiny; serial per thread over LDS would be 256 reads — do a log-step scan)
    for (int o = 1; o < 256; o <<= 1) {
        const int vk = (tid >= o) ? s_kept[tid - o] : 0;
        const int vn = (tid >= o) ? s_np[tid - o] : 0;
        __syncthreads();
        s_kept[tid] += vk;
        s_np[tid] += vn;
        __syncthreads();
    }
    int kbase = s_kept[tid] - kept_cnt, nbase = s_np[tid] - np_cnt;
    for (int e = 0; e < ept; ++e) {
        const int j = j0 + e;
        if (j < T) {
            const long long id = a.input_ids[(size_t)b * T + j];
            const bool kept = a.dense_rows || j <= last_kept;
            a.text_dst[(size_t)b * T + j] = kept ? kbase : -1;
            kbase += kept ? 1 : 0;
            int pid;
            if (a.position_ids) {
                long long p = a.position_ids[(size_t)b * T + j];
                if (p < 0 || p >= a.max_pos) { bad |= 4; p = 0; }
                pid = (int)p;
            } else {
                // create_position_ids_from_input_ids (HF:138-146): cumsum(ids != pad) * (ids != pad) + pad
                const int m = (id != a.pad_id) ? 1 : 0;
                nbase += m;
                pid = nbase * m + a.pad_id;
                if (pid >= a.max_pos) { bad |= 4; pid = a.max_pos - 1; }
            }
            a.emb_pos[(size_t)b * T + j] = pid;
        }
    }
    if (tid == 255) a.ntext[b] = s_kept[255];
    if (bad) atomicOr(a.err_flag, bad);
}

// single workgroup: exclusive scan of document lengths -> stage 0 arrays
__global__ __launch_bounds__(1024) void doc_scan_kernel(PrepArgs a) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    __shared__ unsigned long long s_sq[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    unsigned long long sq = 0;
    __syncthreads();
    for (int base = 0; base < a.B; base += 1024) {
        const int i = base + tid;
        const int len = (i < a.B) ? a.ntext[i] + a.Pv : 0;
        sq += (unsigned long long)len * (unsigned long long)len;
        const int inc = wave_incl_scan(len, lane);
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_w[w];
        const int carry = s_carry;
        if (i < a.B) {
            const int off = carry + wbase + inc - len;
            a.doc_off[i] = off;
            a.x_src[i] = off;
            a.doc_orig[i] = i;
        }
        __syncthreads();
        if (tid == 1023) s_carry = carry + wbase + inc;
        __syncthreads();
    }
    // total len^2
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if (lane == 0) s_sq[wave] = sq;
    __syncthreads();
    if (tid == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += s_sq[w];
        a.doc_off[a.B] = s_carry;
        a.counts->n_docs = a.B;
        a.counts->n_rows = s_carry;
        a.counts->sum_len_sq = t;
    }
}

__global__ __launch_bounds__(256) void row_meta_kernel(PrepArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x, T = a.T;
    const int off = a.doc_off[b];
    const int nt = a.ntext[b];
    const int hi = a.max_2d - 1;
    for (int j = tid; j < T; j += 256) {
        const int dst = a.text_dst[(size_t)b * T + j];
        if (dst >= 0) {
            RowMeta m;
            m.pos = 4 * j;
            long long x0 = a.bbox[((size_t)b * T + j) * 4 + 0], y1 = a.bbox[((size_t)b * T + j) * 4 + 3];
            m.x0 = 4 * (int)(x0 < 0 ? 0 : (x0 > hi ? hi : x0));
            m.y1 = 4 * (int)(y1 < 0 ? 0 : (y1 > hi ? hi : y1));
            const bool valid = a.attention_mask ? (a.attention_mask[(size_t)b * T + j] != 0) : true;
            m.flags = __float_as_int(valid ? 0.0f : kKeyMasked);
            a.meta[off + dst] = m;
        }
    }
    for (int v = tid; v < a.Pv; v += 256) {
        RowMeta m;
        m.pos = 4 * v;
        if (v == 0) {                      // cls_token_box = [1, 1, max_len-1, max_len-1]  (HF:594)
            m.x0 = 4 * 1;
            m.y1 = 4 * 999;
        } else {                           // create_visual_bbox (HF:575-596): trunc(1000*k / grid)
            const int p = v - 1, py = p / a.G, px = p - py * a.G;
            m.x0 = 4 * ((1000 * px) / a.G);
            m.y1 = 4 * ((1000 * (py + 1)) / a.G);
        }
        m.flags = __float_as_int(0.0f);
        a.meta[off + nt + v] = m;
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <int NV, bool FULL = false>
__device__ __forceinline__ void wave_store_row(float* __restrict__ dst, const f32x4 (&x)[NV], int H, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * lane + 256 * i;
        if (FULL || c < H) *reinterpret_cast<f32x4*>(dst + c) = x[i];
    }
}

// the same row as split-f16 planes (the A operand of the first Q|K|V projection, the residual of the first attention output)
template <int NV, bool FULL = false>
__device__ __forceinline__ void wave_store_row_split(void* dst_row, const f32x4 (&x)[NV], int H, int lane, float scale, float& amax) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * lane + 256 * i;
        if (FULL || c < H) store_split4_quad(dst_row, c, x[i], scale, amax, lane);      // split rows exist only where H % 256 == 0: whole waves take the branch
    }
}

// reduce per-wave column sums of the 4 waves through LDS and write one partial row
template <int NV>
__device__ __forceinline__ void block_write_partial(float* lds, float* __restrict__ dst, const f32x4 (&acc)[NV],
                                                    int H, int lane, int wave) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < H) *reinterpret_cast<f32x4*>(lds + wave * H + c) = acc[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) dst[c] = (lds[c] + lds[H + c]) + (lds[2 * H + c] + lds[3 * H + c]);
}

template <int NV, bool FAST, bool POOLED, bool FULL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void embed_text_kernel(EmbedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int T = a.T, H = a.H;
    const int nch = (T + 31) / 32;
    const int b = blockIdx.x / nch, ch = blockIdx.x - b * nch;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cs = a.cs, ss = a.ss, hi = a.max_2d - 1;
    f32x4 acc_t[NV], acc_c[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { acc_t[i] = f32x4{0, 0, 0, 0}; acc_c[i] = f32x4{0, 0, 0, 0}; }
    const int doff = a.doc_off[b];
    constexpr bool pooled = POOLED;                           // the pooled embedding exits (text_part / cat_part) average over every position, padding included
    float amax = 0.f;
    // Per position: ids -> table rows -> two LayerNorms -> store, and a wave walks its 8 positions in turn.  (1) Lane t fetches the ids /
    // box / position of the wave's position t, all 8 at once, and the loop reads them back with v_readlane; (2) the registers stay few (no
    // unrolling over positions, branch-free addressing of the spatial tables): many waves per SIMD.  Round 4 (PMC: 134 M VALU instructions per
    // launch = 0.5 ms of VALU issue, SQ_WAIT_ANY 73 % of the wave cycles): the software prefetch of position t + 1's rows is gone (114 -> 80
    // VGPRs, 0.93 -> 0.78 ms), the reductions are DPP and the `c < H` tests compile away when H == 256 NV.
    const int j0 = ch * 32 + wave * 8;
    int m_id = 0, m_tt = 0, m_pid = 0, m_dst = -1, m_b0 = 0, m_b1 = 0, m_b2 = 0, m_b3 = 0;
    if (lane < 8 && j0 + lane < T) {
        const size_t tok = (size_t)b * T + j0 + lane;
        // out-of-range ids are reported through err_flag by doc_prep_kernel; here they are clamped so that the table
        // lookups stay inside the tables (as bbox and position ids are)
        const long long id = a.input_ids[tok];
        m_id = (int)(id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id));
        const long long ttl = a.token_type_ids ? a.token_type_ids[tok] : 0;
        m_tt = (int)(ttl < 0 ? 0 : (ttl >= a.type_vocab ? a.type_vocab - 1 : ttl));
        m_pid = a.emb_pos[tok];
        m_dst = a.text_dst[tok];
        const long long v0 = a.bbox[tok * 4], v1 = a.bbox[tok * 4 + 1], v2 = a.bbox[tok * 4 + 2], v3 = a.bbox[tok * 4 + 3];
        m_b0 = (int)(v0 < 0 ? 0 : (v0 > hi ? hi : v0));
        m_b1 = (int)(v1 < 0 ? 0 : (v1 > hi ? hi : v1));
        m_b2 = (int)(v2 < 0 ? 0 : (v2 > hi ? hi : v2));
        m_b3 = (int)(v3 < 0 ? 0 : (v3 > hi ? hi : v3));
    }
    // spatial embedding = cat(x0, y0, x1, y1 rows of the x / y tables (cs wide each), h row, w row (ss wide each)), HF:118-134.
    // FAST (cs, ss multiples of 4): this lane's 4 columns of chunk i sit inside ONE of the six segments, fixed for the whole kernel
    const float* seg_tab[NV];
    int seg_sel[NV], seg_stride[NV];
    if constexpr (FAST) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * lane + 256 * i;
            int sel, col;
            if (c < 4 * cs) { sel = c / cs; col = c - sel * cs; }
            else if (c < 4 * cs + ss) { sel = 4; col = c - 4 * cs; }
            else { sel = 5; col = c - 4 * cs - ss; }
            seg_sel[i] = sel;
            seg_stride[i] = sel < 4 ? cs : ss;
            seg_tab[i] = (sel == 4 ? a.htab : sel == 5 ? a.wtab : (sel & 1) ? a.ytab : a.xtab) + col;
        }
    }
    auto load_row = [&](int t, f32x4 (&x)[NV]) __attribute__((always_inline)) {
        const int id = __builtin_amdgcn_readlane(m_id, t), tt = __builtin_amdgcn_readlane(m_tt, t), pid = __builtin_amdgcn_readlane(m_pid, t);
        const int b0 = __builtin_amdgcn_readlane(m_b0, t), b1 = __builtin_amdgcn_readlane(m_b1, t);
        const int b2 = __builtin_amdgcn_readlane(m_b2, t), b3 = __builtin_amdgcn_readlane(m_b3, t);
        // the token's word row, or its row of the caller's inputs_embeds (HF:185-186)
        const float* wrow = a.inputs_embeds ? a.inputs_embeds + ((size_t)b * T + j0 + t) * H : a.word + (size_t)id * H;
        int hidx = b3 - b1; hidx = hidx < 0 ? 0 : (hidx > hi ? hi : hidx);   // clip(y1 - y0, 0, 1023) HF:121
        int widx = b2 - b0; widx = widx < 0 ? 0 : (widx > hi ? hi : widx);   // clip(x1 - x0, 0, 1023) HF:122
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * lane + 256 * i;
            if (FULL || c < H) {
                f32x4 v = *reinterpret_cast<const f32x4*>(wrow + c);
                v += *reinterpret_cast<const f32x4*>(a.type + (size_t)tt * H + c);
                v += *reinterpret_cast<const f32x4*>(a.pos + (size_t)pid * H + c);
                f32x4 sp;
                if constexpr (FAST) {
                    const int sel = seg_sel[i];
                    const int idx = sel == 0 ? b0 : sel == 1 ? b1 : sel == 2 ? b2 : sel == 3 ? b3 : sel == 4 ? hidx : widx;
                    sp = *reinterpret_cast<const f32x4*>(seg_tab[i] + (size_t)idx * seg_stride[i]);
                } else {
                    const int bb[4] = {b0, b1, b2, b3};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int cc = c + e;
                        float s;
                        if (cc < 4 * cs) {
                            const int seg = cc / cs, col = cc - seg * cs;
                            s = ((seg & 1) ? a.ytab : a.xtab)[(size_t)bb[seg] * cs + col];
                        } else if (cc < 4 * cs + ss) {
                            s = a.htab[(size_t)hidx * ss + (cc - 4 * cs)];
                        } else {
                            s = a.wtab[(size_t)widx * ss + (cc - 4 * cs - ss)];
                        }
                        sp[e] = s;
                    }
                }
                x[i] = v + sp;
            } else {
                x[i] = f32x4{0, 0, 0, 0};
            }
        }
    };
    // a padded position has no packed row; it is embedded only when a pooled exit averages over it
    auto wanted = [&](int t) __attribute__((always_inline)) { return j0 + t < T && (pooled || __builtin_amdgcn_readlane(m_dst, t) >= 0); };
    // not unrolled: eight positions' worth of rows in registers is one wave per SIMD, and this kernel lives on occupancy
#pragma unroll 1
    for (int t = 0; t < 8; ++t) {
        const bool want = wanted(t);                          // wave-uniform
        f32x4 x[NV];
        if (!want) continue;
        load_row(t, x);
        const int dst = __builtin_amdgcn_readlane(m_dst, t);
        wave_layernorm<NV, FULL>(x, H, lane, a.ln1_g, a.ln1_b, a.eps1);      // embeddings.LayerNorm
        if constexpr (POOLED) {
#pragma unroll
            for (int i = 0; i < NV; ++i) acc_t[i] += x[i];
        }
        wave_layernorm<NV, FULL>(x, H, lane, a.ln2_g, a.ln2_b, a.eps2);      // layoutlmv3.LayerNorm (after the concat)
        if constexpr (POOLED) {
#pragma unroll
            for (int i = 0; i < NV; ++i) acc_c[i] += x[i];
        }
        if (dst >= 0) {
            if (a.Xs) wave_store_row_split<NV, FULL>(reinterpret_cast<char*>(a.Xs) + (size_t)(doff + dst) * H * 4, x, H, lane, a.split_scale, amax);
            else wave_store_row<NV, FULL>(a.X + (size_t)(doff + dst) * H, x, H, lane);
        }
    }
    if (a.Xs) split_flag_overflow(amax, a.err_flag);
    if constexpr (POOLED) {
        if (a.text_part) block_write_partial<NV>(lds, a.text_part + ((size_t)b * nch + ch) * H, acc_t, H, lane, wave);
        if (a.cat_part) block_write_partial<NV>(lds, a.cat_part + ((size_t)b * a.cat_chunks + ch) * H, acc_c, H, lane, wave);
    }
}

template <int NV, bool FULL = false>
__global__ __launch_bounds__(256) void embed_visual_kernel(EmbedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = a.H, Pv = a.Pv;
    const int nch = (Pv + 31) / 32;
    const int tch = (a.T + 31) / 32;
    const int b = blockIdx.x / nch, ch = blockIdx.x - b * nch;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc_v[NV], acc_c[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { acc_v[i] = f32x4{0, 0, 0, 0}; acc_c[i] = f32x4{0, 0, 0, 0}; }
    const int doff = a.doc_off[b] + a.ntext[b];
    float amax = 0.f;
    for (int t = 0; t < 8; ++t) {
        const int v = ch * 32 + wave * 8 + t;
        if (v >= Pv) break;
        f32x4 x[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * lane + 256 * i;
            if (FULL || c < H) {
                f32x4 e = (v == 0) ? *reinterpret_cast<const f32x4*>(a.cls_token + c)
                                   : *reinterpret_cast<const f32x4*>(a.vis_raw + ((size_t)b * (Pv - 1) + (v - 1)) * H + c);
                x[i] = e + *reinterpret_cast<const f32x4*>(a.pos_embed + (size_t)v * H + c);
            } else {
                x[i] = f32x4{0, 0, 0, 0};
            }
        }
        wave_layernorm<NV, FULL>(x, H, lane, a.ln1_g, a.ln1_b, a.eps1);      // layoutlmv3.norm, eps 1e-6
#pragma unroll
        for (int i = 0; i < NV; ++i) acc_v[i] += x[i];
        wave_layernorm<NV, FULL>(x, H, lane, a.ln2_g, a.ln2_b, a.eps2);
#pragma unroll
        for (int i = 0; i < NV; ++i) acc_c[i] += x[i];
        if (a.Xs) wave_store_row_split<NV, FULL>(reinterpret_cast<char*>(a.Xs) + (size_t)(doff + v) * H * 4, x, H, lane, a.split_scale, amax);
        else wave_store_row<NV, FULL>(a.X + (size_t)(doff + v) * H, x, H, lane);
    }
    if (a.Xs) split_flag_overflow(amax, a.err_flag);
    if (a.vis_part) block_write_partial<NV>(lds, a.vis_part + ((size_t)b * nch + ch) * H, acc_v, H, lane, wave);
    if (a.cat_part) block_write_partial<NV>(lds, a.cat_part + ((size_t)b * a.cat_chunks + tch + ch) * H, acc_c, H, lane, wave);
}

// The same rows without the pooled-exit partial sums (no embedding-level exit configured: nothing needs a fixed row -> workgroup assignment):
// one row per wave and iteration, grid-stride, as ln_rows_kernel.  Per row the arithmetic is the kernel's above, so are the bits.
// Round 4, B = 1024: 0.47 ms (the kernel above: 8 rows per wave in turn, 120 VGPRs) -> 0.39 ms with 4096 workgroups (0.41 with 2048; forcing
// 64 VGPRs spilled: 0.69 ms) -> 0.30 ms with the DPP reductions, the 4-instruction split and FULL (66 VGPRs, 632 -> 422 VALU instructions)
// -> 0.275 ms with 8192 workgroups (16384: 0.271).
template <int NV, bool FULL = false>
__global__ __launch_bounds__(256) void embed_visual_rows_kernel(EmbedArgs a) {
    const int H = a.H, Pv = a.Pv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_rows = a.B * Pv;
    float amax = 0.f;
#pragma unroll 1
    for (int r = blockIdx.x * 4 + wave; r < n_rows; r += gridDim.x * 4) {
        const int b = r / Pv, v = r - b * Pv;
        const int doff = a.doc_off[b] + a.ntext[b];
        f32x4 x[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * lane + 256 * i;
            if (FULL || c < H) {
                f32x4 e = (v == 0) ? *reinterpret_cast<const f32x4*>(a.cls_token + c)
                                   : *reinterpret_cast<const f32x4*>(a.vis_raw + ((size_t)b * (Pv - 1) + (v - 1)) * H + c);
                x[i] = e + *reinterpret_cast<const f32x4*>(a.pos_embed + (size_t)v * H + c);
            } else {
                x[i] = f32x4{0, 0, 0, 0};
            }
        }
        wave_layernorm<NV, FULL>(x, H, lane, a.ln1_g, a.ln1_b, a.eps1);      // layoutlmv3.norm, eps 1e-6
        wave_layernorm<NV, FULL>(x, H, lane, a.ln2_g, a.ln2_b, a.eps2);
        if (a.Xs) wave_store_row_split<NV, FULL>(reinterpret_cast<char*>(a.Xs) + (size_t)(doff + v) * H * 4, x, H, lane, a.split_scale, amax);
        else wave_store_row<NV, FULL>(a.X + (size_t)(doff + v) * H, x, H, lane);
    }
    if (a.Xs) split_flag_overflow(amax, a.err_flag);
}

// pooled[b][c] = (sum over chunks, in chunk order) / count      (x.mean(1), EE/models/LayoutLMv3.py:466, 520, 582)
__global__ __launch_bounds__(256) void pool_finish_kernel(const float* __restrict__ part, int chunks, int H, float count,
                                                          float* __restrict__ pooled) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < H; c += 256) {
        float s = 0.f;
        for (int k = 0; k < chunks; ++k) s += part[((size_t)b * chunks + k) * H + c];
        pooled[(size_t)b * H + c] = s / count;
    }
}

// row LayerNorm over the packed rows of the active stage: dst[r] = LN(src[row_src ? row_src[r] : r]) (dst may be src or
// null); dst_split, when given, receives the same row as split-f16 planes (the A operand of the next split GEMM)
// pre (CLS-probe rows under MMEE_FLAG_XPROBE only): the row is first completed from the n_parts split-K partial planes of the GEMM in
// front (src + p * part_stride, added in order p = 0, 1, ...), its bias and the residual row (split planes scaled by 1 / resid_inv).
struct LnPre {
    int n_parts;
    size_t part_stride;
    const float* bias;
    const char* resid;
    float resid_inv;
};
template <int NV, bool PRE, bool FULL = false>      // PRE is a separate instantiation: the layers' own LayerNorm launches carry none of its code
// (Round 6: the wave's NEXT row fetched while the current one is reduced and stored -- 12 more registers, full occupancy kept -- moved the kernel by -1 % (share 4.74 ->
//  4.69 % of the step) and docs/s by nothing: 7249 / 7229 against 7241 / 7234, tools/lib_ab.sh on one box.  The kernel runs at the memory system's rate; not kept.)
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                      const int* __restrict__ row_src, const int* __restrict__ n_rows_ptr, int H,
                                                      const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                      char* __restrict__ dst_split, float split_scale, int* __restrict__ err_flag, const LnPre pre) {
    const int n_rows = *n_rows_ptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float amax = 0.f;
    for (int r = blockIdx.x * 4 + wave; r < n_rows; r += gridDim.x * 4) {
        f32x4 x[NV];
        const float* p = src + (size_t)(row_src ? row_src[r] : r) * H;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * lane + 256 * i;
            x[i] = (FULL || c < H) ? *reinterpret_cast<const f32x4*>(p + c) : f32x4{0, 0, 0, 0};
        }
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = 4 * lane + 256 * i;
                if (FULL || c < H) {
                    for (int q = 1; q < pre.n_parts; ++q) x[i] += *reinterpret_cast<const f32x4*>(p + (size_t)q * pre.part_stride + c);
                    if (pre.bias) x[i] += *reinterpret_cast<const f32x4*>(pre.bias + c);
                    if (pre.resid) x[i] += load_split4(pre.resid + (size_t)r * H * 4, c, pre.resid_inv);
                }
            }
        }
        wave_layernorm<NV, FULL>(x, H, lane, g, b, eps);
        if (dst) wave_store_row<NV, FULL>(dst + (size_t)r * H, x, H, lane);
        if (dst_split) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = 4 * lane + 256 * i;
                if (FULL || c < H) store_split4_quad(dst_split + (size_t)r * H * 4, c, x[i], split_scale, amax, lane);
            }
        }
    }
    split_flag_overflow(amax, err_flag);
}

// BEiT / DiT embeddings (BeitEmbeddings.forward): X[b*Pv + v] = (v == 0 ? cls_token : patch[b][v-1]) + position_embeddings[v]
__global__ __launch_bounds__(256) void embed_beit_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                                         const float* __restrict__ pos, int B, int Pv, int H,
                                                         float* __restrict__ X) {
    const size_t total = (size_t)B * Pv * (H / 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % (H / 4));
        const size_t row = i / (H / 4);
        const int v = (int)(row % Pv);
        const size_t bidx = row / Pv;
        f32x4 e = (v == 0) ? *reinterpret_cast<const f32x4*>(cls + 4 * c4)
                           : *reinterpret_cast<const f32x4*>(patch + (bidx * (Pv - 1) + (v - 1)) * H + 4 * c4);
        if (pos) e += *reinterpret_cast<const f32x4*>(pos + (size_t)v * H + 4 * c4);
        *reinterpret_cast<f32x4*>(X + row * H + 4 * c4) = e;
    }
}

// BeitPooler with use_mean_pooling: pooled[i] = mean over rows 1..len-1 (patch tokens, CLS excluded) of active doc i
__global__ __launch_bounds__(256) void patch_mean_kernel(const float* __restrict__ X, int H, const int* __restrict__ x_phys,
                                                         const int* __restrict__ doc_off, const int* __restrict__ n_docs_ptr,
                                                         float* __restrict__ pooled) {
    const int n = *n_docs_ptr;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int len = doc_off[i + 1] - doc_off[i];
        const float* base = X + (size_t)x_phys[i] * H;
        for (int c = threadIdx.x; c < H; c += 256) {
            float s = 0.f;
            for (int r = 1; r < len; ++r) s += base[(size_t)r * H + c];
            pooled[(size_t)i * H + c] = s / (float)(len - 1);
        }
    }
}

// image-only (BEiT / DiT) stage 0: every document is Pv rows, no relative-position metadata, every key valid
__global__ __launch_bounds__(256) void prep_uniform_kernel(int B, int Pv, int* __restrict__ doc_off, int* __restrict__ x_src,
                                                           int* __restrict__ doc_orig, RowMeta* __restrict__ meta,
                                                           StageCounts* __restrict__ counts) {
    const int i0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int i = i0; i <= B; i += stride) {
        doc_off[i] = i * Pv;
        if (i < B) { x_src[i] = i * Pv; doc_orig[i] = i; }
    }
    for (int r = i0; r < B * Pv; r += stride) meta[r] = RowMeta{0, 0, 0, __float_as_int(0.0f)};
    if (i0 == 0) {
        counts->n_docs = B;
        counts->n_rows = B * Pv;
        counts->sum_len_sq = (unsigned long long)B * Pv * Pv;
    }
}

void launch_prep_uniform(int B, int Pv, int* doc_off, int* x_src, int* doc_orig, RowMeta* meta, StageCounts* counts, hipStream_t s) {
    int grid = (B * Pv + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(prep_uniform_kernel, dim3(grid), dim3(256), 0, s, B, Pv, doc_off, x_src, doc_orig, meta, counts);
}

// ---------------------------------------------------------------------------------------------------------------
void launch_prep(const PrepArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(doc_prep_kernel, dim3(a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(doc_scan_kernel, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(row_meta_kernel, dim3(a.B), dim3(256), 0, s, a);
}

void launch_embed_text(const EmbedArgs& a, hipStream_t s) {
    const int grid = a.B * ((a.T + 31) / 32);
    const size_t lds = 4 * (size_t)a.H * sizeof(float);
    const int nv = (a.H + 255) / 256;
    const bool fast = (a.cs & 3) == 0 && (a.ss & 3) == 0;
    const bool pooled = a.text_part || a.cat_part;
    const bool full = a.H == 256 * nv;      // every column chunk is whole: the kernels' `c < H` tests compile away
#define MMEE_ET(NV_, F_, P_) hipLaunchKernelGGL((embed_text_kernel<NV_, F_, P_>), dim3(grid), dim3(256), lds, s, a)
#define MMEE_ETF(NV_, P_) hipLaunchKernelGGL((embed_text_kernel<NV_, true, P_, true>), dim3(grid), dim3(256), lds, s, a)
    switch (nv) {
        case 1: MMEE_ET(1, false, true); break;      // tiny test models
        case 2: MMEE_ET(2, false, true); break;
        case 3:
            if (fast && full) { if (pooled) MMEE_ETF(3, true); else MMEE_ETF(3, false); }
            else if (fast && pooled) MMEE_ET(3, true, true); else if (fast) MMEE_ET(3, true, false);
            else if (pooled) MMEE_ET(3, false, true); else MMEE_ET(3, false, false);
            break;
        default:
            if (fast && full) { if (pooled) MMEE_ETF(4, true); else MMEE_ETF(4, false); }
            else if (fast && pooled) MMEE_ET(4, true, true); else if (fast) MMEE_ET(4, true, false);
            else if (pooled) MMEE_ET(4, false, true); else MMEE_ET(4, false, false);
            break;
    }
#undef MMEE_ET
#undef MMEE_ETF
}

void launch_embed_visual(const EmbedArgs& a, hipStream_t s) {
    const int nv = (a.H + 255) / 256;
    const bool full = a.H == 256 * nv;
    if (!a.vis_part && !a.cat_part) {          // no pooled embedding exit: rows in any order
        int g = (a.B * a.Pv + 3) / 4;
        if (g > 8192) g = 8192;
        switch (nv) {
            case 1: hipLaunchKernelGGL(embed_visual_rows_kernel<1>, dim3(g), dim3(256), 0, s, a); break;
            case 2: hipLaunchKernelGGL(embed_visual_rows_kernel<2>, dim3(g), dim3(256), 0, s, a); break;
            case 3: if (full) hipLaunchKernelGGL((embed_visual_rows_kernel<3, true>), dim3(g), dim3(256), 0, s, a);
                    else hipLaunchKernelGGL(embed_visual_rows_kernel<3>, dim3(g), dim3(256), 0, s, a);
                    break;
            default: if (full) hipLaunchKernelGGL((embed_visual_rows_kernel<4, true>), dim3(g), dim3(256), 0, s, a);
                     else hipLaunchKernelGGL(embed_visual_rows_kernel<4>, dim3(g), dim3(256), 0, s, a);
                     break;
        }
        return;
    }
    const int grid = a.B * ((a.Pv + 31) / 32);
    const size_t lds = 4 * (size_t)a.H * sizeof(float);
    switch (nv) {
        case 1: hipLaunchKernelGGL(embed_visual_kernel<1>, dim3(grid), dim3(256), lds, s, a); break;
        case 2: hipLaunchKernelGGL(embed_visual_kernel<2>, dim3(grid), dim3(256), lds, s, a); break;
        case 3: if (full) hipLaunchKernelGGL((embed_visual_kernel<3, true>), dim3(grid), dim3(256), lds, s, a);
                else hipLaunchKernelGGL(embed_visual_kernel<3>, dim3(grid), dim3(256), lds, s, a);
                break;
        default: if (full) hipLaunchKernelGGL((embed_visual_kernel<4, true>), dim3(grid), dim3(256), lds, s, a);
                 else hipLaunchKernelGGL(embed_visual_kernel<4>, dim3(grid), dim3(256), lds, s, a);
                 break;
    }
}

void launch_pool_finish(const float* part, int chunks, int H, float count, float* pooled, int B, hipStream_t s) {
    hipLaunchKernelGGL(pool_finish_kernel, dim3(B), dim3(256), 0, s, part, chunks, H, count, pooled);
}

void launch_ln_rows(const float* src, float* dst, const int* row_src, const int* n_rows_ptr, int max_rows, int H,
                    const float* g, const float* b, float eps, int num_cus, hipStream_t s, void* dst_split_v, float split_scale, int* err_flag,
                    int pre_parts, size_t pre_stride, const float* pre_bias, const void* pre_resid, float pre_resid_inv) {
    char* dst_split = reinterpret_cast<char*>(dst_split_v);
    const LnPre pre{pre_parts, pre_stride, pre_bias, reinterpret_cast<const char*>(pre_resid), pre_resid_inv};
    int grid = (max_rows + 3) / 4;
    const int cap = num_cus * 8;
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    const int nv = (H + 255) / 256;
    const bool full = H == 256 * nv;
#define MMEE_LN(NV_, PRE_, FULL_) hipLaunchKernelGGL((ln_rows_kernel<NV_, PRE_, FULL_>), dim3(grid), dim3(256), 0, s, src, dst, row_src, n_rows_ptr, H, g, b, eps, dst_split, split_scale, err_flag, pre)
    const bool has_pre = pre.n_parts > 0;
    switch (nv) {
        case 1: if (has_pre) MMEE_LN(1, true, false); else if (full) MMEE_LN(1, false, true); else MMEE_LN(1, false, false); break;
        case 2: if (has_pre) MMEE_LN(2, true, false); else MMEE_LN(2, false, false); break;
        case 3: if (has_pre) { if (full) MMEE_LN(3, true, true); else MMEE_LN(3, true, false); } else if (full) MMEE_LN(3, false, true); else MMEE_LN(3, false, false); break;
        default: if (has_pre) { if (full) MMEE_LN(4, true, true); else MMEE_LN(4, true, false); } else if (full) MMEE_LN(4, false, true); else MMEE_LN(4, false, false); break;
    }
#undef MMEE_LN
}

void launch_embed_beit(const float* patch, const float* cls, const float* pos, int B, int Pv, int H, float* X, hipStream_t s) {
    size_t total = (size_t)B * Pv * (H / 4);
    int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(embed_beit_kernel, dim3(grid), dim3(256), 0, s, patch, cls, pos, B, Pv, H, X);
}

void launch_patch_mean(const float* X, int H, const int* x_phys, const int* doc_off, const int* n_docs_ptr, float* pooled,
                       int max_docs, hipStream_t s) {
    int grid = max_docs < 4096 ? max_docs : 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(patch_mean_kernel, dim3(grid), dim3(256), 0, s, X, H, x_phys, doc_off, n_docs_ptr, pooled);
}

}  // namespace mmee
