// Fused LayoutLMv3 self-attention (fp32 parity mode) on the CDNA4 matrix cores.
//
// Replaces LayoutLMv3SelfAttention.forward (HF:235-288) together with the materialised relative-position bias of
// _cal_1d_pos_emb / _cal_2d_pos_emb (HF:415-457, called at EE/models/LayoutLMv3.py:170-179) and the additive mask of
// get_extended_attention_mask (EE/models/LayoutLMv3.py:622-624):
//
//     scores = (Q/sqrt(d)) K^T + (rel_pos + rel_2d_pos)/sqrt(d) + mask ; probs = CogView-softmax(scores) ; ctx = probs V
//
// The reference writes two (B, heads, 709, 709) f32 bias tensors (48 MB per document) and re-reads them in every
// layer, plus (B, heads, S, S) scores and probs.  Here nothing S x S ever reaches HBM:
//   * the bias of a (query, key) pair is three LDS lookups into per-head VALUE tables indexed by
//     pos_k - pos_q, x0_k - x0_q, y1_k - y1_q (bucket LUT composed with the nn.Linear tables and the 1/sqrt(d)
//     scale once, at ee_finalize); the per-row (pos, x0, y1, key-valid) metadata rides along with the packed rows;
//   * flash-style online softmax.  CogView's softmax((s/32 - max(s/32))*32) is the max-shifted softmax exactly
//     (scaling by 32 is exact in binary floating point), so the running-max formulation differs only by rounding.
//   * S^T = K Q^T is computed (keys on MFMA rows, queries on lanes) so every lane owns ONE query: row max / row sum
//     are in-register reductions plus one exchange between the two lane halves; P^T is then directly the B operand
//     of O^T = V^T P^T, whose accumulator again has the query on the lane (rescale = one multiply per register).
//   * documents are ragged (pad rows are never materialised in the packed layout); a work item is
//     (document, head, 128-query tile), persistent grid-stride, sizes read from device memory.
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact f32).  Per 32-query x 32-key tile per wave: 32 MFMAs for S^T (d = 64),
// 32 for O^T (two 32-wide halves of d) = 4096 matrix-pipe cycles, against ~50 LDS lookups + ~25 VALU per score.
#include "mmee_common.h"

namespace mmee {

constexpr int QT = 128;        // queries per workgroup (4 waves x 32)
constexpr int KT = 32;         // keys per tile
constexpr int D = 64;          // head dim (base and large)
constexpr int KSTR = 68;       // padded K row stride (floats): 16 lanes of a ds_read_b128 group on distinct slots
constexpr int VSTR = 64;
constexpr float kMasked = -3.0e38f;
constexpr float kLog2e = 1.44269504088896340736f;

static __host__ __device__ inline size_t attn_lds_floats(int n1, int n2) {
    return (size_t)KT * KSTR + (size_t)KT * VSTR + (size_t)KT * 4 + (size_t)n1 + 2 * (size_t)n2 + 4;   // + work-queue slot
}
size_t attention_f32_lds_bytes(const AttnArgs& a) { return attn_lds_floats(a.n1, a.n2) * sizeof(float); }

// LDS per workgroup: one K tile, one V tile, key metadata and the head's three value tables (38 KB at T = 512), so
// three workgroups share a CU: the softmax of one wave (a long dependent LDS/VALU chain between its two MFMA bursts —
// in-kernel stamps put it at 5.5k of a wave-tile's 14k cycles) overlaps the MFMA bursts of the two other waves on its
// SIMD.  The next tile is prefetched into registers while the current one is consumed; two barriers per tile hand the
// single LDS buffer over.  (A double-buffered variant at 2 workgroups/CU and an in-wave software pipeline
// QK(t+1) || softmax(t) were both measured slower: 68 vs 74 TFLOP/s.)
__global__ __launch_bounds__(256, 3) void attention_f32_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                                  // [KT][KSTR]
    float* Vs = Ks + KT * KSTR;                        // [KT][VSTR]
    RowMeta* Ms = reinterpret_cast<RowMeta*>(Vs + KT * VSTR);   // [KT]
    float* T1 = reinterpret_cast<float*>(Ms + KT);
    float* TX = T1 + a.n1;
    float* TY = TX + a.n2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int n_docs = a.counts->n_docs;
    const int qtiles = (a.max_len + QT - 1) / QT;
    const int n_items = n_docs * a.heads * qtiles;
    int cur_head = -1;

    // Work items (document, head, 128-query tile) differ a lot in cost (ragged lengths), so they are handed out by atomic
    // work queues.  The queues are XCD-local: (document, head) pair p belongs to queue p % 8 and its query tiles are
    // consecutive in that queue, so the K/V rows of a pair (236 KB at 462 rows) are fetched into ONE XCD's L2 and re-used
    // by the pair's 4-6 query tiles instead of being pulled over the fabric by up to 6 different XCDs.  A workgroup whose
    // queue is empty steals from the next XCD's queue (placement affects speed only; every index is popped exactly once).
    int* q_slot = reinterpret_cast<int*>(TY + a.n2);
    const int n_pairs = n_docs * a.heads;
    const int my_xcd = a.item_counter ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u) : 0;   // HW_REG_XCC_ID
    int q_try = 0;
    int item = blockIdx.x;
    float amax = 0.f;
    for (;; item += gridDim.x) {
        int doc, head, qt;
        if (a.item_counter) {
            bool got = false;
            while (q_try < 8) {
                const int q = (my_xcd + q_try) & 7;
                __syncthreads();                       // everyone has read the previous slot value
                if (tid == 0) *q_slot = atomicAdd(a.item_counter + 16 * q, 1);
                __syncthreads();
                const int j = *q_slot;
                const int pl = j / qtiles;             // local pair index inside queue q
                const int pair = q + 8 * pl;
                if (pair < n_pairs) {
                    qt = j - pl * qtiles;
                    doc = pair / a.heads;
                    head = pair - doc * a.heads;
                    got = true;
                    break;
                }
                ++q_try;
            }
            if (!got) break;
        } else {
            if (item >= n_items) break;
            doc = item / (a.heads * qtiles);
            const int rem = item - doc * (a.heads * qtiles);
            head = rem / qtiles;
            qt = rem - head * qtiles;
        }
        const int off = a.doc_off[doc];
        const int len = a.doc_off[doc + 1] - off;
        const int q0 = qt * QT;
        if (q0 >= len) continue;                       // uniform over the workgroup

        __syncthreads();                               // previous item's LDS reads are done
        if (head != cur_head) {                        // per-head value tables -> LDS
            for (int i = tid; i < a.n1; i += 256) T1[i] = a.t1[(size_t)head * a.n1 + i];
            for (int i = tid; i < a.n2; i += 256) {
                TX[i] = a.tx[(size_t)head * a.n2 + i];
                TY[i] = a.ty[(size_t)head * a.n2 + i];
            }
            cur_head = head;
        }

        const int qi = q0 + wave * 32 + l31;           // this lane's query (both lane halves hold the same query)
        const bool wave_active = (q0 + wave * 32) < len;
        const int qrow = off + (qi < len ? qi : len - 1);
        // Q fragment: element c of group g is Q[q][8g + 4hh + c]
        f32x4 qf[8];
        {
            const float* qp = a.qkv + (size_t)qrow * a.ld + head * D + 4 * hh;
#pragma unroll
            for (int g = 0; g < 8; ++g) qf[g] = *reinterpret_cast<const f32x4*>(qp + 8 * g);
        }
        const RowMeta mq = a.meta[qrow];
        // RowMeta carries pos/x0/y1 pre-multiplied by 4 (byte offsets into the float tables): one integer add per lookup
        const char* t1q = reinterpret_cast<const char*>(T1 + a.c1) - mq.pos;
        const char* txq = reinterpret_cast<const char*>(TX + a.c2) - mq.x0;
        const char* tyq = reinterpret_cast<const char*>(TY + a.c2) - mq.y1;

        // staging: K tile and V tile are each 32 rows x 16 float4 = 512 float4 -> 2 per thread per operand
        const int st_row = tid >> 4;                   // 0..15 (+16)
        const int st_c4 = (tid & 15) * 4;
        f32x4 rk[2], rv[2];
        RowMeta rm;
        auto load_tile = [&](int k0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kr = k0 + st_row + 16 * i;
                if (kr < len) {
                    const float* p = a.qkv + (size_t)(off + kr) * a.ld + a.H + head * D + st_c4;
                    rk[i] = *reinterpret_cast<const f32x4*>(p);
                    rv[i] = *reinterpret_cast<const f32x4*>(p + a.H);
                } else {
                    rk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            if (tid < KT) {
                const int kr = k0 + tid;
                if (kr < len) rm = a.meta[off + kr];
                else rm = RowMeta{0, 0, 0, __float_as_int(kMasked)};
            }
        };
        auto store_tile = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<f32x4*>(Ks + (st_row + 16 * i) * KSTR + st_c4) = rk[i];
                *reinterpret_cast<f32x4*>(Vs + (st_row + 16 * i) * VSTR + st_c4) = rv[i];
            }
            if (tid < KT) Ms[tid] = rm;
        };

        float m_run = kMasked, l_run = 0.f;
        f32x16 o0, o1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }

        const int n_kt = (len + KT - 1) / KT;
        load_tile(0);
        store_tile();
        __syncthreads();
        for (int kt = 0; kt < n_kt; ++kt) {
            const bool more = kt + 1 < n_kt;
            if (more) load_tile((kt + 1) * KT);
            if (wave_active) {
                // ---- S^T tile: rows = keys (A operand from LDS), cols = queries (B operand = Q registers) -------
                f32x16 s;
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] = 0.f;
                const float* kb = Ks + l31 * KSTR + 4 * hh;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 kf = *reinterpret_cast<const f32x4*>(kb + 8 * g);
#pragma unroll
                    for (int c = 0; c < 4; ++c) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c], qf[g][c], s, 0, 0, 0);
                }
                // ---- bias, mask, online softmax.  register e <-> key (e&3) + 8*(e>>2) + 4*hh of the tile ---------
                // per score: one broadcast ds_read_b128 (key metadata), three integer adds, three table reads, four
                // float adds.  Invalid keys carry kbias = -3e38 (additive mask, EE/models/LayoutLMv3.py:622-624).
                float tmax = kMasked;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kl = (e & 3) + 8 * (e >> 2) + 4 * hh;
                    const RowMeta mk = Ms[kl];
                    const float b1 = *reinterpret_cast<const float*>(t1q + mk.pos);
                    const float bx = *reinterpret_cast<const float*>(txq + mk.x0);
                    const float by = *reinterpret_cast<const float*>(tyq + mk.y1);
                    const float bias = b1 + (bx + by);              // rel_pos + (rel_pos_x + rel_pos_y), HF:268, 455
                    const float v = (s[e] + bias) + __int_as_float(mk.flags);
                    s[e] = v;
                    tmax = fmaxf(tmax, v);
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run, tmax);
                // exp(x) = 2^(x log2 e) on v_exp_f32; x = s - m <= 0, the product is rounded once (relative error of p
                // <= 4e-8 |x log2 e|, i.e. < 1e-6 for every weight above 1e-7)
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
                float psum = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float p = __builtin_amdgcn_exp2f((s[e] - m_new) * kLog2e);
                    s[e] = p;
                    psum += p;
                }
                l_run = l_run * alpha + psum;
                m_run = m_new;
#pragma unroll
                for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
                // ---- O^T += V^T P^T : A operand = V^T (lane row = d), B operand = P^T (lane col = query) ---------
                const float* vb = Vs + (4 * hh) * VSTR + l31;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kl = (e & 3) + 8 * (e >> 2);
                    const float v0 = vb[kl * VSTR];
                    const float v1 = vb[kl * VSTR + 32];
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[e], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[e], o1, 0, 0, 0);
                }
            }
            if (more) {
                __syncthreads();                       // every wave is done reading tile kt
                store_tile();
                __syncthreads();
            }
        }

        if (wave_active) {
            const float l_tot = l_run + __shfl_xor(l_run, 32, 64);   // the two lane halves hold disjoint keys
            const float inv = 1.0f / l_tot;
            if (qi < len) {
                float* op = a.ctx + (size_t)(off + qi) * a.ldc + head * D + 4 * hh;
                char* row_split = reinterpret_cast<char*>(a.ctx) + (size_t)(off + qi) * a.ldc * 4;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {       // registers 4*q4 .. 4*q4+3 <-> d = 8*q4 + 4*hh + (0..3)
                    f32x4 w0, w1;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { w0[c] = o0[4 * q4 + c] * inv; w1[c] = o1[4 * q4 + c] * inv; }
                    if (a.ctx_split) {                 // the attention-output GEMM reads split-f16 rows
                        store_split4(row_split, head * D + 4 * hh + 8 * q4, w0, a.ctx_scale, amax);
                        store_split4(row_split, head * D + 4 * hh + 8 * q4 + 32, w1, a.ctx_scale, amax);
                    } else {
                        *reinterpret_cast<f32x4*>(op + 8 * q4) = w0;
                        *reinterpret_cast<f32x4*>(op + 8 * q4 + 32) = w1;
                    }
                }
            }
        }
    }
    if (a.ctx_split) split_flag_overflow(amax, a.err_flag);
}

void launch_attention_f32(const AttnArgs& a, int max_docs, int num_cus, hipStream_t s) {
    const size_t lds = attention_f32_lds_bytes(a);
    (void)ensure_dynamic_lds<&attention_f32_kernel>("attention_f32_kernel", 160 * 1024);
    const int qtiles = (a.max_len + QT - 1) / QT;
    long items = (long)max_docs * a.heads * qtiles;
    int grid = 3 * num_cus;
    if (items < grid) grid = (int)items;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(attention_f32_kernel, dim3(grid), dim3(256), lds, s, a);
}

}  // namespace mmee
