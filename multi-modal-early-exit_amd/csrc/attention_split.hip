// Fused LayoutLMv3 self-attention on the f16 matrix cores with split-f16 operands (precision mode MMEE_PREC_F32_SPLIT).
//
// Same function as attention_f32.hip (LayoutLMv3SelfAttention.forward HF:235-288 + the relative-position bias of
// HF:415-457 + the additive mask of EE/models/LayoutLMv3.py:622-624, nothing S x S ever in HBM, flash-style online
// softmax, one query per lane) — what changes is the arithmetic of the two contractions:
//
//   * Q, K, V arrive as split-f16 rows (hi + lo planes, 22 significant bits) straight from the QKV GEMM's epilogue, and
//     both S^T = K Q^T and O^T = V^T P^T are three v_mfma_f32_32x32x16_f16 terms (lo*hi + hi*lo + hi*hi, f32
//     accumulate): 24 MFMAs x 32 cycles per 32-query x 32-key tile instead of 64 MFMAs x 64 cycles of the f32 MFMA.
//   * the probabilities p in [0, 1] are split in registers (hi = f16(1024 p), lo = f16(1024 p - hi)); the accumulator
//     layout of S^T (lane = query, registers = keys) is already the B-operand layout of the next MFMA up to a fixed
//     permutation of the 16 keys of a k-step, and the SAME permutation is applied to the A operand by the addresses of the
//     transposed LDS reads, so P never leaves the registers.
//   * V stays row-major [key][d] in LDS (same staging path as K); the A operand V^T is gathered by ds_read_b64_tr_b16
//     (4 keys x 16 d blocks delivered column-major).  K and V tiles are [32 rows][256 B = hi 64 d | lo 64 d] images
//     with the 16-byte-chunk XOR ch ^ (((row&3)<<2) | ((row>>2)&3)), conflict-free for the ds_read_b128 row reads of K
//     and for the transposed reads of V (cdna_hip_programming.md T10, image (b)).
//   * bias / mask / softmax are unchanged f32 VALU work (three LDS table lookups per score); the score gets its
//     1/(s_q s_k) de-scaling in the same fused multiply-add that adds the bias.
#include <cstdlib>
#include "mmee_common.h"

namespace mmee {

namespace {
constexpr int QT = 128;        // queries per workgroup (4 waves x 32)
constexpr int KT = 32;         // keys per tile
constexpr int D = 64;          // head dim (base and large)
constexpr float kMasked = -3.0e38f;
constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kPScale = 1024.0f;
constexpr int TILE_BYTES = KT * 256;   // one operand tile: 32 rows x (64 hi + 64 lo) f16

typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ unsigned img_off(int row, int ch) {     // byte offset of 16-byte chunk ch (0..15) of a tile row
    return 256u * (unsigned)row + 16u * ((unsigned)ch ^ ((((unsigned)row & 3u) << 2) | (((unsigned)row >> 2) & 3u)));
}
}  // namespace

static size_t attn_split_lds_bytes(const AttnArgs& a) {
    return 2 * (size_t)TILE_BYTES + (size_t)KT * sizeof(RowMeta) + ((size_t)a.n1 + 2 * (size_t)a.n2 + 4) * sizeof(float);
}

template <int WGS>
__global__ __launch_bounds__(256, WGS) void attention_split_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Ks = reinterpret_cast<char*>(smem);          // [KT][256 B]
    char* Vs = Ks + TILE_BYTES;
    RowMeta* Ms = reinterpret_cast<RowMeta*>(Vs + TILE_BYTES);   // [KT]
    float* T1 = reinterpret_cast<float*>(Ms + KT);
    float* TX = T1 + a.n1;
    float* TY = TX + a.n2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int n_docs = a.counts->n_docs;
    const int qtiles = (a.max_len + QT - 1) / QT;
    const int n_items = n_docs * a.heads * qtiles;
    int cur_head = -1;
    const size_t row_bytes = (size_t)a.ld * 4;         // a split row of Q | K | V occupies the bytes of ld floats
    const float inv_qk = 1.0f / (a.qkv_scale * a.qkv_scale);

    int* q_slot = reinterpret_cast<int*>(TY + a.n2);
    const int n_pairs = n_docs * a.heads;
    const int my_xcd = a.item_counter ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u) : 0;   // HW_REG_XCC_ID
    int q_try = 0;
    int item = blockIdx.x;

    // transposed-read addressing of V (constant per lane): 16-lane group g = lane >> 4 serves (h = g >> 1, d block 16 (g & 1));
    // lane 4q + p of the group supplies row key0 + q, d = d0 + 4p .. 4p + 3
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = lane >> 4;

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
        // Q fragments (B operand of S^T = K Q^T): k-step s, element j <-> d = 16 s + 8 hh + j; split group s of the head
        f16x8 qh[4], ql[4];
        {
            const char* qp = reinterpret_cast<const char*>(a.qkv) + (size_t)qrow * row_bytes + (size_t)(head * 4) * 64 + 16 * hh;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qh[s] = *reinterpret_cast<const f16x8*>(qp + 64 * s);
                ql[s] = *reinterpret_cast<const f16x8*>(qp + 64 * s + 32);
            }
        }
        const RowMeta mq = a.meta[qrow];
        const char* t1q = reinterpret_cast<const char*>(T1 + a.c1) - mq.pos;
        const char* txq = reinterpret_cast<const char*>(TX + a.c2) - mq.x0;
        const char* tyq = reinterpret_cast<const char*>(TY + a.c2) - mq.y1;

        // staging: a K (or V) tile is 32 rows x 16 chunks of 16 B; thread t moves chunks (t & 7) and (t & 7) + 8 of row t >> 3.
        // Global chunk gch of the head's 256 contiguous bytes: group gch >> 2 (16 d), sub-chunk gch & 3 = {hi d0-7, hi d8-15,
        // lo d0-7, lo d8-15}  ->  LDS logical chunk 8 * plane + 2 * group + half.
        const int st_row = tid >> 3;
        const int st_g0 = tid & 7;
        uint4 rk[2], rv[2];
        RowMeta rm;
        auto load_tile = [&](int k0) {
            const int kr = k0 + st_row;
            if (kr < len) {
                const char* p = reinterpret_cast<const char*>(a.qkv) + (size_t)(off + kr) * row_bytes + (size_t)(a.H / 16 + head * 4) * 64;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    rk[i] = *reinterpret_cast<const uint4*>(p + 16 * (st_g0 + 8 * i));
                    rv[i] = *reinterpret_cast<const uint4*>(p + (size_t)(a.H / 16) * 64 + 16 * (st_g0 + 8 * i));
                }
            } else {
                rk[0] = rk[1] = rv[0] = rv[1] = uint4{0u, 0u, 0u, 0u};
            }
            if (tid < KT) {
                const int km = k0 + tid;
                if (km < len) rm = a.meta[off + km];
                else rm = RowMeta{0, 0, 0, __float_as_int(kMasked)};
            }
        };
        auto store_tile = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int gch = st_g0 + 8 * i;
                const int ch = 8 * ((gch & 3) >> 1) + 2 * (gch >> 2) + (gch & 1);
                const unsigned o = img_off(st_row, ch);
                *reinterpret_cast<uint4*>(Ks + o) = rk[i];
                *reinterpret_cast<uint4*>(Vs + o) = rv[i];
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
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const f16x8 kh = *reinterpret_cast<const f16x8*>(Ks + img_off(l31, 2 * st + hh));
                    const f16x8 kl = *reinterpret_cast<const f16x8*>(Ks + img_off(l31, 8 + 2 * st + hh));
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[st], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[st], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[st], s, 0, 0, 0);
                }
                // ---- bias, mask, online softmax.  register e <-> key (e&3) + 8*(e>>2) + 4*hh of the tile ---------
                // (two registers at a time: f32x2 arithmetic compiles to v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32)
                float tmax = kMasked;
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    f32x2 bias, mask;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int kl = ((e + t) & 3) + 8 * ((e + t) >> 2) + 4 * hh;
                        const RowMeta mk = Ms[kl];
                        const float b1 = *reinterpret_cast<const float*>(t1q + mk.pos);
                        const float bx = *reinterpret_cast<const float*>(txq + mk.x0);
                        const float by = *reinterpret_cast<const float*>(tyq + mk.y1);
                        bias[t] = b1 + (bx + by);                   // rel_pos + (rel_pos_x + rel_pos_y), HF:268, 455
                        mask[t] = __int_as_float(mk.flags);
                    }
                    const f32x2 v = __builtin_elementwise_fma(f32x2{s[e], s[e + 1]}, (f32x2)(inv_qk), bias) + mask;
                    s[e] = v[0];
                    s[e + 1] = v[1];
                    tmax = fmaxf(tmax, fmaxf(v[0], v[1]));
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
                f32x2 psum2 = f32x2{0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 x = (f32x2{s[e], s[e + 1]} - (f32x2)(m_new)) * (f32x2)(kLog2e);
                    f32x2 p;
                    p[0] = __builtin_amdgcn_exp2f(x[0]);
                    p[1] = __builtin_amdgcn_exp2f(x[1]);
                    s[e] = p[0];
                    s[e + 1] = p[1];
                    psum2 += p;
                }
                l_run = l_run * alpha + (psum2[0] + psum2[1]);
                m_run = m_new;
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 a0 = f32x2{o0[e], o0[e + 1]} * (f32x2)(alpha), a1 = f32x2{o1[e], o1[e + 1]} * (f32x2)(alpha);
                    o0[e] = a0[0]; o0[e + 1] = a0[1];
                    o1[e] = a1[0]; o1[e + 1] = a1[1];
                }
                // ---- O^T += V^T P^T.  B operand = P^T: for k-step st, element j of lane (query, hh) is register 8 st + j, i.e.
                // key 16 st + 4 hh + (j & 3) + 8 (j >> 2); the A operand takes the same key order from two transposed reads ----
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    f16x8 ph, pl;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const f32x2 x = f32x2{s[8 * st + j], s[8 * st + j + 1]} * (f32x2)(kPScale);
                        const f16x2 h = __builtin_convertvector(x, f16x2);
                        const f16x2 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), f16x2);
                        ph[j] = h[0]; ph[j + 1] = h[1];
                        pl[j] = l[0]; pl[j + 1] = l[1];
                    }
#pragma unroll
                    for (int dh = 0; dh < 2; ++dh) {
                        // block rows key0 + tq (key0 = 16 st + 4 h and + 8), d = 32 dh + 16 (tg & 1) + 4 tp .. + 3
                        const int dch = 4 * dh + 2 * (tg & 1) + (tp >> 1);          // hi chunk holding those d
                        const int ka = 16 * st + 4 * (tg >> 1) + tq;
                        const unsigned byte8 = 8u * (unsigned)(tp & 1);
                        const h4 vh0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(Vs + img_off(ka, dch) + byte8));
                        const h4 vh1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(Vs + img_off(ka + 8, dch) + byte8));
                        const h4 vl0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(Vs + img_off(ka, 8 + dch) + byte8));
                        const h4 vl1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(Vs + img_off(ka + 8, 8 + dch) + byte8));
                        f16x8 vh, vl;
                        const f16x4 a0 = __builtin_bit_cast(f16x4, vh0), a1 = __builtin_bit_cast(f16x4, vh1);
                        const f16x4 b0 = __builtin_bit_cast(f16x4, vl0), b1 = __builtin_bit_cast(f16x4, vl1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { vh[j] = a0[j]; vh[4 + j] = a1[j]; vl[j] = b0[j]; vl[4 + j] = b1[j]; }
                        if (dh == 0) {
                            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o0, 0, 0, 0);
                            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o0, 0, 0, 0);
                            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o0, 0, 0, 0);
                        } else {
                            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o1, 0, 0, 0);
                            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o1, 0, 0, 0);
                            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o1, 0, 0, 0);
                        }
                    }
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
            const float inv = 1.0f / (l_tot * kPScale * a.qkv_scale);
            if (qi < len) {
                char* row_split = reinterpret_cast<char*>(a.ctx) + (size_t)(off + qi) * a.ldc * 4;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {       // registers 4*q4 .. 4*q4+3 <-> d = 8*q4 + 4*hh + (0..3)
                    f32x4 w0, w1;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { w0[c] = o0[4 * q4 + c] * inv; w1[c] = o1[4 * q4 + c] * inv; }
                    store_split4(row_split, head * D + 4 * hh + 8 * q4, w0, a.ctx_scale, amax);
                    store_split4(row_split, head * D + 4 * hh + 8 * q4 + 32, w1, a.ctx_scale, amax);
                }
            }
        }
    }
    if (a.ctx_split) split_flag_overflow(amax, a.err_flag);
    if (a.ctx_split) split_flag_overflow(amax, a.err_flag);
}

void launch_attention_split(const AttnArgs& a, int max_docs, int num_cus, hipStream_t s) {
    // MMEE_ATTN_WGS=2: 256-VGPR build at two workgroups per CU (A/B switch); default three workgroups per CU
    static const int wgs = [] { const char* e = getenv("MMEE_ATTN_WGS"); return (e && e[0] == '2') ? 2 : 3; }();
    static bool attr_set = false;
    const size_t lds = attn_split_lds_bytes(a);
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_split_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_split_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int qtiles = (a.max_len + QT - 1) / QT;
    long items = (long)max_docs * a.heads * qtiles;
    int grid = wgs * num_cus;
    if (items < grid) grid = (int)items;
    if (grid < 1) grid = 1;
    if (wgs == 2) hipLaunchKernelGGL(attention_split_kernel<2>, dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(attention_split_kernel<3>, dim3(grid), dim3(256), lds, s, a);
}

}  // namespace mmee
