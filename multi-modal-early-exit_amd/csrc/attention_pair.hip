// Fused LayoutLMv3 self-attention, split-f16 operands, TWO heads per work item (precision mode MMEE_PREC_F32_SPLIT).
//
// Same function as attention_f32.hip (LayoutLMv3SelfAttention.forward HF:235-288 + the relative-position bias of HF:415-457 +
// the additive mask of EE/models/LayoutLMv3.py:622-624; nothing S x S in HBM, online softmax, one query per lane) and the same
// arithmetic of the two contractions (three v_mfma_f32_32x32x16_f16 terms per product, f32 accumulate).  What changes is
// everything around the MFMAs, which is what rocprof showed the one-head kernel to be bound by (LDS 51 % busy of which a third
// bank conflicts, VALU 40 %, matrix pipe 25 %, waves parked 47 % of their cycles):
//
//   * a work item is (document, PAIR of heads, 128 queries).  The relative-position bias depends on the (query, key) pair
//     through three deltas that are the same for every head, so the key metadata reads, the address arithmetic and the table
//     gathers are done once per pair of heads: the three value tables are stored interleaved [delta][2 heads] and one
//     ds_read_b64 returns both heads' values — half the LDS gather instructions, address VALU and metadata reads per head.
//   * the tables are indexed by the delta CLAMPED to +-max_distance (beyond it the bucket, hence the value, is constant,
//     HF:392-413): 10 KB for both heads instead of 40 KB, and every far-apart pair hits one of two addresses (a broadcast
//     instead of a bank conflict).
//   * the bias is not added to the scores: pre-multiplied by s_q*s_k it INITIALISES the accumulator of S^T = K Q^T, so the
//     matrix pipe does the add (and the de-scaling of the split planes folds into the exponent's multiplier).
//   * exp2 with the 2^10 scale of the split probabilities folded into the exponent; the running maximum is only moved (and the
//     accumulators only rescaled) when a score exceeds it by more than 2^5 — probabilities then reach 2^15 < 65504, still inside
//     the split planes; exact otherwise (the factor cancels in O / l).
//   * K / V tiles of both heads arrive by LDS-DMA (global_load_lds_dwordx4, chunk XOR applied on the source side) into a
//     double-buffered ring: the tile after the one being consumed is in flight during the whole compute phase, ONE barrier per
//     32 keys, no staging registers, no ds_write.
//   * two 4-wave workgroups per CU (77 KB of LDS each, <= 256 VGPRs), the two heads' MFMA chains and softmax VALU work of one
//     wave are independent instruction streams the scheduler can interleave.
#include <cstdlib>
#include "mmee_common.h"

#ifndef MMEE_ATTN_HP1_WGS
#define MMEE_ATTN_HP1_WGS 3      // workgroups per CU (= waves per SIMD) of the one-head form
#endif

namespace mmee {

namespace {
constexpr int QT = 128;        // queries per workgroup (4 waves x 32)
constexpr int KT = 32;         // keys per tile
constexpr int D = 64;          // head dim
constexpr int TILE_BYTES = KT * 256;          // one head's K (or V) tile: 32 rows x (64 hi + 64 lo) f16
constexpr int META_BYTES = 1024;              // 64 RowMeta slots per stage (32 used); the queue slot sits in the unused half
constexpr int R1MAX = 128, R2MAX = 256;       // largest max_rel_pos / max_rel_2d_pos the fixed LDS layout holds

// LDS layout for HP heads per work item: value tables (entries of HP floats, index = clamped delta), key metadata ring, K / V ring
template <int HP>
struct Lds {
    static constexpr int ENT = 4 * HP;                                    // bytes per table entry
    static constexpr int STAGE_BYTES = 2 * HP * TILE_BYTES;               // K_0 .. K_HP-1 | V_0 .. V_HP-1
    static constexpr int OFF_TX = 0;
    static constexpr int OFF_TY = ((2 * R2MAX + 1) * ENT + 15) & ~15;
    static constexpr int OFF_T1 = 2 * OFF_TY;
    static constexpr int OFF_META = OFF_T1 + (((2 * R1MAX + 1) * ENT + 15) & ~15);
    static constexpr int OFF_STAGE = (OFF_META + 2 * META_BYTES + 255) & ~255;
    static constexpr int BYTES = OFF_STAGE + 2 * STAGE_BYTES;
    static constexpr int WGS = HP == 1 ? MMEE_ATTN_HP1_WGS : 2;                           // workgroups per CU the layout and the register budget allow
    static_assert(WGS * BYTES <= 160 * 1024, "LDS budget");
};
constexpr int kDefaultHP = 1;
constexpr float kNegBig = -1.0e30f;
constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kPShift = 10.0f;              // probabilities carry 2^10 into the split planes
constexpr float kLazyLog2 = 5.0f;             // the running maximum lags by at most 2^5

typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ unsigned img_off(int row, int ch) {     // byte offset of 16-byte chunk ch (0..15) of a tile row
    return 256u * (unsigned)row + 16u * ((unsigned)ch ^ ((((unsigned)row & 3u) << 2) | (((unsigned)row >> 2) & 3u)));
}

__device__ __forceinline__ void dma16(unsigned voff, unsigned long long base, unsigned lds_addr) {
    unsigned keep;   // m0 is saved and restored: the compiler does not accept it in a clobber list.  s_nop 3: the scalar base may come straight
    // from v_readfirstlane (VALU write of an SGPR -> VMEM read: 5 wait states; the two s_mov count)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(base)
                 : "memory");
}

__device__ __forceinline__ unsigned long long sgpr64(unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
}

// 32-bit LDS addressing done by hand: table bases and clamp bounds fold into per-lane constants, the clamp is ONE v_med3_i32
template <typename T>
__device__ __forceinline__ T lds_load(unsigned addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) T*>((size_t)addr);
}
__device__ __forceinline__ int clamp0(int x, int hi) {      // min(max(x, 0), hi), hi wave-uniform
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct HeadState {
    f32x16 o0, o1;       // O^T accumulators: d 0..31 and 32..63 (rows) x query (lane)
    float mref;          // reference maximum of the exponent (score domain x s_q s_k)
    float l;             // running sum of the 2^10-scaled probabilities of this lane's keys
};

// DIAG build only: in-kernel stamps (s_memtime) around the phases of a key tile
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
}  // namespace

// HP   = heads per work item (2: the gathers are shared by a pair of heads, 2 workgroups per CU; 1: 4 workgroups per CU)
// MODE = 0 the path's kernel; 1 stamped diagnostic build (phase sums go to `stamps`, a buffer nothing else reads; its run time means
//        nothing, the SHARES do); 2 timing variants selected by `dbg` (wrong results).  Modes 1 and 2 are never in the path.
template <int HP, int MODE>
__global__ __launch_bounds__(256, Lds<HP>::WGS) void attention_pair_kernel(const AttnArgs a, const int r1, const int r2, const int any_masked,
                                                                          unsigned long long* __restrict__ stamps, const int dbg) {
    using L = Lds<HP>;
    constexpr bool DIAG = MODE == 1;
    constexpr int ENT = L::ENT;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int n_docs = a.counts->n_docs;
    const int qtiles = (a.max_len + QT - 1) / QT;
    const int n_pairs = n_docs * (a.heads / HP);
    const size_t row_bytes = (size_t)a.ld * 4;         // a split row of Q | K | V occupies the bytes of ld floats
    const float sc2 = a.qkv_scale * a.qkv_scale;       // score accumulators carry s_q * s_k
    const float cexp = kLog2e / sc2;                   // exponent = acc * cexp
    const float lazy = kLazyLog2 / cexp;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // The kernel declares no static LDS, so the dynamic region starts at LDS address 0 and every table / ring offset below is an
    // instruction immediate instead of a register (checked here: a toolchain that breaks the assumption is reported, nothing is gathered)
    if (lds0 != 0) {                            // would be a toolchain change; reported through err_flag (bit 32), nothing is computed
        if (threadIdx.x == 0 && a.err_flag) atomicOr(a.err_flag, 32);
        return;
    }
    // table entry of a key: index = med3(key - query + R, 0, 2 R) (one v_med3_i32: inline 0, scalar bound), byte offset = ENT * index
    const int hi1 = 2 * ENT * r1, hi2 = 2 * ENT * r2;

    int* q_slot = reinterpret_cast<int*>(smem + L::OFF_META + 512);
    const int my_xcd = a.item_counter ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u) : 0;   // HW_REG_XCC_ID
    int q_try = 0;
    int item = blockIdx.x;
    int cur_hp = -1;
    float amax = 0.f;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i, t_prev) if (DIAG) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_now = stamp_now(); __builtin_amdgcn_sched_barrier(0); ph[i] += t_now - t_prev; t_prev = t_now; }

    // ---- per-lane constants of the LDS-DMA: a 1 KiB piece = 4 tile rows x 256 B; lane -> (row 4 j + (lane >> 4), physical chunk
    // lane & 15).  The logical chunk it must fetch is phys ^ swz(row), swz = ((row & 3) << 2) | ((row >> 2) & 3) = (((lane >> 4) & 3) << 2) | (j & 3);
    // logical chunk ch = 8 plane + 2 group + half  <->  global chunk 4 group + 2 plane + half of the head's 256 contiguous bytes.
    unsigned gch16[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned ch = (unsigned)(lane & 15) ^ ((((unsigned)lane >> 4) & 3u) << 2) ^ (unsigned)j;
        gch16[j] = 16u * (4u * ((ch >> 1) & 3u) + 2u * (ch >> 3) + (ch & 1u));
    }
    const unsigned prow = (unsigned)lane >> 4;           // row of the piece this lane fills
    // a stage holds 2 HP images of 8 pieces; wave w fills pieces [pstart, pstart + PPW) of image `img`
    constexpr int PPW = 4 * HP;
    const int img = (wave * 2 * HP) >> 2, pstart = (wave * PPW) & 7;
    // transposed-read addressing of V (constant per lane): 16-lane group g = lane >> 4 serves (h = g >> 1, d block 16 (g & 1));
    // lane 4q + p of the group supplies row key0 + q, d = d0 + 4p .. 4p + 3.  With row0 = 4 (g >> 1) + q and dch0 = 2 (g & 1) + (p >> 1):
    // img_off(16 ks + row0 + 8 x, dch0 + 4 dh + 8 plane) + 8 (p & 1) = (vbase ^ (32 x + 64 dh + 128 plane)) + 2048 x + 4096 ks
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = lane >> 4;
    const unsigned vbase = img_off(4 * (tg >> 1) + tq, 2 * (tg & 1) + (tp >> 1)) + 8u * (unsigned)(tp & 1);
    // K row reads: chunk (2 st + hh) + 8 plane of row l31 = kbase ^ (32 st + 128 plane)
    const unsigned kbase = img_off(l31, hh);

    for (;; item += gridDim.x) {
        int doc, hp, qt;
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
                    hp = pair / n_docs;                // head group is the slow index: a workgroup keeps its tables across documents
                    doc = pair - hp * n_docs;
                    got = true;
                    break;
                }
                ++q_try;
            }
            if (!got) break;
        } else {
            if (item >= n_pairs * qtiles) break;
            const int pair = item / qtiles;
            qt = item - pair * qtiles;
            hp = pair / n_docs;
            doc = pair - hp * n_docs;
        }
        const int off = a.doc_off[doc];                // context rows and row metadata: current numbering
        const int qoff = a.qkv_doc_off ? a.qkv_doc_off[doc] : off;      // Q | K | V rows may still be in the previous stage's (probe-first layers)
        const int len = a.doc_off[doc + 1] - off;
        const int qlen = a.q_limit > 0 && a.q_limit < len ? a.q_limit : len;      // queries wanted (CLS probe: the first block only)
        const int q0 = qt * QT;
        if (q0 >= qlen) continue;                      // uniform over the workgroup
        unsigned long long tprev = 0;
        if (DIAG) tprev = stamp_now();

        __syncthreads();                               // previous item's LDS reads are done
        if (hp != cur_hp) {                            // value tables of the item's heads, interleaved and pre-scaled -> LDS
            float* T1 = reinterpret_cast<float*>(smem + L::OFF_T1);
            float* TX = reinterpret_cast<float*>(smem + L::OFF_TX);
            float* TY = reinterpret_cast<float*>(smem + L::OFF_TY);
            for (int i = tid; i < HP * (2 * r1 + 1); i += 256) {
                const int e = i / HP, hd = i - e * HP;
                T1[i] = a.t1[(size_t)(HP * hp + hd) * a.n1 + (a.c1 - r1 + e)] * sc2;
            }
            for (int i = tid; i < HP * (2 * r2 + 1); i += 256) {
                const int e = i / HP, hd = i - e * HP;
                TX[i] = a.tx[(size_t)(HP * hp + hd) * a.n2 + (a.c2 - r2 + e)] * sc2;
                TY[i] = a.ty[(size_t)(HP * hp + hd) * a.n2 + (a.c2 - r2 + e)] * sc2;
            }
            cur_hp = hp;
        }

        const int qi = q0 + wave * 32 + l31;           // this lane's query (both lane halves hold the same query)
        const bool wave_active = (q0 + wave * 32) < qlen;
        const int qrel = qi < len ? qi : len - 1;
        const int qrow = qoff + qrel;
        // Q fragments (B operand of S^T = K Q^T): k-step s, element j <-> d = 16 s + 8 hh + j; split group s of the head
        f16x8 qh[HP][4], ql[HP][4];
        {
            const char* qp = reinterpret_cast<const char*>(a.qkv) + (size_t)qrow * row_bytes + (size_t)(HP * hp) * 256 + 16 * hh;
#pragma unroll
            for (int hd = 0; hd < HP; ++hd)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    qh[hd][s] = *reinterpret_cast<const f16x8*>(qp + 256 * hd + 64 * s);
                    ql[hd][s] = *reinterpret_cast<const f16x8*>(qp + 256 * hd + 64 * s + 32);
                }
        }
        const RowMeta mq = a.meta[off + qrel];
        const int cq1 = ENT * (r1 - mq.pos / 4), cqx = ENT * (r2 - mq.x0 / 4), cqy = ENT * (r2 - mq.y1 / 4);

        // ---- LDS-DMA of one key tile (the item's heads' K and V + the keys' metadata) into ring slot `buf` -------------------------
        const size_t sect = (size_t)(img / HP + 1) * (size_t)a.H * 4 + (size_t)(HP * hp + img % HP) * 256;
        const char* kv_base = reinterpret_cast<const char*>(a.qkv) + (size_t)qoff * row_bytes + sect;
        const unsigned long long meta_base = sgpr64((unsigned long long)(size_t)(a.meta + off));
        // piece jj (0 .. PPW-1) of this wave's share of tile kt; jj == PPW: the metadata (wave 3)
        auto issue_piece = [&](int kt, int buf, int jj) __attribute__((always_inline)) {
            if (MODE == 2 && (dbg & 4)) return;
            const int k0 = kt * KT;
            if (jj == PPW) {
                if (wave == 3) {
                    int r = k0 + lane;
                    r = r < len ? r : len - 1;
                    if (lane < 32) dma16((unsigned)r * 16u, meta_base, lds0 + (unsigned)L::OFF_META + (unsigned)buf * META_BYTES);
                }
                return;
            }
            const int j = pstart + jj;
            const unsigned dst = lds0 + (unsigned)L::OFF_STAGE + (unsigned)buf * L::STAGE_BYTES + (unsigned)img * TILE_BYTES + 1024u * (unsigned)j;
            if (k0 + KT <= len) {                      // whole tile inside the document: the row goes into the scalar base
                const unsigned long long base = sgpr64((unsigned long long)(size_t)(kv_base + (size_t)(k0 + 4 * j) * row_bytes));
                dma16(prow * (unsigned)row_bytes + gch16[j & 3], base, dst);
            } else {                                   // last tile: rows past the document are clamped to its last row (and masked)
                const unsigned long long base = sgpr64((unsigned long long)(size_t)(kv_base + (size_t)k0 * row_bytes));
                const int lim = len - 1 - k0;
                int r = 4 * j + (int)prow;
                r = r < lim ? r : lim;
                dma16((unsigned)r * (unsigned)row_bytes + gch16[j & 3], base, dst);
            }
        };

        HeadState hs[HP];
#pragma unroll
        for (int hd = 0; hd < HP; ++hd) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { hs[hd].o0[e] = 0.f; hs[hd].o1[e] = 0.f; }
            hs[hd].mref = kNegBig;
            hs[hd].l = 0.f;
        }

        // ---- one head's softmax + P V on a finished score tile -------------------------------------------------------------------
        auto softmax_pv = [&](f32x16& s, HeadState& st, const unsigned Vs) __attribute__((always_inline)) {
            if (!(MODE == 2 && (dbg & 2))) {
                float tmax = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
                for (int e = 3; e < 15; e += 2) tmax = fmaxf(fmaxf(tmax, s[e]), s[e + 1]);      // v_max3_f32 chain
                tmax = fmaxf(tmax, s[15]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                if (__any(tmax > st.mref + lazy)) {                  // rare after the first tiles: move the reference, rescale
                    const float mnew = fmaxf(st.mref, tmax);
                    const float alpha = __builtin_amdgcn_exp2f((st.mref - mnew) * cexp);
                    st.mref = mnew;
                    st.l *= alpha;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { st.o0[e] *= alpha; st.o1[e] *= alpha; }
                }
                const float negm = kPShift - st.mref * cexp;
                float psum = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    s[e] = __builtin_amdgcn_exp2f(fmaf(s[e], cexp, negm));      // 2^10 p, p relative to the reference maximum
                    psum += s[e];
                }
                st.l += psum;
            }
            if (MODE == 2 && (dbg & 8)) { asm volatile("" :: "v"(s[0]), "v"(s[5]), "v"(s[10]), "v"(s[15])); return; }
            unsigned vb = vbase;
            asm volatile("" : "+v"(vb));
            // O^T += V^T P^T.  B operand = P^T: for k-step ks, element j of lane (query, hh) is register 8 ks + j, i.e.
            // key 16 ks + 4 hh + (j & 3) + 8 (j >> 2); the A operand takes the same key order from two transposed reads
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 ph8, pl8;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f32x2 x = f32x2{s[8 * ks + j], s[8 * ks + j + 1]};
                    const f16x2 h = __builtin_convertvector(x, f16x2);
                    const f16x2 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), f16x2);
                    ph8[j] = h[0]; ph8[j + 1] = h[1];
                    pl8[j] = l[0]; pl8[j + 1] = l[1];
                }
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    // rows 16 ks + row0 (+ 8), chunk (dch0 + 4 dh) + 8 plane: address = (vbase ^ (64 dh + 128 plane + 32 x)) + 2048 x + 4096 ks
                    auto trd = [&](unsigned xorc, unsigned addc) __attribute__((always_inline)) {
                        return __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(size_t)(Vs + (vb ^ xorc) + addc));
                    };
                    const h4 vh0 = trd(64u * dh, 4096u * ks);
                    const h4 vh1 = trd(64u * dh + 32u, 4096u * ks + 2048u);
                    const h4 vl0 = trd(64u * dh + 128u, 4096u * ks);
                    const h4 vl1 = trd(64u * dh + 128u + 32u, 4096u * ks + 2048u);
                    f16x8 vh, vl;
                    const f16x4 a0 = __builtin_bit_cast(f16x4, vh0), a1 = __builtin_bit_cast(f16x4, vh1);
                    const f16x4 b0 = __builtin_bit_cast(f16x4, vl0), b1 = __builtin_bit_cast(f16x4, vl1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { vh[j] = a0[j]; vh[4 + j] = a1[j]; vl[j] = b0[j]; vl[4 + j] = b1[j]; }
                    if (dh == 0) {
                        st.o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph8, st.o0, 0, 0, 0);
                        st.o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl8, st.o0, 0, 0, 0);
                        st.o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph8, st.o0, 0, 0, 0);
                    } else {
                        st.o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph8, st.o1, 0, 0, 0);
                        st.o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl8, st.o1, 0, 0, 0);
                        st.o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph8, st.o1, 0, 0, 0);
                    }
                }
            }
        };

        // ---- one key tile out of ring slot `buf`; the LDS-DMA of tile kt + 1 (into the other slot) is issued between the MFMAs ------
        auto compute = [&](int kt, const int buf, const bool more) __attribute__((always_inline)) {
            const unsigned sbase = (unsigned)L::OFF_STAGE + (unsigned)buf * L::STAGE_BYTES;
            const int k0 = kt * KT;
            const bool slow = any_masked || (k0 + KT > len);
            if (!wave_active) {
                if (more) {
#pragma unroll
                    for (int jj = 0; jj <= PPW; ++jj) issue_piece(kt + 1, buf ^ 1, jj);
                }
                return;
            }
            // bias of the item's heads = initial accumulators.  register e <-> key (e & 3) + 8 (e >> 2) + 4 hh of the tile
            f32x16 s[HP];
            if (MODE == 2 && (dbg & 1)) {
#pragma unroll
                for (int hd = 0; hd < HP; ++hd)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[hd][e] = 0.f;
            } else {
                const unsigned mbase = (unsigned)L::OFF_META + (unsigned)buf * META_BYTES + 64u * (unsigned)hh;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const i32x4 mk = lds_load<i32x4>(mbase + 16u * (unsigned)((e & 3) + 8 * (e >> 2)));      // {pos, x0, y1, flags}, each x 4
                    const unsigned a1 = (unsigned)clamp0(mk[0] * (ENT / 4) + cq1, hi1) + (unsigned)L::OFF_T1;
                    const unsigned ax = (unsigned)clamp0(mk[1] * (ENT / 4) + cqx, hi2) + (unsigned)L::OFF_TX;
                    const unsigned ay = (unsigned)clamp0(mk[2] * (ENT / 4) + cqy, hi2) + (unsigned)L::OFF_TY;
                    if constexpr (HP == 2) {
                        const f32x2 b1 = lds_load<f32x2>(a1), bx = lds_load<f32x2>(ax), by = lds_load<f32x2>(ay);
                        s[0][e] = b1[0] + (bx[0] + by[0]);      // rel_pos + (rel_pos_x + rel_pos_y), HF:268, 455
                        s[1][e] = b1[1] + (bx[1] + by[1]);
                    } else {
                        const float b1 = lds_load<float>(a1), bx = lds_load<float>(ax), by = lds_load<float>(ay);
                        s[0][e] = b1 + (bx + by);
                    }
                }
                if (slow) {                               // masked keys (EE/models/LayoutLMv3.py:622-624) and keys past the document
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int kl = (e & 3) + 8 * (e >> 2) + 4 * hh;
                        const int fl = lds_load<int>(mbase + 16u * (unsigned)((e & 3) + 8 * (e >> 2)) + 12u);
                        const bool dead = (fl != 0) || (k0 + kl >= len);
#pragma unroll
                        for (int hd = 0; hd < HP; ++hd) s[hd][e] = dead ? kNegBig : s[hd][e];
                    }
                }
            }
            STAMP(2, tprev)
            // S^T tiles: rows = keys (A operand from LDS), cols = queries (B operand = Q registers).  Chunk (2 st + hh) + 8 plane of row
            // l31 sits at kbase ^ (32 st + 128 plane): one register + one v_xor per read instead of eight address registers
            unsigned kb = kbase;
            asm volatile("" : "+v"(kb));                  // opaque per tile: the XORs are recomputed, not hoisted into registers
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const unsigned ohi = kb ^ (32u * st), olo = kb ^ (32u * st + 128u);
                f16x8 kh[HP], kl[HP];
#pragma unroll
                for (int hd = 0; hd < HP; ++hd) {
                    kh[hd] = lds_load<f16x8>(sbase + hd * TILE_BYTES + ohi);
                    kl[hd] = lds_load<f16x8>(sbase + hd * TILE_BYTES + olo);
                }
#pragma unroll
                for (int hd = 0; hd < HP; ++hd) s[hd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[hd], qh[hd][st], s[hd], 0, 0, 0);
                if (more) {                               // next tile's DMA pieces ride in the shadow of the MFMAs
#pragma unroll
                    for (int jj = st * HP; jj < (st + 1) * HP; ++jj) issue_piece(kt + 1, buf ^ 1, jj);
                }
#pragma unroll
                for (int hd = 0; hd < HP; ++hd) s[hd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[hd], ql[hd][st], s[hd], 0, 0, 0);
#pragma unroll
                for (int hd = 0; hd < HP; ++hd) s[hd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[hd], qh[hd][st], s[hd], 0, 0, 0);
            }
            if (more) issue_piece(kt + 1, buf ^ 1, PPW);
            STAMP(3, tprev)
#pragma unroll
            for (int hd = 0; hd < HP; ++hd) {
                softmax_pv(s[hd], hs[hd], sbase + (HP + hd) * TILE_BYTES);
                STAMP(4 + hd, tprev)
            }
        };

        const int n_kt = (len + KT - 1) / KT;
#pragma unroll
        for (int jj = 0; jj <= PPW; ++jj) issue_piece(0, 0, jj);
        STAMP(6, tprev)                                // item prologue: queue, tables, Q fragments, first DMA issue
        for (int kt = 0; kt < n_kt; kt += 2) {
            // my pieces of tile kt have landed, then the barrier: everyone's have, and everyone is done with tile kt - 1,
            // whose ring slot the DMA issued during this tile overwrites
            if (DIAG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(0, tprev) asm volatile("s_barrier" ::: "memory"); STAMP(7, tprev) }
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            compute(kt, 0, kt + 1 < n_kt);
            if (kt + 1 >= n_kt) break;
            if (DIAG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(0, tprev) asm volatile("s_barrier" ::: "memory"); STAMP(7, tprev) }
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            compute(kt + 1, 1, kt + 2 < n_kt);
        }

        if (wave_active) {
#pragma unroll
            for (int hd = 0; hd < HP; ++hd) {
                const HeadState& st = hs[hd];
                const float l_tot = st.l + __shfl_xor(st.l, 32, 64);   // the two lane halves hold disjoint keys
                const float inv = 1.0f / (l_tot * a.qkv_scale);        // the 2^10 of the probabilities is in l as well
                if (qi < len) {
                    char* row_split = reinterpret_cast<char*>(a.ctx) + (size_t)(off + qi) * a.ldc * 4;
                    const int head = HP * hp + hd;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {       // registers 4*q4 .. 4*q4+3 <-> d = 8*q4 + 4*hh + (0..3)
                        f32x4 w0, w1;
#pragma unroll
                        for (int c = 0; c < 4; ++c) { w0[c] = st.o0[4 * q4 + c] * inv; w1[c] = st.o1[4 * q4 + c] * inv; }
                        store_split4(row_split, head * D + 4 * hh + 8 * q4, w0, a.ctx_scale, amax);
                        store_split4(row_split, head * D + 4 * hh + 8 * q4 + 32, w1, a.ctx_scale, amax);
                    }
                }
            }
        }
    }
    if (a.ctx_split) split_flag_overflow(amax, a.err_flag);
    if (DIAG && stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(stamps + i, ph[i]);
    }
#undef STAMP
}

unsigned long long* g_attn_pair_stamps = nullptr;
// diagnostic: the eight phase sums of the stamped build (device pointer, or null when MMEE_ATTN_STAMPS is not set)
unsigned long long* attention_pair_stamps() { return g_attn_pair_stamps; }

bool attention_pair_supports(const AttnArgs& a, int max_rel_pos, int max_rel_2d_pos) {
    const int r1 = max_rel_pos < a.c1 ? max_rel_pos : a.c1, r2 = max_rel_2d_pos < a.c2 ? max_rel_2d_pos : a.c2;
    return r1 <= R1MAX && r2 <= R2MAX && a.ctx_split;
}

template <int HP>
static void launch_pair_hp(const AttnArgs& a, int max_docs, int num_cus, int r1, int r2, int any_masked, unsigned long long* stamps, int dbg,
                           hipStream_t s) {
    (void)ensure_dynamic_lds<&attention_pair_kernel<HP, 0>>("attention_pair_kernel", Lds<HP>::BYTES);
#ifdef MMEE_DIAG
    (void)ensure_dynamic_lds<&attention_pair_kernel<HP, 1>>("attention_pair_kernel", Lds<HP>::BYTES);
    (void)ensure_dynamic_lds<&attention_pair_kernel<HP, 2>>("attention_pair_kernel", Lds<HP>::BYTES);
#endif
    const int qtiles = (a.max_len + QT - 1) / QT;
    long items = (long)max_docs * (a.heads / HP) * qtiles;
    int grid = Lds<HP>::WGS * num_cus;
    if (items < grid) grid = (int)items;
    if (grid < 1) grid = 1;
    const size_t lds = Lds<HP>::BYTES;
#ifdef MMEE_DIAG      // stamped build and timing variants (wrong results): diagnostic library only
    if (stamps) { hipLaunchKernelGGL((attention_pair_kernel<HP, 1>), dim3(grid), dim3(256), lds, s, a, r1, r2, any_masked, stamps, 0); return; }
    if (dbg) { hipLaunchKernelGGL((attention_pair_kernel<HP, 2>), dim3(grid), dim3(256), lds, s, a, r1, r2, any_masked, (unsigned long long*)nullptr, dbg); return; }
#endif
    (void)stamps; (void)dbg;
    hipLaunchKernelGGL((attention_pair_kernel<HP, 0>), dim3(grid), dim3(256), lds, s, a, r1, r2, any_masked, (unsigned long long*)nullptr, 0);
}

// max_rel_pos / max_rel_2d_pos: the distances at which the 1D / 2D buckets saturate (HF:392-413); the tables are clamped there.
// any_masked: the batch may hold masked keys inside documents (MMEE_FLAG_DENSE_ROWS keeps pad rows); the tail of a document's
// last key tile is always masked.  MMEE_ATTN_HP=1 / 2 picks the heads per work item (A/B switch; 2 needs an even head count).
void launch_attention_pair(const AttnArgs& a, int max_docs, int num_cus, int max_rel_pos, int max_rel_2d_pos, int any_masked, hipStream_t s) {
    const int r1 = max_rel_pos < a.c1 ? max_rel_pos : a.c1, r2 = max_rel_2d_pos < a.c2 ? max_rel_2d_pos : a.c2;
    // diagnostic library only (diag_env_int reads nothing in the release library): MMEE_ATTN_STAMPS=1 stamped build, phase sums readable
    // through ee_debug_attn_stamps; MMEE_ATTN_DBG timing variants (wrong results); MMEE_ATTN_HP heads per work item
    static unsigned long long* stamps = [] {
        unsigned long long* p = nullptr;
        if (diag_env_int("MMEE_ATTN_STAMPS", 0) == 1 && hipMalloc((void**)&p, 64) == hipSuccess) (void)hipMemset(p, 0, 64);
        return p;
    }();
    static const int dbg = diag_env_int("MMEE_ATTN_DBG", 0);
    static const int hp_env = diag_env_int("MMEE_ATTN_HP", 0);
    int hp = hp_env == 1 || hp_env == 2 ? hp_env : kDefaultHP;
    if (a.heads % 2) hp = 1;
    g_attn_pair_stamps = stamps;
    if (hp == 2) launch_pair_hp<2>(a, max_docs, num_cus, r1, r2, any_masked, stamps, dbg, s);
    else launch_pair_hp<1>(a, max_docs, num_cus, r1, r2, any_masked, stamps, dbg, s);
}

}  // namespace mmee
