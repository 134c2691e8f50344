// Fused LayoutLMv3 self-attention, split-f16 operands, relative-position bias from a per-document PAIR INDEX
// (precision mode MMEE_PREC_F32_SPLIT; the default attention of that mode).
//
// Same function as attention_f32.hip and the same arithmetic of the two contractions as attention_pair.hip
// (LayoutLMv3SelfAttention.forward HF:235-288 + relative-position bias HF:415-457 + additive mask EE/models/LayoutLMv3.py:622-624;
// three v_mfma_f32_32x32x16_f16 terms per product, f32 accumulate; nothing S x S of a LAYER ever in HBM).  What changes is where
// the bias comes from.  In-kernel stamps of attention_pair.hip showed the bias phase (key metadata reads -> address arithmetic ->
// three data-dependent LDS gathers per score and head) to be the largest single share of a wave's time (35 %), and every head of
// every layer redid it although the three bucket indices of a (query, key) pair
//     b1 = bucket_1d(pos_k - pos_q), bx = bucket_2d(x0_k - x0_q), by = bucket_2d(y1_k - y1_q)        (HF:392-457)
// depend on the DOCUMENT only.  So:
//
//   * pair_index_kernel runs ONCE per forward: for every document it writes one 32-bit word per (query, key) pair,
//     (4 b1) | (4 bx) << 10 | (4 by) << 20, a masked key (attention_mask == 0, or past the document in its last key tile) getting
//     b1 = bins_1d, an extra table entry that holds -1e30.  4 B x len^2 per document (~0.9 MB), shared by all heads and all
//     layers, stored in the register order of the score tile (lane-major) so a wave fetches a tile's 4 KB as four 1 KiB LDS-DMA
//     pieces.  The reference materialises 2 x 4 B x heads per pair and re-reads it in every layer (HF:415-457).
//   * the attention kernel then needs per score and head: three bit-field extracts, three lookups in the head's RAW bucket tables
//     (33 + 64 + 64 floats: at most 2-way bank conflicts, no key metadata, no clamps) and two adds.  The sum initialises the
//     accumulator of S^T = K Q^T (pre-multiplied by the split planes' scale), so the matrix pipe adds it to the scores.
//   * everything else as attention_pair.hip: exp2 with the 2^10 of the split probabilities folded into the exponent, lazy
//     rescaling (reference maximum moved only when exceeded by 2^5), K / V tiles by LDS-DMA into a double-buffered ring with ONE
//     barrier per 32 keys, the next tile's DMA issued between the MFMAs, 3 workgroups of 4 waves per CU.
#include <cstdlib>
#include <type_traits>
#include "mmee_common.h"

namespace mmee {

namespace {
constexpr int QT = 128;        // queries per workgroup (4 waves x 32)
constexpr int KT = 32;         // keys per tile
constexpr int D = 64;          // head dim
constexpr int TILE_BYTES = KT * 256;          // K (or V) tile: 32 rows x (64 hi + 64 lo) f16
constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // K | V
constexpr int RING = 3;                       // ring depth of the K / V stages
constexpr int BINS_MAX = 64;
// LDS layout: the K / V ring FIRST (slot s at s * 16 KiB: every tile base is a multiple of 8 KiB, so a lane's offset inside a tile and the
// tile base never share a bit and "base + (offset ^ c)" is "(base + offset) ^ c": one add per tile, one XOR per read), then the bucket
// tables and the queue slot
constexpr int OFF_TAB = RING * STAGE_BYTES;   // 49152
constexpr int OFF_T1 = OFF_TAB;               // (bins1 + 1) floats (last = masked-key sentinel)
constexpr int OFF_TX = OFF_TAB + 272;
constexpr int OFF_TY = OFF_TAB + 528;
constexpr int OFF_QSLOT = OFF_TAB + 784;
constexpr int LDS_BYTES = OFF_TAB + 1024;
// IDX16 (round 6): the head's 1-D bias over DELTA = pos_k - pos_q, index delta + c1, behind the bucket tables (up to 1024 entries: T <= 512)
// (its own packing: the LDS of a gfx950 workgroup is allocated in granules of 1280 bytes -- 128 per CU -- so three workgroups fit only up to 42
//  granules = 53760 bytes each; the first build used 54272 and ran TWO workgroups per CU: 11.7 ms against 9.4 ms per forward)
constexpr int OFF_TX16 = OFF_TAB;             // 64 floats
constexpr int OFF_TY16 = OFF_TAB + 256;       // 64 floats
constexpr int OFF_T1V = OFF_TAB + 512;        // T1V_MAX floats
constexpr int T1V_MAX = 1023;                 // 2 * 512 - 1
constexpr int OFF_QSLOT16 = OFF_T1V + 4 * T1V_MAX;
constexpr int LDS_BYTES16 = OFF_QSLOT16 + 4;
#ifndef MMEE_LKG
#define MMEE_LKG 8
#endif
constexpr int LKG = MMEE_LKG ? MMEE_LKG : 8;                 // scores whose three lookups are issued before the first add of the group (IDX16 fast path)
constexpr int WGS = 3;
static_assert(OFF_T1 + 4 * (BINS_MAX + 1) <= OFF_TX && OFF_TX + 4 * BINS_MAX <= OFF_TY && OFF_TY + 4 * BINS_MAX <= OFF_QSLOT, "tables");
static_assert(WGS * LDS_BYTES <= 160 * 1024 && OFF_TY + 4 * BINS_MAX < 65536, "LDS budget / 16-bit instruction offsets");
static_assert(WGS * ((LDS_BYTES16 + 1279) / 1280) <= 128 && WGS * ((LDS_BYTES + 1279) / 1280) <= 128 && OFF_T1V + 4 * 31 < 65536,
              "LDS budget in 1280-byte granules (three workgroups per CU) / 16-bit instruction offsets");
constexpr float kNegBig = -1.0e30f;
constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kPShift = 10.0f;              // probabilities carry 2^10 into the split planes
constexpr float kLazyLog2 = 5.0f;             // the running maximum lags by at most 2^5

typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned img_off(int row, int ch) {     // byte offset of 16-byte chunk ch (0..15) of a tile row
    return 256u * (unsigned)row + 16u * ((unsigned)ch ^ ((((unsigned)row & 3u) << 2) | (((unsigned)row >> 2) & 3u)));
}
__device__ __forceinline__ void dma16(unsigned voff, unsigned long long base, unsigned lds_addr) {
    unsigned keep;   // m0 is saved and restored: the compiler does not accept it in a clobber list
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(base)
                 : "memory");
}
__device__ __forceinline__ unsigned long long sgpr64(unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
}
template <typename T>
__device__ __forceinline__ T lds_load(unsigned addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) T*>((size_t)addr);
}
struct HeadState {
    f32x16 o0, o1;       // O^T accumulators: d 0..31 and 32..63 (rows) x query (lane)
    float mref;          // reference maximum of the exponent (score domain x s_q s_k)
    float l;             // running sum of the 2^10-scaled probabilities of this lane's keys
};
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Pair index of every document of the batch (stage-0 numbering = original document id), once per forward.
// One workgroup per (document, 32-query block); tile (qb, kb) of a document = 1024 words at ((qb * nb + kb) * 1024) in its slab,
// word [piece p][lane][w] <-> score register e = 4 p + w of lane (query qb*32 + (lane & 31), key kb*32 + (e & 3) + 8 (e >> 2) + 4 (lane >> 5)).
// ---------------------------------------------------------------------------------------------------------------
// The document's key metadata (16 B per row) and the two byte LUTs are staged in LDS first: the inner loop then issues no global load (the
// first version fetched 4 metadata records and 12 LUT bytes from global memory per 16-byte store and ran at 1.5 TB/s).
__global__ __launch_bounds__(256) void pair_index_kernel(const RowMeta* __restrict__ meta, const int* __restrict__ doc_off, int n_docs, int nb,
                                                         const unsigned char* __restrict__ lut1, int c1, int n1,
                                                         const unsigned char* __restrict__ lut2, int c2, int n2, int bins1,
                                                         unsigned* __restrict__ out, size_t doc_stride, int max_len) {
    extern __shared__ __attribute__((aligned(16))) char psm[];
    const int doc = blockIdx.x / nb, qb = blockIdx.x - doc * nb;
    if (doc >= n_docs) return;
    const int off = doc_off[doc], len = doc_off[doc + 1] - off;
    if (qb * 32 >= len) return;
    RowMeta* mk_s = reinterpret_cast<RowMeta*>(psm);                       // [max_len]
    unsigned char* l1 = reinterpret_cast<unsigned char*>(psm) + (size_t)max_len * sizeof(RowMeta);
    unsigned char* l2 = l1 + ((n1 + 15) & ~15);
    const int tid = threadIdx.x, lane = tid & 63, p = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    for (int i = tid; i < len; i += 256) mk_s[i] = meta[off + i];
    for (int i = tid; i < (n1 + 3) / 4; i += 256) reinterpret_cast<unsigned*>(l1)[i] = reinterpret_cast<const unsigned*>(lut1)[i];
    for (int i = tid; i < (n2 + 3) / 4; i += 256) reinterpret_cast<unsigned*>(l2)[i] = reinterpret_cast<const unsigned*>(lut2)[i];
    __syncthreads();
    const int q = qb * 32 + l31;
    const RowMeta mq = mk_s[q < len ? q : len - 1];
    const int nkb = (len + 31) / 32;
    unsigned* slab = out + (size_t)doc * doc_stride + (size_t)qb * nb * 1024;
    for (int kb = 0; kb < nkb; ++kb) {
        u32x4 w;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = 4 * p + t;
            const int k = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            unsigned v;
            if (k < len) {
                const RowMeta mk = mk_s[k];
                const unsigned b1 = mk.flags != 0 ? (unsigned)bins1 : (unsigned)l1[(mk.pos - mq.pos) / 4 + c1];
                const unsigned bx = l2[(mk.x0 - mq.x0) / 4 + c2], by = l2[(mk.y1 - mq.y1) / 4 + c2];
                v = (b1 << 2) | (bx << 12) | (by << 22);
            } else {
                v = (unsigned)bins1 << 2;            // past the document: masked
            }
            w[t] = v;
        }
        *reinterpret_cast<u32x4*>(slab + (size_t)kb * 1024 + p * 256 + lane * 4) = w;
    }
}

void launch_pair_index(const RowMeta* meta, const int* doc_off, int n_docs, int nb, const unsigned char* lut1, int c1, int n1,
                       const unsigned char* lut2, int c2, int n2, int bins1, unsigned* out, size_t doc_stride, int max_len, hipStream_t s) {
    const size_t lds = (size_t)max_len * sizeof(RowMeta) + ((n1 + 15) & ~15) + ((n2 + 15) & ~15);
    hipLaunchKernelGGL(pair_index_kernel, dim3(n_docs * nb), dim3(256), lds, s, meta, doc_off, n_docs, nb, lut1, c1, n1, lut2, c2, n2, bins1, out,
                       doc_stride, max_len);
}

#ifdef MMEE_DIAG      // the IDX16 form is an experiment of the diagnostic library (measured -1.3 %, not shipped): the release library holds none of it
// ---------------------------------------------------------------------------------------------------------------
// Round 6, the 16-bit pair index (IDX16).  The 1-D bucket of a pair depends on pos_k - pos_q alone, and since the kept text rows of a document
// are a prefix of its tokens (prep_embed.hip) row j of a document IS token j (visual row j: patch j - n_text): inside a key tile the
// positions are "tile base + key offset", so the attention kernel reads the 1-D bias from a per-head table over DELTA (1023 entries at
// T = 512) with the key offset as the INSTRUCTION's immediate -- no index bits, no extract, no bank conflict (32 consecutive queries read 32
// consecutive words).  What is left per pair is (4 bx, 4 by): two bytes, half the index traffic of the 32-bit word (the index loads were
// 16 % of the kernel's time, round 3).  Masked keys (past the document in its last tile; pad rows under MMEE_FLAG_DENSE_ROWS; holes) are a
// property of the KEY, not of the pair: one 32-bit mask per (document, key tile) + a per-document "has a masked key inside" flag.
//   tile (qb, kb) of a document = 1024 pairs x 2 B at halfword ((qb * nb + kb) * 1024); dword [piece p (0, 1)][lane][w (0..3)] holds the pairs of
//   score registers e = 8 p + 2 w (low half) and e + 1 (high half) of that lane: byte 0 = 4 bx, byte 1 = 4 by.
// The workgroups of query block 0 also write what the X-space probe reads (xprobe.hip: the buckets of query 0 against every key, 32-bit
// words of the old format, [nb][1024] per document) and the key masks.
__global__ __launch_bounds__(256) void pair_index16_kernel(const RowMeta* __restrict__ meta, const int* __restrict__ doc_off, int n_docs, int nb,
                                                           const unsigned char* __restrict__ lut1, int c1, int n1,
                                                           const unsigned char* __restrict__ lut2, int c2, int n2, int bins1,
                                                           unsigned* __restrict__ out16, size_t doc_stride16, unsigned* __restrict__ out_q0,
                                                           unsigned* __restrict__ keymask, int* __restrict__ doc_flags, int max_len) {
    extern __shared__ __attribute__((aligned(16))) char psm[];
    const int doc = blockIdx.x / nb, qb = blockIdx.x - doc * nb;
    if (doc >= n_docs) return;
    const int off = doc_off[doc], len = doc_off[doc + 1] - off;
    if (qb * 32 >= len) return;
    RowMeta* mk_s = reinterpret_cast<RowMeta*>(psm);                       // [max_len]
    unsigned char* l1 = reinterpret_cast<unsigned char*>(psm) + (size_t)max_len * sizeof(RowMeta);
    unsigned char* l2 = l1 + ((n1 + 15) & ~15);
    const int tid = threadIdx.x;
    for (int i = tid; i < len; i += 256) mk_s[i] = meta[off + i];
    if (qb == 0) for (int i = tid; i < (n1 + 3) / 4; i += 256) reinterpret_cast<unsigned*>(l1)[i] = reinterpret_cast<const unsigned*>(lut1)[i];
    for (int i = tid; i < (n2 + 3) / 4; i += 256) reinterpret_cast<unsigned*>(l2)[i] = reinterpret_cast<const unsigned*>(lut2)[i];
    __syncthreads();
    const int nkb = (len + 31) / 32;
    // thread -> (piece p, lane, half h): dwords 2 h, 2 h + 1 of the lane's four = score registers e = 8 p + 4 h + (0..3)
    const int p = tid >> 7, lane = (tid >> 1) & 63, h = tid & 1, l31 = lane & 31, hh = lane >> 5;
    const int q = qb * 32 + l31;
    const RowMeta mq = mk_s[q < len ? q : len - 1];
    unsigned* slab = out16 + (size_t)doc * doc_stride16 + (size_t)qb * nb * 512;      // 512 dwords per tile
    for (int kb = 0; kb < nkb; ++kb) {
        unsigned w[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            unsigned v = 0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int e = 8 * p + 4 * h + 2 * d + t;
                int k = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                k = k < len ? k : len - 1;                   // past the document: any value, the key is masked
                const RowMeta mk = mk_s[k];
                const unsigned bx = l2[(mk.x0 - mq.x0) / 4 + c2], by = l2[(mk.y1 - mq.y1) / 4 + c2];
                v |= ((bx << 2) | (by << 10)) << (16 * t);
            }
            w[d] = v;
        }
        *reinterpret_cast<u32x2*>(slab + (size_t)kb * 512 + p * 256 + lane * 4 + 2 * h) = u32x2{w[0], w[1]};
    }
    if (qb != 0) return;
    // ---- query block 0 only: the old-format words of its tiles (X-space probe) and the key masks ----
    {
        const int p4 = tid >> 6, ln = tid & 63, q0 = ln & 31, h0 = ln >> 5;
        const RowMeta m0 = mk_s[q0 < len ? q0 : len - 1];
        unsigned* q0slab = out_q0 + (size_t)doc * nb * 1024;
        for (int kb = 0; kb < nkb; ++kb) {
            u32x4 w;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = 4 * p4 + t;
                const int k = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h0;
                unsigned v;
                if (k < len) {
                    const RowMeta mk = mk_s[k];
                    const unsigned b1 = mk.flags != 0 ? (unsigned)bins1 : (unsigned)l1[(mk.pos - m0.pos) / 4 + c1];
                    const unsigned bx = l2[(mk.x0 - m0.x0) / 4 + c2], by = l2[(mk.y1 - m0.y1) / 4 + c2];
                    v = (b1 << 2) | (bx << 12) | (by << 22);
                } else {
                    v = (unsigned)bins1 << 2;
                }
                w[t] = v;
            }
            *reinterpret_cast<u32x4*>(q0slab + (size_t)kb * 1024 + p4 * 256 + ln * 4) = w;
        }
        // key masks: bit j of keymask[doc][kb] <-> key 32 kb + j is masked (past the document, or flagged); doc_flags: any flagged key INSIDE
        bool inside = false;
        for (int kb = tid >> 6; kb < nkb; kb += 4) {
            const int k = kb * 32 + (ln & 31);
            const bool fl = k < len && mk_s[k].flags != 0;
            const bool msk = k >= len || fl;
            const unsigned long long bal = __ballot(msk);
            inside |= __any(fl);
            if (ln == 0) keymask[(size_t)doc * nb + kb] = (unsigned)(bal & 0xffffffffull);
        }
        __shared__ int s_in;
        if (tid == 0) s_in = 0;
        __syncthreads();
        if (inside && (tid & 63) == 0) atomicOr(&s_in, 1);
        __syncthreads();
        if (tid == 0) doc_flags[doc] = s_in;
    }
}

void launch_pair_index16(const RowMeta* meta, const int* doc_off, int n_docs, int nb, const unsigned char* lut1, int c1, int n1,
                         const unsigned char* lut2, int c2, int n2, int bins1, unsigned* out16, size_t doc_stride16, unsigned* out_q0,
                         unsigned* keymask, int* doc_flags, int max_len, hipStream_t s) {
    const size_t lds = (size_t)max_len * sizeof(RowMeta) + ((n1 + 15) & ~15) + ((n2 + 15) & ~15);
    hipLaunchKernelGGL(pair_index16_kernel, dim3(n_docs * nb), dim3(256), lds, s, meta, doc_off, n_docs, nb, lut1, c1, n1, lut2, c2, n2, bins1, out16,
                       doc_stride16, out_q0, keymask, doc_flags, max_len);
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// The attention kernel.  MODE = 0 the path's kernel; 1 stamped diagnostic build (phase sums go to `stamps`, a buffer nothing else reads;
// its run time means nothing, the SHARES do); 2 timing variants selected by `dbg` (wrong results).  Modes 1 and 2 exist in the
// diagnostic library only (-DMMEE_DIAG).  BIAS: relative-position bias from the pair index (LayoutLMv3) or none (image-only).
//
// Round-3 timing variants (tools/attn_ab.sh) showed what binds this kernel: removing all 12 Q K^T MFMAs of a tile saves 3 %, removing
// the barrier nothing, while every other piece costs about what its INSTRUCTIONS cost to issue (index loads 16 %, K / V DMA issue 16 %,
// softmax 12 %, P V incl. its fragment reads and conversions 19 %) -- the matrix pipe is hidden, the waves are bound by the vector /
// scalar / LDS instructions they have to issue.  Hence the shape of the loop below:
//   * FOUR compile-time variants of a key tile (HOT: the tile two ahead is a full tile; TAIL: it is the document's partial last tile;
//     PENULT / LAST: nothing left to fetch), picked by the loop structure, so that a tile's body has no branch, no clamp and a constant
//     wait count; the old body spent ~150 scalar instructions and ~30 branches per tile on "is there a next tile, is it whole";
//   * one scalar tile pointer advanced per tile; a piece's row offset rides in the lane offset (one v_xad_u32 per piece);
//   * the ring at LDS 0 (see the layout above): V / K fragment addresses are one XOR each;
//   * the lo plane of P by v_fma_mixlo / mixhi_f16 (x - f16(x) rounded to f16 in one instruction instead of three);
//   * the index words of tile kt + 1 are fetched as soon as the lookups of tile kt have consumed the registers (a whole softmax + P V +
//     barrier + Q K^T ahead of their use); they are read-write asm operands, so input and output are ONE register by construction and
//     the loop's back edge needs no copy (a copy in front of the wait would read words that have not landed).
// Waits are hand-counted: from the Q loads to the end of an item only asm vector-memory operations are in flight (hipcc would wait
// vmcnt(0) at the first use of a compiler-visible load while an LDS-DMA is pending).
enum { V_HOT = 0, V_TAIL = 1, V_PENULT = 2, V_LAST = 3 };

// XP: experiment bits kept as template switches while they are being measured (tools/attn_ab.sh, MMEE_ATTN_XP in the diagnostic library):
//   2 = bias first: the lookups initialise the score accumulator (two adds per score instead of three, no zero init, no mid-tile wait;
//       the index words of tile kt + 1 are fetched right behind the lookups of tile kt, before the DMA of tile kt + 2),
//   16 = s_setprio 1 around the two MFMA phases of a tile (Q K^T with its K reads and DMA issues, P V with its V reads), 0 elsewhere: the
//       waves of a SIMD that are in a matrix phase issue ahead of those in lookups / softmax.  PMC of round 4: the matrix pipe is busy 40 % and
//       the VALU 48 % of the cycles, both at once only 14 %.  9.49 / 9.50 ms against 9.57 / 9.58 ms (+0.9 %, bit-identical); 32 = the reverse
//       (priority to the VALU phases): 9.72 ms.  Shipped (kXP),
//   (64, removed) = FOUR workgroups per CU on the two-slot ring: 128 VGPRs per lane, 136 bytes of scratch per lane -- 13.0 ms against 9.24 ms;
//       tools/check_attn_asm.py caught the first build (the compiler spilled a Q-fragment register while its load was in flight),
//   4 = two ring slots in use instead of three (tile kt + 1 fetched during tile kt, full wait at every tile): round 4, 256 documents x 12
//       layers, 9.37 / 9.42 ms against 9.41 / 9.42 ms with three slots, results bit-identical -- the third slot buys nothing, but nothing
//       that was measured needs its 16 KB either, so the path keeps three.  What the 16 KB was tried for: ONE lookup for rel_pos_x +
//       rel_pos_y in a [bins2 x bins2] table of pre-rounded sums living in the third slot (fetched per head by four LDS-DMA pieces per wave,
//       pair index carrying bx * bins2 + by; logits bit-identical) -- 10.63 ms against 9.40 ms: the 4096-entry gathers conflict ~3.5-way
//       where the 64-entry tables are at most 2-way, and the LDS, not the 32 VALU instructions saved, sets the pace.  Removed,
// Measured and removed (round 3, tools/attn_ab.sh on 256 documents x 12 layers, baseline 10.1 ms; bias first 9.9 ms): two-item tickets
// 10.5 ms, eight precomputed V fragment addresses 10.0 ms, software-pipelined tiles (Q K^T of tile kt + 1 paired with the exp / split of
// tile kt, P V of tile kt with the lookups of tile kt + 1, in one basic block each) 10.1 ms at 168 VGPRs + 18 spilled, K / V DMA and Q
// loads ahead of the table fill with no drain in front 9.9 ms (no change), the next queue ticket drawn at the start of an item 10.4 ms
// (a held ticket starts late: the queue balances worse), six waves per workgroup (192 queries per item, two workgroups per CU, waves 4 and 5
// issue no DMA: correct, but 13.8 ms -- a workgroup's six waves land 2, 2, 1, 1 on the four SIMDs and a second workgroup of 168-VGPR waves
// does not fit beside it, so a CU runs six waves instead of twelve).
// Round 5, the item hand-over (profiles/r05_attention_handover.txt).  Built as two template switches and measured with
// tools/attn_ab.sh on 256 documents x 12 layers against the shipped form on the same box (9.48 / 9.52 ms):
//   (1) the NEXT item's ticket drawn by thread 0 in the second-to-last key tile (a compiler-visible atomic whose first use sits right behind the
//       last tile's full wait) and published through LDS at the last tile's barrier, so an item starts without barrier / atomic / barrier:
//       bit-identical, 9.61 / 9.63 ms (-1.2 %); on 219-row documents 2.86 against 2.88 ms.  The two other workgroups of the CU already cover
//       that latency, and the decode + eight more spilled SGPRs cost more than it returns.
//   (2) on top, the next item's offsets decoded behind the last barrier and its Q fragments fetched behind the last Q K^T MFMA into the SAME
//       registers (read-write asm operands, one load site per path): 168 VGPRs + 4 spilled, and hipcc splits the live ranges of the 32
//       in-flight registers at the loop header (v_mov of words that have not landed) -- tools/check_attn_asm.py rejects the build, and the
//       GPU agrees (max |dlogit| 5.5).  There is no way to tell the register allocator that a register is in flight; 16 index registers
//       survive it, 48 do not.  LDS staging for the next item's Q needs 32 KB per workgroup where 3.3 KB are free.
// The "16.8 % of a wave's time is serial per-item work" of round 4 is a per-WAVE share: with three workgroups per CU the SIMD issues other
// waves meanwhile, so what bounds the kernel is instruction issue (VALU 48 %, matrix pipe 40 % of the cycles), not the item prologue.
// Removed; the kernel below is the round-4 form.
// Round 6 (profiles/r06_attention_conflicts_and_idx16.txt): (1) the IDX = 16 form below (16-bit pair index + 1-D bias from a delta table at immediate offsets):
// bit-identical, lookups -13 %, kernel -1.3 % because of its per-item table -- diagnostic library only.  (2) One ticket per (document, head), the
// workgroup walking its query tiles itself (ticket, table fill and head set-up paid once per (document, head)): bit-identical, 11.2 against 9.3 ms at 256
// documents and 41.6 against 36.9 ms at 1024 (16 tickets per workgroup slot, so not a balance effect: the four query tiles of a (document, head) then stream
// its K / V rows one after the other instead of side by side on four workgroups that share every L2 fill); with it the IDX = 16 form is the faster of the
// two (11.1 / 40.9 ms: its table is amortised) and both lose to the shipped form.  Removed.
constexpr int kXP = 2 | 16;
// TERMS = 1 (MMEE_FLAG_ONE_TERM, a reported low-precision mode, never a parity path): both products on the hi planes only.
// IDX = 16 (round 6): the 16-bit pair index + the 1-D bias as a DELTA table read at "tile base + immediate" (header of pair_index16_kernel):
// 32 instead of 48 bit-field extracts, 2 instead of 4 index loads per key tile, 8 instead of 16 index registers, no bank conflict in the 1-D
// lookups.  A key tile takes the FAST lookups when its 32 keys are 32 consecutive positions and none of them is masked; the tile in which a
// document's text ends and its visual rows begin, a tile with a masked key (the document's partial last tile; pad rows / holes) take the SLOW
// form (a per-score select of the table base and of -1e30): one or two of a document's ~15 tiles.
template <int MODE, bool BIAS, int XP, int TERMS = 3, int IDX = 32>
__global__ __launch_bounds__(256, WGS) void attention_idx_kernel(const AttnArgs a, unsigned long long* __restrict__ stamps, const int dbg) {
    constexpr bool DIAG = MODE == 1;
    constexpr bool I16 = BIAS && IDX == 16;
    static_assert(!I16 || (XP & 2), "IDX16 is built on the bias-first form");
    constexpr bool R2 = (XP & 4) != 0;          // experiment: two ring slots in use, tile kt + 1 fetched during tile kt
    constexpr int AHEAD = R2 ? 1 : 2;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int n_docs = a.counts->n_docs;
    const int qtiles = (a.max_len + QT - 1) / QT;
    const int n_pairs = n_docs * a.heads;
    const unsigned row_bytes = (unsigned)a.ld * 4u;    // a split row of Q | K | V occupies the bytes of ld floats
    const float sc2 = a.qkv_scale * a.qkv_scale;       // score accumulators carry s_q * s_k
    const float cexp = kLog2e / sc2;                   // exponent = acc * cexp
    const float lazy = kLazyLog2 / cexp;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // no static LDS in this kernel: the dynamic region starts at LDS address 0, so every offset below is an instruction immediate
    if (lds0 != 0) {                            // would be a toolchain change; reported through err_flag (bit 32), nothing is computed
        if (threadIdx.x == 0 && a.err_flag) atomicOr(a.err_flag, 32);
        return;
    }

    int* q_slot = reinterpret_cast<int*>(smem + ((BIAS && IDX == 16) ? OFF_QSLOT16 : OFF_QSLOT));
    const int my_xcd = a.item_counter ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u) : 0;   // HW_REG_XCC_ID
    int q_try = 0;
    int item = blockIdx.x;
    int cur_head = -1;
    float amax = 0.f;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i, t_prev) if (DIAG) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_now = stamp_now(); __builtin_amdgcn_sched_barrier(0); ph[i] += t_now - t_prev; t_prev = t_now; }

    // ---- per-lane constants of the K / V LDS-DMA: a 1 KiB piece = 4 tile rows x 256 B; lane -> (row 4 j + (lane >> 4), physical chunk
    // lane & 15).  The logical chunk it must fetch is phys ^ swz(row), swz = ((row & 3) << 2) | ((row >> 2) & 3) = (((lane >> 4) & 3) << 2) | (j & 3);
    // logical chunk ch = 8 plane + 2 group + half  <->  global chunk 4 group + 2 plane + half of the head's 256 contiguous bytes.
    // For piece j the global chunk offset is g0 ^ (16 (j & 1) + 64 ((j >> 1) & 1)) with g0 the j = 0 value, and row_bytes is a multiple of
    // 256, so the lane's whole source offset is ONE register XOR-ed with a constant per piece, plus the piece's row offset.
    const unsigned ch0 = (unsigned)(lane & 15) ^ ((((unsigned)lane >> 4) & 3u) << 2);
    const unsigned vdma0 = ((unsigned)lane >> 4) * row_bytes + 16u * (4u * ((ch0 >> 1) & 3u) + 2u * (ch0 >> 3) + (ch0 & 1u));
    const int img = wave >> 1, pstart = (wave & 1) * 4;  // waves 0, 1 fill K (pieces 0-3, 4-7), waves 2, 3 fill V
    const unsigned dst_wave = (unsigned)img * TILE_BYTES + 1024u * (unsigned)pstart;      // this wave's four pieces inside a ring slot
    // transposed-read addressing of V (constant per lane): 16-lane group g = lane >> 4 serves (h = g >> 1, d block 16 (g & 1));
    // lane 4q + p of the group supplies row key0 + q, d = d0 + 4p .. 4p + 3.  With row0 = 4 (g >> 1) + q and dch0 = 2 (g & 1) + (p >> 1):
    // img_off(16 ks + row0 + 8 x, dch0 + 4 dh + 8 plane) + 8 (p & 1) = (vbase ^ (32 x + 64 dh + 128 plane)) + 2048 x + 4096 ks
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = lane >> 4;
    const unsigned vbase = img_off(4 * (tg >> 1) + tq, 2 * (tg & 1) + (tp >> 1)) + 8u * (unsigned)(tp & 1) + (unsigned)TILE_BYTES;      // V image of slot 0
    // K row reads: chunk (2 st + hh) + 8 plane of row l31 = kbase ^ (32 st + 128 plane)
    const unsigned kbase = img_off(l31, hh);
    const unsigned ivoff = 16u * (unsigned)lane;

    for (;; item += gridDim.x) {
        unsigned long long t_item = 0;
        if (DIAG) t_item = stamp_now();
        int doc, head, qt;
        if (a.item_counter) {
            bool got = false;
            const int per_doc = a.heads * qtiles;
            while (q_try < 8) {
                const int q = (my_xcd + q_try) & 7;
                __syncthreads();                       // everyone has read the previous slot value
                if (tid == 0) *q_slot = atomicAdd(a.item_counter + 16 * q, 1);
                __syncthreads();
                const int j = __builtin_amdgcn_readfirstlane(*q_slot);     // wave-uniform by construction: keep doc / head / tile scalar
                // queue q serves the documents q, q + 8, ...; within a document the order is head-major, query tile fastest: the 12-16
                // heads of a document follow each other on ONE XCD, so its pair index (shared by all heads) is fetched into that L2 once
                const int dl = j / per_doc;
                const int r = j - dl * per_doc;
                const int dq = q + 8 * dl;
                if (dq < n_docs) {
                    doc = dq;
                    head = r / qtiles;
                    qt = r - head * qtiles;
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
            doc = pair / a.heads;
            head = pair - doc * a.heads;
        }
        const int off = a.doc_off[doc];                // context rows are written in the current numbering,
        const int qoff = a.qkv_doc_off ? a.qkv_doc_off[doc] : off;      // Q | K | V rows may still be in the previous stage's (probe-first layers)
        const int len = a.doc_off[doc + 1] - off;
        const int qlen = a.q_limit > 0 && a.q_limit < len ? a.q_limit : len;      // queries wanted (CLS probe: the first block only)
        const int q0 = qt * QT;
        if (q0 >= qlen) continue;                      // uniform over the workgroup
        unsigned long long tprev = 0;
        if (DIAG) { tprev = stamp_now(); ph[1] += tprev - t_item; }      // queue ticket + document offsets

        __syncthreads();                               // previous item's LDS reads are done
        auto fill_tables = [&]() __attribute__((always_inline)) {
            if (BIAS && head != cur_head) {            // the head's raw bucket tables, pre-scaled; entry bins1 of T1 = masked key
                float* T1 = reinterpret_cast<float*>(smem + OFF_T1);
                float* TX = reinterpret_cast<float*>(smem + (I16 ? OFF_TX16 : OFF_TX));
                float* TY = reinterpret_cast<float*>(smem + (I16 ? OFF_TY16 : OFF_TY));
                const float f = a.inv_sqrt_d * sc2;
                if (I16) {                             // T1V[delta + c1] = w1[head][bucket_1d(delta)] * f: the same values the bucket table holds
                    float* T1V = reinterpret_cast<float*>(smem + OFF_T1V);
                    // a.t1 = w1[head][bucket_1d(delta)] / sqrt(d) (ee_finalize's value tables); x s_q s_k, a power of two: the bits of
                    // w1[.] * (1 / sqrt(d) * s_q s_k) that the bucket table of the 32-bit form holds, without the LUT gather per head change
                    for (int i = tid; i < a.n1; i += 256) T1V[i] = a.t1[(size_t)head * a.n1 + i] * sc2;
                } else {
                    if (tid < a.bins1) T1[tid] = a.w1[(size_t)head * a.bins1 + tid] * f;
                    if (tid == a.bins1) T1[tid] = kNegBig;
                }
                if (tid >= 64 && tid < 64 + a.bins2) TX[tid - 64] = a.wx[(size_t)head * a.bins2 + tid - 64] * f;
                if (tid >= 128 && tid < 128 + a.bins2) TY[tid - 128] = a.wy[(size_t)head * a.bins2 + tid - 128] * f;
                cur_head = head;
            }
        };
        fill_tables();

        const int slab = BIAS ? __builtin_amdgcn_readfirstlane(a.doc_orig[doc]) : 0;      // fetched before the counted regime starts
        const int qb = (q0 >> 5) + wave;               // this wave's 32-query block of the document
        const int qi = q0 + wave * 32 + l31;           // this lane's query (both lane halves hold the same query)
        const bool wave_active = (q0 + wave * 32) < qlen;
        const int qrow = qoff + (qi < len ? qi : len - 1);
        // Q fragments (B operand of S^T = K Q^T): k-step s, element j <-> d = 16 s + 8 hh + j; split group s of the head
        f16x8 qh[4], ql[4];
        auto load_q = [&]() __attribute__((always_inline)) {
            const char* qp = reinterpret_cast<const char*>(a.qkv) + (size_t)qrow * row_bytes + (size_t)head * 256 + 16 * hh;
            // asm loads: from here to the end of the item only hand-counted vector memory operations are issued.  Operations that are still
            // in flight from before (the previous item's context stores, the table loads, the queue atomic) are OLDER and the counter
            // retires in issue order, so every counted wait below still covers what it names (it can only wait a little longer).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // start clean (not required for correctness: older operations retire first)
            asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:32\n\t"
                         "global_load_dwordx4 %2, %8, off offset:64\n\tglobal_load_dwordx4 %3, %8, off offset:96\n\t"
                         "global_load_dwordx4 %4, %8, off offset:128\n\tglobal_load_dwordx4 %5, %8, off offset:160\n\t"
                         "global_load_dwordx4 %6, %8, off offset:192\n\tglobal_load_dwordx4 %7, %8, off offset:224"
                         : "=&v"(qh[0]), "=&v"(ql[0]), "=&v"(qh[1]), "=&v"(ql[1]), "=&v"(qh[2]), "=&v"(ql[2]), "=&v"(qh[3]), "=&v"(ql[3])
                         : "v"(qp)
                         : "memory");
        };
        load_q();

        // ---- LDS-DMA of the K / V tiles.  kv_wave: first row of THIS WAVE's four pieces of tile 0 (K or V section of the head) ----
        const size_t sect = (size_t)(img + 1) * (size_t)a.H * 4 + (size_t)head * 256;
        const char* kv_doc = reinterpret_cast<const char*>(a.qkv) + (size_t)qoff * row_bytes + sect;
        const unsigned tile_step = KT * row_bytes, piece_step = 4u * row_bytes;
        const char* kv_wave = kv_doc + (size_t)(4 * pstart) * row_bytes;
        // whole tile inside the document: piece jj = rows 4 (pstart + jj) .. + 3; the row offset rides in the lane offset
        auto issue_full = [&](int kt, unsigned sbase, int jj) __attribute__((always_inline)) {
            if (MODE == 2 && (dbg & 4)) return;
            if (MODE == 2 && (dbg & 16)) kt = 0;       // timing variant: every tile re-fetches the document's first rows (L2-resident source)
            const unsigned cj = 16u * (unsigned)(jj & 1) + 64u * (unsigned)((jj >> 1) & 1);
            unsigned vd = vdma0;
            asm volatile("" : "+v"(vd));               // opaque: recomputed per piece, never a set of precomputed (spillable) registers
            const unsigned long long base = (unsigned long long)(size_t)(kv_wave + (size_t)kt * tile_step);
            dma16((vd ^ cj) + (unsigned)jj * piece_step, base, sbase + dst_wave + 1024u * (unsigned)jj);
        };
        // the document's last, partial tile: rows past the document are clamped to its last row (and masked through the pair index /
        // the tail mask)
        auto issue_tail = [&](int kt, unsigned sbase, int jj) __attribute__((always_inline)) {
            if (MODE == 2 && (dbg & 4)) return;
            if (MODE == 2 && (dbg & 16)) kt = 0;
            const int k0 = kt * KT;
            const unsigned cj = 16u * (unsigned)(jj & 1) + 64u * (unsigned)((jj >> 1) & 1);
            unsigned vd = vdma0;
            asm volatile("" : "+v"(vd));
            const unsigned long long base = (unsigned long long)(size_t)(kv_doc + (size_t)k0 * row_bytes);
            const int lim = len - 1 - k0;
            const int pr = (int)((unsigned)lane >> 4);
            int r = 4 * (pstart + jj) + pr;
            r = r < lim ? r : lim;
            dma16((vd ^ cj) + (unsigned)(r - pr) * row_bytes, base, sbase + dst_wave + 1024u * (unsigned)jj);
        };
        auto issue_any = [&](int kt, unsigned sbase) __attribute__((always_inline)) {      // prologue and idle waves: one branch per tile
            if ((kt + 1) * KT <= len) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) issue_full(kt, sbase, jj);
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) issue_tail(kt, sbase, jj);
            }
        };

        HeadState st;
#pragma unroll
        for (int e = 0; e < 16; ++e) { st.o0[e] = 0.f; st.o1[e] = 0.f; }
        st.mref = kNegBig;
        st.l = 0.f;

        // ---- softmax (lazy rescaling, 2^10 folded into the exponent) + O^T += V^T P^T on a finished score tile; vs = this lane's V read
        // base in the tile's ring slot (vbase + slot base) ----
        auto softmax_pv = [&](f32x16& s, HeadState& st, const unsigned vs) __attribute__((always_inline)) {
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
            unsigned vb = vs;
            asm volatile("" : "+v"(vb));
            if (XP & 16) __builtin_amdgcn_s_setprio(1);
            if (XP & 32) __builtin_amdgcn_s_setprio(0);
            // O^T += V^T P^T.  B operand = P^T: for k-step ks, element j of lane (query, hh) is register 8 ks + j, i.e.
            // key 16 ks + 4 hh + (j & 3) + 8 (j >> 2); the A operand takes the same key order from two transposed reads
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // split of P: hi = f16(x) (v_cvt_pk_f16_f32), lo = f16(x - hi) by v_fma_mixlo / mixhi_f16 (fma(x, 1.0, -hi) in f32 -- exact --
                // rounded to f16: the bits of cvt(x - float(hi)), one instruction per value instead of three per pair)
                unsigned hw[4], lw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x0 = s[8 * ks + 2 * j], x1 = s[8 * ks + 2 * j + 1];
                    const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);
                    const unsigned hb = __builtin_bit_cast(unsigned, h);
                    unsigned lb = 0u;
                    if (TERMS == 3)
                        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
                            "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                            : "=&v"(lb)
                            : "v"(x0), "v"(x1), "v"(hb));
                    hw[j] = hb;
                    lw[j] = lb;
                }
                const u32x4 hq = {hw[0], hw[1], hw[2], hw[3]}, lq = {lw[0], lw[1], lw[2], lw[3]};
                const f16x8 ph8 = __builtin_bit_cast(f16x8, hq), pl8 = __builtin_bit_cast(f16x8, lq);
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    // rows 16 ks + row0 (+ 8), chunk (dch0 + 4 dh) + 8 plane: address = (vb ^ (64 dh + 128 plane + 32 x)) + 2048 x + 4096 ks
                    auto trd = [&](unsigned xorc, unsigned addc) __attribute__((always_inline)) {
                        return __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(size_t)((vb ^ xorc) + addc));
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
                        if (TERMS == 3) {
                            st.o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph8, st.o0, 0, 0, 0);
                            st.o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl8, st.o0, 0, 0, 0);
                        }
                        st.o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph8, st.o0, 0, 0, 0);
                    } else {
                        if (TERMS == 3) {
                            st.o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph8, st.o1, 0, 0, 0);
                            st.o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl8, st.o1, 0, 0, 0);
                        }
                        st.o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph8, st.o1, 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);        // k-step 1's fragments and split are not hoisted over k-step 0 (register pressure)
            }
            if (XP & 16) __builtin_amdgcn_s_setprio(0);
            if (XP & 32) __builtin_amdgcn_s_setprio(1);
        };

        const int n_kt = (len + KT - 1) / KT, n_full = len / KT;
        const bool want_idx = BIAS && !(MODE == 2 && (dbg & (1 | 32)));
        // index words of one key tile: four asm loads from a scalar tile pointer
        const unsigned* idx_base = BIAS ? a.pair_idx + (size_t)slab * a.idx_doc_stride + (size_t)qb * a.idx_nb * (I16 ? 512 : 1024) : nullptr;
        u32x4 iw0 = {0u, 0u, 0u, 0u}, iw1 = iw0, iw2 = iw0, iw3 = iw0;
        // IDX16: row j of a document is token j (j < nt) or patch j - nt; this lane's query position, the byte offset of its delta = 0 entry
        // minus that position (+ 16 bytes for the upper lane half, whose keys sit 4 further), the tile the text ends in, the document's masks
        const int nt16 = I16 ? len - a.n_visual : 0;
        const int qc16 = qi < len ? qi : len - 1;
        const unsigned abase16 = I16 ? (unsigned)(4 * (a.c1 - (qc16 < nt16 ? qc16 : qc16 - nt16)) + 16 * hh) : 0u;
        const int kts16 = I16 ? ((nt16 & 31) ? (nt16 >> 5) : -1) : -1;
        const bool docmask16 = I16 && __builtin_amdgcn_readfirstlane(a.doc_flags[slab]) != 0;      // fetched before the counted regime starts
        auto issue_idx = [&](int kt) __attribute__((always_inline)) {
            if (I16) {
                const unsigned long long ib16 = sgpr64((unsigned long long)(size_t)idx_base + (unsigned long long)kt * 2048ull);
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                             : "+v"(iw0), "+v"(iw1)
                             : "v"(ivoff), "s"(ib16)
                             : "memory");
                return;
            }
            const unsigned long long ib = sgpr64((unsigned long long)(size_t)idx_base + (unsigned long long)kt * 4096ull);
            // s_nop 4: the scalar base may come straight from v_readfirstlane (VALU write of an SGPR -> VMEM read needs 5 wait
            // states, and nothing inside an asm string is padded by the compiler)
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                         "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"
                         : "+v"(iw0), "+v"(iw1), "+v"(iw2), "+v"(iw3)
                         : "v"(ivoff), "s"(ib)
                         : "memory");
        };

        // ---- prologue: tiles 0 and 1 into slots 0 and 1, the index words of tile 0 ----
        issue_any(0, 0u);
        if ((XP & 2) && wave_active && want_idx) issue_idx(0);      // bias first: the index words of a tile are older than the DMA of the tile after it
        if (!R2 && n_kt > 1) issue_any(1, (unsigned)STAGE_BYTES);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the table stores have left this wave before it reaches tile 0's barrier
        STAMP(6, tprev)

        if (!wave_active) {
            // a wave without queries of this item (the document's last query tile): its share of the DMA and the barriers, nothing else
            unsigned sb2 = R2 ? (unsigned)STAGE_BYTES : 2u * STAGE_BYTES;
            for (int kt = 0; kt < n_kt; ++kt) {
                if (!R2 && kt + 1 < n_kt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(MODE == 2 && (dbg & 64))) asm volatile("s_barrier" ::: "memory");
                if (kt + AHEAD < n_kt) issue_any(kt + AHEAD, sb2);
                if (R2) sb2 ^= (unsigned)STAGE_BYTES;
                else sb2 = sb2 == 2u * STAGE_BYTES ? 0u : sb2 + (unsigned)STAGE_BYTES;
            }
            continue;
        }
        if (!(XP & 2) && want_idx) issue_idx(0);

        // ---- one key tile out of ring slot `sb`; VAR says what is left to fetch (compile time: no branch, constant wait counts) ----
        unsigned sb = 0u;                                    // byte base of the slot of tile kt; tile kt + 2 goes into the slot before it
        auto tile = [&](auto var_tag, const int kt) __attribute__((always_inline)) {
            constexpr int VAR = decltype(var_tag)::value;
            constexpr bool more1 = VAR != V_LAST, issue2 = VAR == V_HOT || VAR == V_TAIL;
            const unsigned sb2 = R2 ? (sb ^ (unsigned)STAGE_BYTES) : sb == 0u ? 2u * STAGE_BYTES : sb - (unsigned)STAGE_BYTES;
            // tile kt has landed: everything but the youngest operations -- the four pieces of tile kt + 1 and the four index loads of this
            // tile, issued after them -- is complete; then the barrier: everyone's pieces have, and everyone is done with tile kt - 1,
            // whose slot tile kt + 2 goes into
            constexpr bool BF = (XP & 2) != 0;            // bias first
            if (want_idx && !BF) {
                if (more1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {      // bias first: the youngest four operations are the pieces of tile kt + 1; this tile's index words are older
                if (more1 && !R2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // two-slot ring: this tile's pieces were the youngest
            }
            STAMP(0, tprev)
            if (!(MODE == 2 && (dbg & 64))) asm volatile("s_barrier" ::: "memory");      // 64: no barrier (timing variant: races)
            STAMP(7, tprev)
            f32x16 s;
            auto lookups = [&](bool init) __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 iw[4] = {iw0, iw1, iw2, iw3};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (p == 2) __builtin_amdgcn_sched_barrier(0);      // two groups of 24 lookups in flight, not 48: register pressure
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const unsigned v = iw[p][t];
                        const float b1 = lds_load<float>((v & 0x3ffu) + (unsigned)OFF_T1);
                        const float bx = lds_load<float>(((v >> 10) & 0x3ffu) + (unsigned)OFF_TX);
                        const float by = lds_load<float>((v >> 20) + (unsigned)OFF_TY);
                        const float b = b1 + (bx + by);          // rel_pos + (rel_pos_x + rel_pos_y), HF:268, 455; masked: -1e30
                        if (init) s[4 * p + t] = b;
                        else s[4 * p + t] += b;
                    }
                }
                if (more1) {                     // the registers are free: fetch the next tile's words
                    __builtin_amdgcn_sched_barrier(0);
                    issue_idx(kt + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto lookups16 = [&]() __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned iwv[8] = {iw0[0], iw0[1], iw0[2], iw0[3], iw1[0], iw1[1], iw1[2], iw1[3]};
                const int k0 = kt * KT;
                // masked keys of this tile: past the document (only its last tile can have them), or flagged rows (pad rows under
                // MMEE_FLAG_DENSE_ROWS, holes in the mask: per-document flag, then one scalar load per tile -- the slow path of a rare input)
                unsigned km = 0u;
                if (VAR == V_LAST && k0 + KT > len) km = ~0u << (unsigned)(len - k0);
                if (docmask16) {
                    const unsigned* kmp = a.keymask + (size_t)slab * a.idx_nb + kt;
                    unsigned kv;
                    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(kv) : "s"(kmp) : "memory");
                    km |= kv;
                }
                const bool straddle = kt == kts16;
                // byte offset of (position of the tile's first key) in the delta table, seen from this lane's query
                const unsigned At = abase16 + 4u * (unsigned)(k0 >= nt16 ? k0 - nt16 : k0);
                if (!straddle && km == 0u) {
                    // two halves of 8 scores: all 24 reads of a half are ISSUED before its first add (left to itself hipcc waits lgkmcnt(0) behind
                    // every four reads -- eight LDS round trips per tile instead of two: the first build of this path ran 9.8 ms against 9.2 ms)
#if MMEE_LKG == 0
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        if (p == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int e = 4 * p + t, kk0 = (e & 3) + 8 * (e >> 2);
                            const unsigned w = iwv[e >> 1];
                            const float b1 = lds_load<float>(At + (unsigned)(OFF_T1V + 4 * kk0));
                            const float bx = lds_load<float>(((e & 1) ? ((w >> 16) & 0xffu) : (w & 0xffu)) + (unsigned)OFF_TX16);
                            const float by = lds_load<float>(((e & 1) ? (w >> 24) : ((w >> 8) & 0xffu)) + (unsigned)OFF_TY16);
                            s[e] = b1 + (bx + by);
                        }
                    }
#else
#pragma unroll
                    for (int hf = 0; hf < 16 / LKG; ++hf) {
                        float b1v[LKG], bxv[LKG], byv[LKG];
#pragma unroll
                        for (int i = 0; i < LKG; ++i) {
                            const int e = LKG * hf + i, kk0 = (e & 3) + 8 * (e >> 2);
                            const unsigned w = iwv[e >> 1];
                            b1v[i] = lds_load<float>(At + (unsigned)(OFF_T1V + 4 * kk0));
                            bxv[i] = lds_load<float>(((e & 1) ? ((w >> 16) & 0xffu) : (w & 0xffu)) + (unsigned)OFF_TX16);
                            byv[i] = lds_load<float>(((e & 1) ? (w >> 24) : ((w >> 8) & 0xffu)) + (unsigned)OFF_TY16);
                        }
                        // an empty asm that names every result: all reads of the group are issued (and have landed) in front of it
                        static_assert(LKG == 4 || LKG == 8, "LKG");
                        if (LKG == 8)
                            asm volatile("" : "+v"(b1v[0]), "+v"(b1v[1]), "+v"(b1v[2]), "+v"(b1v[3]), "+v"(b1v[4 % LKG]), "+v"(b1v[5 % LKG]), "+v"(b1v[6 % LKG]), "+v"(b1v[7 % LKG]),
                                              "+v"(bxv[0]), "+v"(bxv[1]), "+v"(bxv[2]), "+v"(bxv[3]), "+v"(bxv[4 % LKG]), "+v"(bxv[5 % LKG]), "+v"(bxv[6 % LKG]), "+v"(bxv[7 % LKG]),
                                              "+v"(byv[0]), "+v"(byv[1]), "+v"(byv[2]), "+v"(byv[3]), "+v"(byv[4 % LKG]), "+v"(byv[5 % LKG]), "+v"(byv[6 % LKG]), "+v"(byv[7 % LKG]));
                        else
                            asm volatile("" : "+v"(b1v[0]), "+v"(b1v[1]), "+v"(b1v[2]), "+v"(b1v[3]), "+v"(bxv[0]), "+v"(bxv[1]), "+v"(bxv[2]), "+v"(bxv[3]),
                                              "+v"(byv[0]), "+v"(byv[1]), "+v"(byv[2]), "+v"(byv[3]));
#pragma unroll
                        for (int i = 0; i < LKG; ++i) s[LKG * hf + i] = b1v[i] + (bxv[i] + byv[i]);
                    }
#endif
                } else {
                    // keys >= m of this tile are visual rows: their positions restart at 0
                    const int m = straddle ? (nt16 & 31) : KT;
                    const unsigned Av = abase16 - 4u * (unsigned)m;
                    const int vthr = m - 4 * hh;
                    const unsigned kml = km >> (4 * hh);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        __builtin_amdgcn_sched_barrier(0);      // four scores at a time: this path is rare, its registers must not set the kernel's budget
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int e = 4 * p + t, kk0 = (e & 3) + 8 * (e >> 2);
                            // the extracts as asm: were they the fast path's expressions, hipcc would hoist all 32 of them above the branch
                            // (32 live registers in front of BOTH paths: 168 VGPRs + spills in the first build)
                            unsigned ox, oy;
                            asm volatile("v_bfe_u32 %0, %2, %3, 8\n\tv_bfe_u32 %1, %2, %4, 8" : "=&v"(ox), "=&v"(oy) : "v"(iwv[e >> 1]), "n"(16 * (e & 1)), "n"(16 * (e & 1) + 8));
                            // (the key offset is added BEFORE the read here: Av alone can be negative for a query near the end of the text)
                            const unsigned Ab = (kk0 >= vthr ? Av : At) + 4u * (unsigned)kk0;
                            const float b1 = lds_load<float>(Ab + (unsigned)OFF_T1V);
                            const float bx = lds_load<float>(ox + (unsigned)OFF_TX16);
                            const float by = lds_load<float>(oy + (unsigned)OFF_TY16);
                            const float b = b1 + (bx + by);
                            s[e] = ((kml >> kk0) & 1u) ? kNegBig : b;
                        }
                    }
                }
                if (more1) {                     // the registers are free: fetch the next tile's words
                    __builtin_amdgcn_sched_barrier(0);
                    issue_idx(kt + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (want_idx && BF && I16) lookups16();
            else if (want_idx && BF) lookups(true);       // the bias IS the initial accumulator: the matrix pipe adds it to the scores
            else {
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] = 0.f;
            }
            // S^T = K Q^T; fragments one k-step ahead of the MFMAs; tile kt + 2's DMA pieces between them
            const unsigned kb = kbase + sb;
            if (XP & 16) __builtin_amdgcn_s_setprio(1);      // the matrix-pipe phases issue ahead of the other waves' VALU work
            if (XP & 32) __builtin_amdgcn_s_setprio(0);      // (measured: the other way round)
            f16x8 kh = lds_load<f16x8>(kb), kl = lds_load<f16x8>(kb ^ 128u);
#pragma unroll
            for (int stp = 0; stp < 4; ++stp) {
                f16x8 khn = kh, kln = kl;
                if (stp < 3) {
                    khn = lds_load<f16x8>(kb ^ (32u * (stp + 1)));
                    kln = lds_load<f16x8>(kb ^ (32u * (stp + 1) + 128u));
                }
                if (!(MODE == 2 && (dbg & 128)) && TERMS == 3) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[stp], s, 0, 0, 0);      // 128: no Q K^T MFMAs
                else asm volatile("" :: "v"(kl), "v"(kh));
                if (VAR == V_HOT) issue_full(kt + AHEAD, sb2, stp);
                if (VAR == V_TAIL) issue_tail(kt + AHEAD, sb2, stp);
                if (!(MODE == 2 && (dbg & 128))) {
                    if (TERMS == 3) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[stp], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[stp], s, 0, 0, 0);
                }
                kh = khn;
                kl = kln;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (XP & 16) __builtin_amdgcn_s_setprio(0);
            if (XP & 32) __builtin_amdgcn_s_setprio(1);
            STAMP(3, tprev)
            // bias behind the Q K^T MFMAs: the index words are back when all but the pieces issued above are (they are older than those)
            if (want_idx && !BF) {
                // wait-only statements WITHOUT operands: naming the index registers here lets the register allocator copy them in front
                // of the wait (seen in the .s: v_mov of words that had not landed).  The sched_barrier keeps their first use below it.
                if (issue2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lookups(false);
            } else if (!BIAS && VAR == V_LAST) {          // no pair index (image-only model): mask the keys past the document here
                const int k0 = kt * KT;
                if (k0 + KT > len) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[e] = (k0 + (e & 3) + 8 * (e >> 2) + 4 * hh >= len) ? kNegBig : s[e];
                }
            }
            STAMP(2, tprev)
            softmax_pv(s, st, vbase + sb);
            STAMP(4, tprev)
            if (R2) sb ^= (unsigned)STAGE_BYTES;
            else sb = sb == 2u * STAGE_BYTES ? 0u : sb + (unsigned)STAGE_BYTES;
        };
        {
            int kt = 0;
            for (; kt + AHEAD < n_full; ++kt) tile(std::integral_constant<int, V_HOT>{}, kt);           // tile kt + AHEAD is a whole tile
            if (kt + AHEAD < n_kt) { tile(std::integral_constant<int, V_TAIL>{}, kt); ++kt; }            // ... is the partial last tile
            if (!R2 && kt + 1 < n_kt) { tile(std::integral_constant<int, V_PENULT>{}, kt); ++kt; }
            tile(std::integral_constant<int, V_LAST>{}, kt);
        }

        {
            const float l_tot = st.l + __shfl_xor(st.l, 32, 64);   // the two lane halves hold disjoint keys
            const float inv = 1.0f / (l_tot * a.qkv_scale);        // the 2^10 of the probabilities is in l as well
            const float inv_s = inv * a.ctx_scale;
            // context row of this lane's query as split planes.  The two lane halves of a query hold neighbouring 4-column groups (hh = 0:
            // columns 8 q4 .. + 3, hh = 1: + 4 .. + 7), i.e. adjacent 8-byte pieces of the hi plane and of the lo plane.  One
            // v_permlane32_swap per dword gives the lower half both hi pieces and the upper half both lo pieces: 8 stores of 16 bytes per
            // lane instead of 16 of 8 (the epilogue was store-issue bound: guide T21).
            char* row_split = reinterpret_cast<char*>(a.ctx) + (size_t)(off + (qi < len ? qi : len - 1)) * a.ldc * 4 + 32 * hh;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {       // registers 4*q4 .. 4*q4+3 <-> d = 8*q4 + 4*hh + (0..3)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    // o * (inv * ctx_scale): the plane scale is a power of two, so this is (o * inv) * ctx_scale bit for bit; the split is the
                    // GEMM epilogues' 4-instruction form (same hi / lo bits as split_f16x4 below the overflow bound, overflow flagged alike)
                    f32x4 w;
#pragma unroll
                    for (int c = 0; c < 4; ++c) w[c] = (half ? st.o1[4 * q4 + c] : st.o0[4 * q4 + c]) * inv_s;
                    unsigned h01, l01, h23, l23;
                    split_pair(w[0], w[1], h01, l01, amax);
                    split_pair(w[2], w[3], h23, l23, amax);
                    const u32x2 hx = {h01, h23}, lx = {l01, l23};
                    // lanes 32-63 of the first operand swap with lanes 0-31 of the second: lower half = [own hi | partner's hi],
                    // upper half = [partner's lo | own lo]
                    const auto s0 = __builtin_amdgcn_permlane32_swap(hx[0], lx[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(hx[1], lx[1], false, false);
                    const u32x4 piece = {(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
                    const int col = head * D + 8 * q4 + 32 * half;           // first of the pair's 8 columns
                    if (qi < len) *reinterpret_cast<u32x4*>(row_split + (col >> 4) * 64 + (col & 15) * 2) = piece;
                }
            }
        }
        STAMP(5, tprev)                                                       // normalisation, split conversion, context stores
    }
    if (a.ctx_split && MODE != 2) split_flag_overflow(amax, a.err_flag);      // the timing variants compute garbage by design
    if (DIAG && stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(stamps + i, ph[i]);
    }
#undef STAMP
}

static unsigned long long* g_attn_idx_stamps = nullptr;
unsigned long long* attention_idx_stamps() { return g_attn_idx_stamps; }

bool attention_idx_supports(const AttnArgs& a) {
    if (a.idx16 && !(a.pair_idx && a.lut1 && a.keymask && a.doc_flags && a.n1 >= 1 && a.n1 <= T1V_MAX && a.n_visual >= 0)) return false;
    return a.ctx_split && (a.pair_idx == nullptr || (a.bins1 >= 1 && a.bins1 <= BINS_MAX && a.bins2 >= 1 && a.bins2 <= BINS_MAX));
}
// can a handle with these bucket tables / this delta range run the 16-bit index?  (capi.hip decides once per handle)
bool attention_idx16_fits(int bins1, int bins2, int n1) { return bins1 >= 1 && bins1 <= BINS_MAX && bins2 >= 1 && bins2 <= BINS_MAX && n1 >= 1 && n1 <= T1V_MAX; }

template <bool BIAS, int XP, int TERMS = 3, int IDX = 32>
static void launch_idx(const AttnArgs& a, int max_docs, int num_cus, unsigned long long* stamps, int dbg, hipStream_t s) {
    constexpr int LDSB = (BIAS && IDX == 16) ? LDS_BYTES16 : LDS_BYTES;
    (void)ensure_dynamic_lds<&attention_idx_kernel<0, BIAS, XP, TERMS, IDX>>("attention_idx_kernel", LDSB);
#ifdef MMEE_DIAG
    (void)ensure_dynamic_lds<&attention_idx_kernel<1, BIAS, XP, TERMS, IDX>>("attention_idx_kernel", LDSB);
    (void)ensure_dynamic_lds<&attention_idx_kernel<2, BIAS, XP, TERMS, IDX>>("attention_idx_kernel", LDSB);
#endif
    const int qtiles = (a.max_len + QT - 1) / QT;
    long items = (long)max_docs * a.heads * qtiles;
    int grid = WGS * num_cus;
    if (items < grid) grid = (int)items;
    if (grid < 1) grid = 1;
#ifdef MMEE_DIAG      // stamped build and timing variants (wrong results): diagnostic library only
    if (stamps) { hipLaunchKernelGGL((attention_idx_kernel<1, BIAS, XP, TERMS, IDX>), dim3(grid), dim3(256), LDSB, s, a, stamps, 0); return; }
    if (dbg) { hipLaunchKernelGGL((attention_idx_kernel<2, BIAS, XP, TERMS, IDX>), dim3(grid), dim3(256), LDSB, s, a, (unsigned long long*)nullptr, dbg); return; }
#endif
    (void)stamps; (void)dbg;
    hipLaunchKernelGGL((attention_idx_kernel<0, BIAS, XP, TERMS, IDX>), dim3(grid), dim3(256), LDSB, s, a, (unsigned long long*)nullptr, 0);
}

// a.pair_idx == nullptr: no relative-position bias (image-only model); only the tail of a document's last key tile is masked.
// The release library runs ONE form of the kernel.  The diagnostic library (make diag, -DMMEE_DIAG) also carries the stamped build
// (MMEE_ATTN_STAMPS=1), the timing variants (MMEE_ATTN_DBG=<bits>, wrong results) and the experiment forms (MMEE_ATTN_XP=<bits>).
void launch_attention_idx(const AttnArgs& a, int max_docs, int num_cus, hipStream_t s) {
    unsigned long long* stamps = nullptr;
    int dbg = 0;
#ifdef MMEE_DIAG
    static unsigned long long* stamps_buf = [] {
        unsigned long long* p = nullptr;
        if (diag_env_int("MMEE_ATTN_STAMPS", 0) == 1 && hipMalloc((void**)&p, 64) == hipSuccess) (void)hipMemset(p, 0, 64);
        return p;
    }();
    static const int dbg_env = diag_env_int("MMEE_ATTN_DBG", 0);
    static const int xp = diag_env_int("MMEE_ATTN_XP", kXP);
    stamps = stamps_buf;
    dbg = dbg_env;
    g_attn_idx_stamps = stamps;
    if (a.pair_idx && xp != kXP && !a.idx16) {
        switch (xp) {
            case 0: launch_idx<true, 0>(a, max_docs, num_cus, stamps, dbg, s); return;
            case 2: launch_idx<true, 2>(a, max_docs, num_cus, stamps, dbg, s); return;
            case 22: launch_idx<true, 22>(a, max_docs, num_cus, stamps, dbg, s); return;
            case 6: launch_idx<true, 6>(a, max_docs, num_cus, stamps, dbg, s); return;
            case 18: launch_idx<true, 18>(a, max_docs, num_cus, stamps, dbg, s); return;
            case 34: launch_idx<true, 34>(a, max_docs, num_cus, stamps, dbg, s); return;

            default: break;
        }
    }
#endif
#ifdef MMEE_DIAG
    if (a.pair_idx && a.idx16) {         // round 6 experiment (diagnostic library, MMEE_ATTN_IDX=16): 16-bit pair index + delta table
        if (a.terms == 1 && !stamps && !dbg) launch_idx<true, kXP, 1, 16>(a, max_docs, num_cus, nullptr, 0, s);      // MMEE_FLAG_ONE_TERM
        else launch_idx<true, kXP, 3, 16>(a, max_docs, num_cus, stamps, dbg, s);
        return;
    }
#endif
    if (a.terms == 1 && a.pair_idx && !stamps && !dbg) { launch_idx<true, kXP, 1>(a, max_docs, num_cus, nullptr, 0, s); return; }      // MMEE_FLAG_ONE_TERM
    if (a.pair_idx) launch_idx<true, kXP>(a, max_docs, num_cus, stamps, dbg, s);
    else launch_idx<false, (kXP & ~2)>(a, max_docs, num_cus, stamps, dbg, s);      // no bias to put first
}

}  // namespace mmee
