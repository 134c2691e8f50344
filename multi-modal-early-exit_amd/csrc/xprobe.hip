// The CLS probe of an exit layer in X SPACE (precision mode MMEE_PREC_F32_SPLIT, LayoutLMv3; MMEE_FLAG_XPROBE).
//
// An exit head reads ONE row of a layer's output, the CLS row (EE/models/LayoutLMv3.py:226, 757-768), and probe-first layers decide from that
// row before the rest of the layer runs (capi.hip).  The probe of rounds 1-2 still projected Q | K | V for EVERY row of the stage first (the
// CLS query attends to all keys and values of its document) and then streamed all K | V rows once more (1.46 GB at the first exit of the
// bench batch).  But the CLS context does not need K and V as matrices.  With q = (W_q x_cls + b_q) / sqrt(d) of one head (HF:243-263):
//     score_j = q . k_j = q . (W_k x_j + b_k) = (W_k^T q) . x_j + q . b_k                       u := W_k^T q  (H values per head)
//     ctx     = sum_j p_j v_j = sum_j p_j (W_v x_j + b_v) = W_v (sum_j p_j x_j) + b_v          c := sum_j p_j x_j  (H values per head)
// (p = softmax of the biased, masked scores, HF:265-288; sum p = 1).  So the probe reads the LayerNorm rows x_j themselves -- 3 KB per row
// instead of 6 KB of K | V, and no Q | K | V projection has to exist yet: the layer's Q | K | V GEMM then runs for the documents that STAY
// only, and not at all in the last layer.  Four small launches:
//   1. Q of the CLS rows: the split GEMM on one row per document (capi.hip; W_q is the first third of the fused Q | K | V weight),
//   2. xprobe_u_kernel     u[d][h][:] = W_k,h^T q[d][h], s0[d][h] = q . b_k, the power-of-two plane scale of u, and the documents'
//                          order by falling length (weights streamed once per 8 documents),
//   3. xprobe_attn_kernel  one workgroup per document, longest first; its rows cross HBM -> LDS once (LDS-DMA, 16-row tiles, ring of
//                          3 / 2): scores against the heads' u on the f16 matrix cores (x rows ARE split-f16 planes; u is split in
//                          registers) + q . b_k + relative-position bias of query 0 from the pair index (attention_idx.hip) -> online
//                          softmax -> c += P^T X on the matrix cores from the same LDS tile,
//   4. xprobe_v_kernel     ctx[d][h] = W_v,h c[d][h] + b_v on the matrix cores -> the document's context row as split planes, where
//                          the probe's attention-output GEMM expects it.
// Shapes: 12 heads x 768 (LayoutLMv3-base) and 16 x 1024 (-large); xprobe_supports() says no to everything else.
// The result is a re-association of the whole-layer arithmetic (~1e-6 apart, inside the 1e-4 bar): with this flag "early exit == dump-all
// row bit for bit" holds to tolerance, not to the bit; MMEE_FLAG_WHOLE_LAYERS and the default probe stay bit-identical to each other.
#include <type_traits>
#include "mmee_kernels.h"

namespace mmee {

namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 xp_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
constexpr int XP_THREADS = 512;            // 8 waves
constexpr float kNegBig = -1.0e30f;
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// u[d][h][c] = sum_t q[d][h*64 + t] W_k[h*64 + t][c];  s0[d][h] = sum_t q[d][h*64 + t] b_k[h*64 + t]
// grid (heads, ceil(max_docs / 8)); thread <-> columns c = tid + 256 i
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xprobe_u_kernel(const XProbeArgs a) {
    const int n_docs = a.counts->n_docs, H = a.H;
    const int h = blockIdx.x, d0 = blockIdx.y * 8;
    const int tid = threadIdx.x;
    // the ticket counter of xprobe_attn_kernel is reset by EVERY launch, also when the stage is empty (n_docs == 0: the workspace is not
    // zeroed by the allocator, and a stale negative ticket would pass the `ticket < n_docs` test of the consumer)
    if (h == 0 && blockIdx.y == 0 && tid == 0) *a.ticket = 0;
    if (d0 >= n_docs) return;
    if (h == 0) {
        // work order of xprobe_attn_kernel: longest document first (its workgroups draw tickets; the makespan of a ragged stage is then the
        // mean, not two long documents on one CU).  Rank by counting: 32 lanes per document of this block's eight.
        const int d = d0 + (tid >> 5), sub = tid & 31;
        const int len = d < n_docs ? a.doc_off[d + 1] - a.doc_off[d] : 0;
        int rank = 0;
        for (int e = sub; e < n_docs; e += 32) {
            const int le = a.doc_off[e + 1] - a.doc_off[e];
            rank += (le > len || (le == len && e < d)) ? 1 : 0;
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) rank += __shfl_xor(rank, o, 64);
        if (sub == 0 && d < n_docs) a.order[rank] = d;
    }
    __shared__ float q_s[8][64];
    for (int i = tid; i < 512; i += 256) {
        const int g = i >> 6, t = i & 63, d = d0 + g;
        q_s[g][t] = d < n_docs ? a.qc[(size_t)d * H + h * 64 + t] : 0.f;
    }
    __syncthreads();
    float acc[8][4];
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[g][i] = 0.f;
#pragma unroll 8
    for (int t = 0; t < 64; ++t) {
        const float* wr = a.wk + (size_t)(h * 64 + t) * H;
        float w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = tid + 256 * i < H ? wr[tid + 256 * i] : 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float qv = q_s[g][t];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[g][i] = fmaf(qv, w[i], acc[g][i]);
        }
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int d = d0 + g;
        if (d >= n_docs) break;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (tid + 256 * i < H) a.u[((size_t)d * a.heads + h) * H + tid + 256 * i] = acc[g][i];
    }
    // q . b_k and the power-of-two scale that puts max |s u| into [2^12, 2^13) (the planes of u are formed in xprobe_attn_kernel)
    __shared__ float mx_s[4][8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) mx = fmaxf(mx, fabsf(acc[g][i]));
        mx = wave_max(mx);
        if ((tid & 63) == 0) mx_s[tid >> 6][g] = mx;
    }
    __syncthreads();
    if (tid < 8 && d0 + tid < n_docs) {
        float s = 0.f;
        for (int t = 0; t < 64; ++t) s = fmaf(q_s[tid][t], a.bk[h * 64 + t], s);
        const float mx = fmaxf(fmaxf(mx_s[0][tid], mx_s[1][tid]), fmaxf(mx_s[2][tid], mx_s[3][tid]));
        int ex = 0;
        (void)frexpf(mx, &ex);                                     // mx = f 2^ex, f in [0.5, 1)
        a.s0[((size_t)(d0 + tid) * a.heads + h) * 2] = s;
        a.s0[((size_t)(d0 + tid) * a.heads + h) * 2 + 1] = mx > 0.f ? ldexpf(1.0f, 13 - ex) : 1.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One workgroup (8 waves) per document; every row of the document crosses HBM -> LDS ONCE.
//   * 16-row tiles of the split rows arrive by LDS-DMA (global_load_lds_dwordx4, six 1-KiB pieces per wave) into a 3-deep ring: the DMA
//     of tile t+2 is issued when tile t starts, a tile waits vmcnt(6) (its own pieces, never the younger tile's).  Chunk c of tile row r
//     sits at chunk c ^ sw(r), sw(r) = (r & 7) << 1 | r >> 3, applied on the SOURCE side of the DMA: the row reads of the scores and the
//     transposed reads of the weighted sums are both conflict-free on unpadded 3072-byte rows (three tiles + tables = 160 KB, no room for pads).
//   * scores S[row][head] = x_row . u_head on the f16 matrix cores (v_mfma_f32_16x16x32_f16, three split terms); the 24 k-steps are
//     divided over the 8 waves -- a wave keeps the u fragments of ITS three k-steps in registers for the whole document -- the partial
//     tiles meet in LDS and EVERY wave adds them in the same order, so all waves hold the same scores, apply the same bias (bucket
//     bytes of query 0 from the pair index, staged in LDS once per document) and the same online softmax (running maximum that only
//     moves when exceeded by 2^5, as attention_idx.hip) -- no cross-wave softmax state.
//   * c[head][col] += sum_row p[row][head] x[row][col] is the second product of flash attention with M = heads: the accumulator layout of
//     the scores IS the A-operand layout of v_mfma_f32_16x16x16_f16 (p split in registers, scaled by 2^10), x comes out of the same LDS
//     tile through transposed reads (ds_read_b64_tr_b16).  A wave owns 96 of the 768 columns for all heads: no merge at the end.
// Dynamic LDS: three tiles | bias tables [heads][tstr] | partial score tiles [8 waves][4][12] f32x4 | bucket bytes b1, bx, by [npad] | ticket.
// Documents are drawn from a ticket counter in the order xprobe_u_kernel wrote: longest first.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long xp_sgpr64(unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
}
__device__ __forceinline__ void xp_dma16(unsigned voff, unsigned long long base, unsigned lds_addr) {
    unsigned keep;   // m0 is saved and restored; s_nop 4: the scalar base comes straight from v_readfirstlane
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(base)
                 : "memory");
}
__device__ __forceinline__ constexpr int xp_sw(int r) { return ((r & 7) << 1) | ((r >> 3) & 1); }

// HEADS x HH = 12 x 768 (LayoutLMv3-base, 3-deep ring) or 16 x 1024 (LayoutLMv3-large: a tile is 64 KB, so the ring is 2 deep: the DMA of
// tile t + 1 is issued when tile t starts and waited for with vmcnt(0)).
template <int HEADS, int HH, int RING>
__global__ __launch_bounds__(XP_THREADS) void xprobe_attn_kernel(const XProbeArgs a, const int tstr, const int off_tab, const int off_part, const int off_idx,
                                                                const int npad) {
    extern __shared__ __attribute__((aligned(1024))) char xsm[];
    constexpr int H = HH, NW = XP_THREADS / 64, ROWB = H * 4, XP_TILE = 16 * ROWB;
    constexpr int KSW = H / 32 / NW;          // k-steps of the scores per wave (3 / 4)
    constexpr int CGW = H / 16 / NW;          // 16-column groups of the weighted sums per wave (6 / 8)
    constexpr int PPR = ROWB / 1024;          // 1-KiB DMA pieces per tile row (3 / 4)
    constexpr int PIECES = 2 * PPR;           // pieces per wave and tile: two rows (6 / 8)
    static_assert(NW == 8 && KSW * NW * 32 == H && CGW * NW * 16 == H && HEADS * 64 == H && HEADS <= 16, "k-steps and column groups are divided over eight waves");
    static_assert((RING == 2 || RING == 3) && (PIECES == 6 || PIECES == 8), "ring depth / wait immediates");
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)xsm;
    const int n_docs = a.counts->n_docs, heads = a.heads;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* tab = reinterpret_cast<float*>(xsm + off_tab);         // [heads][tstr]: T1 (bins1 + 1) | TX (bins2) | TY (bins2), x log2(e) / sqrt(d)
    f32x4* part = reinterpret_cast<f32x4*>(xsm + off_part);       // [8 waves][4 kq][HEADS]
    unsigned char* ib1 = reinterpret_cast<unsigned char*>(xsm + off_idx);      // buckets of (query 0, key j)
    unsigned char* ibx = ib1 + npad;
    unsigned char* iby = ibx + npad;
    constexpr float kL2e = 1.44269504088896340736f;
    const int m = lane & 15, kq = lane >> 4;
    const int mh = m < HEADS ? m : m - HEADS;                     // lanes of absent heads carry a real head's u; their scores are discarded
    int* tick_s = reinterpret_cast<int*>(xsm + off_idx + 3 * npad);
    for (;;) {
        __syncthreads();                                           // the previous document's LDS reads (and its ticket) are done with
        if (tid == 0) *tick_s = atomicAdd(a.ticket, 1);
        __syncthreads();
        const int tk = __builtin_amdgcn_readfirstlane(*tick_s);
        if ((unsigned)tk >= (unsigned)n_docs) break;
        const int d = __builtin_amdgcn_readfirstlane(a.order[tk]);
        const int off = a.doc_off[d], len = a.doc_off[d + 1] - off;
        const char* x0 = a.xs + (size_t)a.x_phys[d] * ROWB;
        const unsigned* slab = a.pair_idx + (size_t)a.doc_orig[d] * a.idx_doc_stride;      // query block 0 of the document
        const int n_rt = (len + 15) >> 4;
        auto issue_tile = [&](int rt) __attribute__((always_inline)) {
            const int buf = rt % RING;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int row = 2 * wave + i / PPR, pc = i % PPR;
                int src = rt * 16 + row;
                src = src < len ? src : len - 1;                   // rows past the end: a copy of the last row, weighted with p = 0
                const unsigned long long base = xp_sgpr64((unsigned long long)(size_t)(x0 + (size_t)src * ROWB + pc * 1024));
                xp_dma16((unsigned)(lane ^ xp_sw(row)) * 16u, base, lds0 + (unsigned)(buf * XP_TILE + row * ROWB + pc * 1024));
            }
        };
        issue_tile(0);
        if (RING == 3 && n_rt > 1) issue_tile(1);
        // ---- a. the u fragments of the wave's three k-steps as split-f16 planes (scale of xprobe_u_kernel: max |s u| in [2^12, 2^13)) ----
        f16x8 bh[KSW], bl[KSW];
        float inv, s0;
        {
            const float us = a.s0[((size_t)d * heads + mh) * 2 + 1];
            inv = kL2e * a.xs_inv / us;                            // x planes carry 1 / xs_inv (= 16); scores are kept in log2 units
            s0 = a.s0[((size_t)d * heads + mh) * 2] * kL2e;
            const float* ur = a.u + ((size_t)d * heads + mh) * H + 8 * kq;
#pragma unroll
            for (int i = 0; i < KSW; ++i) {
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(ur + 32 * (KSW * wave + i));
                const f32x4 u1 = *reinterpret_cast<const f32x4*>(ur + 32 * (KSW * wave + i) + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (e < 4 ? u0[e] : u1[e - 4]) * us;
                    const _Float16 hi = (_Float16)v;
                    bh[i][e] = hi;
                    bl[i][e] = (_Float16)(v - (float)hi);
                }
            }
        }
        // ---- b. the heads' raw bucket tables (rel_pos + rel_2d_pos are divided by sqrt(d) like the scores, HF:265-268), in log2 units ----
        for (int i = tid; i < heads * tstr; i += XP_THREADS) {
            const int h = i / tstr, r = i - h * tstr;
            float v;
            if (r < a.bins1) v = a.w1[h * a.bins1 + r] * a.inv_sqrt_d * kL2e;
            else if (r == a.bins1) v = kNegBig;                    // masked key (sentinel bucket of the pair index)
            else if (r < a.bins1 + 1 + a.bins2) v = a.wx[h * a.bins2 + r - a.bins1 - 1] * a.inv_sqrt_d * kL2e;
            else v = a.wy[h * a.bins2 + r - a.bins1 - 1 - a.bins2] * a.inv_sqrt_d * kL2e;
            tab[i] = v;
        }
        // ---- c. buckets of (query 0, key j): pair-index tile (0, j >> 5), lane 32 hh, register e (attention_idx.hip, pair_index_kernel) ----
        for (int j = tid; j < n_rt * 16; j += XP_THREADS) {
            unsigned w = 0u;
            if (j < len) {
                const int kb = j >> 5, ko = j & 31, hh = (ko >> 2) & 1, e = (ko & 3) + 4 * (ko >> 3);
                w = slab[(size_t)kb * 1024 + (e >> 2) * 256 + 128 * hh + (e & 3)];
            }
            ib1[j] = (unsigned char)((w & 0x3ffu) >> 2);
            ibx[j] = (unsigned char)((w >> 12) & 0xffu);
            iby[j] = (unsigned char)(w >> 22);
        }
        // ---- d. the tiles ----
        f32x4 cacc[CGW];
#pragma unroll
        for (int i = 0; i < CGW; ++i) cacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m_run = kNegBig, l_run = 0.f;                        // of head m; l over this lane's rows 4 kq + r of every tile
        const float* th = tab + (m < heads ? m : 0) * tstr;
        const int swm = xp_sw(m);                                  // row reads: lane <-> tile row m
        const int rowt = 4 * kq + (m >> 2), swt = xp_sw(rowt);     // transposed reads: lane 4 q + p of group kq supplies row 4 kq + q, columns 4 p ..
        for (int rt = 0; rt < n_rt; ++rt) {
            const int buf = rt % RING;
            if (RING == 3 && rt + 1 < n_rt) {                      // this wave's pieces of tile rt have landed (tile rt + 1 may fly)
                if (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();                                        // ... and everybody's; tile rt - 1 is no longer read
            if (rt + RING - 1 < n_rt) issue_tile(rt + RING - 1);
            const char* tile = xsm + buf * XP_TILE;
            // partial scores of the wave's three k-steps
            {
                const char* pa = tile + m * ROWB;
                f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < KSW; ++i) {
                    const int c = 8 * (KSW * wave + i) + 4 * (kq >> 1) + (kq & 1);     // 16-byte chunk of the hi plane; lo two chunks on
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(pa + 16 * (c ^ swm));
                    const f16x8 al = *reinterpret_cast<const f16x8*>(pa + 16 * ((c + 2) ^ swm));
                    ps = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[i], ps, 0, 0, 0);
                    ps = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[i], ps, 0, 0, 0);
                    ps = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[i], ps, 0, 0, 0);
                }
                if (m < HEADS) part[(wave * 4 + kq) * HEADS + m] = ps;
            }
            __syncthreads();
            f32x4 sc = part[kq * HEADS + mh];
#pragma unroll
            for (int w = 1; w < NW; ++w) sc += part[(w * 4 + kq) * HEADS + mh];
            // sc[r] <-> (row 16 rt + 4 kq + r, head m): bias of query 0, online softmax per head
            const int r0 = rt * 16;
            float sv[4];
            float tmax = kNegBig;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = r0 + 4 * kq + r;
                const float b1 = th[ib1[j]], bx = th[a.bins1 + 1 + ibx[j]], by = th[a.bins1 + 1 + a.bins2 + iby[j]];
                const float v = (j < len && m < heads) ? sc[r] * inv + s0 + (b1 + (bx + by)) : kNegBig;
                sv[r] = v;
                tmax = fmaxf(tmax, v);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));          // over the tile's 16 rows, for head m
            float alpha = 1.0f;
            if (tmax > m_run + 5.0f) {                             // lazy: the reference only moves when exceeded by 2^5
                alpha = __builtin_amdgcn_exp2f(m_run - tmax);
                m_run = tmax;
                l_run *= alpha;
            }
            if (__any(alpha != 1.0f)) {                            // accumulators of head 4 kq + r
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float f = __shfl(alpha, 4 * kq + r, 64);
#pragma unroll
                    for (int i = 0; i < CGW; ++i) cacc[i][r] *= f;
                }
            }
            f16x4 p_hi, p_lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(sv[r] - m_run + 10.0f);     // x 2^10: the lo plane stays normal; < 2^15
                l_run += p;
                p_hi[r] = (_Float16)p;
                p_lo[r] = (_Float16)(p - (float)p_hi[r]);
            }
            // weighted sums: D[head][col] += P^T[head][row] X[row][col]; 16-lane group kq takes rows 4 kq .. 4 kq + 3 of the 16 columns
            const unsigned tb = lds0 + (unsigned)(buf * XP_TILE + rowt * ROWB + 8 * (m & 1));
#pragma unroll
            for (int i = 0; i < CGW; ++i) {
                const int c = 4 * (CGW * wave + i) + ((m >> 1) & 1);               // chunk of columns 16 g + 4 p .. (hi); lo two chunks on
                const unsigned ah = tb + 16u * (unsigned)(c ^ swt), al = tb + 16u * (unsigned)((c + 2) ^ swt);
                const f16x4 xh = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) xp_h4*)(size_t)ah));
                const f16x4 xl = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) xp_h4*)(size_t)al));
                cacc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(p_lo, xh, cacc[i], 0, 0, 0);
                cacc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(p_hi, xl, cacc[i], 0, 0, 0);
                cacc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(p_hi, xh, cacc[i], 0, 0, 0);
            }
        }
        // ---- e. normalise: lane (col m of the group, heads 4 kq + r) ----
        l_run += __shfl_xor(l_run, 16, 64);
        l_run += __shfl_xor(l_run, 32, 64);
        auto store_head = [&](auto rc) __attribute__((always_inline)) {      // (a lambda per register: `#pragma unroll` over r trips -Wpass-failed here)
            constexpr int r = decltype(rc)::value;
            const int h = 4 * kq + r;
            const float lh = __shfl(l_run, h, 64);
            if (h < heads) {
                const float f = a.xs_inv / lh;                     // the 2^10 of p cancels in c / l
#pragma unroll
                for (int i = 0; i < CGW; ++i) a.cvec[((size_t)d * heads + h) * H + (CGW * wave + i) * 16 + m] = cacc[i][r] * f;
            }
        };
        store_head(std::integral_constant<int, 0>{});
        store_head(std::integral_constant<int, 1>{});
        store_head(std::integral_constant<int, 2>{});
        store_head(std::integral_constant<int, 3>{});
    }
}

// ---------------------------------------------------------------------------------------------------------------
// ctx[d][h*64 + t] = sum_c W_v[h*64 + t][c] c[d][h][c] + b_v[h*64 + t]  ->  split planes at context row doc_off[d]
// A [documents x H] . [H x 64] product per head on the f16 matrix cores: one wave = 16 documents x the 64 outputs of a head.  A = c
// (f32 from xprobe_attn_kernel, split in registers with the scale of the LayerNorm planes: |c| <= max |x|), B = the rows of the layer's
// split-f16 value weights (the fused Q | K | V weight of ee_finalize, 16 bytes per lane and k-step straight from global memory / L2);
// three terms per product as everywhere.  grid (heads, ceil(max_docs / 64)), 4 waves.  (The first version walked the f32 weights with
// VALU dot products, 8 documents per workgroup: 122 us at 1024 documents against 27 for this one.)
// ---------------------------------------------------------------------------------------------------------------
template <int HH>
__global__ __launch_bounds__(256) void xprobe_v_kernel(const XProbeArgs a) {
    constexpr int H = HH;                                          // xprobe_supports: 768 or 1024
    const int n_docs = a.counts->n_docs;
    const int h = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d0 = (blockIdx.y * 4 + wave) * 16;
    if (d0 >= n_docs) return;                                      // wave-uniform
    const int m = lane & 15, kq = lane >> 4;
    const int dm = d0 + m < n_docs ? d0 + m : n_docs - 1;
    const float* cp = a.cvec + ((size_t)dm * a.heads + h) * H + 8 * kq;
    const char* wp = reinterpret_cast<const char*>(a.wv_s) + (size_t)(h * 64 + m) * H * 4 + (kq >> 1) * 64 + (kq & 1) * 16;
    const size_t wtile = (size_t)16 * H * 4;                       // 16 output rows further
    constexpr float kSc = 16.0f;                                   // = kSplitScaleX
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int ks = 0; ks < H / 32; ++ks) {
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp + 32 * ks), c1 = *reinterpret_cast<const f32x4*>(cp + 32 * ks + 4);
        f16x8 ah, al;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (e < 4 ? c0[e] : c1[e - 4]) * kSc;
            const _Float16 hi = (_Float16)v;
            ah[e] = hi;
            al[e] = (_Float16)(v - (float)hi);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(wp + j * wtile + 128 * ks);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(wp + j * wtile + 128 * ks + 32);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[j], 0, 0, 0);
        }
    }
    // acc[j][r] <-> (document d0 + 4 kq + r, output 16 j + m): bias, then the split planes of the context row (one f16 per plane and lane:
    // 16 lanes write 32 contiguous bytes)
    const float inv = a.wv_inv / kSc;
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = h * 64 + 16 * j + m;
        const float bias = a.bv[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = d0 + 4 * kq + r;
            if (d < n_docs) {
                const float v = (acc[j][r] * inv + bias) * a.ctx_scale;
                amax = fmaxf(amax, fabsf(v));
                const float vc = fminf(fmaxf(v, -kSplitClamp), kSplitClamp);
                const _Float16 hi = (_Float16)vc;
                const _Float16 lo = (_Float16)(vc - (float)hi);
                char* row = reinterpret_cast<char*>(a.ctx) + (size_t)a.doc_off[d] * H * 4 + (size_t)(col >> 4) * 64 + (size_t)(col & 15) * 2;
                *reinterpret_cast<_Float16*>(row) = hi;
                *reinterpret_cast<_Float16*>(row + 32) = lo;
            }
        }
    }
    split_flag_overflow(amax, a.err_flag);
}

namespace {
struct XpLayout { int tstr, off_tab, off_part, off_idx, npad, lds; };
XpLayout xp_layout(const XProbeArgs& a, int max_len) {
    XpLayout L;
    const int ring = a.H == 768 ? 3 : 2;
    L.tstr = a.bins1 + 1 + 2 * a.bins2;
    L.npad = (max_len + 15) & ~15;
    L.off_tab = ring * 16 * a.H * 4;
    L.off_part = (L.off_tab + a.heads * L.tstr * 4 + 15) & ~15;
    L.off_idx = L.off_part + 8 * 4 * a.heads * 16;
    L.lds = L.off_idx + 3 * L.npad + 16;
    return L;
}
}  // namespace

bool xprobe_supports(const XProbeArgs& a, int max_len) {
    // built for 12 heads x 768 columns (LayoutLMv3-base) and 16 x 1024 (LayoutLMv3-large): eight waves share H / 32 k-steps and H / 16
    // column groups; other models run the probe of attention_idx.hip
    const bool shape = (a.H == 768 && a.heads == 12) || (a.H == 1024 && a.heads == 16);
    return shape && a.pair_idx != nullptr && a.bins1 <= 255 && a.bins2 <= 255 && xp_layout(a, max_len).lds <= 160 * 1024;
}

template <int HEADS, int HH, int RING>
static void launch_xprobe_attn(const XProbeArgs& a, const XpLayout& L, int grid, hipStream_t s) {
    // per kernel AND device (a process-wide high-water mark left a second device without its opt-in); a failure shows up as the
    // launch error ee_forward reports
    (void)ensure_dynamic_lds<&xprobe_attn_kernel<HEADS, HH, RING>>("xprobe_attn_kernel", L.lds);
    hipLaunchKernelGGL((xprobe_attn_kernel<HEADS, HH, RING>), dim3(grid), dim3(XP_THREADS), L.lds, s, a, L.tstr, L.off_tab, L.off_part, L.off_idx, L.npad);
}

void launch_xprobe(const XProbeArgs& a, int max_docs, int max_len, int num_cus, hipStream_t s) {
    const int groups = (max_docs + 7) / 8;
    hipLaunchKernelGGL(xprobe_u_kernel, dim3(a.heads, groups), dim3(256), 0, s, a);
    const XpLayout L = xp_layout(a, max_len);
    int grid = max_docs < num_cus ? max_docs : num_cus;
    if (grid < 1) grid = 1;
    if (a.H == 768) {
        launch_xprobe_attn<12, 768, 3>(a, L, grid, s);
        hipLaunchKernelGGL(xprobe_v_kernel<768>, dim3(a.heads, (max_docs + 63) / 64), dim3(256), 0, s, a);
    } else {
        launch_xprobe_attn<16, 1024, 2>(a, L, grid, s);
        hipLaunchKernelGGL(xprobe_v_kernel<1024>, dim3(a.heads, (max_docs + 63) / 64), dim3(256), 0, s, a);
    }
}

}  // namespace mmee
