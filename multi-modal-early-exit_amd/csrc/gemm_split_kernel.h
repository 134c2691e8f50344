// The split-precision GEMM kernel template and its launcher (see gemm_split.hip for the design notes).  A header because the kernel is
// instantiated in TWO translation units: gemm_split.hip (every instantiation but one) and gemm_split_ffn_up.hip, which holds the FFN-up
// instantiation (256 x 256 tile, GELU epilogue, split output, three terms) and is compiled with -mllvm -amdgpu-sched-strategy=max-ilp: that
// scheduler is worth +1.3-1.7 % on this instantiation and -0.4...-0.9 % on the others (round 4, tools/gemm_ab.py and tools/lib_ab.sh;
// profiles/r04_gemm_epilogue_experiments.txt section 22).  Same source, same arithmetic, same bits.
#pragma once
#include <cstdlib>
#include "mmee_common.h"

namespace mmee {


// Tile configurations.  ROWB = bytes of one LDS row = one k-stage of one operand row: 64 (k = 16: [hi 16 | lo 16]) or
// 128 (k = 32: two such groups).  Each wave owns a 64x64 sub-tile (2x2 MFMA tiles); WM x WN waves per workgroup.
//   CfgA  128x256, k16 stages, 3-deep ring (72 KiB), 8 waves, 2 workgroups per CU: one workgroup's epilogue runs under the
//         other's MFMA stream, at the price of 64-byte DMA rows.
//   CfgB  256x256, k32 stages, 2-deep ring (128 KiB), 16 waves, 1 workgroup per CU (default): whole 128-byte lines per
//         DMA row (half the L2 requests per byte of CfgA, which rocprof showed at 43 % of the L2 request slots) and a third
//         less global -> LDS traffic per MAC; the epilogue is exposed but the kernel is power-limited (shader clock
//         1.35-1.6 GHz under f16 MFMA load), so the energy saved on data movement wins: 390 vs 362 TFLOP/s on the
//         bias epilogue, 317 vs 297 on GELU + split output (tools/gemm_split_epi.py).
template <int BM_, int BN_, int ROWB_, int NST_, int WM_, int WN_, int WGS_, int MF_ = 32>
struct SplitCfg {
    static constexpr int BM = BM_, BN = BN_, ROWB = ROWB_, NST = NST_, WM = WM_, WN = WN_, WGS = WGS_;
    static constexpr int MF = MF_;                               // MFMA shape: 32 = 32x32x16, 16 = 16x16x32 (needs ROWB = 128)
    static_assert(MF == 32 || (MF == 16 && ROWB == 128), "the 16x16x32 MFMA consumes one 128-byte row (k = 32) per step");
    static constexpr int NW = WM * WN, THREADS = NW * 64;
    static constexpr int KSTAGE = ROWB / 4;                      // k values per stage (16 or 32)
    static constexpr int PROWS = 1024 / ROWB;                    // rows per 1 KiB DMA piece (16 or 8)
    static constexpr int PA = BM / PROWS / NW, PW = BN / PROWS / NW;   // pieces per wave per stage
    static constexpr int PP = PA + PW;
    static constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB, LOOP_BYTES = NST * STAGE_BYTES;
    static constexpr int EPI_BYTES = NW * 32 * 64 * 4;
    static_assert(BM == WM * 64 && BN == WN * 64, "one 64x64 sub-tile per wave");
    static_assert(PA * PROWS * NW == BM && PW * PROWS * NW == BN && PA >= 1 && PW >= 1, "DMA pieces must tile the stage");
    static_assert(EPI_BYTES <= LOOP_BYTES, "epilogue staging must fit in the stage ring");
    static_assert(NST >= 2 && NST <= 4, "ring depth");
    // refused at compile time rather than at launch: a configuration the CU cannot hold (round 2's 128 x 64 sub-tile experiment died with
    // SIGABRT inside ee_debug_gemm_split's GELU / split-output case and left no diagnostic)
    static_assert(THREADS <= 1024, "a workgroup is at most 16 waves");
    static_assert(WGS * (LOOP_BYTES + 16) <= 160 * 1024, "LDS budget of a CU (160 KiB) for WGS workgroups");
    static_assert(WGS * NW <= 32, "wave slots of a CU");
};
using CfgA = SplitCfg<128, 256, 64, 3, 2, 4, 2>;
using CfgB = SplitCfg<256, 256, 128, 2, 4, 4, 1>;
// CfgC = CfgB on v_mfma_f32_16x16x32_f16.  The kernel is power-limited (tools/mfma_f16_peak.hip: a bare 32x32x16 loop on
// random operands holds 1.62 GHz = 1674 TFLOP/s, a bare 16x16x32 loop 1.89 GHz = 1945 TFLOP/s): the 16x16x32 form moves
// half the accumulator bits per MAC, and the clock the chip can hold rises with it.
using CfgC = SplitCfg<256, 256, 128, 2, 4, 4, 1, 16>;
// CfgP: the CLS-probe GEMMs (M = documents of the stage, a few hundred rows).  Those launches are a handful of tiles whose k-loop
// is bound by the latency of a stage, not by the matrix pipe.  Measured on the six probes of one bench step (three GEMMs each):
// CfgC 1.80 ms, 64x128 on 2 waves 1.73 ms, 128x256 on 8 waves 1.26 ms, 128x128 on 4 waves with the 3-deep ring 0.91 ms (one wave
// per SIMD, two stages in flight).  Round 3: a 4-deep ring (three stages = 96 KB in flight per CU; the ring code takes NST = 4) is
// SLOWER end to end, 6669 against 6704 docs/s on one box (tools/lib_ab.sh).  Per output element the MFMA sequence is CfgC's (same
// 16x16x32 form, same k order, same term order), so the results are CfgC's bit for bit.
using CfgP = SplitCfg<128, 128, 128, 3, 2, 2, 1, 16>;


__device__ __forceinline__ void dma_piece(unsigned voff, unsigned long long base, unsigned lds_addr) {
    unsigned keep;   // m0 is saved and restored: the compiler does not accept it in a clobber list
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(base)
                 : "memory");
}

template <int EPI, bool OUT_SPLIT, int WN>
__device__ __forceinline__ void split_store_tile(const GemmArgs& g, float* smem, f32x16 (&acc)[2][2], int m0, int n0, int M,
                                                 int wave, int lane) {
    const int wr = wave / WN, wc = wave % WN;
    const int l31 = lane & 31, hh = lane >> 5;
    float* stg = smem + wave * (32 * 64);           // 8 KB per wave, one 32-row half of its sub-tile at a time
    const int c4 = (lane & 15) * 4;                 // 4 consecutive columns of the wave's 64
    const int col = n0 + wc * 64 + c4;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + col);
    const float sc = (col < g.scale_cols) ? g.scale : 1.0f;      // scale_cols is a multiple of 64
    f32x4 lam = f32x4{1.f, 1.f, 1.f, 1.f};
    if (g.col_scale) lam = *reinterpret_cast<const f32x4*>(g.col_scale + col);
    const float alpha = g.alpha;
    float amax = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * hh;
                stg[r * 64 + ni * 32 + l31] = acc[mi][ni][e];
            }
        asm volatile("" ::: "memory");      // float writes, f32x4 reads: keep the compiler from hoisting the reads (see split_store_tile16)
        const int rbase = m0 + wr * 64 + mi * 32 + (lane >> 4);
#pragma unroll 4
        for (int j = 0; j < 8; ++j) {
            const int rl = (lane >> 4) + 4 * j;
            const int row = rbase + 4 * j;
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * 64 + c4);
            if (row < M) {
#pragma unroll
                for (int t = 0; t < 4; t += 2) {      // two columns at a time: packed f32 VALU
                    f32x2 x = __builtin_elementwise_fma(f32x2{v[t], v[t + 1]}, (f32x2)(alpha), f32x2{bv[t], bv[t + 1]}) * (f32x2)(sc);
                    if (EPI == EPI_GELU) x = gelu_erf2(x);
                    if (EPI == EPI_TANH) { x[0] = tanhf(x[0]); x[1] = tanhf(x[1]); }
                    x = x * f32x2{lam[t], lam[t + 1]};
                    v[t] = x[0];
                    v[t + 1] = x[1];
                }
                if (EPI == EPI_RESID) {
                    const int rs = g.resid_row_src ? g.resid_row_src[row] : row;
                    if (g.resid_split_inv != 0.f) {      // split planes: hi + lo by v_fma_mix_f32, the power-of-two 1 / scale folded into the add
                        const f32x4 r = load_split4_sum(g.resid + (size_t)rs * g.ldr, col);
#pragma unroll
                        for (int t = 0; t < 4; t += 2) {
                            const f32x2 y = __builtin_elementwise_fma(f32x2{r[t], r[t + 1]}, (f32x2)(g.resid_split_inv), f32x2{v[t], v[t + 1]});
                            v[t] = y[0];
                            v[t + 1] = y[1];
                        }
                    } else {
                        v += *reinterpret_cast<const f32x4*>(g.resid + (size_t)rs * g.ldr + col);
                    }
                }
                if (!OUT_SPLIT) *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + col) = v;
            }
            if (OUT_SPLIT) {
                // The four lanes of a quad hold columns 4q .. 4q+3 of one 16-column group, whose 64 output bytes are
                // [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15]: a quad permute hands lane q the 16-byte piece number q, so the
                // row leaves as one fully coalesced 16-byte-per-lane store (as the f32 output does) instead of two 8-byte
                // scatters per lane.  Every lane takes part in the permute (rows >= M only skip the store).
                f16x4 hi, lo;
                split_f16x4(row < M ? v : f32x4{0.f, 0.f, 0.f, 0.f}, g.out_scale, hi, lo, amax);
                const int2 h2 = __builtin_bit_cast(int2, hi), l2 = __builtin_bit_cast(int2, lo);
                const bool take_lo = (lane & 2) != 0;
                int4 piece;       // lane q: q = 0 -> hi of lanes 0,1; 1 -> hi of lanes 2,3; 2 -> lo of lanes 0,1; 3 -> lo of lanes 2,3
                {
                    const int a0 = __builtin_amdgcn_mov_dpp(h2.x, 0x88, 0xf, 0xf, true), a1 = __builtin_amdgcn_mov_dpp(h2.y, 0x88, 0xf, 0xf, true);
                    const int b0 = __builtin_amdgcn_mov_dpp(l2.x, 0x88, 0xf, 0xf, true), b1 = __builtin_amdgcn_mov_dpp(l2.y, 0x88, 0xf, 0xf, true);
                    const int c0 = __builtin_amdgcn_mov_dpp(h2.x, 0xDD, 0xf, 0xf, true), c1 = __builtin_amdgcn_mov_dpp(h2.y, 0xDD, 0xf, 0xf, true);
                    const int d0 = __builtin_amdgcn_mov_dpp(l2.x, 0xDD, 0xf, 0xf, true), d1 = __builtin_amdgcn_mov_dpp(l2.y, 0xDD, 0xf, 0xf, true);
                    piece.x = take_lo ? b0 : a0;
                    piece.y = take_lo ? b1 : a1;
                    piece.z = take_lo ? d0 : c0;
                    piece.w = take_lo ? d1 : c1;
                }
                if (row < M)
                    *reinterpret_cast<int4*>(reinterpret_cast<char*>(g.C) + (size_t)row * g.ldc * 4 + (size_t)(col >> 4) * 64 + (lane & 3) * 16) = piece;
            }
        }
    }
    if (OUT_SPLIT) split_flag_overflow(amax, g.err_flag);
}

// Accumulator staging of the 16x16x32 form: tile (mi, ni) of the wave's 4x4 has col = lane & 15, rows 4 (lane >> 4) + reg.
// One 32-row half (mi = 2 half, 2 half + 1) is written to the wave's [32][64] f32 staging block; the column index is XOR-ed with
// 16 on rows whose (row >> 2) is odd, so the two 16-lane groups of a 32-lane write group hit disjoint banks.
// RB = rows staged at a time (32: 8 KB per wave; 16: 4 KB per wave, so that the 16 waves' staging fits ONE ring slot and the other
// slot can already receive the next tile's first stage while this epilogue runs).
// The residual epilogue (attention-output and FFN-down GEMMs) is the expensive one: N = 768, K = 768 runs at 290-295 algorithmic TFLOP/s
// with it and at 380 with the bias epilogue on the same operands, N = 2304 at 341 against 411 (tools/gemm_attn_out_why.py,
// profiles/r03_gemm_attn_out_why.txt): ~19 us per 256 x 256 tile for 256 KB of residual that the k-loop of the ONE workgroup on the CU
// cannot hide.  Measured on top of this form, none moved the kernel: (1) touching the sub-tile's 128 residual lines eight stages before the
// end of the k-loop (dword LDS-DMA into a scratch strip: no register, no compiler wait) -- 284 vs 284; (2) branch-free fetches of 16 rows
// at a time issued as soon as the previous 16 rows' registers are free -- 290, +5 spilled registers at the 128-VGPR cap; (3) the same with
// two register sets -- 21 spills; (4) the add moved into the LayerNorm kernel that follows (bias epilogue + f32 store + residual there: the
// same bits, all 57 GPU tests green): attention-output share of the step 8.1 -> 6.3 %, FFN-down 22.5 -> 21.2 %, LayerNorm 5.0 -> 8.0 %,
// 6750 vs 6757 docs/s -- the 0.73 GB cost the same HBM time wherever they are read, so the residual stays here.  What remains is holding
// the first rows' residual across the last k-stages, which needs 16 registers the 16-wave configuration does not have, or 4 KB of LDS
// per wave where 2 KB are free.
template <int EPI, bool OUT_SPLIT, int WN, int RB>
__device__ __forceinline__ void split_store_tile16(const GemmArgs& g, float* smem, f32x4 (&acc)[4][4], int m0, int n0, int M,
                                                   int wave, int lane, const f32x4 bv, const f32x4 lam, const size_t c_shift = 0) {
    const int wr = wave / WN, wc = wave % WN;
    const int l15 = lane & 15, gq = lane >> 4;
    float* stg = smem + wave * (RB * 64);
    const int c4 = (lane & 15) * 4;
    const int col = n0 + wc * 64 + c4;
    const float sc = (col < g.scale_cols) ? g.scale : 1.0f;      // scale_cols is a multiple of 64: uniform over the wave
    // Round 4: the epilogue is VALU-issue bound (one wave64 instruction per four cycles per SIMD, the matrix pipe idle), so everything that
    // can be folded is: a power-of-two Q scale and the (power-of-two) plane scale of a split output go into alpha and the bias -- exact --
    // so a bias element is ONE fma; the activation forms take the unscaled value and produce the scaled one (gelu_scaled2); a Q scale that is
    // not a power of two (no supported model) and the BEiT per-column factor sit behind wave-uniform branches
    const float s_out = OUT_SPLIT ? g.out_scale : 1.0f;
    constexpr bool ACT = EPI == EPI_GELU || EPI == EPI_TANH;
    const bool sc_p2 = (__float_as_uint(sc) & 0x007fffffu) == 0u;
    const float fold = ACT ? 1.0f : s_out * (sc_p2 ? sc : 1.0f);
    const float post = ACT ? sc : (sc_p2 ? 1.0f : sc);
    const bool need_post = __builtin_amdgcn_readfirstlane((int)__float_as_uint(post)) != 0x3f800000;
    const bool need_lam = g.col_scale != nullptr;
    const float alpha_e = g.alpha * fold;
    const f32x4 be = bv * fold;
    const float gelu_c0 = __builtin_log2f(s_out) - 1.0f, gelu_hs = 0.5f * s_out;
    float amax = 0.f;
#pragma unroll
    for (int half = 0; half < 64 / RB; ++half) {
#pragma unroll
        for (int m2 = 0; m2 < RB / 16; ++m2)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 16 * m2 + 4 * gq + e;                      // (r >> 2) & 1 == gq & 1
                    stg[r * 64 + ((ni * 16 + l15) ^ (16 * (gq & 1)))] = acc[(RB / 16) * half + m2][ni][e];
                }
        // The staging block is written as floats and read back as f32x4: without this compiler barrier hipcc may (and, after the round-4
        // edits, did) hoist the first read above half of the writes -- type-based alias analysis sees no conflict.  The hardware needs
        // nothing: a wave's LDS operations execute in order.
        asm volatile("" ::: "memory");
        const int rbase = m0 + wr * 64 + half * RB + (lane >> 4);
#pragma unroll
        for (int j = 0; j < RB / 4; ++j) {
            const int rl = (lane >> 4) + 4 * j;
            const int row = rbase + 4 * j;
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * 64 + (c4 ^ (16 * ((rl >> 2) & 1))));
            // (rows past M compute on whatever the clamped A rows gave and are not stored; the quad permute below needs every lane)
#pragma unroll
            for (int t = 0; t < 4; t += 2) {
                f32x2 x = __builtin_elementwise_fma(f32x2{v[t], v[t + 1]}, (f32x2)(alpha_e), f32x2{be[t], be[t + 1]});
                if (need_post) x = x * (f32x2)(post);
                if (EPI == EPI_GELU) x = gelu_scaled2(x, gelu_c0, gelu_hs);
                if (EPI == EPI_TANH) { x[0] = tanhf(x[0]) * s_out; x[1] = tanhf(x[1]) * s_out; }
                if (need_lam) x = x * f32x2{lam[t], lam[t + 1]};
                v[t] = x[0];
                v[t + 1] = x[1];
            }
            if (row < M) {
                if (EPI == EPI_RESID) {
                    const int rs = g.resid_row_src ? g.resid_row_src[row] : row;
                    if (g.resid_split_inv != 0.f) {      // split planes: hi + lo by v_fma_mix_f32, the power-of-two 1 / scale folded into the add
                        const f32x4 r = load_split4_sum(g.resid + (size_t)rs * g.ldr, col);
#pragma unroll
                        for (int t = 0; t < 4; t += 2) {
                            const f32x2 y = __builtin_elementwise_fma(f32x2{r[t], r[t + 1]}, (f32x2)(g.resid_split_inv), f32x2{v[t], v[t + 1]});
                            v[t] = y[0];
                            v[t + 1] = y[1];
                        }
                    } else {
                        v += *reinterpret_cast<const f32x4*>(g.resid + (size_t)rs * g.ldr + col);
                    }
                }
                if (!OUT_SPLIT) *reinterpret_cast<f32x4*>(g.C + c_shift + (size_t)row * g.ldc + col) = v;
            }
            if (OUT_SPLIT) {
                // The four lanes of a quad hold columns 4q .. 4q+3 of one 16-column group, whose 64 output bytes are
                // [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15]: a quad permute hands lane q the 16-byte piece number q, so the row leaves as
                // one fully coalesced 16-byte-per-lane store: consecutive lanes on consecutive bytes, 4 rows x 256 B per instruction -- the
                // only shape the store path takes at speed (tools/store_rate.hip).  Every lane takes part in the permute.
                int2 h2, l2;
                {
                    unsigned h01, h23, l01, l23;
                    split_pair(v[0], v[1], h01, l01, amax);
                    split_pair(v[2], v[3], h23, l23, amax);
                    h2.x = (int)h01; h2.y = (int)h23; l2.x = (int)l01; l2.y = (int)l23;
                }
                const bool take_lo = (lane & 2) != 0;
                int4 piece;       // lane q: q = 0 -> hi of lanes 0,1; 1 -> hi of lanes 2,3; 2 -> lo of lanes 0,1; 3 -> lo of lanes 2,3
                {
                    const int a0 = __builtin_amdgcn_mov_dpp(h2.x, 0x88, 0xf, 0xf, true), a1 = __builtin_amdgcn_mov_dpp(h2.y, 0x88, 0xf, 0xf, true);
                    const int b0 = __builtin_amdgcn_mov_dpp(l2.x, 0x88, 0xf, 0xf, true), b1 = __builtin_amdgcn_mov_dpp(l2.y, 0x88, 0xf, 0xf, true);
                    const int c0 = __builtin_amdgcn_mov_dpp(h2.x, 0xDD, 0xf, 0xf, true), c1 = __builtin_amdgcn_mov_dpp(h2.y, 0xDD, 0xf, 0xf, true);
                    const int d0 = __builtin_amdgcn_mov_dpp(l2.x, 0xDD, 0xf, 0xf, true), d1 = __builtin_amdgcn_mov_dpp(l2.y, 0xDD, 0xf, 0xf, true);
                    piece.x = take_lo ? b0 : a0;
                    piece.y = take_lo ? b1 : a1;
                    piece.z = take_lo ? d0 : c0;
                    piece.w = take_lo ? d1 : c1;
                }
                if (row < M)
                    *reinterpret_cast<int4*>(reinterpret_cast<char*>(g.C) + (size_t)row * g.ldc * 4 + (size_t)(col >> 4) * 64 + (lane & 3) * 16) = piece;
            }
        }
        asm volatile("" ::: "memory");      // ... and the next slice's writes stay behind this slice's reads
    }
    if (OUT_SPLIT) split_flag_overflow(amax, g.err_flag);
}

// TAG only names the instantiation (1 = the CLS-probe launches of capi.hip, so that a profiler keeps them apart from the
// layer's own GEMMs); the code is the same, and so is every result bit.
// TERMS = 1 (MMEE_FLAG_ONE_TERM): only hi x hi -- plain f16 operands, f32 accumulate; the lo planes are fetched with their rows but never read.
template <typename Cfg, int EPI, bool OUT_SPLIT, bool DIAG = false, int TAG = 0, int TERMS = 3>
__global__ __launch_bounds__(Cfg::THREADS, 4) void gemm_split_kernel(const GemmArgs g) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, ROWB = Cfg::ROWB, NST = Cfg::NST, WN = Cfg::WN, PA = Cfg::PA, PW = Cfg::PW;
    constexpr int STAGE_BYTES = Cfg::STAGE_BYTES, A_BYTES = Cfg::A_BYTES;
    const int dbg = DIAG ? g.dbg_noload : 0;     // timing diagnostics (ee_debug_gemm_split), compiled out of the path's kernels
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = g.m_ptr ? *g.m_ptr : g.m_static;
    const int tiles_m = (M + BM - 1) / BM;
    const int tiles_n = g.N / BN;
    // split-K (the CLS-probe launches, TAG 1, only): K is divided over k_splits workgroups per tile, part p goes to C + p * split_stride
    constexpr bool KSPLIT = TAG == 1;
    const int ksp = KSPLIT && g.k_splits > 1 ? g.k_splits : 1;
    const int n_tiles = tiles_m * tiles_n * ksp;
    const int nk = g.K / Cfg::KSTAGE / ksp;
    int ks_pop = 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int l31 = lane & 31, hh = lane >> 5;

    unsigned long long clk0 = 0, rt0 = 0;
    if (DIAG && g.clk_probe) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int GM = 8;      // (round 4: 4 is equal on all four layer shapes, 16 is 0-2.5 % slower; tools/gemm_ab.py)
    int* q_slot = reinterpret_cast<int*>(reinterpret_cast<char*>(smem) + Cfg::LOOP_BYTES);
    const int n_groups = (tiles_m + GM - 1) / GM;
    const int my_xcd = g.tile_counter ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u) : 0;   // HW_REG_XCC_ID
    int q_try = 0;
    int tile = blockIdx.x;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    // DMA: a 1 KiB piece is PROWS rows x ROWB bytes written linearly; the lane that fills physical 16-byte chunk p of row r
    // fetches logical chunk p ^ swz(r), and the fragment reads apply the same XOR (conflict-free ds_read_b128):
    //   ROWB = 64:  4 chunks per row, swz(r) = (r >> 2) & 3 (pieces start at multiples of 16 rows: depends on the lane only)
    //   ROWB = 128: 8 chunks per row, swz(r) = (r >> 1) & 7 (pieces are 8 rows: odd pieces add 4)
    constexpr int CPR = ROWB / 16;                                   // chunks per row
    const int p_row = lane / CPR, p_chunk = lane % CPR;
    const int swz_even = ROWB == 64 ? ((p_row >> 2) & 3) : ((p_row >> 1) & 7);
    const unsigned chunk_even = 16u * (unsigned)(p_chunk ^ swz_even);
    const unsigned chunk_odd = ROWB == 64 ? chunk_even : 16u * (unsigned)(p_chunk ^ ((swz_even + 4) & 7));
    // fragment reads: lane (r = l31, h = hh) takes k = 8h .. 8h+7 of k-step s: logical chunks 4s + h (hi) and 4s + 2 + h (lo)
    const unsigned rswz = ROWB == 64 ? (unsigned)((l31 >> 2) & 3) : (unsigned)((l31 >> 1) & 7);
    const unsigned a_row = (unsigned)(wr * 64 + l31) * ROWB;
    const unsigned w_row = (unsigned)A_BYTES + (unsigned)(wc * 64 + l31) * ROWB;
    // 16x16x32 form: lane (r = lane & 15, q = lane >> 4) takes k = 8q .. 8q+7 of the 32: logical chunks (q & 1) + 4 (q >> 1)
    // (hi) and + 2 (lo) of row r of a 16-row tile; the row XOR only sees (r >> 1) & 7 (tiles start at multiples of 16)
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned swz16 = (unsigned)((l15 >> 1) & 7);
    const unsigned c16_hi = 16u * (((unsigned)(lq & 1) + 4u * (unsigned)(lq >> 1)) ^ swz16);
    const unsigned c16_lo = 16u * (((unsigned)(lq & 1) + 4u * (unsigned)(lq >> 1) + 2u) ^ swz16);
    const unsigned a_row16 = (unsigned)(wr * 64 + l15) * ROWB;
    const unsigned w_row16 = (unsigned)A_BYTES + (unsigned)(wc * 64 + l15) * ROWB;
    const char* sbytes = reinterpret_cast<const char*>(smem);

    // HANDOVER (the default configuration): the epilogue stages through ONE ring slot (4 KB per wave), so the next tile is drawn
    // from the queue as soon as the k-loop ends and its first stage lands in the other slot while the epilogue runs.
    constexpr bool HANDOVER = Cfg::MF == 16 && NST == 2 && Cfg::NW * 4096 <= STAGE_BYTES;

    auto pop = [&](int& tm, int& tn) __attribute__((always_inline)) -> bool {
        if (g.tile_counter) {
            while (q_try < 8) {
                const int q = (my_xcd + q_try) & 7;
                if (tid == 0) *q_slot = atomicAdd(g.tile_counter + 16 * q, 1);
                __syncthreads();
                const int j = *q_slot;
                __syncthreads();
                if (g.tile_order >= 2) {
                    // Round 6 experiments (profiles/r06_gemm_xcd_schedules.txt).  Orders 0 / 1 walk ONE group of GM M-tiles through
                    // all N-tiles before the queue moves to its next group: the 32 CUs of an XCD run GM M-tiles x 4 N-tiles at a time, the next 32
                    // entries keep the A panels (GM x 256 rows x K: 6.3 MB at K = 768, more than the 4 MB L2) and change the W tiles -- both are
                    // refetched.  tile_order 2, W-STATIONARY: the queue keeps a block of CW = 4 N-tiles (3 MB of W at K = 768) and sweeps ALL of
                    // its M-groups under it before it takes the next block: W stays in the XCD's L2, only A streams.  tile_order 3, the verdict's
                    // A-PANEL order: one 256-row A panel (0.77 MB) at a time through all its N-tiles (GM = 1, N fastest).
                    const int gq = q < n_groups ? (n_groups - q + 7) / 8 : 0;      // M-groups this queue owns: q, q + 8, ...
                    if (g.tile_order == 2) {
                        constexpr int CW = 4;
                        const int nfull = tiles_n / CW, per_cb = gq * GM * CW;
                        int cb, jj, w;
                        if (per_cb > 0 && j < nfull * per_cb) { cb = j / per_cb; jj = j - cb * per_cb; w = CW; }
                        else { cb = nfull; jj = j - nfull * per_cb; w = tiles_n - CW * nfull; }
                        if (w > 0 && jj < gq * GM * w) {
                            const int gl2 = jj / (GM * w), r2 = jj - gl2 * GM * w;
                            tn = cb * CW + r2 / GM;
                            tm = (q + 8 * gl2) * GM + (r2 - (r2 / GM) * GM);
                            if (tm < tiles_m) return true;
                            continue;
                        }
                    } else {
                        const int per_q = gq * GM * tiles_n;
                        if (j < per_q) {
                            const int pl = j / tiles_n;                            // this queue's pl-th A panel
                            tn = j - pl * tiles_n;
                            tm = (q + 8 * (pl / GM)) * GM + (pl - (pl / GM) * GM);
                            if (tm < tiles_m) return true;
                            continue;
                        }
                    }
                    ++q_try;
                    continue;
                }
                const int per_group = GM * tiles_n;
                const int gl = j / per_group, r = j - gl * per_group;
                const int grp = q + 8 * gl;
                if (grp < n_groups) {
                    if (g.tile_order == 1) {
                        const int mi = r / tiles_n;
                        tn = r - mi * tiles_n;
                        tm = grp * GM + mi;
                    } else {
                        tn = r / GM;
                        tm = grp * GM + (r - tn * GM);
                    }
                    if (tm < tiles_m) return true;
                    continue;
                }
                ++q_try;
            }
            return false;
        }
        if (tile >= n_tiles) return false;
        int t2 = tile;
        if (KSPLIT) { t2 = tile / ksp; ks_pop = tile - t2 * ksp; }
        tm = t2 / tiles_n;
        tn = t2 - tm * tiles_n;
        tile += gridDim.x;
        return true;
    };

    // per-tile DMA sources: SGPR base + 32-bit lane offset (A rows may be gathered; row_src is increasing).
    // Wave w owns A pieces w, w + NW, ... and W pieces likewise.
    struct TileCtx {
        int m0, n0;
        int kt0;                                     // split-K: first k-stage of this workgroup's part
        unsigned a_voff[PA], w_voff[PW];
        unsigned long long a_base, w_base;
    };
    auto setup = [&](int tm, int tn, TileCtx& t) __attribute__((always_inline)) {
        t.m0 = __builtin_amdgcn_readfirstlane(tm * BM);
        t.n0 = __builtin_amdgcn_readfirstlane(tn * BN);
        t.kt0 = KSPLIT ? __builtin_amdgcn_readfirstlane(ks_pop * nk) : 0;
        int first = g.row_src ? g.row_src[t.m0] : t.m0;
        if (DIAG && (dbg & 16)) first = 0;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int piece = wave + Cfg::NW * j;
            int ra = t.m0 + Cfg::PROWS * piece + p_row;
            ra = ra < M ? ra : M - 1;
            if (DIAG && (dbg & 16)) ra = Cfg::PROWS * piece + p_row;      // timing variant (wrong results): every tile streams A panel 0, an L2-resident A operand
            const int sa = g.row_src ? g.row_src[ra] : ra;
            t.a_voff[j] = (unsigned)(sa - first) * (unsigned)g.lda * 4u + ((piece & 1) ? chunk_odd : chunk_even);
        }
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int piece = wave + Cfg::NW * j;
            t.w_voff[j] = (unsigned)(Cfg::PROWS * piece + p_row) * (unsigned)g.K * 4u + ((piece & 1) ? chunk_odd : chunk_even);
        }
        const unsigned long long a_base_v = (unsigned long long)(size_t)g.A + (unsigned long long)first * (unsigned)g.lda * 4ull;
        const unsigned long long w_base_v = (unsigned long long)(size_t)g.W + (unsigned long long)t.n0 * (unsigned)g.K * 4ull;
        t.a_base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a_base_v >> 32)) << 32) |
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(a_base_v & 0xffffffffu));
        t.w_base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w_base_v >> 32)) << 32) |
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(w_base_v & 0xffffffffu));
    };
    auto issue = [&](const TileCtx& t, int kt, int buf) __attribute__((always_inline)) {
        const unsigned long long koff = (unsigned long long)(kt + (KSPLIT ? t.kt0 : 0)) * (unsigned)ROWB;
        const unsigned dst = lds0 + (unsigned)buf * STAGE_BYTES + (unsigned)wave * 1024u;
#pragma unroll
        for (int j = 0; j < PA; ++j) dma_piece(t.a_voff[j], t.a_base + koff, dst + (unsigned)(Cfg::NW * j) * 1024u);
#pragma unroll
        for (int j = 0; j < PW; ++j) dma_piece(t.w_voff[j], t.w_base + koff, dst + A_BYTES + (unsigned)(Cfg::NW * j) * 1024u);
    };

    TileCtx cur;
    int buf0 = 0;                                    // ring slot of the current tile's stage 0
    {
        int tm, tn;
        if (!pop(tm, tn)) return;                    // uniform over the workgroup (diagnostic clock probe: nothing to report)
        setup(tm, tn, cur);
    }
    // ring: stage kt lives in slot (buf0 + kt) % NST; NST - 1 stages are in flight ahead of the one being consumed
    issue(cur, 0, buf0);
    if (NST >= 3 && nk > 1) issue(cur, 1, 1);
    if (NST >= 4 && nk > 2) issue(cur, 2, 2);
    for (;;) {
        const int m0 = cur.m0, n0 = cur.n0;
        const size_t c_shift = KSPLIT && ksp > 1 ? (size_t)(cur.kt0 / nk) * g.split_stride : 0;
        f32x16 acc[2][2];
        f32x4 acc16[4][4];
        if constexpr (Cfg::MF == 32) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        int buf = buf0, bufn = (buf0 + NST - 1) % NST;     // slots of stage kt and of stage kt + NST - 1
        // (measured, round 3: forcing the k-loop's first instruction onto a 32 / 64 / 256-byte boundary with .p2align changes nothing: 6564-6593
        // docs/s for all four builds on one box, tools/lib_ab.sh)
        for (int kt = 0; kt < nk; ++kt) {
            // my pieces of stage kt have landed (NST == 3: the pieces of stage kt + 1 may still be in flight), then the
            // barrier: everyone's have, and everyone has left stage kt - 1, whose slot the next issue overwrites
            if ((dbg & 6) == 6) {                    // diagnostic: neither the DMA wait nor the barrier
            } else if (dbg & 2) {                    // diagnostic: no barrier (results wrong)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (dbg & 4) {                    // diagnostic: no DMA wait
                asm volatile("s_barrier" ::: "memory");
            } else if (NST == 4 && kt + 2 < nk) {      // two younger stages may still be in flight
                static_assert(NST != 4 || Cfg::PP == 8, "vmcnt immediate: two stages' pieces per wave");
                asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
            } else if (NST >= 3 && kt + 1 < nk) {
                static_assert(Cfg::PP == 3 || Cfg::PP == 4 || Cfg::PP == 8, "vmcnt immediate: one stage's pieces per wave");
                if (Cfg::PP == 3) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
                else if (Cfg::PP == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
            // (measured, round 4: issuing this DMA behind the MFMAs of the first / second / third W fragment instead costs 1-4 % / 3-8 % / 4-8 % on
            // the four layer shapes -- the lead of a stage's DMA matters; profiles/r04_gemm_epilogue_experiments.txt section 10)
            if (kt + NST - 1 < nk && !(dbg & 1)) issue(cur, kt + NST - 1, bufn);
            const char* sb = sbytes + buf * STAGE_BYTES;
            if constexpr (Cfg::MF == 16) {
                // A fragments of the four 16-row tiles stay in registers; W fragments come one 16-column tile at a time
                f16x8 ah[4], al[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(sb + a_row16 + c16_hi + i * 16 * ROWB);
                    al[i] = *reinterpret_cast<const f16x8*>(sb + a_row16 + c16_lo + i * 16 * ROWB);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(sb + w_row16 + c16_hi + j * 16 * ROWB);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(sb + w_row16 + c16_lo + j * 16 * ROWB);
                    if (TERMS == 3) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], wh, acc16[i][j], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], wl, acc16[i][j], 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], wh, acc16[i][j], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int ks = 0; ks < ROWB / 64; ++ks) {
                const unsigned c_hi = 16u * (((unsigned)(4 * ks) + (unsigned)hh) ^ rswz);
                const unsigned c_lo = 16u * (((unsigned)(4 * ks + 2) + (unsigned)hh) ^ rswz);
                f16x8 ah[2], al[2], wh[2], wl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(sb + a_row + c_hi + i * 32 * ROWB);
                    al[i] = *reinterpret_cast<const f16x8*>(sb + a_row + c_lo + i * 32 * ROWB);
                    wh[i] = *reinterpret_cast<const f16x8*>(sb + w_row + c_hi + i * 32 * ROWB);
                    wl[i] = *reinterpret_cast<const f16x8*>(sb + w_row + c_lo + i * 32 * ROWB);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
                    }
            }
            buf = buf == NST - 1 ? 0 : buf + 1;
            bufn = bufn == NST - 1 ? 0 : bufn + 1;
        }
        __syncthreads();                             // every wave is done with the ring before (part of) it becomes the staging area
        // the epilogue's per-column constants are fetched BEFORE the hand-over DMA is issued: hipcc waits vmcnt(0) at the first use of
        // an ordinary load, which would otherwise also wait for the LDS-DMA pieces issued below
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f}, lam = f32x4{1.f, 1.f, 1.f, 1.f};
        if constexpr (Cfg::MF == 16) {
            const int col = n0 + wc * 64 + (lane & 15) * 4;
            if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + col);
            if (g.col_scale) lam = *reinterpret_cast<const f32x4*>(g.col_scale + col);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv), "+v"(lam) :: "memory");
        }
        // draw the next tile now; with HANDOVER its first stage goes into the slot the last stage did NOT use, and the epilogue
        // below stages through the last stage's slot only
        TileCtx nxt;
        bool have;
        {
            int tm, tn;
            have = pop(tm, tn);
            if (have) setup(tm, tn, nxt);
        }
        const int last_slot = (buf0 + nk - 1) % NST;
        const int nbuf0 = HANDOVER ? (last_slot ^ 1) : 0;
        if (HANDOVER && have) issue(nxt, 0, nbuf0);
        if (!(dbg & 8)) {
            if constexpr (Cfg::MF == 16) {
                if constexpr (HANDOVER)
                    split_store_tile16<EPI, OUT_SPLIT, WN, 16>(g, smem + last_slot * (STAGE_BYTES / 4), acc16, m0, n0, M, wave, lane, bv, lam);
                else
                    split_store_tile16<EPI, OUT_SPLIT, WN, 32>(g, smem, acc16, m0, n0, M, wave, lane, bv, lam, c_shift);
            } else {
                split_store_tile<EPI, OUT_SPLIT, WN>(g, smem, acc, m0, n0, M, wave, lane);
            }
        }
        __syncthreads();
        if (!have) break;
        cur = nxt;
        buf0 = nbuf0;
        if (!HANDOVER) {
            issue(cur, 0, 0);
            if (NST >= 3 && nk > 1) issue(cur, 1, 1);
            if (NST >= 4 && nk > 2) issue(cur, 2, 2);
        }
    }
    if (DIAG && g.clk_probe && threadIdx.x == 0) {      // diagnostic: shader clock = d(memtime) / d(memrealtime) * 100 MHz
        g.clk_probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
        g.clk_probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

template <typename Cfg, int EPI, bool OUT_SPLIT, bool DIAG = false, int TAG = 0, int TERMS = 3>
static void launch_split_one(const GemmArgs& a, int max_m, int num_cus, hipStream_t s) {
    const size_t lds = Cfg::LOOP_BYTES + 16;
    (void)ensure_dynamic_lds<&gemm_split_kernel<Cfg, EPI, OUT_SPLIT, DIAG, TAG, TERMS>>("gemm_split_kernel", (int)lds);
    const int tiles = ((max_m + Cfg::BM - 1) / Cfg::BM) * (a.N / Cfg::BN) * (TAG == 1 && a.k_splits > 1 ? a.k_splits : 1);
    int grid = Cfg::WGS * num_cus;
    if (grid > tiles) grid = tiles;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((gemm_split_kernel<Cfg, EPI, OUT_SPLIT, DIAG, TAG, TERMS>), dim3(grid), dim3(Cfg::THREADS), lds, s, a);
}

// CfgC is the default for every GEMM (measured end to end: 5672 docs/s, CfgB 5425, CfgA for the GELU GEMM + CfgB 5283), CfgP for the CLS-probe
// GEMMs.  The release library holds exactly these; the other configurations (MMEE_SPLIT_CFG=1 / 2 for CfgA / CfgB) and the timing
// diagnostics (GemmArgs::dbg_noload, wrong results) are compiled into the diagnostic library only (make diag, -DMMEE_DIAG).
}  // namespace mmee
