// fp32 GEMM on the CDNA4 matrix cores: C[M,N] = A[M,K] * W[N,K]^T (+ bias, + fused epilogue).
//
// Replaces every nn.Linear / Conv2d of the encoder path (precision "fp32"; with the default split precision only the patch
// embedding and the exit heads):
//   QKV projection  HF:243-258 | attention output dense HF:299-303 | FFN up + GELU HF:485-497 | FFN down HF:508-512
//   patch embedding Conv2d(k = s = 16) HF:71-83 (AMODE_IM2COL) | exit-head / classifier dense + tanh
//   (EE/models/LayoutLMv3.py:86-93, HF:799-823) on gathered CLS rows.
//
// Design (gfx950):
//   * v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate (bitwise an fmaf chain) — the only way to meet the
//     1e-4 logit tolerance over 12-24 layers; peak 157 TFLOP/s (MI355X_MICROARCH.md, Matrix cores).
//   * 128x128 output tile per 256-thread workgroup, 2x2 waves, each wave a 64x64 sub-tile = 2x2 MFMA tiles
//     (64 accumulator registers); BK = 32 per stage.  64 MFMAs (4096 matrix-pipe cycles) per wave per stage against
//     16 ds_read_b128 + 8 global_load_dwordx4: the loop is matrix-pipe bound, two workgroups per CU interleave.
//   * K is permuted consistently for both operands: lane half h of MFMA step c in k-group g consumes
//     k = 8g + 4h + c, so one ds_read_b128 per operand feeds four MFMAs.
//   * LDS rows padded to 36 floats: the 16 lanes of a ds_read_b128 group start on 16 distinct 4-bank slots.
//   * two kernels with the same tiling and epilogue: gemm_f32_dma_kernel (default; global_load_lds straight into unpadded,
//     XOR-swizzled LDS rows, fragment double buffering with counted lgkmcnt waits, the stage barrier placed between k-groups 2
//     and 3 of the previous stage) and gemm_f32_kernel (register-staged double buffering into padded rows: global_load ->
//     VGPR during compute, ds_write after; MMEE_GEMM_DMA=0 selects it).  The AMODE_IM2COL patch-embedding GEMM, the exit
//     heads and precision "fp32" run here; the big layer GEMMs of the default precision run in gemm_split.hip.
//   * persistent grid-stride over tiles; M is read from device memory (rows of the still-active documents), so the
//     host never synchronises to size a launch after an exit stage.
//   * optional row gather on A and on the residual: the stream compaction after an exit is fused into the next
//     layer's loads instead of moving 3 KB per row through HBM.
#include <cstdlib>
#include <type_traits>
#include "mmee_common.h"

namespace mmee {

#ifndef MMEE_GEMM_BK
#define MMEE_GEMM_BK 32
#endif
constexpr int BM = 128, BN = 128, BK = MMEE_GEMM_BK, LDS_STRIDE = BK + 4;
constexpr int STAGE_FLOATS = (BM + BN) * LDS_STRIDE;
constexpr int LD_TPR = BK / 4;              // threads per staged row (one float4 each)
constexpr int LD_RPP = 256 / LD_TPR;        // rows per staging pass
constexpr int LD_NP = BM / LD_RPP;          // passes per operand
constexpr int GEMM_WGS = (BK == 16) ? 3 : 2;   // workgroups per CU the LDS / register budget is sized for
constexpr size_t EPI_STAGE_FLOATS = 4 * 32 * 64;   // epilogue transpose: 4 waves x 32 rows x 64 cols

constexpr size_t LOOP_LDS_FLOATS = 2 * (size_t)STAGE_FLOATS;
constexpr size_t MAIN_LDS_FLOATS = LOOP_LDS_FLOATS > EPI_STAGE_FLOATS ? LOOP_LDS_FLOATS : EPI_STAGE_FLOATS;

// + 16 bytes at the end of the dynamic region for the work-queue slot (a static __shared__ object would shift the
// dynamic base off its 16-byte alignment, cdna_hip_programming.md Guideline 17)
size_t gemm_f32_lds_bytes() { return MAIN_LDS_FLOATS * sizeof(float) + 16; }

// Epilogue shared by the GEMM kernels.  C layout of a 32x32 MFMA tile: col = lane & 31, row = (reg&3) + 8*(reg>>2) +
// 4*(lane>>5), i.e. a lane owns one COLUMN — storing from registers is 64 scalar 4-byte store instructions per wave, and
// the store-issue tail then costs 4-8 stages of matrix-pipe time per tile (in-kernel stamps: 17k-33k cycles).  The staging
// buffers are dead after the k-loop's last barrier, so each wave transposes its 64x64 sub-tile, 32 rows at a time, through
// its own 8 KB of LDS (ds_write_b32: 32 consecutive floats per half-wave, conflict-free) and leaves with whole rows:
// ds_read_b128 + one global_store_dwordx4 per 4 rows x 256 B (16 stores per wave, fully coalesced); bias / layer scale /
// residual are read as float4 on the same row-contiguous layout.
template <int EPI>
__device__ __forceinline__ void gemm_store_tile(const GemmArgs& g, float* smem, f32x16 (&acc)[2][2], int m0, int n0, int M,
                                                int wave, int lane) {
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    {
    float* stg = smem + wave * (32 * 64);           // 8 KB per wave, one 32-row half of its sub-tile at a time
    const int c4 = (lane & 15) * 4;                 // 4 consecutive columns of the wave's 64
    const int col = n0 + wc * 64 + c4;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + col);
    const float sc = (col < g.scale_cols) ? g.scale : 1.0f;      // scale_cols is a multiple of 128
    f32x4 lam = f32x4{1.f, 1.f, 1.f, 1.f};
    if (g.col_scale) lam = *reinterpret_cast<const f32x4*>(g.col_scale + col);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * hh;
                stg[r * 64 + ni * 32 + l31] = acc[mi][ni][e];
            }
        const int rbase = m0 + wr * 64 + mi * 32 + (lane >> 4);
#pragma unroll 4
        for (int j = 0; j < 8; ++j) {
            const int rl = (lane >> 4) + 4 * j;
            const int row = rbase + 4 * j;
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * 64 + c4);
            if (row < M) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float x = (v[t] + bv[t]) * sc;
                    if (EPI == EPI_GELU) x = x * 0.5f * (1.0f + fast_erff(x * 0.70710678118654752440f));
                    if (EPI == EPI_TANH) x = tanhf(x);
                    v[t] = x * lam[t];
                }
                if (EPI == EPI_RESID) {
                    const int rs = g.resid_row_src ? g.resid_row_src[row] : row;
                    v += *reinterpret_cast<const f32x4*>(g.resid + (size_t)rs * g.ldr + col);
                }
                *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + col) = v;
            }
        }
    }
}
}

#define STAMP(var)                                                        \
    if (STAMPS) {                                                         \
        __builtin_amdgcn_sched_barrier(0);                                \
        var = __builtin_amdgcn_s_memtime();                               \
        __builtin_amdgcn_s_waitcnt(0xC07F);                               \
        __builtin_amdgcn_sched_barrier(0);                                \
    }

template <int EPI, int AMODE, bool STAMPS = false>
__global__ __launch_bounds__(256, GEMM_WGS) void gemm_f32_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = g.m_ptr ? *g.m_ptr : g.m_static;
    const int tiles_m = (M + BM - 1) / BM;
    const int tiles_n = g.N / BN;
    const int n_tiles = tiles_m * tiles_n;
    const int nk = g.K / BK;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    const int ld_row = tid / LD_TPR;             // row inside a staging pass
    const int ld_c4 = (tid % LD_TPR) * 4;        // float column inside the BK slab

    // The two waves that share a SIMD (one from each resident workgroup) otherwise alternate MFMAs fairly, finish their
    // 64-MFMA bursts together and then sit in their load-issue / ds_write / barrier phases together, leaving the matrix
    // pipe idle ~10 % of the time.  A static priority for the wave in the odd hardware wave slot de-synchronises them: it
    // owns the pipe while it computes and the other wave fills every gap; the work queue absorbs the speed difference.
    if (g.prio_mode) {
        const unsigned wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 15u;   // HW_REG_HW_ID.WAVE_ID
        if ((g.prio_mode == 1 && (wave_slot & 1u)) || (g.prio_mode == 2 && (blockIdx.x >= gridDim.x / 2)))
            __builtin_amdgcn_s_setprio(1);
    }
    unsigned long long clk0 = 0, rt0 = 0;
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
    unsigned long long acc_issue = 0, acc_comp = 0, acc_store = 0, acc_bar = 0, acc_pro = 0, acc_epi = 0;
    if (g.clk_probe) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    // Tiles are handed out by atomic work queues (zeroed by the host before the launch) instead of a static grid stride:
    // workgroups run at slightly different speeds, and with ~87 tiles each a static split left the average workgroup
    // idle for 6.7 % of the kernel waiting for the slowest one.
    // The queues are XCD-local (each XCD has its own 4 MB L2): m-panels are dealt to the 8 XCDs in groups of GM = 8, and
    // inside a queue the order is group -> n -> m, so the 64 workgroups of an XCD work on 8 A panels x 8 W panels at a
    // time and walk along n with the A panels staying in L2 — instead of every A panel being fetched by every XCD
    // (measured 9.4 GB fetched per FFN-up launch against 0.5 GB of A).  A workgroup whose queue is empty steals from the
    // next XCD's queue, so placement only affects speed, never coverage: every tile index of every queue is popped once.
    constexpr int GM = 8;
    int* q_slot = reinterpret_cast<int*>(smem + MAIN_LDS_FLOATS);
    const int n_groups = (tiles_m + GM - 1) / GM;
    const int my_xcd = g.tile_counter ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u) : 0;   // HW_REG_XCC_ID
    int q_try = 0;                                  // queues tried since the last successful pop (uniform over the WG)
    int tile = blockIdx.x;
    for (;; tile += gridDim.x) {
        int tm, tn;
        if (g.tile_counter) {
            bool got = false;
            while (q_try < 8) {
                const int q = (my_xcd + q_try) & 7;
                if (tid == 0) *q_slot = atomicAdd(g.tile_counter + 16 * q, 1);     // one counter per 64-byte line
                __syncthreads();
                const int j = *q_slot;
                __syncthreads();
                // queue q owns groups q, q+8, q+16, ...; each group is GM m-panels x tiles_n n-tiles (m fastest)
                const int per_group = GM * tiles_n;
                const int gl = j / per_group, r = j - gl * per_group;
                const int grp = q + 8 * gl;
                if (grp < n_groups) {
                    tn = r / GM;
                    tm = grp * GM + (r - tn * GM);
                    if (tm < tiles_m) { got = true; break; }
                    continue;                       // padding slot of the last (partial) group: pop again
                }
                ++q_try;                            // this queue is exhausted for good: move on to the next XCD's
            }
            if (!got) break;
        } else {
            if (tile >= n_tiles) break;
            tm = tile / tiles_n;
            tn = tile - tm * tiles_n;
        }
        const int m0 = tm * BM, n0 = tn * BN;

        // ---- per-thread source pointers of the 4 A rows and 4 W rows this thread stages -------------------------
        // Rows past M are clamped to the last valid row (their results are never stored): unconditional loads keep
        // the staging code free of exec-mask branches and of the conservative vmcnt waits hipcc puts around them.
        const float* a_ptr[LD_NP];
        const float* w_ptr[LD_NP];
#pragma unroll
        for (int i = 0; i < LD_NP; ++i) {
            int r = m0 + ld_row + LD_RPP * i;
            r = r < M ? r : M - 1;
            if (AMODE == AMODE_ROWS) {
                const int src = g.row_src ? g.row_src[r] : r;
                a_ptr[i] = g.A + (size_t)src * g.lda + ld_c4;
            } else {
                const int np = g.G * g.G;
                const int b = r / np, p = r - b * np;
                const int py = p / g.G, px = p - py * g.G;
                // k = ld_c4 + k0 with k0 a multiple of 32: c = k / P^2, ky = (k % P^2) / P, kx = k % P; kx is fixed
                a_ptr[i] = g.pix + (size_t)b * g.C_in * g.R * g.R + (size_t)(py * g.P) * g.R + px * g.P + (ld_c4 % g.P);
            }
            w_ptr[i] = g.W + (size_t)(n0 + ld_row + LD_RPP * i) * g.K + ld_c4;
        }

        f32x4 ra[LD_NP], rw[LD_NP];
        auto load_stage = [&](int k0) {
            size_t a_off;
            if (AMODE == AMODE_ROWS) {
                a_off = (size_t)k0;
            } else {
                const int k = k0 + ld_c4;
                const int pp = g.P * g.P;
                const int c = k / pp, rem = k - c * pp;
                const int ky = rem / g.P;
                a_off = (size_t)c * g.R * g.R + (size_t)ky * g.R;
            }
#pragma unroll
            for (int i = 0; i < LD_NP; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + a_off);
#pragma unroll
            for (int i = 0; i < LD_NP; ++i) rw[i] = *reinterpret_cast<const f32x4*>(w_ptr[i] + k0);
        };
        auto store_stage = [&](int buf) {
            float* As = smem + buf * STAGE_FLOATS;
            float* Ws = As + BM * LDS_STRIDE;
#pragma unroll
            for (int i = 0; i < LD_NP; ++i) {
                *reinterpret_cast<f32x4*>(As + (ld_row + LD_RPP * i) * LDS_STRIDE + ld_c4) = ra[i];
                *reinterpret_cast<f32x4*>(Ws + (ld_row + LD_RPP * i) * LDS_STRIDE + ld_c4) = rw[i];
            }
        };

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        STAMP(ts0);
        load_stage(0);
        store_stage(0);
        __syncthreads();
        STAMP(ts1);
        if (STAMPS) acc_pro += ts1 - ts0;

        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            STAMP(ts0);
            if (more && !g.dbg_noload) load_stage((kt + 1) * BK);
            STAMP(ts1);
            const float* As = smem + (kt & 1) * STAGE_FLOATS;
            const float* Ws = As + BM * LDS_STRIDE;
            const float* a_base = As + (wr * 64 + l31) * LDS_STRIDE + 4 * hh;
            const float* w_base = Ws + (wc * 64 + l31) * LDS_STRIDE + 4 * hh;
#pragma unroll
            for (int gk = 0; gk < BK / 8; ++gk) {
                f32x4 af[2], wf[2];
                af[0] = *reinterpret_cast<const f32x4*>(a_base + 8 * gk);
                af[1] = *reinterpret_cast<const f32x4*>(a_base + 32 * LDS_STRIDE + 8 * gk);
                wf[0] = *reinterpret_cast<const f32x4*>(w_base + 8 * gk);
                wf[1] = *reinterpret_cast<const f32x4*>(w_base + 32 * LDS_STRIDE + 8 * gk);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][c], wf[0][c], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][c], wf[1][c], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][c], wf[0][c], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][c], wf[1][c], acc[1][1], 0, 0, 0);
                }
            }
            STAMP(ts2);
            if (more) store_stage((kt + 1) & 1);
            STAMP(ts3);
            __syncthreads();
            STAMP(ts4);
            if (STAMPS) { acc_issue += ts1 - ts0; acc_comp += ts2 - ts1; acc_store += ts3 - ts2; acc_bar += ts4 - ts3; }
        }
        STAMP(ts0);

        gemm_store_tile<EPI>(g, smem, acc, m0, n0, M, wave, lane);
        __syncthreads();      // the next tile's prologue overwrites the staging area
        STAMP(ts1);
        if (STAMPS) acc_epi += ts1 - ts0;
    }
    if (g.clk_probe && threadIdx.x == 0) {      // diagnostic only: shader clock = d(memtime) / d(memrealtime) * 100 MHz
        const int st = STAMPS ? 8 : 2;
        g.clk_probe[st * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
        g.clk_probe[st * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
        if (STAMPS) {
            g.clk_probe[8 * blockIdx.x + 2] = acc_issue;
            g.clk_probe[8 * blockIdx.x + 3] = acc_comp;
            g.clk_probe[8 * blockIdx.x + 4] = acc_store;
            g.clk_probe[8 * blockIdx.x + 5] = acc_bar;
            g.clk_probe[8 * blockIdx.x + 6] = acc_pro;
            g.clk_probe[8 * blockIdx.x + 7] = acc_epi;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant: global -> LDS directly (global_load_lds_dwordx4), no staging registers and no ds_write pass.
// The per-stage non-MFMA phase of a wave (load issue + wait + ds_write + barrier, ~930 cycles in the register-staged
// kernel) shrinks to "issue 8 DMA pieces + barrier", so the window in which the two waves of a SIMD can both be outside
// their MFMA bursts shrinks with it.  A DMA piece writes 64 lanes x 16 B = 1 KiB LINEARLY (8 rows of the 128-byte BK slab),
// so the rows cannot be padded; bank conflicts of the ds_read_b128 fragment reads are avoided by an XOR swizzle applied on
// BOTH sides: the lane that fills physical chunk p of row r fetches logical chunk p ^ ((r >> 1) & 7) from global memory,
// and the fragment read of logical chunk q of row r reads physical chunk q ^ ((r >> 1) & 7) (16 lanes x 16 B then cover
// all 64 banks exactly once).
// ---------------------------------------------------------------------------------------------------------------
constexpr int DMA_STAGE_FLOATS = (BM + BN) * 32;

template <int EPI, int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_f32_dma_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = g.m_ptr ? *g.m_ptr : g.m_static;
    const int tiles_m = (M + BM - 1) / BM;
    const int tiles_n = g.N / BN;
    const int n_tiles = tiles_m * tiles_n;
    const int nk = g.K / 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR: LDS-DMA bases stay scalar
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    if (g.prio_mode) {
        const unsigned wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 15u;   // HW_REG_HW_ID.WAVE_ID
        if ((g.prio_mode == 1 && (wave_slot & 1u)) || (g.prio_mode == 2 && (blockIdx.x >= gridDim.x / 2)))
            __builtin_amdgcn_s_setprio(1);
    }
    if (g.dbg_noload >> 8) {   // diagnostic: start the workgroups in odd wave slots late, so the two workgroups of a CU run in anti-phase
        const unsigned wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 15u;
        if (wave_slot & 1u)
            for (int i = 0; i < (g.dbg_noload >> 8); ++i) __builtin_amdgcn_s_sleep(127);
    }
    unsigned long long clk0 = 0, rt0 = 0;
    if (g.clk_probe) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int GM = 8;
    int* q_slot = reinterpret_cast<int*>(smem + MAIN_LDS_FLOATS);
    const int n_groups = (tiles_m + GM - 1) / GM;
    const int my_xcd = g.tile_counter ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u) : 0;   // HW_REG_XCC_ID
    int q_try = 0;
    int tile = blockIdx.x;
    // this lane's slot inside a DMA piece: row (lane >> 3) of the piece's 8 rows, physical chunk (lane & 7)
    const int p_row = lane >> 3, p_chunk = lane & 7;
    for (;; tile += gridDim.x) {
        int tm, tn;
        if (g.tile_counter) {
            bool got = false;
            while (q_try < 8) {
                const int q = (my_xcd + q_try) & 7;
                if (tid == 0) *q_slot = atomicAdd(g.tile_counter + 16 * q, 1);
                __syncthreads();
                const int j = *q_slot;
                __syncthreads();
                const int per_group = GM * tiles_n;
                const int gl = j / per_group, r = j - gl * per_group;
                const int grp = q + 8 * gl;
                if (grp < n_groups) {
                    tn = r / GM;
                    tm = grp * GM + (r - tn * GM);
                    if (tm < tiles_m) { got = true; break; }
                    continue;
                }
                ++q_try;
            }
            if (!got) break;
        } else {
            if (tile >= n_tiles) break;
            tm = tile / tiles_n;
            tn = tile - tm * tiles_n;
        }
        const int m0 = tm * BM, n0 = tn * BN;

        // each wave issues 4 A pieces and 4 W pieces per stage: piece i = wave + 4*j covers tile rows 8i .. 8i+7
        const float* a_src[4];
        const float* w_src[4];
        int im_kx = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rl = 8 * (wave + 4 * j) + p_row;                 // row inside the tile
            const int lc = p_chunk ^ ((rl >> 1) & 7);                  // logical 16-byte chunk this lane fetches
            int r = m0 + rl;
            r = r < M ? r : M - 1;
            if (AMODE == AMODE_ROWS) {
                const int src = g.row_src ? g.row_src[r] : r;
                a_src[j] = g.A + (size_t)src * g.lda + 4 * lc;
            } else {
                const int np = g.G * g.G;
                const int b = r / np, p = r - b * np;
                const int py = p / g.G, px = p - py * g.G;
                a_src[j] = g.pix + (size_t)b * g.C_in * g.R * g.R + (size_t)(py * g.P) * g.R + px * g.P;
                im_kx = 4 * lc;                                        // k offset of this lane's chunk inside the 32-wide slab
            }
            w_src[j] = g.W + (size_t)(n0 + rl) * g.K + 4 * lc;
        }
        auto issue_a = [&](int k0, int buf, int j) {      // A piece number (wave + 4j) of one stage
            float* As = smem + buf * DMA_STAGE_FLOATS;
            const int piece = wave + 4 * j;
            const float* ap;
            if (AMODE == AMODE_ROWS) {
                ap = a_src[j] + k0;
            } else {
                const int rl = 8 * piece + p_row;
                const int k = k0 + 4 * (p_chunk ^ ((rl >> 1) & 7));
                const int pp = g.P * g.P;
                const int c = k / pp, rem = k - c * pp;
                const int ky = rem / g.P, kx = rem - ky * g.P;
                ap = a_src[j] + (size_t)c * g.R * g.R + (size_t)ky * g.R + kx;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ap,
                                             (__attribute__((address_space(3))) void*)(As + piece * 256), 16, 0, 0);
        };
        auto issue_w = [&](int k0, int buf, int j) {      // W piece number (wave + 4j)
            float* Ws = smem + buf * DMA_STAGE_FLOATS + BM * 32;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[j] + k0),
                                             (__attribute__((address_space(3))) void*)(Ws + (wave + 4 * j) * 256), 16, 0, 0);
        };
        (void)im_kx;

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // LDS byte addresses of this lane's fragment chunks for the 4 k-groups of a stage (stage 0; toggled per stage)
        const unsigned sw = (unsigned)((l31 >> 1) & 7);
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
        unsigned a_addr[4], w_addr[4];
#pragma unroll
        for (int gk = 0; gk < 4; ++gk) {
            const unsigned pc = 16u * ((unsigned)(2 * gk + hh) ^ sw);
            a_addr[gk] = lds0 + (unsigned)(wr * 64 + l31) * 128u + pc;
            w_addr[gk] = lds0 + (unsigned)(BM * 128) + (unsigned)(wc * 64 + l31) * 128u + pc;
        }
        // two fragment sets, explicit ds_read_b128 + counted lgkmcnt: the reads of k-group gk+1 are in flight while the 16
        // MFMAs of k-group gk issue (left to itself the scheduler sinks the reads behind the MFMAs to save 16 VGPRs and
        // then waits lgkmcnt(0) on every group).  The wait statement lists the fragments it retires as in/out operands so
        // the MFMAs that consume them cannot be hoisted above it.
        f32x4 af[2][2], wf[2][2];
#define MMEE_READ_FRAGS(gk, set)                                                                                         \
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\t"                                       \
                     "ds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:4096"                                            \
                     : "=&v"(af[set][0]), "=&v"(af[set][1]), "=&v"(wf[set][0]), "=&v"(wf[set][1])                        \
                     : "v"(a_addr[gk]), "v"(w_addr[gk]) : "memory")
#define MMEE_WAIT_FRAGS(n, set)                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(af[set][0]), "+v"(af[set][1]), "+v"(wf[set][0]), "+v"(wf[set][1]))
#define MMEE_MFMA8(set, c0)                                                                                             \
        _Pragma("unroll") for (int c = c0; c < c0 + 2; ++c) {                                                            \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][0][c], wf[set][0][c], acc[0][0], 0, 0, 0);          \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][0][c], wf[set][1][c], acc[0][1], 0, 0, 0);          \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][1][c], wf[set][0][c], acc[1][0], 0, 0, 0);          \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][1][c], wf[set][1][c], acc[1][1], 0, 0, 0);          \
        }
#define MMEE_MFMA16(set)                                                                                                 \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                  \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][0][c], wf[set][0][c], acc[0][0], 0, 0, 0);          \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][0][c], wf[set][1][c], acc[0][1], 0, 0, 0);          \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][1][c], wf[set][0][c], acc[1][0], 0, 0, 0);          \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][1][c], wf[set][1][c], acc[1][1], 0, 0, 0);          \
        }
        // Software pipeline over the K stages (2 LDS buffers).  The per-stage barrier sits between k-groups 2 and 3 of the
        // PREVIOUS stage: at that point this wave's fragment reads of stage kt-1 have all retired (so after the barrier
        // nobody reads buffer (kt-1)&1 any more and the DMA of stage kt+1 may overwrite it) and its DMA pieces of stage kt
        // have landed (so after the barrier stage kt is readable).  The 16 MFMAs of k-group 3 are still queued behind the
        // barrier, which covers the barrier skew, the DMA issue and the LDS latency of the first fragments of stage kt.
        // Steady state of one trip:  wait | barrier | DMA kt+1 (spread) | read (kt,0) | MFMA (kt-1,3) | read (kt,1) |
        // MFMA (kt,0) | read (kt,2) | MFMA (kt,1) | read (kt,3) | MFMA (kt,2).
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            issue_a(0, 0, j);
            issue_w(0, 0, j);
        }
        // One DMA piece per 8 MFMAs: a global_load_lds holds the wave's issue for roughly one MFMA's duration, so a single
        // piece hides in the shadow of the MFMA issued just before it, while two or more back to back do not.
#define MMEE_PIN() __builtin_amdgcn_sched_barrier(0)
#define MMEE_GROUP(set, j, HAVE)                                                                                         \
        if (HAVE) { MMEE_MFMA8(set, 0) }                                                                                 \
        MMEE_PIN();                                                                                                      \
        if (more) issue_a(k1, b1, j);                                                                                    \
        MMEE_PIN();                                                                                                      \
        if (HAVE) { MMEE_MFMA8(set, 2) }                                                                                 \
        MMEE_PIN();                                                                                                      \
        if (more) issue_w(k1, b1, j);                                                                                    \
        MMEE_PIN();
        auto stage = [&](int kt, auto first) {
            constexpr bool kFirst = decltype(first)::value;
            if constexpr (!kFirst) { MMEE_WAIT_FRAGS(0, 1); }          // (kt-1, 3) retired: my reads of the old buffer are over
            if (!(g.dbg_noload & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of stage kt have landed
            if (!(g.dbg_noload & 2)) __builtin_amdgcn_s_barrier();
            const bool more = kt + 1 < nk && !(g.dbg_noload & 1);
            const int k1 = (kt + 1) * 32, b1 = (kt + 1) & 1;
            MMEE_READ_FRAGS(0, 0);
            MMEE_GROUP(1, 0, !kFirst)
            MMEE_READ_FRAGS(1, 1);
            MMEE_WAIT_FRAGS(4, 0);
            MMEE_GROUP(0, 1, true)
            MMEE_READ_FRAGS(2, 0);
            MMEE_WAIT_FRAGS(4, 1);
            MMEE_GROUP(1, 2, true)
            MMEE_READ_FRAGS(3, 1);
            MMEE_WAIT_FRAGS(4, 0);
            MMEE_GROUP(0, 3, true)
            const unsigned flip = (kt & 1) ? (unsigned)(-(int)(DMA_STAGE_FLOATS * 4)) : (unsigned)(DMA_STAGE_FLOATS * 4);
#pragma unroll
            for (int gk = 0; gk < 4; ++gk) {
                a_addr[gk] += flip;
                w_addr[gk] += flip;
            }
        };
        stage(0, std::true_type{});
        for (int kt = 1; kt < nk; ++kt) stage(kt, std::false_type{});
        MMEE_WAIT_FRAGS(0, 1);
        MMEE_MFMA16(1)
#undef MMEE_GROUP
#undef MMEE_PIN
#undef MMEE_MFMA8
#undef MMEE_READ_FRAGS
#undef MMEE_WAIT_FRAGS
#undef MMEE_MFMA16
        __syncthreads();                                               // all fragment reads done before the staging reuse
        gemm_store_tile<EPI>(g, smem, acc, m0, n0, M, wave, lane);
        __syncthreads();
    }
    if (g.clk_probe && threadIdx.x == 0) {      // diagnostic only: shader clock = d(memtime) / d(memrealtime) * 100 MHz
        g.clk_probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
        g.clk_probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

template <int EPI, int AMODE>
static void launch_one_dma(const GemmArgs& a, int grid, hipStream_t s) {
    const size_t lds = gemm_f32_lds_bytes();
    (void)ensure_dynamic_lds<&gemm_f32_dma_kernel<EPI, AMODE>>("gemm_f32_dma_kernel", (int)lds);
    hipLaunchKernelGGL((gemm_f32_dma_kernel<EPI, AMODE>), dim3(grid), dim3(256), lds, s, a);
}

template <int EPI, int AMODE>
static void launch_one(const GemmArgs& a, int grid, hipStream_t s) {
    const size_t lds = gemm_f32_lds_bytes();
    (void)ensure_dynamic_lds<&gemm_f32_kernel<EPI, AMODE>>("gemm_f32_kernel", (int)lds);
    hipLaunchKernelGGL((gemm_f32_kernel<EPI, AMODE>), dim3(grid), dim3(256), lds, s, a);
}

static int g_gemm_wgs_per_cu = 0;   // 0 = the tile configuration's own choice
void set_gemm_wgs_per_cu(int n) { g_gemm_wgs_per_cu = n < 0 ? 0 : n; }

void launch_gemm_f32_stamped(const GemmArgs& a, int epi, int grid, hipStream_t s) {
    const size_t lds = gemm_f32_lds_bytes();
    if (epi == EPI_GELU) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<EPI_GELU, AMODE_ROWS, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm_f32_kernel<EPI_GELU, AMODE_ROWS, true>), dim3(grid), dim3(256), lds, s, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<EPI_BIAS, AMODE_ROWS, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm_f32_kernel<EPI_BIAS, AMODE_ROWS, true>), dim3(grid), dim3(256), lds, s, a);
    }
}

void launch_gemm_f32(const GemmArgs& a, int epi, int amode, int max_m, int num_cus, hipStream_t s) {
    const int tiles = ((max_m + BM - 1) / BM) * (a.N / BN);
    int grid = (g_gemm_wgs_per_cu > 0 ? g_gemm_wgs_per_cu : GEMM_WGS) * num_cus;
    if (tiles < grid) grid = tiles;
    if (grid < 1) grid = 1;
    static const int env_dma = diag_env_int("MMEE_GEMM_DMA", 1);   // diagnostic library only: MMEE_GEMM_DMA=0 = register-staged kernel (A/B switch)
    if (a.use_dma == 1 || (a.use_dma == 0 && env_dma)) {
        if (amode == AMODE_IM2COL) { launch_one_dma<EPI_BIAS, AMODE_IM2COL>(a, grid, s); return; }
        switch (epi) {
            case EPI_BIAS: launch_one_dma<EPI_BIAS, AMODE_ROWS>(a, grid, s); break;
            case EPI_GELU: launch_one_dma<EPI_GELU, AMODE_ROWS>(a, grid, s); break;
            case EPI_RESID: launch_one_dma<EPI_RESID, AMODE_ROWS>(a, grid, s); break;
            default: launch_one_dma<EPI_TANH, AMODE_ROWS>(a, grid, s); break;
        }
        return;
    }
    if (amode == AMODE_IM2COL) {
        launch_one<EPI_BIAS, AMODE_IM2COL>(a, grid, s);
        return;
    }
    switch (epi) {
        case EPI_BIAS: launch_one<EPI_BIAS, AMODE_ROWS>(a, grid, s); break;
        case EPI_GELU: launch_one<EPI_GELU, AMODE_ROWS>(a, grid, s); break;
        case EPI_RESID: launch_one<EPI_RESID, AMODE_ROWS>(a, grid, s); break;
        default: launch_one<EPI_TANH, AMODE_ROWS>(a, grid, s); break;
    }
}

}  // namespace mmee
