// fp32 GEMM on the CDNA4 matrix cores: C[M,N] = A[M,K] * W[N,K]^T (+ bias, + fused epilogue).
//
// Replaces every nn.Linear / Conv2d of the encoder path in parity mode:
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
//   * register-staged double buffering (global_load -> VGPR during compute, ds_write after), one barrier per stage.
//   * persistent grid-stride over tiles; M is read from device memory (rows of the still-active documents), so the
//     host never synchronises to size a launch after an exit stage.
//   * optional row gather on A and on the residual: the stream compaction after an exit is fused into the next
//     layer's loads instead of moving 3 KB per row through HBM.
#include "mmee_common.h"

namespace mmee {

constexpr int BM = 128, BN = 128, BK = 32, LDS_STRIDE = 36;
constexpr int STAGE_FLOATS = (BM + BN) * LDS_STRIDE;

size_t gemm_f32_lds_bytes() { return 2 * STAGE_FLOATS * sizeof(float); }

template <int EPI, int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmArgs g) {
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
    const int ld_row = tid >> 3;          // 0..31
    const int ld_c4 = (tid & 7) * 4;      // float column inside the BK slab

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;

        // ---- per-thread source pointers of the 4 A rows and 4 W rows this thread stages -------------------------
        const float* a_ptr[4];
        bool a_ok[4];
        const float* w_ptr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = m0 + ld_row + 32 * i;
            a_ok[i] = r < M;
            if (AMODE == AMODE_ROWS) {
                const int src = a_ok[i] ? (g.row_src ? g.row_src[r] : r) : 0;
                a_ptr[i] = g.A + (size_t)src * g.lda + ld_c4;
            } else {
                const int rr = a_ok[i] ? r : 0;
                const int np = g.G * g.G;
                const int b = rr / np, p = rr - b * np;
                const int py = p / g.G, px = p - py * g.G;
                a_ptr[i] = g.pix + (size_t)b * g.C_in * g.R * g.R + (size_t)(py * g.P) * g.R + px * g.P;
            }
            w_ptr[i] = g.W + (size_t)(n0 + ld_row + 32 * i) * g.K + ld_c4;
        }

        f32x4 ra[4], rw[4];
        auto load_stage = [&](int k0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (AMODE == AMODE_ROWS) {
                    ra[i] = a_ok[i] ? *reinterpret_cast<const f32x4*>(a_ptr[i] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    const int k = k0 + ld_c4;
                    const int pp = g.P * g.P;
                    const int c = k / pp, rem = k - c * pp;
                    const int ky = rem / g.P, kx = rem - ky * g.P;
                    const float* p = a_ptr[i] + (size_t)c * g.R * g.R + ky * g.R + kx;
                    ra[i] = a_ok[i] ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                rw[i] = *reinterpret_cast<const f32x4*>(w_ptr[i] + k0);
            }
        };
        auto store_stage = [&](int buf) {
            float* As = smem + buf * STAGE_FLOATS;
            float* Ws = As + BM * LDS_STRIDE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(As + (ld_row + 32 * i) * LDS_STRIDE + ld_c4) = ra[i];
                *reinterpret_cast<f32x4*>(Ws + (ld_row + 32 * i) * LDS_STRIDE + ld_c4) = rw[i];
            }
        };

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        load_stage(0);
        store_stage(0);
        __syncthreads();

        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if (more) load_stage((kt + 1) * BK);
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
            if (more) store_stage((kt + 1) & 1);
            __syncthreads();
        }

        // ---- epilogue: C layout of a 32x32 MFMA tile is col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int col = n0 + wc * 64 + ni * 32 + l31;
                const float bv = g.bias ? g.bias[col] : 0.f;
                const float sc = (col < g.scale_cols) ? g.scale : 1.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wr * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (row < M) {
                        float v = (acc[mi][ni][e] + bv) * sc;
                        if (EPI == EPI_GELU) v = v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
                        if (EPI == EPI_TANH) v = tanhf(v);
                        if (EPI == EPI_RESID) {
                            const int rs = g.resid_row_src ? g.resid_row_src[row] : row;
                            v += g.resid[(size_t)rs * g.ldr + col];
                        }
                        g.C[(size_t)row * g.ldc + col] = v;
                    }
                }
            }
        }
    }
}

template <int EPI, int AMODE>
static void launch_one(const GemmArgs& a, int grid, hipStream_t s) {
    static bool attr_set = false;
    const size_t lds = gemm_f32_lds_bytes();
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<EPI, AMODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<EPI, AMODE>), dim3(grid), dim3(256), lds, s, a);
}

void launch_gemm_f32(const GemmArgs& a, int epi, int amode, int max_m, int num_cus, hipStream_t s) {
    const int tiles = ((max_m + BM - 1) / BM) * (a.N / BN);
    int grid = 2 * num_cus;
    if (tiles < grid) grid = tiles;
    if (grid < 1) grid = 1;
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
