// f32-grade GEMM on the f16 matrix cores by operand splitting: C[M,N] = A[M,K] * W[N,K]^T (+ fused epilogue).
//
// Same role as gemm_f32.hip for the four big Linear layers of an encoder layer (QKV HF:243-258, attention output
// HF:299-303, FFN up + GELU HF:485-497, FFN down HF:508-512) in precision mode MMEE_PRECISION_F32_SPLIT.
//
// Why: v_mfma_f32_32x32x2_f32 peaks at 157 TFLOP/s; v_mfma_f32_32x32x16_f16 at 2.5 PFLOP/s.  Every f32 operand is kept
// as two f16 planes hi + lo (22 significant bits, mmee_common.h "Split-f16 operand rows") and the product is formed as
//     a*w  ~=  a_lo*w_hi + a_hi*w_lo + a_hi*w_hi          (a_lo*w_lo ~ 2^-22 relative is dropped)
// with three f16 MFMAs accumulating in f32: 3/16 of the matrix-pipe time of the f32 MFMA for the same MACs.  Measured
// against an f64 reference (tools/split_accuracy.hip, K = 768 / 3072, N(0,1) x N(0,0.02) operands): rms relative error
// 3.2e-7 / 6.1e-7 for this scheme against 5.0e-7 / 9.9e-7 for the f32 MFMA chain — the 16-long dot product inside the
// f16 MFMA is summed with fewer roundings than eight chained 2-long f32 MFMAs, so the split result is the MORE accurate
// of the two.  Both meet the 1e-4 logit tolerance with the same margin; exit indices stay bit-exact on the fixtures.
//
// Design (gfx950):
//   * 128 x 256 output tile per 512-thread workgroup (8 waves as 2 x 4, each a 64x64 sub-tile = 2x2 MFMA tiles),
//     two workgroups per CU (4 waves per SIMD, <= 128 VGPRs): one workgroup's epilogue (GELU is ~35 VALU instructions
//     per element) runs under the other's MFMA stream, and at 4 waves per SIMD the LDS-DMA issue cost of one wave
//     (~25-35 cycles per 1 KiB piece, tools/mfma_peak.hip) is covered by the other three.
//   * K stage = 16 (one MFMA k-step): a row contributes 32 B of hi and 32 B of lo = one 64-byte LDS row
//     [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15]; stage = (128 + 256) rows x 64 B = 24 KiB, 3-deep ring (72 KiB).
//   * global -> LDS by global_load_lds_dwordx4 with an SGPR base and a 32-bit per-lane offset (the cheapest form to
//     issue): a piece is 16 rows x 64 B, 3 pieces per wave per stage; LDS rows are unpadded (the DMA writes linearly),
//     bank conflicts of the ds_read_b128 fragment reads are removed by XOR-ing the 16-byte chunk index with (row>>2)&3
//     on both the DMA source side and the read side.
//   * ring protocol per stage kt:  s_waitcnt vmcnt(3) (my pieces of stage kt landed, stage kt+1 may still fly) |
//     s_barrier (everyone's landed; everyone left stage kt-1) | issue stage kt+2 into the buffer stage kt-1 used |
//     8 ds_read_b128 + 12 MFMA on stage kt.
//   * persistent workgroups pulling tiles from the XCD-local queues of gemm_f32.hip; M read from device memory.
//   * epilogue through LDS as in gemm_f32.hip (whole-row 16-byte stores); the FFN-up epilogue writes its output as
//     split rows directly (the only consumer is the FFN-down GEMM), so H1 never exists in f32.
#include <cstdlib>
#include "mmee_common.h"

namespace mmee {

constexpr int SBM = 128, SBN = 256, SBK = 16, SNST = 3;
constexpr int S_A_BYTES = SBM * 64;                           // 8 KiB
constexpr int S_STAGE_BYTES = (SBM + SBN) * 64;               // 24 KiB
constexpr int S_LOOP_BYTES = SNST * S_STAGE_BYTES;            // 72 KiB
constexpr int S_EPI_BYTES = 8 * 32 * 64 * 4;                  // 64 KiB: 8 waves x 32 rows x 64 f32
static_assert(S_EPI_BYTES <= S_LOOP_BYTES, "epilogue staging must fit in the stage ring");
static size_t gemm_split_lds_bytes() { return S_LOOP_BYTES + 16; }

bool gemm_split_supports(int N, int K) { return N > 0 && K > 0 && N % SBN == 0 && K % SBK == 0; }

__device__ __forceinline__ void dma_piece(unsigned voff, unsigned long long base, unsigned lds_addr) {
    unsigned keep;   // m0 is saved and restored: the compiler does not accept it in a clobber list
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(base)
                 : "memory");
}

template <int EPI, bool OUT_SPLIT>
__device__ __forceinline__ void split_store_tile(const GemmArgs& g, float* smem, f32x16 (&acc)[2][2], int m0, int n0, int M,
                                                 int wave, int lane) {
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, hh = lane >> 5;
    float* stg = smem + wave * (32 * 64);           // 8 KB per wave, one 32-row half of its sub-tile at a time
    const int c4 = (lane & 15) * 4;                 // 4 consecutive columns of the wave's 64
    const int col = n0 + wc * 64 + c4;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + col);
    const float sc = (col < g.scale_cols) ? g.scale : 1.0f;      // scale_cols is a multiple of 256 or >= N... (host checks % 64)
    f32x4 lam = f32x4{1.f, 1.f, 1.f, 1.f};
    if (g.col_scale) lam = *reinterpret_cast<const f32x4*>(g.col_scale + col);
    const float alpha = g.alpha;
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
                    float x = fmaf(v[t], alpha, bv[t]) * sc;
                    if (EPI == EPI_GELU) x = x * 0.5f * (1.0f + fast_erff(x * 0.70710678118654752440f));
                    if (EPI == EPI_TANH) x = tanhf(x);
                    v[t] = x * lam[t];
                }
                if (EPI == EPI_RESID) {
                    const int rs = g.resid_row_src ? g.resid_row_src[row] : row;
                    v += *reinterpret_cast<const f32x4*>(g.resid + (size_t)rs * g.ldr + col);
                }
                if (OUT_SPLIT)
                    store_split4(reinterpret_cast<char*>(g.C) + (size_t)row * g.ldc * 4, g.ldc, col, v, g.out_scale);
                else
                    *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + col) = v;
            }
        }
    }
}

template <int EPI, bool OUT_SPLIT>
__global__ __launch_bounds__(512, 4) void gemm_split_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = g.m_ptr ? *g.m_ptr : g.m_static;
    const int tiles_m = (M + SBM - 1) / SBM;
    const int tiles_n = g.N / SBN;
    const int n_tiles = tiles_m * tiles_n;
    const int nk = g.K / SBK;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, hh = lane >> 5;

    constexpr int GM = 8;
    int* q_slot = reinterpret_cast<int*>(reinterpret_cast<char*>(smem) + S_LOOP_BYTES);
    const int n_groups = (tiles_m + GM - 1) / GM;
    const int my_xcd = g.tile_counter ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u) : 0;   // HW_REG_XCC_ID
    int q_try = 0;
    int tile = blockIdx.x;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    // DMA: lane (p_row, p_chunk) of a 16-row x 64-byte piece fills physical chunk p_chunk of its row with logical chunk
    // p_chunk ^ ((row >> 2) & 3); pieces start at multiples of 16 rows, so the swizzle only depends on p_row.
    const int p_row = lane >> 2;
    const int src_chunk = (lane & 3) ^ ((p_row >> 2) & 3);
    const unsigned chunk_off = src_chunk < 2 ? 16u * src_chunk : 2u * (unsigned)g.K + 16u * (src_chunk - 2);
    // fragment reads: lane (r = l31, h = hh) takes k = 8h..8h+7 of row r: logical chunks h (hi) and 2 + h (lo)
    const unsigned swz = (unsigned)((l31 >> 2) & 3);
    const unsigned a_hi = (unsigned)(wr * 64 + l31) * 64u + 16u * ((unsigned)hh ^ swz);
    const unsigned a_lo = (unsigned)(wr * 64 + l31) * 64u + 16u * ((2u | (unsigned)hh) ^ swz);
    const unsigned w_hi = (unsigned)S_A_BYTES + (unsigned)(wc * 64 + l31) * 64u + 16u * ((unsigned)hh ^ swz);
    const unsigned w_lo = (unsigned)S_A_BYTES + (unsigned)(wc * 64 + l31) * 64u + 16u * ((2u | (unsigned)hh) ^ swz);
    const char* sbytes = reinterpret_cast<const char*>(smem);

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
        const int m0 = __builtin_amdgcn_readfirstlane(tm * SBM), n0 = __builtin_amdgcn_readfirstlane(tn * SBN);

        // per-tile DMA sources: SGPR base + 32-bit lane offset (A rows may be gathered; row_src is increasing)
        const int first = g.row_src ? g.row_src[m0] : m0;
        int ra = m0 + 16 * wave + p_row;
        ra = ra < M ? ra : M - 1;
        const int sa = g.row_src ? g.row_src[ra] : ra;
        const unsigned a_voff = (unsigned)(sa - first) * (unsigned)g.lda * 4u + chunk_off;
        const unsigned w_voff0 = (unsigned)(16 * wave + p_row) * (unsigned)g.K * 4u + chunk_off;
        const unsigned w_voff1 = w_voff0 + 128u * (unsigned)g.K * 4u;
        const unsigned long long a_base_v = (unsigned long long)(size_t)g.A + (unsigned long long)first * (unsigned)g.lda * 4ull;
        const unsigned long long w_base_v = (unsigned long long)(size_t)g.W + (unsigned long long)n0 * (unsigned)g.K * 4ull;
        const unsigned long long a_base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a_base_v >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)(a_base_v & 0xffffffffu));
        const unsigned long long w_base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w_base_v >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)(w_base_v & 0xffffffffu));
        auto issue = [&](int kt, int buf) {
            const unsigned long long koff = (unsigned long long)kt * (2u * SBK);
            const unsigned dst = lds0 + (unsigned)buf * S_STAGE_BYTES + (unsigned)wave * 1024u;
            dma_piece(a_voff, a_base + koff, dst);
            dma_piece(w_voff0, w_base + koff, dst + S_A_BYTES);
            dma_piece(w_voff1, w_base + koff, dst + S_A_BYTES + 8192u);
        };

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        issue(0, 0);
        if (nk > 1) issue(1, 1);
        int buf = 0, buf2 = 2;                       // ring slots of stage kt and stage kt + 2
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 2 < nk) issue(kt + 2, buf2);
            const char* sb = sbytes + buf * S_STAGE_BYTES;
            f16x8 ah[2], al[2], wh[2], wl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(sb + a_hi + i * 2048);
                al[i] = *reinterpret_cast<const f16x8*>(sb + a_lo + i * 2048);
                wh[i] = *reinterpret_cast<const f16x8*>(sb + w_hi + i * 2048);
                wl[i] = *reinterpret_cast<const f16x8*>(sb + w_lo + i * 2048);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
                }
            buf = buf == 2 ? 0 : buf + 1;
            buf2 = buf2 == 2 ? 0 : buf2 + 1;
        }
        __syncthreads();                             // every wave is done with the ring before it becomes the staging area
        split_store_tile<EPI, OUT_SPLIT>(g, smem, acc, m0, n0, M, wave, lane);
        __syncthreads();
    }
}

template <int EPI, bool OUT_SPLIT>
static void launch_split_one(const GemmArgs& a, int grid, hipStream_t s) {
    static bool attr_set = false;
    const size_t lds = gemm_split_lds_bytes();
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<EPI, OUT_SPLIT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_split_kernel<EPI, OUT_SPLIT>), dim3(grid), dim3(512), lds, s, a);
}

void launch_gemm_split(const GemmArgs& a, int epi, int max_m, int num_cus, hipStream_t s) {
    const int tiles = ((max_m + SBM - 1) / SBM) * (a.N / SBN);
    int grid = 2 * num_cus;
    if (grid > tiles) grid = tiles;
    if (grid < 1) grid = 1;
    if (a.out_split) {
        if (epi == EPI_GELU) launch_split_one<EPI_GELU, true>(a, grid, s);
        else launch_split_one<EPI_BIAS, true>(a, grid, s);
        return;
    }
    switch (epi) {
        case EPI_BIAS: launch_split_one<EPI_BIAS, false>(a, grid, s); break;
        case EPI_GELU: launch_split_one<EPI_GELU, false>(a, grid, s); break;
        case EPI_RESID: launch_split_one<EPI_RESID, false>(a, grid, s); break;
        default: launch_split_one<EPI_TANH, false>(a, grid, s); break;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// f32 rows -> split rows, and the |max| reduction that picks a weight tensor's scale
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, char* __restrict__ dst, const int* __restrict__ n_rows_ptr,
                                                         int n_rows_static, int K, float scale) {
    const int n_rows = n_rows_ptr ? *n_rows_ptr : n_rows_static;
    const int k4 = K / 4;
    const size_t total = (size_t)n_rows * k4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / k4;
        const int c = (int)(i - r * k4) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * K + c);
        store_split4(dst + r * (size_t)K * 4, K, c, v, scale);
    }
}

void launch_split_rows(const float* src, void* dst, const int* n_rows_ptr, int n_rows_static, int max_rows, int K, float scale, int num_cus,
                       hipStream_t s) {
    size_t total = (size_t)max_rows * (K / 4);
    size_t blocks = (total + 255) / 256;
    int grid = (int)(blocks < (size_t)num_cus * 16 ? blocks : (size_t)num_cus * 16);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(split_rows_kernel, dim3(grid), dim3(256), 0, s, src, reinterpret_cast<char*>(dst), n_rows_ptr, n_rows_static, K, scale);
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ src, size_t n, float* __restrict__ out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(src[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));   // non-negative floats order as uints
}

void launch_absmax(const float* src, size_t n, float* out_dev, hipStream_t s) {
    (void)hipMemsetAsync(out_dev, 0, sizeof(float), s);
    size_t blocks = (n + 255) / 256;
    int grid = (int)(blocks < 1024 ? blocks : 1024);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, s, src, n, out_dev);
}

}  // namespace mmee
