// Internal declarations shared by the HIP translation units of libmmee_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>

namespace mmee {

// A/B and diagnostic switches exist only in the DIAGNOSTIC library (`make diag` -> libmmee_hip_diag.so, built with -DMMEE_DIAG, loaded by
// tools/ through MMEE_LIB): the release library never reads the environment, so no variable can change what it computes.
inline int diag_env_int(const char* name, int dflt) {
#ifdef MMEE_DIAG
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE).  Round 5: the opt-in is cached PER LAUNCHER
// INSTANTIATION -- a function-local table of one atomic per device, keyed by the kernel symbol as a template argument -- so a launch costs
// one hipGetDevice and one relaxed load (no mutex, no search, no table that can fill up: the diagnostic build multiplies instantiations),
// and a failed opt-in is neither discarded nor left to surface as a generic launch error: it is recorded with the kernel's name and the
// byte count, and ee_forward / the stand-alone entry points report it (take_lds_error).  Round 6: the slot is PER THREAD -- a
// launcher runs on the thread of the entry point that called it, so the failure is reported by THAT call (every entry point that launches
// drains the slot before it returns: launch_status() in capi.hip), never by an unrelated forward of another handle or thread.
struct LdsOptInError {
    std::mutex mu;
    int code = 0;
    char what[192] = {0};
};
inline LdsOptInError& lds_optin_error() {
    static thread_local LdsOptInError e;
    return e;
}
// returns the recorded failure (and clears it), or nullptr
inline const char* take_lds_error(char* buf, size_t cap) {
    LdsOptInError& e = lds_optin_error();
    std::lock_guard<std::mutex> lock(e.mu);
    if (!e.code) return nullptr;
    snprintf(buf, cap, "%s", e.what);
    e.code = 0;
    return buf;
}
template <auto Kernel>
inline hipError_t ensure_dynamic_lds(const char* name, int bytes) {
    constexpr int kMaxDev = 64;
    static std::atomic<int> granted[kMaxDev];      // zero-initialised: bytes this kernel may use on device d
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDev;
    if (known && granted[dev].load(std::memory_order_relaxed) >= bytes) return hipSuccess;
    const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (rc == hipSuccess) {
        if (known) granted[dev].store(bytes, std::memory_order_relaxed);
    } else {
        LdsOptInError& e = lds_optin_error();
        std::lock_guard<std::mutex> lock(e.mu);
        e.code = (int)rc;
        snprintf(e.what, sizeof(e.what), "hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d bytes) failed for %s on device %d: %s", bytes, name,
                 dev, hipGetErrorString(rc));
    }
    return rc;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Per-row metadata of the packed (ragged) sequence layout: what the in-kernel relative-position bias needs, in the form
// the attention kernel consumes directly.
//   pos   : 4 * position index used by the 1D bias (token index j for text rows, patch index v for visual rows;
//           EE/models/LayoutLMv3.py:559-563 builds arange(T) ++ arange(197), NOT the pad-aware embedding positions)
//   x0,y1 : 4 * bbox[...,0] and 4 * bbox[...,3] — the 2D bias buckets x0 and y1 (HF:433-434)
//           (the factor 4 makes the differences byte offsets into the float value tables)
//   flags : float bits of the additive key mask: 0.0f for a valid attention KEY (attention_mask != 0; visual rows
//           always), -3e38f for a masked one (EE/models/LayoutLMv3.py:622-624 adds finfo.min)
struct RowMeta {
    int pos, x0, y1, flags;
};
constexpr float kKeyMasked = -3.0e38f;

// Device-resident description of one exit stage (documents still active when the stage starts).
struct StageCounts {
    int n_docs;                    // active documents
    int n_rows;                    // packed rows of the active documents
    unsigned long long sum_len_sq; // sum over active documents of len^2 (attention FLOP accounting)
};

// ---------------------------------------------------------------------------------------------------------------
// wave helpers (wavefront = 64)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// Sum over the 64 lanes by DPP, returned wave-uniform (through an SGPR): quad xor 1, quad xor 2, row_half_mirror, row_mirror leave the sum of
// its 16-lane row in every lane (each step adds two lanes that already hold equal partial sums of disjoint lane sets, so both partners
// compute the same bits); row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3 put the total into lane 63.  Six v_add_f32_dpp and
// one v_readlane instead of six ds_bpermute round trips with their address arithmetic: PMC showed the LayerNorm and embedding kernels bound by
// VALU issue and LDS-crossbar latency, not by HBM (round 4).  A fixed order, so a fixed result for a given row.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or_zero(float v) {      // lanes in rows outside ROW_MASK read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_uniform(float v) {
    v += dpp_or_zero<0xB1, 0xf>(v);        // quad_perm [1,0,3,2]
    v += dpp_or_zero<0x4E, 0xf>(v);        // quad_perm [2,3,0,1]
    v += dpp_or_zero<0x141, 0xf>(v);       // row_half_mirror
    v += dpp_or_zero<0x140, 0xf>(v);       // row_mirror
    v += dpp_or_zero<0x142, 0xa>(v);       // row_bcast15 -> rows 1, 3
    v += dpp_or_zero<0x143, 0xc>(v);       // row_bcast31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Row-wise LayerNorm on a row held by one wave: lane owns columns c = 4*lane + 256*i + e (i < NV, e < 4).
// Two-pass (mean, then centred variance), biased variance, as torch.nn.LayerNorm.  FULL: H == 256 * NV, no column is ever masked (the
// per-chunk `c < H` tests cost an exec-mask branch and register copies each when H is a run-time value).
template <int NV, bool FULL = false>
__device__ __forceinline__ void wave_layernorm(f32x4 (&x)[NV], int H, int lane, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * lane + 256 * i;
        if (FULL || c < H) s += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
    }
    const float mean = wave_sum_uniform(s) / (float)H;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * lane + 256 * i;
        if (FULL || c < H) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = x[i][e] - mean;
                v += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum_uniform(v) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = 4 * lane + 256 * i;
        if (FULL || c < H) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[i][e] = (x[i][e] - mean) * rstd * g[e] + b[e];
        }
    }
}

constexpr int kMaxNV = 4;  // hidden_size <= 1024, multiple of 4

// erf for the GELU epilogue: erf(x) = sign(x) * (1 - 2^(-q(|x|))), q = degree-8 polynomial fitted to -log2(erfc(t)) on
// [0, 4] with the error weighted by erfc (so it is the ABSOLUTE error of erf that is minimised).  Evaluated in float32:
// max |erf error| 1.03e-7, GELU max abs error 5.0e-7 against float64 (an exactly rounded float32 erf gives 4.5e-7), at
// ~14 instructions instead of the ~50 of the library erff, which was 60 % of the FFN-up epilogue (in-kernel stamps).
__device__ __forceinline__ float fast_erff(float x) {
    const float a = fminf(fabsf(x), 4.0f);
    float q = 5.389074067e-05f;
    q = fmaf(q, a, -5.102792056e-04f);
    q = fmaf(q, a, 1.682463451e-03f);
    q = fmaf(q, a, 4.861298949e-04f);
    q = fmaf(q, a, -2.802465111e-02f);
    q = fmaf(q, a, 1.483877152e-01f);
    q = fmaf(q, a, 9.184340239e-01f);
    q = fmaf(q, a, 1.627907515e+00f);
    q = q * a;
    const float r = 1.0f - __builtin_amdgcn_exp2f(-q);
    return copysignf(r, x);
}

// Two-wide forms of the same arithmetic: on gfx950 the compiler turns f32x2 multiply / add / fma into v_pk_mul_f32 /
// v_pk_add_f32 / v_pk_fma_f32 (two f32 results per lane per instruction at the one-result issue cost), which halves the
// VALU time of the epilogues that hang ~25 such operations on every output element.  Bitwise identical to fast_erff.
__device__ __forceinline__ f32x2 fast_erf2(f32x2 x) {
    const f32x2 a = __builtin_elementwise_min(__builtin_elementwise_abs(x), (f32x2)(4.0f));
    f32x2 q = (f32x2)(5.389074067e-05f);
    q = __builtin_elementwise_fma(q, a, (f32x2)(-5.102792056e-04f));
    q = __builtin_elementwise_fma(q, a, (f32x2)(1.682463451e-03f));
    q = __builtin_elementwise_fma(q, a, (f32x2)(4.861298949e-04f));
    q = __builtin_elementwise_fma(q, a, (f32x2)(-2.802465111e-02f));
    q = __builtin_elementwise_fma(q, a, (f32x2)(1.483877152e-01f));
    q = __builtin_elementwise_fma(q, a, (f32x2)(9.184340239e-01f));
    q = __builtin_elementwise_fma(q, a, (f32x2)(1.627907515e+00f));
    q = q * a;
    f32x2 r;
    r[0] = 1.0f - __builtin_amdgcn_exp2f(-q[0]);
    r[1] = 1.0f - __builtin_amdgcn_exp2f(-q[1]);
    r[0] = copysignf(r[0], x[0]);
    r[1] = copysignf(r[1], x[1]);
    return r;
}
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {      // x * 0.5 * (1 + erf(x / sqrt 2)), same operation order as the scalar form
    const f32x2 e = fast_erf2(x * (f32x2)(0.70710678118654752440f));
    return (x * (f32x2)(0.5f)) * ((f32x2)(1.0f) + e);
}

// Round 4: s * GELU(x) with fewer instructions (a wave64 VALU instruction holds its SIMD for four cycles, packed or not, and the matrix pipe
// idles during the GEMM epilogue, so the epilogue costs its instruction count): with erf(z) = sign(z) (1 - 2^-q(|z|)) and t = 2^-q / 2,
//     GELU(x) = x Phi(x) = x (1 - t) for x >= 0, x t for x < 0   =   fma(|x|, 1/2 - t, x / 2),
// so there is no copysign and no 1 - e; the x / sqrt 2 of the argument is folded into the coefficients (polynomial in |x| directly, clamp at
// 4 sqrt 2) and a power-of-two output scale s into the exponent (s t = 2^(log2 s - 1 - q)): 16 instructions per PAIR against 22.  Max abs
// error against float64 3.7e-7 (the form above: 5.0e-7).  c0 = log2(s) - 1, hs = s / 2.
constexpr float kGeluD1 = -1.151104450e+00f, kGeluD2 = -4.592170119e-01f, kGeluD3 = -5.246298015e-02f, kGeluD4 = 7.006162778e-03f,
                kGeluD5 = -8.593643724e-05f, kGeluD6 = -2.103079314e-04f, kGeluD7 = 4.510273720e-05f, kGeluD8 = -3.368171292e-06f;
constexpr float kGeluClamp = 5.656854249f;       // 4 sqrt 2: the fit of q covers |x| / sqrt 2 in [0, 4]; erfc(4) = 1.5e-8
__device__ __forceinline__ f32x2 gelu_scaled2(const f32x2 x, const float c0, const float hs) {
    f32x2 a;
    a[0] = fminf(fabsf(x[0]), kGeluClamp);
    a[1] = fminf(fabsf(x[1]), kGeluClamp);
    f32x2 p = (f32x2)(kGeluD8);
    p = __builtin_elementwise_fma(p, a, (f32x2)(kGeluD7));
    p = __builtin_elementwise_fma(p, a, (f32x2)(kGeluD6));
    p = __builtin_elementwise_fma(p, a, (f32x2)(kGeluD5));
    p = __builtin_elementwise_fma(p, a, (f32x2)(kGeluD4));
    p = __builtin_elementwise_fma(p, a, (f32x2)(kGeluD3));
    p = __builtin_elementwise_fma(p, a, (f32x2)(kGeluD2));
    p = __builtin_elementwise_fma(p, a, (f32x2)(kGeluD1));
    const f32x2 u = __builtin_elementwise_fma(p, a, (f32x2)(c0));      // log2(s t) = log2 s - 1 - q
    f32x2 t;
    t[0] = __builtin_amdgcn_exp2f(u[0]);
    t[1] = __builtin_amdgcn_exp2f(u[1]);
    const f32x2 h = (f32x2)(hs) - t;
    const f32x2 xh = x * (f32x2)(hs);
    f32x2 r;
    r[0] = fmaf(fabsf(x[0]), h[0], xh[0]);
    r[1] = fmaf(fabsf(x[1]), h[1], xh[1]);
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// Split-f16 operand rows (precision mode MMEE_PREC_F32_SPLIT, gemm_split.hip).  A row of K f32 values x[k] is kept in the same
// 4*K bytes as K/16 groups of 64 bytes: group j = [hi[16j .. 16j+15] (32 B) | lo[16j .. 16j+15] (32 B)] with
// hi[k] = f16(s*x[k]), lo[k] = f16(s*x[k] - hi[k]) and s a power of two chosen per tensor so that the lo plane stays in the
// f16 normal range for every element that matters (|s*x| >= 2^-3) and nothing overflows (|s*x| is clamped to 60000 <
// 65504).  hi + lo carries 22 significant bits; the GEMM forms hi*hi + hi*lo + lo*hi on the f16 matrix cores and multiplies
// by 1/(s_a*s_w) (exact) afterwards.  A group is one MFMA k-step of both planes, so the GEMM's global -> LDS pieces read
// whole contiguous 64-byte runs.
// ---------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float kSplitClamp = 60000.0f;     // < 65504 = largest finite f16
constexpr int kErrSplitOverflow = 16;       // err_flag bit: a value left the range of the split-f16 planes (it was clamped)
// `amax` collects max |scale * x| BEFORE the clamp (one v_max3_f32 per pair): the caller raises kErrSplitOverflow in the
// handle's err_flag when it exceeds kSplitClamp, so that a clamped (= wrong) result never goes unnoticed.
__device__ __forceinline__ void split_f16x4(const f32x4& v, float scale, f16x4& hi, f16x4& lo, float& amax) {
#pragma unroll
    for (int t = 0; t < 4; t += 2) {          // pairs: v_pk_mul_f32, v_cvt_pk_f16_f32, v_pk_add_f32
        f32x2 x = f32x2{v[t], v[t + 1]} * (f32x2)(scale);
        amax = fmaxf(amax, fmaxf(fabsf(x[0]), fabsf(x[1])));
        x = __builtin_elementwise_min(__builtin_elementwise_max(x, (f32x2)(-60000.0f)), (f32x2)(60000.0f));
        const f16x2 h = __builtin_convertvector(x, f16x2);
        const f32x2 hf = __builtin_convertvector(h, f32x2);
        const f16x2 l = __builtin_convertvector(x - hf, f16x2);
        hi[t] = h[0]; hi[t + 1] = h[1];
        lo[t] = l[0]; lo[t + 1] = l[1];
    }
}
// Round 4, the GEMM epilogues' form: (x0, x1) ALREADY scaled -> packed hi pair and packed lo pair in 4 instructions (v_max3_f32,
// v_cvt_pk_f16_f32, and lo = f16(x - float(hi)) by one v_fma_mixlo / mixhi_f16 each: fma(x, 1.0, -hi) in f32 is exact and rounds once) against
// 11 for the form above.  No clamp: |x| > 65504 becomes inf in the hi plane and the forward is REFUSED through kErrSplitOverflow exactly as
// before (amax sees every element); values in (60000, 65504] are now converted correctly instead of clamped.
__device__ __forceinline__ void split_pair(const float x0, const float x1, unsigned& hb, unsigned& lb, float& amax) {
    amax = fmaxf(amax, fmaxf(fabsf(x0), fabsf(x1)));
    const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);
    hb = __builtin_bit_cast(unsigned, h);
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"       // the GEMM epilogue reads `lb` with v_mov_b32_dpp next: a DPP source needs two wait states after the VALU write, and the
                        // hazard recognizer does not see a VALU write inside an asm statement (rows 2-3 of every 16 came out stale without it)
        : "=&v"(lb)
        : "v"(x0), "v"(x1), "v"(hb));
}
// store 4 consecutive columns [col, col+4) (col % 4 == 0) of a split row
__device__ __forceinline__ void split_flag_overflow(float amax, int* err_flag) {
    if (err_flag && !(amax <= kSplitClamp)) atomicOr(err_flag, kErrSplitOverflow);      // NaN counts as overflow
}
__device__ __forceinline__ void store_split4(void* row_base, int col, const f32x4& v, float scale, float& amax) {
    f16x4 hi, lo;
    split_f16x4(v, scale, hi, lo, amax);
    char* p = reinterpret_cast<char*>(row_base) + (col >> 4) * 64 + (col & 15) * 2;
    *reinterpret_cast<f16x4*>(p) = hi;
    *reinterpret_cast<f16x4*>(p + 32) = lo;
}
// The same row piece, stored as ONE contiguous 16 bytes per lane: lanes 4i .. 4i+3 of a wave hold the 16 columns of one 64-byte group
// (col = 4 * lane + const), so a quad-permute hands lane 4i the group's hi[0..7], 4i+1 hi[8..15], 4i+2 lo[0..7], 4i+3 lo[8..15]: a wave's
// store instruction then writes whole contiguous kilobytes instead of 8-byte pieces 32 bytes apart (the GEMM epilogue's form).  All four
// lanes of a quad must be active and col must be 4 * lane + a multiple of 16.
__device__ __forceinline__ void store_split4_quad(void* row_base, int col, const f32x4& v, float scale, float& amax, int lane) {
    // split_pair: 4 instructions per pair (the GEMM epilogues' form; the same hi / lo bits as split_f16x4 for every value that does not
    // overflow, and overflow is flagged either way)
    unsigned h01, l01, h23, l23;
    split_pair(v[0] * scale, v[1] * scale, h01, l01, amax);
    split_pair(v[2] * scale, v[3] * scale, h23, l23, amax);
    const bool take_lo = (lane & 2) != 0;
    const int a0 = __builtin_amdgcn_mov_dpp((int)h01, 0x88, 0xf, 0xf, true), a1 = __builtin_amdgcn_mov_dpp((int)h23, 0x88, 0xf, 0xf, true);
    const int b0 = __builtin_amdgcn_mov_dpp((int)l01, 0x88, 0xf, 0xf, true), b1 = __builtin_amdgcn_mov_dpp((int)l23, 0x88, 0xf, 0xf, true);
    const int c0 = __builtin_amdgcn_mov_dpp((int)h01, 0xDD, 0xf, 0xf, true), c1 = __builtin_amdgcn_mov_dpp((int)h23, 0xDD, 0xf, 0xf, true);
    const int d0 = __builtin_amdgcn_mov_dpp((int)l01, 0xDD, 0xf, 0xf, true), d1 = __builtin_amdgcn_mov_dpp((int)l23, 0xDD, 0xf, 0xf, true);
    int4 piece;
    piece.x = take_lo ? b0 : a0;
    piece.y = take_lo ? b1 : a1;
    piece.z = take_lo ? d0 : c0;
    piece.w = take_lo ? d1 : c1;
    *reinterpret_cast<int4*>(reinterpret_cast<char*>(row_base) + (size_t)(col >> 4) * 64 + (lane & 3) * 16) = piece;
}
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// 4 consecutive columns [col, col+4) of a split row back to f32.  hi + lo is exact in f32 (it carries <= 23 significant bits) and ONE
// v_fma_mix_f32 per value computes it straight from the two f16 halves (fma(hi, 1.0, lo) with f16 sources 0 and 2) instead of two
// conversions and an add; the plane scale is a power of two, so multiplying by its inverse -- or folding it into an fma with the value the
// row is added to, as the GEMM's residual epilogue does -- rounds nothing: the same bits as ((float)hi + (float)lo) * inv_scale.
__device__ __forceinline__ f32x4 load_split4_sum(const void* row_base, int col) {      // hi + lo, still carrying the plane scale
    const char* p = reinterpret_cast<const char*>(row_base) + (col >> 4) * 64 + (col & 15) * 2;
    const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(p), lo = *reinterpret_cast<const u32x2_t*>(p + 32);
    float r0, r1, r2, r3;
    asm("v_fma_mix_f32 %0, %4, 1.0, %6 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_fma_mix_f32 %1, %4, 1.0, %6 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n\t"
        "v_fma_mix_f32 %2, %5, 1.0, %7 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_fma_mix_f32 %3, %5, 1.0, %7 op_sel:[1,0,1] op_sel_hi:[1,0,1]"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(hi[0]), "v"(hi[1]), "v"(lo[0]), "v"(lo[1]));
    return f32x4{r0, r1, r2, r3};
}
__device__ __forceinline__ f32x4 load_split4(const void* row_base, int col, float inv_scale) {
    return load_split4_sum(row_base, col) * inv_scale;
}
// activation scales of the split planes (powers of two; see DESIGN.md "split precision")
constexpr float kSplitScaleX = 16.0f;     // LayerNorm outputs (|x| up to a few tens)
constexpr float kSplitScaleCtx = 64.0f;   // attention context (convex combinations of V rows)
constexpr float kSplitScaleH1 = 16.0f;    // GELU outputs
constexpr float kSplitScaleQKV = 16.0f;   // Q / sqrt(d), K, V

// ---------------------------------------------------------------------------------------------------------------
// launch parameter blocks
// ---------------------------------------------------------------------------------------------------------------
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_TANH = 3 };
enum { AMODE_ROWS = 0, AMODE_IM2COL = 1 };

struct GemmArgs {
    const float* A;
    int lda;
    const int* row_src;          // optional gather of A rows: A row of output row r is A[row_src[r]]
    const float* W;              // [N][K] (torch Linear layout)
    const float* bias;           // [N] or null
    float* C;
    int ldc;
    const float* resid;          // EPI_RESID: C = acc + bias + resid[resid_row_src ? resid_row_src[r] : r]
    int ldr;
    float resid_split_inv;       // != 0: resid points to split-f16 rows (of ldr columns) scaled by 1 / resid_split_inv; 0: f32 rows
    const int* resid_row_src;
    const int* m_ptr;            // device pointer to M (rows or docs of the active stage); null -> m_static
    int m_static;
    int N, K;
    int scale_cols;              // columns [0, scale_cols) are multiplied by `scale` after the bias (Q / sqrt(d))
    float scale;
    const float* col_scale;      // optional per-column factor applied to (acc + bias) before the residual (BEiT lambda_1/2)
    int k_splits;                // CLS-probe launches of the split kernel only (probe = 1): > 1 divides K over that many workgroups per tile; part p is
    size_t split_stride;         //   written (bias epilogue, f32) at C + p * split_stride; the LayerNorm kernel that follows adds the parts in order
    // AMODE_IM2COL: A row (b, p) = patch p of image b, k = (c, ky, kx)   (Conv2d k = s = patch, HF:71-83)
    const float* pix;
    int C_in, R, P, G;
    // split-f16 kernel (gemm_split.hip): A and W point to split rows (same row strides in bytes as f32 rows)
    float alpha;                     // 1 / (s_a * s_w), applied to the accumulator before the bias
    int out_split;                   // 1: C is written as split rows (planes of ldc columns) scaled by out_scale
    float out_scale;
    int use_dma;                     // 0 = default (LDS-DMA kernel unless MMEE_GEMM_DMA=0), 1 = LDS-DMA kernel, 2 = register-staged kernel
    int probe;                       // split kernel: 1 = the CLS-probe instantiation (same code, its own kernel name for the profiler)
    int role_tag;                    // split kernel, residual epilogue: 2 = attention output, 3 = FFN down (same code, own kernel names: the two
                                     // share every template argument, and a profiler could not tell them apart)
    int tile_order;                  // work-queue order inside a group of 8 M-tiles: 0 = M fastest (eight tiles share a W tile back to back),
                                     // 1 = N fastest (the N-tiles of one M-panel follow each other: they share the A panel)
    int prio_mode;                   // 0 none, 1 raise the priority of odd hardware wave slots, 2 of the second half of the grid
    int terms;                       // split kernel: 1 = ONE f16 MFMA term per MAC (hi x hi only: MMEE_FLAG_ONE_TERM, the reported low-precision
                                     //   mode, never a parity path); anything else = the three terms of the split precision
    int dbg_noload;                  // diagnostic: skip the in-loop global loads (results are garbage; timing only)
    int* tile_counter;               // work-queue head (device int, zeroed before the launch); null -> static grid stride
    unsigned long long* clk_probe;   // diagnostic (ee_debug_gemm): per workgroup {shader cycles, 100 MHz ticks}; null in the path
    int* err_flag;                   // split output: kErrSplitOverflow is raised when a value had to be clamped; may be null
};

struct AttnArgs {
    const float* qkv;            // [rows][3H], Q already divided by sqrt(d)
    int ld;
    float* ctx;                  // [rows][H] f32, or split-f16 rows (ctx_split) scaled by ctx_scale
    int ldc;
    const RowMeta* meta;
    const int* doc_off;          // [n_docs + 1] dense row offsets of the active stage
    const StageCounts* counts;
    const float* t1;             // [heads][n1]  rel_pos_bias[h][bucket1(delta)] / sqrt(d), index delta + c1
    const float* tx;             // [heads][n2]  rel_pos_x_bias ...                          index delta + c2
    const float* ty;             // [heads][n2]
    int n1, c1, n2, c2;
    int H, heads, max_len;
    int* item_counter;               // work-queue head (device int, zeroed before the launch); null -> static grid stride
    int ctx_split;
    float ctx_scale;
    float qkv_scale;                 // split-precision attention: scale of the split Q | K | V rows (qkv then points to split rows)
    int* err_flag;                   // ctx_split: kErrSplitOverflow when a context value had to be clamped; may be null
    // attention_idx.hip: per-document pair index (one word per (query, key) pair, built once per forward) + the raw bucket tables
    const unsigned* pair_idx;        // null: no relative-position bias (image-only model)
    size_t idx_doc_stride;           // words per document slab = idx_nb * idx_nb * 1024
    int idx_nb;                      // 32-row blocks per side of a slab
    const int* doc_orig;             // [n_docs] original document id of the active stage's documents (slab index)
    const float *w1, *wx, *wy;       // rel_pos_bias / rel_pos_x_bias / rel_pos_y_bias weights, [heads][bins]
    int bins1, bins2;
    float inv_sqrt_d;
    // probe-first layers (capi.hip): the split attention kernels only
    const int* qkv_doc_off;          // [n_docs] row offsets of the documents' Q | K | V rows when those are still in the previous stage's
                                     // numbering; null: doc_off
    int q_limit;                     // > 0: only queries < q_limit of every document (the CLS probe asks for the first block)
    int terms;                       // attention_idx.hip: 1 = one f16 MFMA term per product (MMEE_FLAG_ONE_TERM); else three
    // round 6, the 16-bit pair index (attention_idx.hip, IDX16): pair_idx then holds two bytes (4 bx, 4 by) per pair, idx_doc_stride counts dwords
    int idx16;
    const unsigned char* lut1;       // [n1] 1-D bucket of delta + c1 (the delta table of a head is w1[head][lut1[.]])
    int n_visual;                    // visual rows at the end of every document (row j of a document: token j, then patch j - n_text)
    const unsigned* keymask;         // [orig doc][idx_nb]: bit j <-> key 32 kb + j masked (past the document, pad row, hole)
    const int* doc_flags;            // [orig doc]: != 0 when a key INSIDE the document is masked (rare: MMEE_FLAG_DENSE_ROWS, holes)
};

// ---------------------------------------------------------------------------------------------------------------
// host-side launchers (implemented next to their kernels)
// ---------------------------------------------------------------------------------------------------------------
void launch_gemm_f32(const GemmArgs& a, int epi, int amode, int max_m, int num_cus, hipStream_t s);
void launch_gemm_split(const GemmArgs& a, int epi, int max_m, int num_cus, hipStream_t s);
bool gemm_split_supports(int N, int K);
// f32 rows -> split rows (n_rows_ptr null -> n_rows_static); src row r is src[row_src ? row_src[r] : r]
void launch_split_rows(const float* src, void* dst, const int* n_rows_ptr, int n_rows_static, int max_rows, int K, float scale,
                       int num_cus, hipStream_t s, int* err_flag = nullptr);
// pixel_values -> split rows of flattened patches (the patch projection's A operand); needs patch_size % 4 == 0
void launch_patch_split(const float* pix, void* dst, int n_docs, int C, int R, int P, float scale, int num_cus, hipStream_t s, int* err_flag);
void launch_absmax(const float* src, size_t n, float* out_dev, hipStream_t s);   // *out_dev = max |src[i]| (out zeroed by the launcher)
void launch_attention_f32(const AttnArgs& a, int max_docs, int num_cus, hipStream_t s);
unsigned long long* attention_pair_stamps();
unsigned long long* attention_idx_stamps();
bool attention_idx_supports(const AttnArgs& a);
bool attention_idx16_fits(int bins1, int bins2, int n1);
void launch_pair_index16(const RowMeta* meta, const int* doc_off, int n_docs, int nb, const unsigned char* lut1, int c1, int n1,
                         const unsigned char* lut2, int c2, int n2, int bins1, unsigned* out16, size_t doc_stride16, unsigned* out_q0,
                         unsigned* keymask, int* doc_flags, int max_len, hipStream_t s);
void launch_attention_idx(const AttnArgs& a, int max_docs, int num_cus, hipStream_t s);
void launch_pair_index(const RowMeta* meta, const int* doc_off, int n_docs, int nb, const unsigned char* lut1, int c1, int n1,
                       const unsigned char* lut2, int c2, int n2, int bins1, unsigned* out, size_t doc_stride, int max_len, hipStream_t s);
bool attention_pair_supports(const AttnArgs& a, int max_rel_pos, int max_rel_2d_pos);
void launch_attention_pair(const AttnArgs& a, int max_docs, int num_cus, int max_rel_pos, int max_rel_2d_pos, int any_masked, hipStream_t s);
size_t gemm_f32_lds_bytes();
void set_gemm_wgs_per_cu(int n);
void launch_gemm_f32_stamped(const GemmArgs& a, int epi, int grid, hipStream_t s);   // diagnostic build with in-kernel stamps
size_t attention_f32_lds_bytes(const AttnArgs& a);

}  // namespace mmee
