// `output_attentions` and `head_mask` of the reference signature (EE/models/LayoutLMv3.py:382-385, 631-641 -> LayoutLMv3EncoderEE.forward
// :156-157, 185, 219-220 -> LayoutLMv3SelfAttention.forward of transformers 4.26: `attention_probs = softmax(...)`, `attention_probs =
// attention_probs * head_mask`, `context = attention_probs @ value`, the (masked) probabilities returned per layer).
//
// Neither belongs to the hot path: the evaluation loop never passes them (EE/utils.py:179), and the fused attention kernels never
// materialise an S x S map.  They are served by two SIDE kernels that only run when the caller asked -- in the dump-all / whole-layers /
// dense-rows mode `model.forward` runs anyway -- so the kernels of the path stay untouched:
//   * attention_probs_kernel: one workgroup per (query row, head, document) recomputes the row of probabilities from the layer's Q | K rows
//     (f32, or split-f16 planes), the per-head VALUE tables of the f32 attention kernel (bucket LUT o nn.Linear table o 1/sqrt(d), composed at
//     ee_finalize) and the additive key mask, and writes it to (B, heads, S, S) -- 24 MB per document and layer at S = 709, which is why
//     nothing else ever does this;
//   * head_scale_ctx_kernel: context columns of head h times head_mask[l][h] after the fused attention kernel (probs * m @ V == m * (probs @ V)).
#include "mmee_common.h"
#include "mmee_kernels.h"

namespace mmee {

namespace {
constexpr int D = 64;

__device__ __forceinline__ float qkv_elem(const float* qkv, size_t row, int ld, int col, int split, float inv_scale) {
    if (!split) return qkv[row * (size_t)ld + col];
    // split rows: the 4 * ld bytes of a row are 64-byte groups [hi 16 x f16 | lo 16 x f16] (mmee_common.h)
    const char* p = reinterpret_cast<const char*>(qkv) + row * (size_t)ld * 4 + (size_t)(col >> 4) * 64 + (size_t)(col & 15) * 2;
    const _Float16 hi = *reinterpret_cast<const _Float16*>(p), lo = *reinterpret_cast<const _Float16*>(p + 32);
    return ((float)hi + (float)lo) * inv_scale;
}

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o, 64);
        v = is_max ? fmaxf(v, w) : v + w;
    }
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}
}  // namespace

__global__ __launch_bounds__(256) void attention_probs_kernel(const float* __restrict__ qkv, int ld, int split, float inv_scale,
                                                              const RowMeta* __restrict__ meta, const int* __restrict__ doc_off,
                                                              const float* __restrict__ t1, const float* __restrict__ tx,
                                                              const float* __restrict__ ty, int n1, int c1, int n2, int c2, int H, int heads,
                                                              int S, const float* __restrict__ head_scale, float* __restrict__ out) {
    __shared__ float q_s[D];
    __shared__ float red[4];
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int off = doc_off[b], len = doc_off[b + 1] - off;
    float* orow = out + (((size_t)b * heads + h) * S + i) * (size_t)S;
    if (i >= len) {                       // (cannot happen in the dense layout this kernel is launched for; keeps the output defined)
        for (int j = threadIdx.x; j < S; j += 256) orow[j] = 0.f;
        return;
    }
    if (threadIdx.x < D) q_s[threadIdx.x] = qkv_elem(qkv, (size_t)(off + i), ld, h * D + threadIdx.x, split, inv_scale);      // Q / sqrt(d) already
    __syncthreads();
    const RowMeta mq = meta[off + i];
    const float* T1 = t1 + (size_t)h * n1;
    const float* TX = tx + (size_t)h * n2;
    const float* TY = ty + (size_t)h * n2;
    constexpr int PER = 5;               // keys per thread: S <= 1280
    float sc[PER];
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const int j = threadIdx.x + 256 * r;
        float v = -3.0e38f;
        if (j < len) {
            float dot = 0.f;
            for (int d = 0; d < D; ++d) dot = fmaf(q_s[d], qkv_elem(qkv, (size_t)(off + j), ld, H + h * D + d, split, inv_scale), dot);
            const RowMeta mk = meta[off + j];
            const float bias = T1[c1 + (mk.pos - mq.pos) / 4] + (TX[c2 + (mk.x0 - mq.x0) / 4] + TY[c2 + (mk.y1 - mq.y1) / 4]);
            v = (dot + bias) + __int_as_float(mk.flags);        // additive mask: 0 or -3e38 (EE/models/LayoutLMv3.py:622-624 adds finfo.min)
        }
        sc[r] = v;
        mx = fmaxf(mx, v);
    }
    mx = block_reduce(mx, red, true);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const int j = threadIdx.x + 256 * r;
        sc[r] = j < len ? expf(sc[r] - mx) : 0.f;               // a masked key: exp(-3e38 - max) == 0 exactly, as the reference's
        sum += sc[r];
    }
    sum = block_reduce(sum, red, false);
    const float m = head_scale ? head_scale[h] : 1.0f;
    const float inv = 1.0f / sum;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const int j = threadIdx.x + 256 * r;
        if (j < S) orow[j] = j < len ? (sc[r] * inv) * m : 0.f;
    }
}

bool attention_probs_supports(int S) { return S >= 1 && S <= 1280; }

void launch_attention_probs(const float* qkv, int ld, int split, float qkv_scale, const RowMeta* meta, const int* doc_off, const float* t1,
                            const float* tx, const float* ty, int n1, int c1, int n2, int c2, int H, int heads, int S, int B,
                            const float* head_scale, float* out, hipStream_t s) {
    hipLaunchKernelGGL(attention_probs_kernel, dim3(S, heads, B), dim3(256), 0, s, qkv, ld, split, split ? 1.0f / qkv_scale : 1.0f, meta, doc_off,
                       t1, tx, ty, n1, c1, n2, c2, H, heads, S, head_scale, out);
}

// ctx[row][64 h .. 64 h + 63] *= head_scale[h]; rows are f32 or split planes (re-split after the multiplication: exact for the 0 / 1 masks
// head pruning studies use, one rounding of the hi + lo sum otherwise)
__global__ __launch_bounds__(256) void head_scale_ctx_kernel(float* __restrict__ ctx, int ld, const int* __restrict__ n_rows_ptr, int H,
                                                             const float* __restrict__ head_scale, int split, float scale,
                                                             int* __restrict__ err_flag) {
    const int n_rows = *n_rows_ptr;
    const int h4 = H / 4;
    const size_t total = (size_t)n_rows * h4;
    float amax = 0.f;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t r = idx / h4;
        const int c = (int)(idx - r * h4) * 4;
        const float m = head_scale[c / D];
        if (split) {
            char* row = reinterpret_cast<char*>(ctx) + r * (size_t)ld * 4;
            f32x4 v = load_split4(row, c, 1.0f / scale);
            v = v * m;
            store_split4(row, c, v, scale, amax);
        } else {
            f32x4* p = reinterpret_cast<f32x4*>(ctx + r * (size_t)ld + c);
            *p = *p * m;
        }
    }
    if (split) split_flag_overflow(amax, err_flag);
}

void launch_head_scale_ctx(float* ctx, int ld, const int* n_rows_ptr, int max_rows, int H, const float* head_scale, int split, float scale,
                           int num_cus, int* err_flag, hipStream_t s) {
    size_t blocks = ((size_t)max_rows * (H / 4) + 255) / 256;
    int grid = (int)(blocks < (size_t)num_cus * 16 ? blocks : (size_t)num_cus * 16);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(head_scale_ctx_kernel, dim3(grid), dim3(256), 0, s, ctx, ld, n_rows_ptr, H, head_scale, split, scale, err_flag);
}

}  // namespace mmee
