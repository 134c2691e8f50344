// Device-side input feed: image resize + rescale + normalise, and max_length collation.
//
//   * resize  : Pillow's Image.resize(resample=BILINEAR) 8-bit path (libImaging/Resample.c: precompute_coeffs,
//               normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc, ImagingResampleVertical_8bpc) — the resampler behind
//               the HF LayoutLMv3 image processor the reference builds (EE/models/LayoutLMv3.py:674-677) and applies in
//               EE/data/RVL_CDIP.py:246-262.  Antialiased triangle filter with support max(in/out, 1), 22-bit fixed-point
//               weights, each pass rounded to 8 bits: reproduced bit for bit (coefficients in float64 like Pillow).
//   * rescale / normalise: uint8 -> float32 through a 256-entry table built on the host exactly as HF does
//               (u * (1/255) in float64 -> float32, then (v - 0.5f) / 0.5f).
//   * collate : DataCollatorWithPadding(padding="max_length") of EE/utils.py:93-98: pad input_ids with <pad>,
//               attention_mask with 0, bbox with [0,0,0,0] up to T.
// Everything here is HBM/latency-bound byte and integer work (3 MB of pixels in, 0.6 MB out per document).
#include "mmee_kernels.h"

namespace mmee {

constexpr int kPrecisionBits = 32 - 8 - 2;

// one thread per output index: bounds[(doc*2+axis)*R + xx] = {xmin, xmax}; kk[((doc*2+axis)*R + xx)*KMAX + x]
__global__ __launch_bounds__(256) void resize_coeffs_kernel(const ImageDesc* __restrict__ desc, int R, int KMAX,
                                                            int2* __restrict__ bounds, int* __restrict__ kk) {
    const int doc = blockIdx.x, axis = blockIdx.y;
    const int in_size = axis == 0 ? desc[doc].w : desc[doc].h;
    const double scale = (double)in_size / (double)R;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double ss = 1.0 / filterscale;
    for (int xx = threadIdx.x; xx < R; xx += blockDim.x) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        if (xmax > KMAX) xmax = KMAX;                       // guarded on the host (in/out <= (KMAX-1)/2)
        int* k = kk + ((size_t)(doc * 2 + axis) * R + xx) * KMAX;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double w = a < 1.0 ? 1.0 - a : 0.0;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            double w = a < 1.0 ? 1.0 - a : 0.0;
            if (ww != 0.0) w /= ww;
            k[x] = w < 0 ? (int)(-0.5 + w * (double)(1 << kPrecisionBits)) : (int)(0.5 + w * (double)(1 << kPrecisionBits));
        }
        bounds[(size_t)(doc * 2 + axis) * R + xx] = make_int2(xmin, xmax);
    }
}

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= kPrecisionBits;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: tmp[doc][y][xx][c] (uint8, row stride R*3, plane stride max_h*R*3)
__global__ __launch_bounds__(256) void resize_horizontal_kernel(const unsigned char* __restrict__ images, const ImageDesc* __restrict__ desc,
                                                                int R, int KMAX, int max_h, const int2* __restrict__ bounds,
                                                                const int* __restrict__ kk, unsigned char* __restrict__ tmp) {
    const int doc = blockIdx.y;
    const ImageDesc d = desc[doc];
    const unsigned char* img = images + d.offset;
    const int total = d.h * R;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / R, xx = i - y * R;
        const int2 b = bounds[(size_t)(doc * 2 + 0) * R + xx];
        const int* k = kk + ((size_t)(doc * 2 + 0) * R + xx) * KMAX;
        unsigned char* o = tmp + ((size_t)doc * max_h + y) * R * 3 + (size_t)xx * 3;
        if (d.c == 1) {
            int s0 = 1 << (kPrecisionBits - 1);
            const unsigned char* p = img + (size_t)y * d.w + b.x;
            for (int x = 0; x < b.y; ++x) s0 += (int)p[x] * k[x];
            const unsigned char v = clip8(s0);
            o[0] = v; o[1] = v; o[2] = v;                    // convert("RGB") of an "L" image replicates the channel
        } else {
            int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
            const unsigned char* p = img + ((size_t)y * d.w + b.x) * 3;
            for (int x = 0; x < b.y; ++x) {
                s0 += (int)p[3 * x] * k[x];
                s1 += (int)p[3 * x + 1] * k[x];
                s2 += (int)p[3 * x + 2] * k[x];
            }
            o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
        }
    }
}

// vertical pass + uint8 -> float table + HWC -> CHW: out[doc][c][yy][xx]
__global__ __launch_bounds__(256) void resize_vertical_kernel(const unsigned char* __restrict__ tmp, int R, int KMAX, int max_h,
                                                              const int2* __restrict__ bounds, const int* __restrict__ kk,
                                                              const float* __restrict__ lut, float* __restrict__ out,
                                                              unsigned char* __restrict__ out_u8) {
    const int doc = blockIdx.y;
    const int total = R * R;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int yy = i / R, xx = i - yy * R;
        const int2 b = bounds[(size_t)(doc * 2 + 1) * R + yy];
        const int* k = kk + ((size_t)(doc * 2 + 1) * R + yy) * KMAX;
        int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
        const unsigned char* p = tmp + ((size_t)doc * max_h + b.x) * R * 3 + (size_t)xx * 3;
        for (int y = 0; y < b.y; ++y) {
            const unsigned char* q = p + (size_t)y * R * 3;
            s0 += (int)q[0] * k[y];
            s1 += (int)q[1] * k[y];
            s2 += (int)q[2] * k[y];
        }
        const unsigned char v0 = clip8(s0), v1 = clip8(s1), v2 = clip8(s2);
        float* o = out + (size_t)doc * 3 * total + i;
        o[0] = lut[v0];
        o[(size_t)total] = lut[v1];
        o[2 * (size_t)total] = lut[v2];
        if (out_u8) {
            unsigned char* u = out_u8 + ((size_t)doc * total + i) * 3;
            u[0] = v0; u[1] = v1; u[2] = v2;
        }
    }
}

void launch_preprocess_images(const unsigned char* images, const ImageDesc* desc, int B, int R, int KMAX, int max_h, int2* bounds,
                              int* kk, unsigned char* tmp, const float* lut, float* out, unsigned char* out_u8, hipStream_t s) {
    hipLaunchKernelGGL(resize_coeffs_kernel, dim3(B, 2), dim3(256), 0, s, desc, R, KMAX, bounds, kk);
    int gx = (max_h * R + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(resize_horizontal_kernel, dim3(gx, B), dim3(256), 0, s, images, desc, R, KMAX, max_h, bounds, kk, tmp);
    int gy = (R * R + 255) / 256;
    hipLaunchKernelGGL(resize_vertical_kernel, dim3(gy, B), dim3(256), 0, s, tmp, R, KMAX, max_h, bounds, kk, lut, out, out_u8);
}

// ragged token streams -> max_length tensors
__global__ __launch_bounds__(256) void collate_pad_kernel(const long long* __restrict__ ids, const long long* __restrict__ boxes,
                                                          const long long* __restrict__ offsets, int T, long long pad_id,
                                                          long long* __restrict__ out_ids, long long* __restrict__ out_mask,
                                                          long long* __restrict__ out_bbox) {
    const int b = blockIdx.x;
    const long long o = offsets[b];
    long long n = offsets[b + 1] - o;
    if (n > T) n = T;                                       // truncation=True
    for (int j = threadIdx.x; j < T; j += blockDim.x) {
        const bool v = j < n;
        out_ids[(size_t)b * T + j] = v ? ids[o + j] : pad_id;
        out_mask[(size_t)b * T + j] = v ? 1 : 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) out_bbox[((size_t)b * T + j) * 4 + c] = v ? boxes[(o + j) * 4 + c] : 0;
    }
}

void launch_collate_pad(const long long* ids, const long long* boxes, const long long* offsets, int B, int T, long long pad_id,
                        long long* out_ids, long long* out_mask, long long* out_bbox, hipStream_t s) {
    hipLaunchKernelGGL(collate_pad_kernel, dim3(B), dim3(256), 0, s, ids, boxes, offsets, T, pad_id, out_ids, out_mask, out_bbox);
}

}  // namespace mmee
