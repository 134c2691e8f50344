// Feasibility probe: f32-grade GEMM results from the f16 / bf16 matrix cores by operand splitting (GPU box only).
//   mode 0: v_mfma_f32_32x32x2_f32 on the f32 operands (what the path uses today)
//   mode 1: f16 hi/lo split, 3 terms   (hi*hi + hi*lo + lo*hi),  v_mfma_f32_32x32x16_f16
//   mode 2: bf16 3-way split, 6 terms,                            v_mfma_f32_32x32x16_bf16
//   mode 3: bf16 hi/lo split, 3 terms
// Each is compared with an f64 host reference of C = A * W^T on the same f32 inputs.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline __bf16 to_bf16(float x) { return (__bf16)x; }

template <int MODE>
__global__ void gemm_probe(const float* A, const float* W, float* C, int M, int N, int K, float sa, float sw) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* a = A + (size_t)(m0 + r) * K;
    const float* w = W + (size_t)(n0 + r) * K;
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k + h], w[k + h], acc, 0, 0, 0);
    } else if (MODE == 1) {
        for (int k = 0; k < K; k += 16) {
            f16x8 ah, al, wh, wl;
            for (int j = 0; j < 8; ++j) {
                const float av = a[k + 8 * h + j] * sa, wv = w[k + 8 * h + j] * sw;
                ah[j] = (_Float16)av; al[j] = (_Float16)(av - (float)ah[j]);
                wh[j] = (_Float16)wv; wl[j] = (_Float16)(wv - (float)wh[j]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc, 0, 0, 0);
        }
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a1, a2, a3, w1, w2, w3;
            for (int j = 0; j < 8; ++j) {
                const float av = a[k + 8 * h + j], wv = w[k + 8 * h + j];
                a1[j] = to_bf16(av); a2[j] = to_bf16(av - (float)a1[j]); a3[j] = to_bf16(av - (float)a1[j] - (float)a2[j]);
                w1[j] = to_bf16(wv); w2[j] = to_bf16(wv - (float)w1[j]); w3[j] = to_bf16(wv - (float)w1[j] - (float)w2[j]);
            }
            if (MODE == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, w1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w3, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w2, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w1, acc, 0, 0, 0);
        }
    }
    const float inv = (MODE == 1) ? 1.f / (sa * sw) : 1.f;
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        C[(size_t)(m0 + row) * N + n0 + r] = acc[e] * inv;
    }
}

static double frand() { return (double)rand() / RAND_MAX; }
static double nrand() { return std::sqrt(-2.0 * std::log(frand() + 1e-300)) * std::cos(6.283185307179586 * frand()); }

template <int MODE>
static void run(const char* name, const std::vector<float>& A, const std::vector<float>& W, const std::vector<double>& ref, int M, int N, int K,
                float sa, float sw) {
    float *dA, *dW, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gemm_probe<MODE>, dim3(N / 32, M / 32), dim3(64), 0, 0, dA, dW, dC, M, N, K, sa, sw);
    std::vector<float> C((size_t)M * N);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double maxabs = 0, sumsq = 0, refsq = 0;
    for (size_t i = 0; i < C.size(); ++i) {
        const double d = (double)C[i] - ref[i];
        maxabs = std::fmax(maxabs, std::fabs(d));
        sumsq += d * d; refsq += ref[i] * ref[i];
    }
    printf("  %-34s max abs err %.3e   rms err / rms ref %.3e\n", name, maxabs, std::sqrt(sumsq / refsq));
    hipFree(dA); hipFree(dW); hipFree(dC);
}

int main() {
    const int M = 128, N = 128;
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int K = cfg == 1 ? 3072 : 768;
        const double a_scale = cfg == 2 ? 0.05 : 1.0, w_scale = 0.02;
        srand(7 + cfg);
        std::vector<float> A((size_t)M * K), W((size_t)N * K);
        for (auto& v : A) v = (float)(nrand() * a_scale);
        for (auto& v : W) v = (float)(nrand() * w_scale);
        std::vector<double> ref((size_t)M * N);
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)A[(size_t)i * K + k] * (double)W[(size_t)j * K + k];
                ref[(size_t)i * N + j] = s;
            }
        printf("K=%d  A ~ N(0,%g)  W ~ N(0,%g)\n", K, a_scale, w_scale);
        run<0>("f32 MFMA 32x32x2", A, W, ref, M, N, K, 1.f, 1.f);
        run<1>("f16 hi/lo, 3 terms, no scaling", A, W, ref, M, N, K, 1.f, 1.f);
        run<1>("f16 hi/lo, 3 terms, W x 2^8", A, W, ref, M, N, K, 1.f, 256.f);
        run<1>("f16 hi/lo, 3 terms, A x 2^4, W x 2^8", A, W, ref, M, N, K, 16.f, 256.f);
        run<2>("bf16 3-way, 6 terms", A, W, ref, M, N, K, 1.f, 1.f);
        run<3>("bf16 hi/lo, 3 terms", A, W, ref, M, N, K, 1.f, 1.f);
    }
    return 0;
}
