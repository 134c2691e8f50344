// Power-limited ceiling probe: bare f16 MFMA loops on random operands, 32x32x16 vs 16x16x32, with the shader clock the chip
// holds under that load (GPU box only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(float* out, const f16x8* in, int iters, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
    float s = 0.f;
    if (MODE == 0) {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c & 3], b[(c >> 2) & 3], acc[c & 3], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    } else {
        f32x4 acc[16];
        for (int j = 0; j < 16; ++j) for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 32; ++c) acc[c & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c & 3], b[(c >> 2) & 3], acc[c & 15], 0, 0, 0);
        }
        for (int j = 0; j < 16; ++j) for (int e = 0; e < 4; ++e) s += acc[j][e];
    }
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int MODE>
static void run(const char* name, int wgs_per_cu) {
    const int grid = 256 * wgs_per_cu, iters = 40000;
    float* out; f16x8* in; unsigned long long* clk;
    hipMalloc(&out, 4); hipMalloc(&in, 256 * 8 * 16); hipMalloc(&clk, grid * 16);
    std::vector<_Float16> h(256 * 8 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)(((float)rand() / RAND_MAX - 0.5f) * 4.0f);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, in, 100, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, in, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(grid * 2);
    hipMemcpy(hc.data(), clk, grid * 16, hipMemcpyDeviceToHost);
    double g = 0; for (int i = 0; i < grid; ++i) g += (double)hc[2 * i] / hc[2 * i + 1] * 0.1;
    g /= grid;
    const double flops = (double)grid * 4 * iters * 16.0 * 32768.0;
    printf("%-22s %d WG/CU x 4 waves: %.2f ms  %.0f TFLOP/s (%.1f%% of 2500), clock %.2f GHz -> %.1f%% of the peak at that clock\n", name, wgs_per_cu, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 2500 * 100, g, flops / ms / 1e9 / (2500 * g / 2.4) * 100);
    hipFree(out); hipFree(in); hipFree(clk);
}

int main() {
    run<0>("f16 32x32x16", 1); run<0>("f16 32x32x16", 2); run<0>("f16 32x32x16", 4);
    run<1>("f16 16x16x32", 1); run<1>("f16 16x16x32", 2); run<1>("f16 16x16x32", 4);
    return 0;
}
