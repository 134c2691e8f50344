// Does a CU-masked stream confine a persistent kernel, and can an MFMA-heavy (power-limited) kernel on 3/4 of the CUs run
// beside a latency-bound kernel on the remaining quarter at no cost?  (GPU box only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void hog(float* out, const f16x8* in, int iters, unsigned* xcc_hist) {
    if (threadIdx.x == 0) atomicAdd(&xcc_hist[__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u], 1u);
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c & 3], b[(c >> 2) & 3], acc[c & 15], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) for (int e = 0; e < 4; ++e) s += acc[j][e];
    if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void chase(const int* next, int* out, int iters, unsigned* xcc_hist) {   // dependent loads: latency-bound
    if (threadIdx.x == 0) atomicAdd(&xcc_hist[__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u], 1u);
    int p = (blockIdx.x * 256 + threadIdx.x) & 0xfffff;
    for (int i = 0; i < iters; ++i) p = next[p];
    if (p == -1) out[0] = p;
}

static float timed(hipStream_t s, void (*launch)(hipStream_t)) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s); launch(s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

static float* g_out; static f16x8* g_in; static int* g_next; static int* g_iout; static unsigned* g_hist;
static int g_hog_grid = 256 * 4, g_chase_grid = 64 * 8;
static void launch_hog(hipStream_t s) { hipLaunchKernelGGL(hog, dim3(g_hog_grid), dim3(256), 0, s, g_out, g_in, 20000, g_hist); }
static void launch_chase(hipStream_t s) { hipLaunchKernelGGL(chase, dim3(g_chase_grid), dim3(256), 0, s, g_next, g_iout, 20000, g_hist + 8); }

int main() {
    hipMalloc(&g_out, 4); hipMalloc(&g_in, 256 * 8 * 16); hipMalloc(&g_next, 4 << 20); hipMalloc(&g_iout, 4); hipMalloc(&g_hist, 64);
    std::vector<_Float16> h(256 * 8 * 8); srand(1);
    for (auto& v : h) v = (_Float16)(((float)rand() / RAND_MAX - 0.5f) * 4.0f);
    hipMemcpy(g_in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::vector<int> nx(1 << 20);
    for (int i = 0; i < (1 << 20); ++i) nx[i] = (int)(((long long)i * 40503 + 12345) & 0xfffff);
    hipMemcpy(g_next, nx.data(), 4 << 20, hipMemcpyHostToDevice);
    // masks: 256 CUs = 8 words.  "GEMM" set: 24 of every 32 CUs; "attention" set: the other 8 of every 32
    uint32_t m_big[8], m_small[8], m_all[8];
    for (int i = 0; i < 8; ++i) { m_big[i] = 0x00ffffffu; m_small[i] = 0xff000000u; m_all[i] = 0xffffffffu; }
    hipStream_t s_all, s_big, s_small;
    if (hipExtStreamCreateWithCUMask(&s_all, 8, m_all) != hipSuccess || hipExtStreamCreateWithCUMask(&s_big, 8, m_big) != hipSuccess ||
        hipExtStreamCreateWithCUMask(&s_small, 8, m_small) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return 1; }
    auto hist = [&](const char* tag) {
        unsigned hh[16]; hipMemcpy(hh, g_hist, 64, hipMemcpyDeviceToHost);
        printf("   %s workgroups per XCD: hog [", tag); for (int i = 0; i < 8; ++i) printf("%u ", hh[i]);
        printf("]  chase ["); for (int i = 0; i < 8; ++i) printf("%u ", hh[8 + i]); printf("]\n");
        hipMemset(g_hist, 0, 64);
    };
    hipMemset(g_hist, 0, 64);
    launch_hog(s_all); hipDeviceSynchronize(); hipMemset(g_hist, 0, 64);
    g_hog_grid = 256 * 4; printf("hog alone, all 256 CUs (grid 1024):     %.2f ms\n", timed(s_all, launch_hog)); hist("");
    g_hog_grid = 192 * 4; printf("hog alone, 192-CU mask (grid 768 = 3/4 of the work): %.2f ms\n", timed(s_big, launch_hog)); hist("");
    g_hog_grid = 256 * 4; printf("hog alone, 192-CU mask (grid 1024):     %.2f ms\n", timed(s_big, launch_hog)); hist("");
    printf("chase alone, all CUs:                   %.2f ms\n", timed(s_all, launch_chase));
    printf("chase alone, 64-CU mask:                %.2f ms\n", timed(s_small, launch_chase)); hist("");
    // concurrent
    hipEvent_t a0, a1, b0, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
    hipDeviceSynchronize();
    hipEventRecord(a0, s_big); launch_hog(s_big); hipEventRecord(a1, s_big);
    hipEventRecord(b0, s_small); launch_chase(s_small); hipEventRecord(b1, s_small);
    hipDeviceSynchronize();
    float ta, tb; hipEventElapsedTime(&ta, a0, a1); hipEventElapsedTime(&tb, b0, b1);
    printf("concurrent: hog on 192 CUs %.2f ms, chase on 64 CUs %.2f ms\n", ta, tb); hist("");
    return 0;
}
