// How fast does ONE compute unit get a 256 x 256 output tile (256 KB) out -- 16 waves x 16 stores of 16 bytes per lane, then
// s_waitcnt vmcnt(0) -- for the address patterns of the GEMM epilogues, alone and beside other storing CUs, with and without idle time
// between bursts?   (GPU box only: hipcc --offload-arch=gfx950 -O3 tools/store_rate.hip -o /tmp/store_rate && /tmp/store_rate)
// Round 4: the split GEMM's epilogue + the wait for its slowest wave cost ~21k cycles per tile on every CU (stamps in gemm_split.hip),
// although the chip's write path is idle most of the time; this separates the address pattern / cache policy from the rest.
//   pattern 0: 1 KiB contiguous per wave instruction (streaming)
//   pattern 1: rounds 1-3 epilogue: an instruction writes 4 rows x 256 B of a row-major [M][ld] output (full 128-byte lines)
//   pattern 2: round-4 epilogue_t: an instruction writes 16 rows x 64 B (half lines; the neighbouring instruction writes the other half)
//   pattern 3: round-4 epilogue_t after the lane-half exchange: an instruction writes 8 rows x 128 B (whole lines, one per row)
// policy 0 plain, 1 nt, 2 sc1, 3 sc0 sc1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ void st16(i32x4* p, i32x4 v) {
    if (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

template <int POLICY>
__global__ __launch_bounds__(1024) void store_burst(char* out, size_t ld_bytes, int tiles_n, int rounds, int pattern, int idle_sleeps,
                                                    unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, gq = lane >> 4;
    const i32x4 v = {lane, wave, pattern, rounds};
    unsigned long long burst = 0;
    // random-ish initial phase so that the CUs do not burst together when idle time separates the bursts
    if (idle_sleeps) for (int k = 0; k < (int)((blockIdx.x * 2654435761u >> 20) % (unsigned)idle_sleeps); ++k) __builtin_amdgcn_s_sleep(127);
    for (int r = 0; r < rounds; ++r) {
        const int tile = blockIdx.x + r * gridDim.x;
        const int tm = tile / tiles_n, tn = tile % tiles_n;
        char* base = out + (size_t)tm * 256 * ld_bytes + (size_t)tn * 1024;
        for (int k = 0; k < idle_sleeps; ++k) __builtin_amdgcn_s_sleep(127);      // ~ 127 * 64 cycles each
        __builtin_amdgcn_s_barrier();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            char* p;
            if (pattern == 0) p = out + ((size_t)tile * 256 + wave * 16 + s) * 1024 + lane * 16;
            else if (pattern == 1) p = base + (size_t)(wr * 64 + 4 * s + gq) * ld_bytes + wc * 256 + l15 * 16;
            else if (pattern == 2) { const int i = s >> 2, j = s & 3; p = base + (size_t)(wr * 64 + 16 * i + l15) * ld_bytes + wc * 256 + j * 64 + gq * 16; }
            else { const int i = s >> 2, jp = (s >> 1) & 1, hb = s & 1; p = base + (size_t)(wr * 64 + 16 * i + 8 * hb + (l15 & 7)) * ld_bytes + wc * 256 + jp * 128 + (l15 >> 3) * 64 + gq * 16; }
            st16<POLICY>(reinterpret_cast<i32x4*>(p), v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        burst += __builtin_amdgcn_s_memtime() - t0;
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = burst;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2304, tiles_n = N / 256, rounds = 32;
    printf("N = %d (row stride %d bytes)\n", N, N * 4);
    const size_t ld = (size_t)N * 4;
    const size_t rows = (size_t)((256 * rounds + tiles_n - 1) / tiles_n + 1) * 256;
    char* out;
    unsigned long long* cyc;
    hipMalloc(&out, rows * ld + (size_t)256 * rounds * 262144);
    hipMalloc(&cyc, 256 * 8);
    std::vector<unsigned long long> h(256);
    for (int idle : {12})
        for (int grid : {1, 256})
            for (int policy = 0; policy < 2; ++policy)
                for (int pattern = 0; pattern < 4; ++pattern) {
                    if (policy > 0 && pattern == 0) continue;
                    auto k = policy == 0 ? store_burst<0> : policy == 1 ? store_burst<1> : policy == 2 ? store_burst<2> : store_burst<3>;
                    hipEvent_t e0, e1;
                    hipEventCreate(&e0); hipEventCreate(&e1);
                    hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 0, 0, out, ld, tiles_n, rounds, pattern, idle, cyc);
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 0, 0, out, ld, tiles_n, rounds, pattern, idle, cyc);
                    hipEventRecord(e1);
                    hipDeviceSynchronize();
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
                    double sum = 0;
                    for (int i = 0; i < grid; ++i) sum += (double)h[i];
                    printf("idle %2d grid %3d policy %d pattern %d: %.3f ms, %.0f shader cycles per 256 KB burst (%.1f B/clk per CU)\n", idle, grid, policy,
                           pattern, ms, sum / grid / rounds, 262144.0 / (sum / grid / rounds));
                }
    return 0;
}
