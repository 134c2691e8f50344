// Ceiling probe: how fast does v_mfma_f32_32x32x2_f32 issue on gfx950 with nothing else in the loop, and with the
// GEMM's LDS fragment reads beside it?  (diagnostic for DESIGN.md's GEMM roofline section; GPU box only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void peak_kernel(float* out, int iters, float seed, const float* src) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) smem[i] = seed * (float)(i & 7);
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = seed, b = seed * 0.5f;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 22, 0x00020000);
    const f32x4* frag = reinterpret_cast<const f32x4*>(smem) + lane * 8 + (threadIdx.x >> 6) * 512;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c & 3], 0, 0, 0);
        } else if (MODE >= 2) {
            // same reads, two fragment sets, counted waits: the reads of group g+1 fly while group g's MFMAs issue
            const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem + lane * 128 + (threadIdx.x >> 6) * 8192;
            f32x4 f[2][4];
#define RD(set, off) asm volatile("ds_read_b128 %0, %4 offset:" #off "\n\tds_read_b128 %1, %4 offset:" #off "+16\n\tds_read_b128 %2, %4 offset:" #off "+32\n\tds_read_b128 %3, %4 offset:" #off "+48" \
                     : "=&v"(f[set][0]), "=&v"(f[set][1]), "=&v"(f[set][2]), "=&v"(f[set][3]) : "v"(base) : "memory")
#define WT(n, set) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[set][0]), "+v"(f[set][1]), "+v"(f[set][2]), "+v"(f[set][3]))
#define MM(set) _Pragma("unroll") for (int c = 0; c < 4; ++c) { \
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[set][0][c], f[set][2][c], acc[0], 0, 0, 0); \
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[set][0][c], f[set][3][c], acc[1], 0, 0, 0); \
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[set][1][c], f[set][2][c], acc[2], 0, 0, 0); \
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[set][1][c], f[set][3][c], acc[3], 0, 0, 0); }
            // MODE 3/4/5: + one LDS-DMA piece (1 KiB) per 8 MFMAs from an L2-resident buffer; 3 = 64-bit VGPR address,
            // 4 = SGPR base + 32-bit VGPR offset, 5 = buffer_load ... lds (descriptor + 32-bit VGPR offset)
            const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            __attribute__((address_space(3))) float* dst = (__attribute__((address_space(3))) float*)smem + 8192 + wv * 1024;
            const unsigned voff = (unsigned)(lane * 16 + ((blockIdx.x * 4 + wv) & 255) * 4096);
#define H2(set, c0) _Pragma("unroll") for (int c = c0; c < c0 + 2; ++c) { \
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[set][0][c], f[set][2][c], acc[0], 0, 0, 0); \
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[set][0][c], f[set][3][c], acc[1], 0, 0, 0); \
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[set][1][c], f[set][2][c], acc[2], 0, 0, 0); \
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[set][1][c], f[set][3][c], acc[3], 0, 0, 0); }
            auto dma = [&](int j) {
                const unsigned o = voff + ((it + j) & 3) * 1024;
                if (MODE == 3) {
                    const char* pp = (const char*)src + (size_t)o + (size_t)(it & 1) * 0;   // per-lane 64-bit address
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pp, (__attribute__((address_space(3))) void*)(dst + 256 * j), 16, 0, 0);
                } else if (MODE == 4) {
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(o), "s"(src), "s"((unsigned)(size_t)(dst + 256 * j)) : "memory", "m0");
                } else if (MODE == 5) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + 256 * j), 16, o, 0, 0, 0);
                }
            };
            if (it == 0) { RD(0, 0); }
            RD(1, 64);
            WT(4, 0);
            if (MODE == 2) { MM(0) } else {
                H2(0, 0) __builtin_amdgcn_sched_barrier(0); dma(0); __builtin_amdgcn_sched_barrier(0);
                H2(0, 2) __builtin_amdgcn_sched_barrier(0); dma(1); __builtin_amdgcn_sched_barrier(0);
            }
            RD(0, 0);
            WT(4, 1);
            if (MODE == 2) { MM(1) } else {
                H2(1, 0) __builtin_amdgcn_sched_barrier(0); dma(2); __builtin_amdgcn_sched_barrier(0);
                H2(1, 2) __builtin_amdgcn_sched_barrier(0); dma(3); __builtin_amdgcn_sched_barrier(0);
            }
            ++it;   // two k-groups per trip
        } else {
            // 4 ds_read_b128 per 16 MFMAs, as in the GEMM's k-group (conflict-free: 64 lanes x 16 B, stride 128 B... per-lane rows)
            f32x4 f0 = frag[(it & 1) * 4 + 0], f1 = frag[(it & 1) * 4 + 1], f2 = frag[(it & 1) * 4 + 2], f3 = frag[(it & 1) * 4 + 3];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0[c], f2[c], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0[c], f3[c], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f1[c], f2[c], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f1[c], f3[c], acc[3], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char* name, int wgs_per_cu, int threads) {
    float* out;
    float* src;
    hipMalloc(&out, 4);
    hipMalloc(&src, 1 << 22);
    hipMemset(src, 0, 1 << 22);
    const int iters = 20000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(peak_kernel<MODE>, dim3(grid), dim3(threads), 65536, 0, out, 100, 1.0f, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(peak_kernel<MODE>, dim3(grid), dim3(threads), 65536, 0, out, iters, 1.0f, src);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * (threads / 64) * iters * 16.0 * 4096.0;
    printf("%-28s %d WG/CU x %d waves: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", name, wgs_per_cu, threads / 64, ms, flops / ms / 1e9,
           flops / ms / 1e9 / 157.3 * 100);
    hipFree(out);
    hipFree(src);
}

int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&peak_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&peak_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&peak_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&peak_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&peak_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&peak_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<0>("mfma only", 1, 256);
    run<0>("mfma only", 2, 256);
    run<1>("mfma + 4 ds_read_b128/16", 1, 256);
    run<1>("mfma + 4 ds_read_b128/16", 2, 256);
    run<2>("same, 2 fragment sets", 1, 256);
    run<2>("same, 2 fragment sets", 2, 256);
    run<3>("+DMA/8 MFMA, 64-bit vaddr", 1, 256);
    run<3>("+DMA/8 MFMA, 64-bit vaddr", 2, 256);
    run<4>("+DMA/8 MFMA, saddr+voff", 1, 256);
    run<4>("+DMA/8 MFMA, saddr+voff", 2, 256);
    run<5>("+DMA/8 MFMA, buffer lds", 1, 256);
    run<5>("+DMA/8 MFMA, buffer lds", 2, 256);
    return 0;
}
