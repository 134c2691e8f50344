#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    const int va = 1000 + lane, vb = 2000 + lane;
    const int da = __builtin_amdgcn_update_dpp(va, vb, 0x128, 0xf, 0xc, false);
    const int db = __builtin_amdgcn_update_dpp(vb, va, 0x128, 0xf, 0x3, false);
    out[lane] = da; out[64 + lane] = db;
}
int main() {
    int* d; hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("da:"); for (int i = 0; i < 32; ++i) printf(" %d", h[i]); printf("\ndb:"); for (int i = 0; i < 32; ++i) printf(" %d", h[64 + i]); printf("\n");
}
