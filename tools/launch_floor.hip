// Diagnostic: back-to-back launch time of an EMPTY kernel at the grid / block / LDS shapes of the rollout kernels (the part of
// a kernel's launch-to-launch time no kernel code can remove).  hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_floor tools/launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ char lds[];
__global__ void k_empty(int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 1u << 30) p[0] = lds[0]; }
int main() {
    struct { const char* name; int grid, block, lds; } cfg[] = {
        {"1 x 64, no LDS", 1, 64, 0}, {"k_env    1024 x 256, 27 KB", 1024, 256, 27648}, {"k_head    255 x 512, 136 KB", 255, 512, 139264},
        {"k_encode  251 x 512, 60 KB", 251, 512, 61440}, {"2048 x 256, 8 KB", 2048, 256, 8192}};
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (auto& c : cfg) {
        hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), c.lds, 0, nullptr);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), c.lds, 0, nullptr);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-30s %6.2f us / launch\n", c.name, ms * 1000 / 200);
    }
    return 0;
}
