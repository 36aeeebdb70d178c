#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out, int bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned* w = reinterpret_cast<unsigned*>(lds);
    const int n = bytes / 4;
    for (int i = threadIdx.x; i < n; i += blockDim.x) w[i] = 0xA5000000u + i;
    __syncthreads();
    unsigned bad = 0, first = 0xFFFFFFFFu;
    for (int i = threadIdx.x; i < n; i += blockDim.x) if (w[i] != 0xA5000000u + i) { bad++; if ((unsigned)i < first) first = i; }
    atomicAdd(&out[0], bad); atomicMin(&out[1], first);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  sharedMemPerBlockOptin %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin);
    unsigned* out; hipMalloc(&out, 8);
    for (int kb : {64, 120, 128, 129, 136, 139, 140, 144, 152, 160}) {
        int bytes = kb * 1024;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        unsigned init[2] = {0, 0xFFFFFFFFu}; hipMemcpy(out, init, 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(512), bytes, 0, out, bytes);
        hipError_t e2 = hipDeviceSynchronize(); hipError_t e3 = hipGetLastError();
        unsigned r[2]; hipMemcpy(r, out, 8, hipMemcpyDeviceToHost);
        printf("%3d KiB: attr %d sync %d last %d  bad words %u first bad byte %u\n", kb, (int)e, (int)e2, (int)e3, r[0], r[1] == 0xFFFFFFFFu ? 0 : r[1] * 4);
    }
    return 0;
}
