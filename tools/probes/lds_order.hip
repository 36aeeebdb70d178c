// Do LDS reads of one wave return in issue order when they address both sides of the 64 KiB line of a 160 KiB allocation?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(unsigned* out, int lo_first, int hi_base, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned* w = reinterpret_cast<unsigned*>(lds);
    const int n = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < n; i += blockDim.x) w[i] = 0xA5000000u + i;
    __syncthreads();
    unsigned bad = 0;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
        // a few reads of the "other" side first to keep that side's queue busy, then the probed read, then more of the other side
        const unsigned a_lo = ((wave * 64 + lane) * 16 + (unsigned)(it % 7) * 8192u) % 61440u;           // < 60 KiB
        const unsigned a_hi = (unsigned)hi_base + ((wave * 64 + lane) * 16 + (unsigned)(it % 5) * 8192u) % 57344u;
        const unsigned first = lo_first ? a_lo : a_hi, second = lo_first ? a_hi : a_lo;
        u32x4 x = {0u, 0u, 0u, 0u}, y0, y1, y2, y3;
        asm volatile(
            "ds_read_b128 %0, %5\n\t"
            "ds_read_b128 %1, %6\n\t"
            "ds_read_b128 %2, %6 offset:1024\n\t"
            "ds_read_b128 %3, %6 offset:2048\n\t"
            "ds_read_b128 %4, %6 offset:3072\n\t"
            "s_waitcnt lgkmcnt(4)"
            : "+v"(x), "=v"(y0), "=v"(y1), "=v"(y2), "=v"(y3) : "v"(first), "v"(second) : "memory");
        // x must hold the word at `first` now (its register was zero before the read)
        const unsigned snap = x[0]; const unsigned snap3 = x[3];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (snap != 0xA5000000u + first / 4) bad++;
        if (snap3 != 0xA5000000u + first / 4 + 3) bad += 1000;
        if (y0[0] != 0xA5000000u + second / 4 || y3[0] != 0xA5000000u + (second + 3072) / 4) bad += 1000000;
    }
    atomicAdd(out, bad);
}
int main() {
    unsigned* out; hipMalloc(&out, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int hi_base : {65536, 98304, 16384}) for (int lo_first : {0, 1}) {
        hipMemset(out, 0, 4);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 160 * 1024, 0, out, lo_first, hi_base, 2000);
        hipError_t e = hipDeviceSynchronize();
        unsigned r; hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost);
        printf("other side at %6d, %s read first: sync %d, stale first reads %u of %u\n", hi_base, lo_first ? "low " : "high", (int)e, r, 256u * 512u * 2000u);
    }
    return 0;
}
