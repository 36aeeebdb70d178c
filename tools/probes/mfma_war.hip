// Can an LDS load issued right BEHIND a queue of MFMAs overwrite the B operand of the last one before that MFMA has read it?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int Q>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2 * 4];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned* mine = lds + wave * 512;
    // first record: all halves 1.0 (0x3C00); second record: all halves 2.0 (0x4000)
    for (int j = 0; j < 4; ++j) { mine[lane * 4 + j] = 0x3C003C00u; mine[256 + lane * 4 + j] = 0x40004000u; }
    __syncthreads();
    const unsigned a1 = (unsigned)(size_t)(mine + lane * 4) & 0xFFFFu ? 0 : 0;   // (placeholder, address computed below)
    (void)a1;
    unsigned addr_one, addr_two;
    addr_one = (unsigned)(__builtin_amdgcn_readfirstlane(0)) + (unsigned)((size_t)(void*)(mine + lane * 4));
    addr_two = addr_one + 1024;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        u32x4 A, B, X0, X1, X2, X3;
        f32x4 z = {0.f, 0.f, 0.f, 0.f}, r;
        f32x4 q0 = z, q1 = z, q2 = z, q3 = z, q4 = z, q5 = z, q6 = z, q7 = z;
        asm volatile(
            "ds_read_b128 %0, %13\n\t"           // A = ones
            "ds_read_b128 %1, %13\n\t"           // B = ones
            "ds_read_b128 %2, %13\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            ".if %15 > 0\n\t v_mfma_f32_16x16x32_f16 %5, %0, %2, %5\n\t .endif\n\t"
            ".if %15 > 1\n\t v_mfma_f32_16x16x32_f16 %6, %0, %2, %6\n\t .endif\n\t"
            ".if %15 > 2\n\t v_mfma_f32_16x16x32_f16 %7, %0, %2, %7\n\t .endif\n\t"
            ".if %15 > 3\n\t v_mfma_f32_16x16x32_f16 %8, %0, %2, %8\n\t .endif\n\t"
            ".if %15 > 4\n\t v_mfma_f32_16x16x32_f16 %9, %0, %2, %9\n\t .endif\n\t"
            ".if %15 > 5\n\t v_mfma_f32_16x16x32_f16 %10, %0, %2, %10\n\t .endif\n\t"
            ".if %15 > 6\n\t v_mfma_f32_16x16x32_f16 %11, %0, %2, %11\n\t .endif\n\t"
            ".if %15 > 7\n\t v_mfma_f32_16x16x32_f16 %12, %0, %2, %12\n\t .endif\n\t"
            "v_mfma_f32_16x16x32_f16 %4, %0, %1, %3\n\t"     // the probed product: ones x B(ones) = 32
            "ds_read_b128 %1, %14\n\t"                        // B <- twos, right behind it
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 15\n\t s_nop 15\n\t"
            : "=&v"(A), "=&v"(B), "=&v"(X0), "+v"(z), "=&v"(r), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7)
            : "v"(addr_one), "v"(addr_two), "n"(Q) : "memory");
        (void)X1; (void)X2; (void)X3;
        if (r[0] != 32.f || r[1] != 32.f || r[2] != 32.f || r[3] != 32.f) bad++;
        if (B[0] != 0x40004000u) bad += 1000000;
    }
    atomicAdd(out, bad);
}
template <int Q> void run(unsigned* out) {
    hipMemset(out, 0, 4);
    hipLaunchKernelGGL(k<Q>, dim3(1024), dim3(256), 0, 0, out, 200);
    hipError_t e = hipDeviceSynchronize();
    unsigned r; hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost);
    printf("MFMAs queued ahead %d: sync %d, wrong products %u of %u\n", Q, (int)e, r, 1024u * 256u * 200u);
}
int main() {
    unsigned* out; hipMalloc(&out, 4);
    run<0>(out); run<1>(out); run<2>(out); run<4>(out); run<6>(out); run<8>(out);
    return 0;
}
