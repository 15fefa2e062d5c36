// Does VALU work issued between dependent MFMAs of one wave hide under them on gfx950?  One wave per SIMD (256 threads),
// loop of { v_mfma_f32_32x32x16_bf16 (same accumulator) + N x v_exp_f32 | v_fma_f32 } ; cycles per iteration by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
template <int N, int KIND>
__global__ void k(unsigned long long *out, float seed) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = seed;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    float e[8];
    for (int i = 0; i < 8; ++i) e[i] = seed * 0.01f + i;
    unsigned long long t0, t1;
    __syncthreads();
    STAMP(t0);
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j % 8]));
            else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[j % 8]));
        }
    }
    STAMP(t1);
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    float s = 0; for (int i = 0; i < 8; ++i) s += e[i];
    if (acc[3] + s == 12345.678f) out[100] = 1;
}
template <int N, int KIND> void run(unsigned long long *d, int threads) {
    hipLaunchKernelGGL((k<N, KIND>), dim3(1), dim3(threads), 0, 0, d, 1.0f);
    hipLaunchKernelGGL((k<N, KIND>), dim3(1), dim3(threads), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s x %2d per MFMA, %2d waves/CU: %7.1f cycles per iteration (wave 0), %7.1f (last wave)\n", KIND ? "v_fma_f32" : "v_exp_f32", N, threads / 64, h[0] / 256.0, h[threads / 64 - 1] / 256.0);
}
int main() {
    unsigned long long *d; hipMalloc(&d, 4096);
    for (int threads : {256, 512, 768}) {
        run<0, 0>(d, threads); run<1, 0>(d, threads); run<2, 0>(d, threads); run<3, 0>(d, threads); run<4, 0>(d, threads); run<6, 0>(d, threads); run<8, 0>(d, threads);
        run<2, 1>(d, threads); run<4, 1>(d, threads); run<6, 1>(d, threads); run<8, 1>(d, threads); run<12, 1>(d, threads);
    }
    return 0;
}
