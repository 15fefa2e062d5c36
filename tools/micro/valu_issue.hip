// Issue cost of the softmax-relevant VALU instructions on gfx950, one wave alone on its SIMD and two / three
// waves sharing a SIMD (waves w and w + 4 of a workgroup land on the same SIMD).  s_memtime stamps around
// 256 independent instructions.  Build + run: hipcc --offload-arch=gfx950 -O2 valu_issue.hip -o valu_issue && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP16(x) x x x x x x x x x x x x x x x x
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
template <int KIND>
__global__ void k(unsigned long long *out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long t0, t1;
    __syncthreads();
    STAMP(t0);
    for (int it = 0; it < 16; ++it) {
        if (KIND == 0) { REP16(asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(a0), "+v"(a1));) }
        if (KIND == 1) { REP16(asm volatile("v_fma_f32 %0, %0, %2, %2\n\tv_fma_f32 %1, %1, %2, %2" : "+v"(a0), "+v"(a1) : "v"(a2));) }
        if (KIND == 2) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double *)&a0) : "v"(*(double *)&a2)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double *)&a4) : "v"(*(double *)&a6));) }
        if (KIND == 3) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_cvt_pk_bf16_f32 %3, %1, %2" : "=v"(a4), "+v"(a0), "+v"(a1), "=v"(a5));) }
        if (KIND == 4) { REP16(asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));) }
        if (KIND == 5) { REP16(asm volatile("v_or3_b32 %0, %0, %2, %3\n\tv_or3_b32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(a2), "v"(a3));) }
        if (KIND == 6) { REP16(asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(a2));) }
        if (KIND == 7) { REP16(asm volatile("v_exp_f16 %0, %0\n\tv_exp_f16 %1, %1" : "+v"(a0), "+v"(a1));) }
        if (KIND == 8) { REP16(asm volatile("v_mul_f32 %0, %0, %2\n\tv_exp_f32 %1, %1" : "+v"(a0), "+v"(a1) : "v"(a2));) }
    }
    STAMP(t1);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[1000] = 1;
}
template <int KIND>
void run(const char *name, unsigned long long *d) {
    for (int threads : {64, 512, 768}) {       // 1 wave; 8 waves = 2 per SIMD; 12 waves = 3 per SIMD
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, 1.0f);
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-18s %2d waves/CU: %6.2f cycles per wave-instruction (wave 0), %6.2f (last wave)\n", name, threads / 64,
               h[0] / 512.0, h[threads / 64 - 1] / 512.0);
    }
}
int main() {
    unsigned long long *d;
    hipMalloc(&d, 8192 * 8);
    run<0>("v_exp_f32", d); run<7>("v_exp_f16", d); run<1>("v_fma_f32", d); run<2>("v_pk_fma_f32", d); run<3>("v_cvt_pk_bf16_f32", d);
    run<4>("v_max3_f32", d); run<5>("v_or3_b32", d); run<6>("v_add_f32", d); run<8>("v_mul+v_exp pair", d);
    return 0;
}
