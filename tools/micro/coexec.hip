// How much VALU issue is left on a gfx950 SIMD beside a saturating MFMA stream of ANOTHER wave?  (round 3)
// Every CU runs one 512-thread workgroup (waves w and w + 4 share a SIMD); waves 0-3 run role A, waves 4-7 role B,
// each role a loop of independent instructions of one kind.  Whole-chip launch (256 workgroups), random-ish operands,
// time from s_memtime per wave (median over workgroups) AND the launch's wall time by HIP events; the MFMA-only
// rows calibrate cycles (32 per v_mfma_f32_32x32x16_bf16, 16 per v_mfma_f32_16x16x32_bf16 when back to back).
//   hipcc --offload-arch=gfx950 -O3 -o coexec coexec.hip && ./coexec
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

enum { R_IDLE = 0, R_MFMA32 = 1, R_MFMA16 = 2, R_FMA = 3, R_EXP = 4, R_MIX = 5, R_CVT = 6 };

template <int ROLE>
__device__ __forceinline__ void body(int iters, float seed, float *sink) {
    if constexpr (ROLE == R_MFMA32) {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = seed * (i + j);
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * 0.37f + i * 0.11f); b[i] = (__bf16)(seed * 0.21f - i * 0.07f); }
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        }
        float s = 0; for (int j = 0; j < 4; ++j) s += acc[j][3];
        if (s == 12345.678f) *sink = s;
    } else if constexpr (ROLE == R_MFMA16) {
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) acc[j][i] = seed * (i + j);
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * 0.37f + i * 0.11f); b[i] = (__bf16)(seed * 0.21f - i * 0.07f); }
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        }
        float s = 0; for (int j = 0; j < 8; ++j) s += acc[j][1];
        if (s == 12345.678f) *sink = s;
    } else if constexpr (ROLE == R_FMA || ROLE == R_EXP || ROLE == R_MIX || ROLE == R_CVT) {
        float e[16];
        for (int i = 0; i < 16; ++i) e[i] = seed * 0.01f + i * 0.001f;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (ROLE == R_FMA) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[j]));
                else if (ROLE == R_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j]));
                else if (ROLE == R_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(e[j]));
                else { if (j % 4 == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[j])); }
            }
        }
        float s = 0; for (int i = 0; i < 16; ++i) s += e[i];
        if (s == 12345.678f) *sink = s;
    }
}

template <int RA, int RB>
__global__ void __launch_bounds__(512) k(unsigned long long *out, int itA, int itB, float seed, float *sink) {
    const int wave = threadIdx.x >> 6;
    unsigned long long t0, t1;
    __syncthreads();
    STAMP(t0);
    if (wave < 4) body<RA>(itA, seed, sink); else body<RB>(itB, seed, sink);
    STAMP(t1);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

static const char *NAME[] = {"idle", "mfma32x32x16", "mfma16x16x32", "v_fma", "v_exp", "3fma+1exp", "v_cvt_pk"};
template <int RA, int RB>
void run(unsigned long long *d, float *sink, int itA, int itB, int perA, int perB) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, d, itA, itB, 1.0f, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, d, itA, itB, 1.0f, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int g = 0; g < 256; ++g) { for (int w = 0; w < 4; ++w) a.push_back((double)h[g * 8 + w]); for (int w = 4; w < 8; ++w) b.push_back((double)h[g * 8 + w]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    const double ta = a[a.size() / 2], tb = b[b.size() / 2];
    printf("A=%-13s B=%-13s | A: %8.0f ticks = %6.2f per instr | B: %8.0f ticks = %6.2f per instr | wall %7.1f us\n",
           NAME[RA], NAME[RB], ta, RA ? ta / ((double)itA * perA) : 0.0, tb, RB ? tb / ((double)itB * perB) : 0.0, ms * 1e3);
}

int main() {
    unsigned long long *d; hipMalloc(&d, 256 * 8 * 8);
    float *sink; hipMalloc(&sink, 4);
    const int IT = 20000;
    // calibration: one role alone (the other half of the waves idle)
    run<R_MFMA32, R_IDLE>(d, sink, IT, 0, 4, 1);
    run<R_MFMA16, R_IDLE>(d, sink, IT, 0, 8, 1);
    run<R_FMA, R_IDLE>(d, sink, IT, 0, 16, 1);
    run<R_EXP, R_IDLE>(d, sink, IT, 0, 16, 1);
    run<R_MIX, R_IDLE>(d, sink, IT, 0, 16, 1);
    run<R_CVT, R_IDLE>(d, sink, IT, 0, 16, 1);
    // the same role on both waves of every SIMD
    run<R_MFMA32, R_MFMA32>(d, sink, IT, IT, 4, 4);
    run<R_MFMA16, R_MFMA16>(d, sink, IT, IT, 8, 8);
    run<R_FMA, R_FMA>(d, sink, IT, IT, 16, 16);
    run<R_EXP, R_EXP>(d, sink, IT, IT, 16, 16);
    // matrix wave beside vector wave: B sized to finish first (then longer), so A's slowdown and B's rate both show
    run<R_MFMA32, R_FMA>(d, sink, IT, IT / 2, 4, 16);
    run<R_MFMA32, R_FMA>(d, sink, IT, IT * 2, 4, 16);
    run<R_MFMA32, R_EXP>(d, sink, IT, IT / 2, 4, 16);
    run<R_MFMA32, R_MIX>(d, sink, IT, IT, 4, 16);
    run<R_MFMA16, R_FMA>(d, sink, IT, IT / 2, 8, 16);
    run<R_MFMA16, R_FMA>(d, sink, IT, IT * 2, 8, 16);
    run<R_MFMA16, R_EXP>(d, sink, IT, IT / 2, 8, 16);
    run<R_MFMA16, R_MIX>(d, sink, IT, IT, 8, 16);
    run<R_MFMA16, R_CVT>(d, sink, IT, IT, 8, 16);
    return 0;
}
