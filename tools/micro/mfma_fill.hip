// What does ONE wave per SIMD get when it interleaves its own 32x32x16 MFMAs with vector "fillers"?  (round 4, attention)
// 256 workgroups x 256 threads (one wave per SIMD), every wave runs `iters` x 8 gaps; a gap = one v_mfma_f32_32x32x16_bf16
// followed by a filler group.  Variants: accumulators in AGPRs or VGPRs, independent or chained the way k_attention2q chains
// them, filler groups of the softmax (2 v_fma, 2 v_exp, 1 v_cvt_pk), of plain FMAs, packed FMAs, or nothing.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_fill mfma_fill.hip && ./mfma_fill
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
enum { F_NONE = 0, F_SOFTMAX, F_FMA5, F_EXP2, F_PK2, F_FMA2, F_SOFTMAX_PK, F_N };
enum { A_AGPR = 0, A_VGPR, A_CHAIN, A_NOMFMA, A_N };

template <int FILL>
__device__ __forceinline__ void filler(float (&e)[8], f32x2 &pp) {
    if constexpr (FILL == F_SOFTMAX) {
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e[0]) : "v"(e[4]), "v"(e[5]), "v"(e[6]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e[1]) : "v"(e[4]), "v"(e[5]), "v"(e[7]));
        asm volatile("v_exp_f32 %0, %1" : "=v"(e[2]) : "v"(e[0]));
        asm volatile("v_exp_f32 %0, %1" : "=v"(e[3]) : "v"(e[1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(e[4]) : "v"(e[2]), "v"(e[3]));
    } else if constexpr (FILL == F_SOFTMAX_PK) {
        asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(pp) : "v"(pp));
        asm volatile("v_exp_f32 %0, %1" : "=v"(e[2]) : "v"(e[0]));
        asm volatile("v_exp_f32 %0, %1" : "=v"(e[3]) : "v"(e[1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(e[4]) : "v"(e[2]), "v"(e[3]));
    } else if constexpr (FILL == F_FMA5) {
#pragma unroll
        for (int j = 0; j < 5; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[j]));
    } else if constexpr (FILL == F_FMA2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[j]));
    } else if constexpr (FILL == F_EXP2) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(e[0]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(e[1]));
    } else if constexpr (FILL == F_PK2) {
        asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(pp));
    }
}

template <int ACC, int FILL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k(int iters, float *sink, long long *cyc) {
    f32x16 acc[8], init;
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.001f * (i + j + threadIdx.x);
    for (int i = 0; i < 16; ++i) init[i] = 0.5f * i;
    u32x4 a = {0x3F803F80u + threadIdx.x, 0x3F803F80u, 0x3F003F00u, 0x3E803E80u}, b = {0x3F003F00u, 0x3F803F80u + threadIdx.x, 0x3F003F00u, 0x3E803E80u};
    float e[8];
    for (int i = 0; i < 8; ++i) e[i] = 0.01f * (i + 1) + 0.001f * (threadIdx.x & 63);
    f32x2 pp = {e[0], e[1]};
    long long t0, t1;
    __syncthreads();
    STAMP(t0);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if constexpr (ACC == A_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(a), "v"(b));
            else if constexpr (ACC == A_VGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
            else if constexpr (ACC == A_CHAIN) {
                // the k_attention2q pattern: S (VGPR) = init at gap 0, chained at gaps 2, 4, 6; O0 / O1 (AGPR) alternate on the odd gaps
                if (g == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc[0]) : "v"(a), "v"(b), "v"(init));
                else if ((g & 1) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[1 + ((g >> 1) & 1)]) : "v"(a), "v"(b));
            }
            filler<FILL>(e, pp);
        }
    }
    STAMP(t1);
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][3] + e[j];
    s += pp.x + pp.y;
    if (s == 12345.678f) *sink = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// the exact half step of k_attention2q: 10 MFMAs (S chain in VGPRs from an init operand, O0 / O1 / L in AGPRs) with the
// software-pipelined softmax stream (per gap: 2 fma of pair j + 1, 2 exp of pair j, cvt_pk of pair j - 1), the OR tree and the
// vote's compare.  MODE 0: as in the kernel; 1: no MFMAs; 2: MFMAs only; 3: all ten MFMAs on independent AGPR accumulators;
// 4: as 0 but without the s_nop / compare tail
template <int MODE, int UNROLL = 1>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) khalf(int iters, float *sink, long long *cyc) {
    f32x16 S[2], O0, O1, Lr, GW, X[6];
    for (int i = 0; i < 16; ++i) { S[0][i] = 0.01f * i; S[1][i] = 0.02f * i; O0[i] = O1[i] = Lr[i] = 0.f; GW[i] = 0.1f * i; for (int j = 0; j < 6; ++j) X[j][i] = 0.f; }
    u32x4 a = {0x3F803F80u + threadIdx.x, 0x3F803F80u, 0x3F003F00u, 0x3E803E80u}, b = {0x3F003F00u, 0x3F803F80u + threadIdx.x, 0x3F003F00u, 0x3E803E80u};
    const float cexp = 0.18f; float off = -0.5f - 0.001f * (threadIdx.x & 63);
    asm volatile("" : "+v"(off));
    unsigned acc_or = 0;
    long long t0, t1;
    __syncthreads();
    STAMP(t0);
#define MFI(D, C) do { if (MODE != 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(D) : "v"(a), "v"(b), "v"(C)); } while (0)
#define MFV(D) do { if (MODE != 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "v"(a), "v"(b)); } while (0)
#define MFA(D) do { if (MODE != 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(D) : "v"(a), "v"(b)); } while (0)
#define FMA2(A, j) do { if (MODE != 2) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(A.x) : "v"(Sc[2 * (j)]), "s"(cexp), "v"(off)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(A.y) : "v"(Sc[2 * (j) + 1]), "s"(cexp), "v"(off)); } } while (0)
#define EXP2(E0, E1, A) do { if (MODE != 2) { asm volatile("v_exp_f32 %0, %1" : "=v"(E0) : "v"(A.x)); asm volatile("v_exp_f32 %0, %1" : "=v"(E1) : "v"(A.y)); } } while (0)
#define CVT(P, E0, E1) do { if (MODE != 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(P) : "v"(E0), "v"(E1)); } while (0)
#define OR3(R, A_, B_, C_) do { if (MODE != 2) asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(R) : "v"(A_), "v"(B_), "v"(C_)); } while (0)
#pragma unroll 1
    for (int it = 0; it < iters; it += UNROLL) {
#pragma unroll
        for (int h = 0; h < 2 * UNROLL; ++h) {
            f32x16 &Sc = S[h & 1], &Sx = S[(h & 1) ^ 1];
            f32x2 a0 = {0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
            float e0 = 0, e1 = 0, e2 = 0, e3 = 0; unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, u0 = 0, u1 = 0;
            if (MODE == 3) MFA(X[0]); else MFI(Sx, GW);
            FMA2(a0, 0); FMA2(a1, 1); EXP2(e0, e1, a0);
            MFA(O0);
            FMA2(a2, 2); EXP2(e2, e3, a1); CVT(pk[0], e0, e1);
            if (MODE == 3) MFA(X[1]); else MFV(Sx);
            FMA2(a3, 3); EXP2(e0, e1, a2); CVT(pk[1], e2, e3);
            MFA(O1);
            FMA2(a4, 4); EXP2(e2, e3, a3); CVT(pk[2], e0, e1);
            if (MODE == 3) MFA(X[2]); else MFV(Sx);
            FMA2(a5, 5); EXP2(e0, e1, a4); CVT(pk[3], e2, e3);
            if (MODE == 3) MFA(X[4]); else MFA(O0);
            FMA2(a6, 6); EXP2(e2, e3, a5); CVT(pk[4], e0, e1);
            if (MODE == 3) MFA(X[3]); else MFV(Sx);
            FMA2(a7, 7); EXP2(e0, e1, a6); CVT(pk[5], e2, e3);
            if (MODE == 3) MFA(X[5]); else MFA(O1);
            EXP2(e2, e3, a7); CVT(pk[6], e0, e1); OR3(u0, pk[0], pk[1], pk[2]);
            MFA(Lr);
            OR3(u1, pk[3], pk[4], pk[5]); CVT(pk[7], e2, e3);
            if (MODE == 3) MFA(O0); else MFA(Lr);
            OR3(u0, u0, u1, pk[6]);
            if (MODE == 0) {
                unsigned m;
                asm volatile("v_or_b32 %0, %1, %2" : "=v"(m) : "v"(u0), "v"(pk[7]));
                acc_or |= m;
                if (__builtin_expect(__any((m & 0x40004000u) == 0x40004000u), 0)) off += 1.0f;
            } else acc_or |= u0 | pk[7];
        }
    }
    STAMP(t1);
    float s = 0;
    for (int i = 0; i < 16; ++i) s += S[0][i] + S[1][i] + O0[i] + O1[i] + Lr[i] + X[0][i] + X[1][i] + X[2][i] + X[3][i] + X[4][i] + X[5][i];
    if (s == 12345.678f || acc_or == 0x12345u) *sink = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE, int UNROLL = 1>
static void runh(const char *name, int iters, float *sink, long long *dcyc) {
    hipLaunchKernelGGL((khalf<MODE, UNROLL>), dim3(256), dim3(256), 0, 0, iters / 8, sink, dcyc);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((khalf<MODE, UNROLL>), dim3(256), dim3(256), 0, 0, iters, sink, dcyc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(1024);
    (void)hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-58s %7.1f cycles per half step (10 gaps)\n", name, (double)h[512] / iters / 2);
}

template <int ACC, int FILL>
static void run(const char *name, int iters, float *sink, long long *dcyc) {
    hipLaunchKernelGGL((k<ACC, FILL>), dim3(256), dim3(256), 0, 0, iters / 8, sink, dcyc);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<ACC, FILL>), dim3(256), dim3(256), 0, 0, iters, sink, dcyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(1024);
    (void)hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[512];
    printf("%-58s %7.1f cycles per gap   (wall %.3f ms, %.2f GHz)\n", name, med / iters / 8, ms, med / (ms * 1e6));
}
int main() {
    float *sink; long long *dcyc;
    (void)hipMalloc(&sink, 4); (void)hipMalloc(&dcyc, 1024 * 8);
    const int iters = 4000;
    run<A_AGPR, F_NONE>("MFMA (8 AGPR accumulators) alone", iters, sink, dcyc);
    run<A_VGPR, F_NONE>("MFMA (4 VGPR accumulators) alone", iters, sink, dcyc);
    run<A_CHAIN, F_NONE>("MFMA in the k_attention2q chain pattern alone", iters, sink, dcyc);
    run<A_NOMFMA, F_SOFTMAX>("no MFMA, softmax group (2 fma, 2 exp, 1 cvt_pk)", iters, sink, dcyc);
    run<A_NOMFMA, F_FMA5>("no MFMA, 5 fma", iters, sink, dcyc);
    run<A_AGPR, F_FMA2>("MFMA (AGPR) + 2 fma", iters, sink, dcyc);
    run<A_AGPR, F_FMA5>("MFMA (AGPR) + 5 fma", iters, sink, dcyc);
    run<A_AGPR, F_EXP2>("MFMA (AGPR) + 2 exp", iters, sink, dcyc);
    run<A_AGPR, F_PK2>("MFMA (AGPR) + 1 pk_fma", iters, sink, dcyc);
    run<A_AGPR, F_SOFTMAX>("MFMA (AGPR) + softmax group", iters, sink, dcyc);
    run<A_AGPR, F_SOFTMAX_PK>("MFMA (AGPR) + softmax group with pk_fma", iters, sink, dcyc);
    run<A_VGPR, F_SOFTMAX>("MFMA (VGPR) + softmax group", iters, sink, dcyc);
    run<A_CHAIN, F_SOFTMAX>("MFMA (k_attention2q chain) + softmax group", iters, sink, dcyc);
    runh<0>("half step as in k_attention2q", 2000, sink, dcyc);
    runh<4>("half step, no vote tail", 2000, sink, dcyc);
    runh<1>("half step, vector stream only", 2000, sink, dcyc);
    runh<2>("half step, MFMAs only", 2000, sink, dcyc);
    runh<3>("half step, ten independent AGPR MFMAs", 2000, sink, dcyc);
    runh<4, 4>("half step, no vote tail, loop body x4 (~4 KB)", 2000, sink, dcyc);
    runh<4, 8>("half step, no vote tail, loop body x8 (~8 KB)", 2000, sink, dcyc);
    runh<1, 8>("vector stream only, loop body x8", 2000, sink, dcyc);
    runh<2, 8>("MFMAs only, loop body x8", 2000, sink, dcyc);
    runh<0, 8>("half step as in k_attention2q, loop body x8", 2000, sink, dcyc);
    return 0;
}
