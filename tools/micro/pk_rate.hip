// What does a packed f32 VALU instruction cost on a gfx950 SIMD?  (round 4)
// The GEMM epilogues (folded LayerNorm, bias, 2^P5(|x|) GELU) are pure-VALU phases with two waves per SIMD and no MFMA
// in flight.  This measures, for exactly that situation (whole-chip launch, one 256- or 512-thread workgroup per CU =
// one or two waves per SIMD, every wave the same role), the cycles per instruction of
//   v_fma_f32 / v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_exp_f32 / v_max_f32 / v_and_b32 / v_cvt_pk_bf16_f32
// as 16 independent chains (issue-bound) and as ONE dependent chain (latency-bound), and two whole GELU bodies on 16
// values per lane: the scalar form hipcc emits today and the hand-packed form (explicit pairs).
//   hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

enum { R_FMA = 0, R_PKFMA, R_PKMUL, R_PKADD, R_EXP, R_MAX, R_AND, R_CVT, R_FMA_DEP, R_PKFMA_DEP, R_GELU_S, R_GELU_P, R_MAXI, R_MOV, R_CVTH, R_ADD, R_MAX3, R_N };
static const char *NAMES[R_N] = {"v_fma_f32 x16 indep", "v_pk_fma_f32 x8 indep", "v_pk_mul_f32 x8 indep", "v_pk_add_f32 x8 indep",
                                 "v_exp_f32 x16 indep", "v_max_f32 x16 indep", "v_and_b32 x16 indep", "v_cvt_pk_bf16 x16 indep",
                                 "v_fma_f32 1 chain", "v_pk_fma_f32 1 chain", "GELU scalar (16 values)", "GELU packed (16 values)",
                                 "v_max_i32 x16 indep", "v_mov_b32 x16 indep", "v_cvt_pk_f16_f32 x16", "v_add_f32 x16 indep", "v_max3_f32 x16 indep"};
// instructions per iteration (for the per-instruction column) and elements per iteration (for the per-element column)
static const int NINSTR[R_N] = {16, 8, 8, 8, 16, 16, 16, 16, 16, 8, 16 * 9, 8 * 11, 16, 16, 16, 16, 16};

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// d = a * b(broadcast scalar in an SGPR pair's low half) + c(sgpr broadcast)
__device__ __forceinline__ f32x2 pk_fma_vvs(f32x2 a, f32x2 b, float c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}

template <int ROLE>
__device__ __forceinline__ void body(int iters, float seed, float *sink) {
    float e[16];
    for (int i = 0; i < 16; ++i) e[i] = seed * 0.01f + i * 0.001f;
    if constexpr (ROLE == R_GELU_S) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float x = e[j];
                float p;
                asm volatile("v_fma_f32 %0, |%1|, %2, %3" : "=v"(p) : "v"(x), "s"(-4.732933965e-04f), "v"(7.084452800e-03f));
                asm volatile("v_fma_f32 %0, %0, |%1|, %2" : "+v"(p) : "v"(x), "s"(-5.182713611e-02f));
                asm volatile("v_fma_f32 %0, %0, |%1|, %2" : "+v"(p) : "v"(x), "s"(-4.599926953e-01f));
                asm volatile("v_fma_f32 %0, %0, |%1|, %2" : "+v"(p) : "v"(x), "s"(-1.150787739e+00f));
                asm volatile("v_fma_f32 %0, %0, |%1|, %2" : "+v"(p) : "v"(x), "s"(-1.000037638e+00f));
                asm volatile("v_exp_f32 %0, %0" : "+v"(p));
                float a, m;
                asm volatile("v_and_b32 %0, 0x7fffffff, %1" : "=v"(a) : "v"(x));
                asm volatile("v_max_f32 %0, 0, %1" : "=v"(m) : "v"(x));
                asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(e[j]) : "v"(a), "v"(p), "v"(m));
            }
        }
    } else if constexpr (ROLE == R_GELU_P) {
        const f32x2 C5 = {-4.732933965e-04f, -4.732933965e-04f}, C4 = {7.084452800e-03f, 7.084452800e-03f},
                    C3 = {-5.182713611e-02f, -5.182713611e-02f}, C2 = {-4.599926953e-01f, -4.599926953e-01f},
                    C1 = {-1.150787739e+00f, -1.150787739e+00f}, C0 = {-1.000037638e+00f, -1.000037638e+00f};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                f32x2 x = {e[j], e[j + 1]}, a, m, p;
                asm volatile("v_and_b32 %0, 0x7fffffff, %1" : "=v"(a.x) : "v"(x.x));
                asm volatile("v_and_b32 %0, 0x7fffffff, %1" : "=v"(a.y) : "v"(x.y));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "v"(a), "s"(C5), "v"(C4));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(a), "s"(C3));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(a), "s"(C2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(a), "s"(C1));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(a), "s"(C0));
                asm volatile("v_exp_f32 %0, %0" : "+v"(p.x));
                asm volatile("v_exp_f32 %0, %0" : "+v"(p.y));
                asm volatile("v_max_f32 %0, 0, %1" : "=v"(m.x) : "v"(x.x));
                asm volatile("v_max_f32 %0, 0, %1" : "=v"(m.y) : "v"(x.y));
                f32x2 r;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(p), "v"(m));
                e[j] = r.x; e[j + 1] = r.y;
            }
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            if constexpr (ROLE == R_FMA_DEP) {
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[0]));
            } else if constexpr (ROLE == R_PKFMA_DEP) {
                f32x2 v = {e[0], e[1]};
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
                e[0] = v.x; e[1] = v.y;
            } else if constexpr (ROLE == R_PKFMA || ROLE == R_PKMUL || ROLE == R_PKADD) {
                f32x2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j].x = e[2 * j]; v[j].y = e[2 * j + 1]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (ROLE == R_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v[j]));
                    else if (ROLE == R_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(v[j]));
                    else asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(v[j]));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { e[2 * j] = v[j].x; e[2 * j + 1] = v[j].y; }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (ROLE == R_FMA) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[j]));
                    else if (ROLE == R_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j]));
                    else if (ROLE == R_MAX) asm volatile("v_max_f32 %0, 0, %0" : "+v"(e[j]));
                    else if (ROLE == R_AND) asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(e[j]));
                    else if (ROLE == R_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(e[j]));
                    else if (ROLE == R_MAXI) asm volatile("v_max_i32 %0, 0, %0" : "+v"(e[j]));
                    else if (ROLE == R_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(e[j]) : "v"(e[(j + 1) & 15]));
                    else if (ROLE == R_CVTH) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(e[j]));
                    else if (ROLE == R_ADD) asm volatile("v_add_f32 %0, %0, %0" : "+v"(e[j]));
                    else if (ROLE == R_MAX3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(e[j]));
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += e[i];
    if (s == 12345.678f) *sink = s;
}

template <int ROLE>
__global__ void __launch_bounds__(1024) k(int iters, float *sink, long long *cyc) {
    const float seed = 1.0f + (threadIdx.x & 63) * 0.01f;
    long long t0, t1;
    __syncthreads();
    STAMP(t0);
    body<ROLE>(iters, seed, sink);
    STAMP(t1);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int ROLE>
static void run(int threads, int iters, float *sink, long long *dcyc) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256;
    hipLaunchKernelGGL(k<ROLE>, dim3(blocks), dim3(threads), 0, 0, iters / 8, sink, dcyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<ROLE>, dim3(blocks), dim3(threads), 0, 0, iters, sink, dcyc);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const int nw = threads / 64;
    std::vector<long long> h(blocks * 16);
    hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<long long> v;
    for (int bI = 0; bI < blocks; ++bI) for (int w = 0; w < nw; ++w) v.push_back(h[bI * 16 + w]);
    std::sort(v.begin(), v.end());
    const double med = (double)v[v.size() / 2], per_it = med / iters;
    printf("%-26s %d wave(s)/SIMD: %8.1f cyc/iter  %6.2f cyc/instr/wave  %6.2f SIMD cyc per value   wall %.3f ms (%.2f GHz)\n",
           NAMES[ROLE], nw / 4, per_it, per_it / NINSTR[ROLE], per_it / 16.0 / (nw / 4), ms, med / (ms * 1e6));
}

#define RUN(R) run<R>(threads, iters, sink, dcyc)
int main() {
    float *sink; long long *dcyc;
    hipMalloc(&sink, 4); hipMalloc(&dcyc, 256 * 16 * 8);
    const int iters = 20000;
    for (int threads : {256, 512, 768, 1024}) {
        RUN(R_FMA); RUN(R_PKFMA); RUN(R_PKMUL); RUN(R_PKADD); RUN(R_EXP); RUN(R_MAX); RUN(R_AND); RUN(R_CVT);
        RUN(R_FMA_DEP); RUN(R_PKFMA_DEP); RUN(R_GELU_S); RUN(R_GELU_P);
        RUN(R_MAXI); RUN(R_MOV); RUN(R_CVTH); RUN(R_ADD); RUN(R_MAX3);
        printf("\n");
    }
    return 0;
}
