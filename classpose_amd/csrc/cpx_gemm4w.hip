// PROTOTYPE (debug build only, round 4): the 256 x 256 x 64 GEMM tile with ONE wave per SIMD.
//
// The production kernel (cpx_gemm.hip: k_gemm256p) runs a 256^2 tile with 8 waves -- two per SIMD, 128 x 64 outputs each, handing the matrix
// pipe to each other in half-phases between barriers.  The vendor's hand-written kernel for the same macro-tile uses 4 waves, one per SIMD,
// 128 x 128 outputs each with the accumulators in the 256 AGPRs: half the fragment bytes per MFMA (32 ds_read_b128 per 128 MFMAs instead of
// 24-28 per 64) and no hand-over inside a SIMD.  This file measures what that structure reaches when it is written in HIP:
//   * 256 threads, wave (wm, wn) = (wave >> 1, wave & 1) owns token rows wm * 128 .. + 127 and channels wn * 128 .. + 127: 8 x 8 accumulators
//     of 16 x 16 (f32x4 each, 256 registers), operand fragments of one k-substep (32 k) = 8 + 8 ds_read_b128, two fragment sets;
//   * the same LDS image as k_gemm256p: per K tile four 16 KB items (X rows 0-127, X rows 128-255, W rows 0-127, W rows 128-255) of 128-byte
//     rows, 16-byte chunk c of row r at position c ^ (r & 7), two K-tile buffers, landed by LDS-DMA (buffer loads, scalar offsets): 16 requests
//     per wave and K tile;
//   * ONE barrier per K tile, in its middle: [reads of k-substep 1 -> set B; 64 MFMAs on set A] wait(B landed, K tile t + 1 landed) barrier
//     [requests of K tile t + 2 into the buffer just consumed; reads of (t + 1, k-substep 0) -> set A; 64 MFMAs on set B];
//   * bias epilogue only (bf16 out), stored straight from the registers (v_permlane16_swap -> 16-byte buffer stores).
// Same accumulation order per output element as k_gemm256p (k-substep 0 then 1 of every K tile) -> bitwise equal results.
// One workgroup per tile (no persistent loop): the measurement is about the main loop, on the K = 4096 shape most of all.
#ifdef CPX_DEBUG
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cpx_common.h"
#include "cpx_internal.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

#define G4_THREADS 256
#define G4_ITEM 16384
#define G4_BUF 65536
#define G4_LDS_BYTES (2 * G4_BUF + 1024)          // + the tile's 256 bias values

struct Gemm4wArgs {
    const unsigned short *A, *W;
    const float *bias;
    unsigned short *out;
    int M, N, K, ld_out, tiles_n, n_blocks;
};

template <int OFF>
__device__ __forceinline__ u32x4 g4_read128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// 8 row blocks of 16 rows x 128 bytes of one 128-row item: one k-substep's fragments
__device__ __forceinline__ void g4_read8(u32x4 (&f)[8], unsigned a) {
    f[0] = g4_read128<0 * 2048>(a); f[1] = g4_read128<1 * 2048>(a); f[2] = g4_read128<2 * 2048>(a); f[3] = g4_read128<3 * 2048>(a);
    f[4] = g4_read128<4 * 2048>(a); f[5] = g4_read128<5 * 2048>(a); f[6] = g4_read128<6 * 2048>(a); f[7] = g4_read128<7 * 2048>(a);
}
template <bool FIRST>
__device__ __forceinline__ void g4_mma(f32x4 (&acc)[8][8], const u32x4 (&fx)[8], const u32x4 (&fw)[8]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[nb]), __builtin_bit_cast(bf16x8, fx[mb]),
                                                                  FIRST ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mb][nb], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}
#define G4_WAIT_LGKM() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__global__ void __launch_bounds__(G4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_gemm4w(Gemm4wArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int K = g.K, nk = K / 64;
    // XCD-aware tile order: workgroup b runs on XCD b % 8; each XCD sweeps 8 x 4 super-tiles of its contiguous tile range
    int tile_m, tile_n;
    {
        const int nblk = g.n_blocks, tiles_m = nblk / g.tiles_n;
        const int v = blockIdx.x, nxcd = 8, q = nblk / nxcd, r = nblk % nxcd, x = v % nxcd;
        const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + v / nxcd;
        if ((tiles_m & 7) == 0 && (g.tiles_n & 3) == 0) {
            const int grp = bid >> 5, w_ = bid & 31, cgn = g.tiles_n >> 2;
            const int rg = grp / cgn, cg = grp - rg * cgn;
            tile_m = rg * 8 + (w_ >> 2); tile_n = cg * 4 + (w_ & 3);
        } else {
            tile_m = bid / g.tiles_n; tile_n = bid - tile_m * g.tiles_n;
        }
    }
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    // ---- LDS-DMA: a request of the workgroup lands 32 rows x 128 bytes (256 lanes x 16 bytes); a 128-row item = 4 requests
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void *)g.A, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void *)g.W, 0, 0x7FFFFFFF, 0x00020000);
    const int srow = tid >> 3, kc = (tid & 7) ^ (srow & 7);                     // row inside a 32-row group, swizzled source chunk
    const unsigned voff = ((unsigned)srow * (unsigned)K + (unsigned)kc * 8u) * 2u;
    const unsigned sX = (unsigned)m0 * (unsigned)K * 2u, sW = (unsigned)n0 * (unsigned)K * 2u, k32b = (unsigned)K * 64u;   // 32 rows in bytes
    char *sdst = smem + wave * 1024;
    auto stage = [&](int t) {                                                    // all four items of K tile t -> buffer t & 1: 16 requests
        char *d = sdst + (t & 1) * G4_BUF;
        const unsigned kb = (unsigned)t * 128u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {                                            // X rows 32 j .. 32 j + 31
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (__attribute__((address_space(3))) void *)(d + j * 4096), 16, voff, sX + (unsigned)j * k32b + kb, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void *)(d + 2 * G4_ITEM + j * 4096), 16, voff, sW + (unsigned)j * k32b + kb, 0, 0);
        }
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    unsigned xa[2][2], wa[2][2];                                                 // [buffer][k-substep]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned sw = (unsigned)(((ks * 4 + fq) ^ (fr & 7)) << 4);
        xa[0][ks] = lds0 + (unsigned)(wm * G4_ITEM + fr * 128) + sw;
        wa[0][ks] = lds0 + (unsigned)((2 + wn) * G4_ITEM + fr * 128) + sw;
        xa[1][ks] = xa[0][ks] + G4_BUF; wa[1][ks] = wa[0][ks] + G4_BUF;
    }
    float bias_v = 0.f;
    if (g.bias) bias_v = g.bias[n0 + tid];
    stage(0);
    if (nk > 1) stage(1);
    *reinterpret_cast<float *>(smem + 2 * G4_BUF + tid * 4) = bias_v;
    if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x4 acc[8][8];
    u32x4 fxA[8], fwA[8], fxB[8], fwB[8];
    g4_read8(fxA, xa[0][0]); g4_read8(fwA, wa[0][0]);
    G4_WAIT_LGKM();
    // one row block of MFMAs (token rows mb * 16 .. + 15 of the wave, all eight channel blocks): 8 MFMAs = 128 cycles of the matrix pipe, during
    // which the wave has ~100 spare issue cycles -- the fragment reads and the LDS-DMA requests are placed between the row blocks
// (inline-asm MFMAs with the accumulator tied to itself in an AGPR quad: with the builtin, hipcc's allocator rotates the 256 loop-carried
// accumulator registers between the unrolled K-tile bodies and pays for it with v_accvgpr_mov chains between the MFMAs)
#define G4_MM(MB, NB, FX, FW, FIRST)                                                                             \
    {                                                                                                           \
        if (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[MB][NB]) : "v"(FW[NB]), "v"(FX[MB])); \
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[MB][NB]) : "v"(FW[NB]), "v"(FX[MB])); \
    }
#define G4_SB() __builtin_amdgcn_sched_barrier(0)
// the fragment set a half multiplies with stays LIVE to the end of that half: otherwise the allocator hands the registers of a fragment whose
// last MFMA has just been issued to the next inline-asm LDS read (legal in program order -- and the outputs were wrong by 2 % when it did)
#define G4_KEEP(FX, FW)                                                                                         \
    { asm volatile("" ::"v"(FX[0]), "v"(FX[1]), "v"(FX[2]), "v"(FX[3]), "v"(FX[4]), "v"(FX[5]), "v"(FX[6]), "v"(FX[7]),                      \
                   "v"(FW[0]), "v"(FW[1]), "v"(FW[2]), "v"(FW[3]), "v"(FW[4]), "v"(FW[5]), "v"(FW[6]), "v"(FW[7])); }
#define G4_DMAX(T2, I)                                                                                          \
    { __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (__attribute__((address_space(3))) void *)(sdst + ((T2) & 1) * G4_BUF + (I) * 4096), 16, voff, sX + (unsigned)(I) * k32b + (unsigned)(T2) * 128u, 0, 0); G4_SB(); }
#define G4_DMAW(T2, I)                                                                                          \
    { __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void *)(sdst + ((T2) & 1) * G4_BUF + 2 * G4_ITEM + (I) * 4096), 16, voff, sW + (unsigned)(I) * k32b + (unsigned)(T2) * 128u, 0, 0); G4_SB(); }
// one row block (8 MFMAs) with ONE other instruction behind every second MFMA: a lone wave issues in order, so a fragment read or an LDS-DMA
// request only hides in the ~12 cycles the matrix pipe still needs for the MFMA in front of it
#define G4_HALF1(B, FIRST, I)                                                                                   \
    {                                                                                                           \
        G4_MM(I, 0, fxA, fwA, FIRST) G4_MM(I, 1, fxA, fwA, FIRST) fxB[I] = g4_read128<(I) * 2048>(xa[B][1]);    \
        G4_MM(I, 2, fxA, fwA, FIRST) G4_MM(I, 3, fxA, fwA, FIRST) G4_MM(I, 4, fxA, fwA, FIRST) G4_MM(I, 5, fxA, fwA, FIRST) \
        fwB[I] = g4_read128<(I) * 2048>(wa[B][1]);                                                              \
        G4_MM(I, 6, fxA, fwA, FIRST) G4_MM(I, 7, fxA, fwA, FIRST) G4_SB();                                      \
    }
// KIND (compile time, no branch inside a K-tile body): 0 = steady state (t + 2 < nk), 1 = K tile nk - 2 (nothing left to request), 2 = the last K tile
#define G4_HALF2(T, B, I, KIND)                                                                                 \
    {                                                                                                           \
        G4_MM(I, 0, fxB, fwB, false) G4_MM(I, 1, fxB, fwB, false) if ((KIND) < 2) fxA[I] = g4_read128<(I) * 2048>(xa[(B) ^ 1][0]); \
        G4_MM(I, 2, fxB, fwB, false) G4_MM(I, 3, fxB, fwB, false) G4_SB(); if ((KIND) < 1) G4_DMAX((T) + 2, I)  \
        G4_MM(I, 4, fxB, fwB, false) G4_MM(I, 5, fxB, fwB, false) if ((KIND) < 2) fwA[I] = g4_read128<(I) * 2048>(wa[(B) ^ 1][0]); \
        G4_MM(I, 6, fxB, fwB, false) G4_MM(I, 7, fxB, fwB, false) G4_SB(); if ((KIND) < 1) G4_DMAW((T) + 2, I)  \
    }
#define G4_KTILE(T, B, FIRST, KIND)                                                                            \
    {                                                                                                          \
        const int t_ = (T);                                                                                    \
        G4_HALF1(B, FIRST, 0) G4_HALF1(B, FIRST, 1) G4_HALF1(B, FIRST, 2) G4_HALF1(B, FIRST, 3)                \
        G4_HALF1(B, FIRST, 4) G4_HALF1(B, FIRST, 5) G4_HALF1(B, FIRST, 6) G4_HALF1(B, FIRST, 7)                \
        G4_KEEP(fxA, fwA)                                                                                      \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      /* set B and K tile t + 1 have landed */ \
        __builtin_amdgcn_s_barrier();                                                                          \
        G4_SB();                                                                                               \
        G4_HALF2(t_, B, 0, KIND) G4_HALF2(t_, B, 1, KIND) G4_HALF2(t_, B, 2, KIND) G4_HALF2(t_, B, 3, KIND)    \
        G4_HALF2(t_, B, 4, KIND) G4_HALF2(t_, B, 5, KIND) G4_HALF2(t_, B, 6, KIND) G4_HALF2(t_, B, 7, KIND)    \
        G4_KEEP(fxB, fwB)                                                                                      \
        G4_WAIT_LGKM();                                                                                        \
    }
    // (nk >= 4, even: the launcher checks)
    G4_KTILE(0, 0, true, 0)
    G4_KTILE(1, 1, false, 0)
    for (int t = 2; t + 2 < nk; t += 2) {
        G4_KTILE(t, 0, false, 0)
        G4_KTILE(t + 1, 1, false, 0)
    }
    G4_KTILE(nk - 2, 0, false, 1)
    G4_KTILE(nk - 1, 1, false, 2)
#undef G4_HALF1
#undef G4_HALF2
#undef G4_MM
#undef G4_DMAX
#undef G4_DMAW
#undef G4_SB
#undef G4_KEEP
#undef G4_KTILE
    // the last inline-asm MFMAs' results: the compiler does not see the MFMA -> v_accvgpr_read hazard, and without the scheduling barriers it
    // hoisted some of the epilogue's accumulator reads in front of the pad (one element of two channel blocks lost part of the last K tile)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // ... and every accumulator is re-defined BEHIND the pad (empty asm, "+a"): an accumulator's last MFMA is issued up to 63 MFMAs before the
    // end of the loop, and the compiler is free to read it right there, one instruction behind an MFMA that takes 32 cycles
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
        asm volatile("" : "+a"(acc[mb][0]), "+a"(acc[mb][1]), "+a"(acc[mb][2]), "+a"(acc[mb][3]), "+a"(acc[mb][4]), "+a"(acc[mb][5]), "+a"(acc[mb][6]), "+a"(acc[mb][7]));
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: bias, one conversion per pair, lane rows swapped into 16-byte pieces, buffer stores (as the direct-store epilogue of k_gemm256p)
    const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc((void *)g.out, 0, 0x7FFFFFFF, 0x00020000);
    const unsigned ldb = (unsigned)g.ld_out * 2u;
    const unsigned ovoff = (unsigned)(wm * 128 + fr) * ldb + (unsigned)(wn * 128 + ((fq & 1) << 4) + ((fq >> 1) << 3)) * 2u;
    const unsigned so0 = (unsigned)m0 * ldb + (unsigned)n0 * 2u;
#pragma unroll
    for (int np = 0; np < 4; ++np) {                                             // channel block pairs (2 np, 2 np + 1): 32 channels
        const float4 b0 = *reinterpret_cast<const float4 *>(smem + 2 * G4_BUF + (wn * 128 + (2 * np) * 16 + fq * 4) * 4);
        const float4 b1 = *reinterpret_cast<const float4 *>(smem + 2 * G4_BUF + (wn * 128 + (2 * np + 1) * 16 + fq * 4) * 4);
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const f32x4 v0 = acc[mb][2 * np], v1 = acc[mb][2 * np + 1];
            const f32x2_t p00 = {v0[0] + b0.x, v0[1] + b0.y}, p01 = {v0[2] + b0.z, v0[3] + b0.w};
            const f32x2_t p10 = {v1[0] + b1.x, v1[1] + b1.y}, p11 = {v1[2] + b1.z, v1[3] + b1.w};
            const unsigned a0 = __builtin_bit_cast(unsigned, __builtin_convertvector(p00, bf16x2_t)), a1 = __builtin_bit_cast(unsigned, __builtin_convertvector(p01, bf16x2_t));
            const unsigned c0 = __builtin_bit_cast(unsigned, __builtin_convertvector(p10, bf16x2_t)), c1 = __builtin_bit_cast(unsigned, __builtin_convertvector(p11, bf16x2_t));
            const auto r0 = __builtin_amdgcn_permlane16_swap(a0, c0, false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(a1, c1, false, false);
            const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, ovoff, so0 + (unsigned)(mb * 16) * ldb + (unsigned)np * 64u, 0);
        }
    }
}

// out[M][ld_out] (bf16) = A[M][K] . W[N][K]^T + bias; M, N multiples of 256, K a multiple of 128, every operand below 2 GiB
extern "C" int cpx_gemm4w(const void *A, const void *W, int M, int N, int K, const float *bias, void *out, int ld_out, void *stream) {
    if (M % 256 || N % 256 || K % 128 || K < 256) return CPX_EINVAL;
    if ((size_t)M * K * 2 >= ((size_t)1 << 31) || (size_t)N * K * 2 >= ((size_t)1 << 31) || (size_t)M * ld_out * 2 >= ((size_t)1 << 31)) return CPX_EINVAL;
    Gemm4wArgs a;
    a.A = (const unsigned short *)A; a.W = (const unsigned short *)W; a.bias = bias; a.out = (unsigned short *)out;
    a.M = M; a.N = N; a.K = K; a.ld_out = ld_out; a.tiles_n = N / 256; a.n_blocks = (M / 256) * (N / 256);
    static CpxOncePerDevice once;
    once([] { (void)hipFuncSetAttribute((const void *)k_gemm4w, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES); });
    hipLaunchKernelGGL(k_gemm4w, dim3(a.n_blocks), dim3(G4_THREADS), G4_LDS_BYTES, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : CPX_EHIP;
}
#endif  // CPX_DEBUG
