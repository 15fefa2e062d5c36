// The 256 x 256 x 64 GEMM tile with ONE wave per SIMD (round 5: persistent, three fragment sets, requests two K tiles ahead).
//
// The 8-wave kernel (cpx_gemm.hip: k_gemm256p) runs two waves per SIMD, 128 x 64 outputs each, handing the matrix pipe to each other in
// half-phases between barriers.  Here a workgroup is 256 threads: wave (wm, wn) = (wave >> 1, wave & 1) owns token rows wm * 128 .. + 127 and
// channels wn * 128 .. + 127 -- 8 x 8 accumulators of 16 x 16 (f32x4 each) in the 256 AGPRs, half the fragment bytes per MFMA (32 ds_read_b128
// per 128 MFMAs instead of 24-28 per 64) and no hand-over inside a SIMD.
//
// LDS image = k_gemm256p's: per K tile four 16 KB items (X rows 0-127, X rows 128-255, W rows 0-127, W rows 128-255) of 128-byte rows, 16-byte
// chunk c of row r at position c ^ (r & 7), two K-tile buffers, landed by LDS-DMA (buffer loads with scalar offsets, 16 requests per wave and
// K tile).  What the round-4 prototype of this file lacked was prefetch distance: it read the k-substep-1 fragments of K tile t during the
// first half of t, so buffer t & 1 was free only at mid(t), its refill was requested during the second half of t and had to have landed by
// mid(t + 1): 0.5 - 1 K tile (~0.5 - 1 us), less than a loaded HBM / Infinity Cache round trip.  Here:
//   * THREE fragment sets (192 VGPRs beside the 256 AGPRs): P always holds k-substep 0, the k-substep-1 set alternates between Q (even K
//     tiles) and R (odd) -- period 2, like the buffers.  BOTH sets of K tile t + 1 are read during the second half of K tile t (32 reads
//     beside 64 MFMAs), so buffer (t + 1) & 1 is free at end(t);
//   * the requests of K tile t + 2 go into buffer t & 1 during K tile t itself: the 8 X requests in its first half, the 8 W requests in its
//     second, one behind every eighth MFMA (all 16 in the first half -- one per fourth MFMA, the texture path saturated for half of the time and
//     idle for the other -- measured 1 - 4 % slower: profiles/r05_ablate_gemm4w.txt), waited for at mid(t + 1) behind a counted vmcnt(8) (the 8
//     youngest = the X requests of K tile t + 3): 1 - 3 k-substeps of cover, the long end for the activation rows, which are the operand that
//     comes from HBM or the Infinity Cache;
//   * two barriers per K tile: mid(t) [K tile t + 1 has landed for everybody -> reads], end(t) [everybody has read buffer (t + 1) & 1 ->
//     the requests of K tile t + 1 may overwrite it];
//   * persistent: the request stream simply runs on across output tiles (the last two K tiles of a tile request K tiles 0 / 1 of the NEXT
//     tile, the last half reads the next tile's first fragments), the epilogue is a direct store from the registers and touches neither
//     buffer, and the per-tile vectors (bias, LayerNorm row statistics, column sums) arrive by LDS-DMA in a 10 KB tail: no ordinary global
//     load anywhere, so hipcc never drains the request queue.
// Same accumulation order per output element as k_gemm256p (k-substep 0 then 1 of every K tile) and the same epilogue arithmetic -> bitwise
// equal results.  Epilogues: bias only (the debug-build entry point cpx_gemm4w: A/B tools, bitwise tests) and, PRODUCTION, folded LayerNorm +
// bias + erf-GELU (mlp.lin1 of the bf16 network).  One wave per SIMD is the one regime where v_pk_fma_f32 doubles the f32 rate
// (tools/micro/pk_rate.hip), so the LayerNorm fold and the GELU polynomial are written on float2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cpx_common.h"
#include "cpx_internal.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
// two f32 -> one dword of two halves (round to nearest even), bf16 or fp16
template <bool F16>
__device__ __forceinline__ unsigned g4_pack2(f32x2_t v) {
    if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

#define G4_THREADS 256
#define G4_ITEM 16384
#define G4_BUF 65536
#define G4_TAIL (2 * G4_BUF)                      // [0, 8192) raw LayerNorm row statistics, [8192, 9216) bias, [9216, 10240) column sums
#define G4_TAIL_BIAS (G4_TAIL + 8192)
#define G4_TAIL_CSUM (G4_TAIL + 9216)
#define G4_LDS_BYTES (G4_TAIL + 10240)

struct Gemm4wArgs {
    const unsigned short *A, *W;
    const float *bias, *ln_stats, *ln_colsum;     // ln_*: folded LayerNorm of the K = 1024 input rows (G4_EPI_GELU_LN), as in cpx_gemm.hip
    unsigned short *out;
    const unsigned short *resid;                  // G4_EPI_RESID_STATS: residual rows [M][ld_out] (may be `out` itself: every lane reads its 16 bytes before it writes them)
    float *stats_out;                             // G4_EPI_RESID_STATS: [M][4][2] partial (sum, sum of squares) of the OUTPUT rows, slot = column tile (N = 1024)
    int M, N, K, ld_out, tiles_n, n_blocks;
};
#define G4_EPI_BIAS 0
#define G4_EPI_GELU_LN 1
#define G4_EPI_RESID_STATS 2

template <int OFF>
__device__ __forceinline__ u32x4 g4_read128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
#define G4_SB() __builtin_amdgcn_sched_barrier(0)
#define G4_LDSP(p) ((__attribute__((address_space(3))) void *)(p))

// a 16-byte buffer load the COMPILER does not see as a load (with an LDS-DMA in flight hipcc waits vmcnt(0) in front of the first use of any
// ordinary load): the caller waits with g4_vmwait<N>, which carries the destinations as "+v" operands so that no consumer can move in front
__device__ __forceinline__ u32x4 g4_load128(u32x4 rsrc, unsigned voff, unsigned soff) {
    u32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void g4_vmwait(u32x4 (&r)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(N) : "memory");
}
// x + (x of the lane 32 / 16 lanes away): v_permlane32_swap / v_permlane16_swap on two copies leave (lo, lo) and (hi, hi) halves / row pairs
__device__ __forceinline__ float g4_add_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __fadd_rn(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float g4_add_xor16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __fadd_rn(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// erf-GELU of cpx_gemm.hip (gelu_erf: max(x, 0) - |x| 2^P5(|x|)) on two values: the polynomial as v_pk_fma_f32, the rest per value
__device__ __forceinline__ f32x2_t g4_gelu2(f32x2_t x) {
    const f32x2_t ax = __builtin_elementwise_abs(x);
    f32x2_t p = {-4.732933965e-04f, -4.732933965e-04f};
    p = __builtin_elementwise_fma(p, ax, (f32x2_t){7.084452800e-03f, 7.084452800e-03f});
    p = __builtin_elementwise_fma(p, ax, (f32x2_t){-5.182713611e-02f, -5.182713611e-02f});
    p = __builtin_elementwise_fma(p, ax, (f32x2_t){-4.599926953e-01f, -4.599926953e-01f});
    p = __builtin_elementwise_fma(p, ax, (f32x2_t){-1.150787739e+00f, -1.150787739e+00f});
    p = __builtin_elementwise_fma(p, ax, (f32x2_t){-1.000037638e+00f, -1.000037638e+00f});
    const float q0 = __builtin_amdgcn_exp2f(p[0]), q1 = __builtin_amdgcn_exp2f(p[1]);
    return (f32x2_t){__fmaf_rn(-ax[0], q0, fmaxf(x[0], 0.0f)), __fmaf_rn(-ax[1], q1, fmaxf(x[1], 0.0f))};
}

// VAR (debug build; 0 in production).  Timing-only ablations, results are garbage: 1 = no LDS-DMA requests inside the loop, 2 = no fragment
// reads, 4 = no barriers, 8 = no MFMAs, 16 = the requests as ORDINARY buffer loads into 16 staging registers (consumed by an empty asm in the
// second half), 32 = ... and written to LDS by ds_write_b128 there, 64 = every workgroup walks K from its own starting K tile.
// 128 (results valid): all 16 requests of a K tile in its first half (the first form of this kernel), vmcnt(16) at mid.  256 (results valid): start skew (below).
template <int EPI, int VAR, bool F16 = false>
__global__ void __launch_bounds__(G4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_gemm4w(Gemm4wArgs g) {
    constexpr bool LN = EPI == G4_EPI_GELU_LN;
    constexpr bool SPLIT = !(VAR & 128);
    // the residual epilogue needs the registers of the next tile's first two fragment sets (8 + 8 residual rows in flight beside the
    // statistics): its last K tile reads nothing ahead, the sets are read behind the epilogue like in the prologue (~300 cycles per tile)
    constexpr bool LATE_FRAGS = EPI == G4_EPI_RESID_STATS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int K = g.K, nk = K / 64, nblk = g.n_blocks, tiles_m = nblk / g.tiles_n;
    // XCD-aware tile order: workgroup ids with equal id % 8 share an XCD and take a contiguous range of tiles, swept 8 x 4 super-tile by super-tile
    auto coords = [&](int v, int &m0_, int &n0_) {
        const int nxcd = 8, q = nblk / nxcd, r = nblk % nxcd, x = v % nxcd;
        const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + v / nxcd;
        int tile_m, tile_n;
        if ((tiles_m & 7) == 0 && (g.tiles_n & 3) == 0) {
            const int grp = bid >> 5, w_ = bid & 31, cgn = g.tiles_n >> 2;
            const int rg = grp / cgn, cg = grp - rg * cgn;
            tile_m = rg * 8 + (w_ >> 2); tile_n = cg * 4 + (w_ & 3);
        } else {
            tile_m = bid / g.tiles_n; tile_n = bid - tile_m * g.tiles_n;
        }
        m0_ = tile_m * 256; n0_ = tile_n * 256;
    };
    // ---- LDS-DMA: a request of the workgroup lands 32 rows x 128 bytes (256 lanes x 16 bytes); a 128-row item = 4 requests
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void *)g.A, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void *)g.W, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void *)g.bias, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcS = __builtin_amdgcn_make_buffer_rsrc((void *)g.ln_stats, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc((void *)g.ln_colsum, 0, 0x7FFFFFFF, 0x00020000);
    const int srow = tid >> 3, kc = (tid & 7) ^ (srow & 7);                     // row inside a 32-row group, swizzled source chunk
    const unsigned voff = ((unsigned)srow * (unsigned)K + (unsigned)kc * 8u) * 2u;
    const unsigned vlin = (unsigned)lane * 16u;                                  // linear 1 KB copies (the tail)
    const unsigned k32b = (unsigned)K * 64u;                                     // 32 rows in bytes
    char *sdst = smem + wave * 1024;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    // the per-tile vectors as linear 1 KB LDS-DMA copies into the tail: wave w brings the raw row statistics of token rows 64 w .. 64 w + 63 (two
    // requests of 32 rows x 32 bytes), wave 0 the bias, wave 1 the column sums.  Issued at the tile top, i.e. OLDER than the K-tile requests
    // the counted waits of the main loop leave in flight
#define G4_TAIL_REQUESTS(M0_, N0_)                                                                                                            \
    {                                                                                                                                         \
        if (LN) {                                                                                                                             \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcS, G4_LDSP(smem + G4_TAIL + (2 * wave) * 1024), 16, vlin, (unsigned)(M0_) * 32u + (unsigned)(2 * wave) * 1024u, 0, 0); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcS, G4_LDSP(smem + G4_TAIL + (2 * wave + 1) * 1024), 16, vlin, (unsigned)(M0_) * 32u + (unsigned)(2 * wave + 1) * 1024u, 0, 0); \
            if (wave == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcC, G4_LDSP(smem + G4_TAIL_CSUM), 16, vlin, (unsigned)(N0_) * 4u, 0, 0); \
        }                                                                                                                                     \
        if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, G4_LDSP(smem + G4_TAIL_BIAS), 16, vlin, (unsigned)(N0_) * 4u, 0, 0);   \
    }
    unsigned xa[2][2], wa[2][2];                                                 // [buffer][k-substep]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned sw = (unsigned)(((ks * 4 + fq) ^ (fr & 7)) << 4);
        xa[0][ks] = lds0 + (unsigned)(wm * G4_ITEM + fr * 128) + sw;
        wa[0][ks] = lds0 + (unsigned)((2 + wn) * G4_ITEM + fr * 128) + sw;
        xa[1][ks] = xa[0][ks] + G4_BUF; wa[1][ks] = wa[0][ks] + G4_BUF;
    }
    f32x4 acc[8][8];
    u32x4 PX[8], PW[8], QX[8], QW[8], RX[8], RW[8];
    [[maybe_unused]] u32x4 SG[16];                                               // (ablation 16 / 32 only)

// ---- building blocks (macros: every index is a compile-time constant, nothing in a K-tile body branches)
// inline-asm MFMAs with the accumulator tied to itself in an AGPR quad: with the builtin, hipcc's allocator rotates the 256 loop-carried
// accumulator registers between the unrolled K-tile bodies and pays for it with v_accvgpr_mov chains between the MFMAs
#define G4_MM(MB, NB, FX, FW, FIRST)                                                                             \
    {                                                                                                           \
        if (VAR & 8) { if (FIRST) asm volatile("; no mfma %0 %1 %2" : "=a"(acc[MB][NB]) : "v"(FW[NB]), "v"(FX[MB])); else asm volatile("; no mfma %0 %1 %2" : "+a"(acc[MB][NB]) : "v"(FW[NB]), "v"(FX[MB])); } \
        else if (F16) { if (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc[MB][NB]) : "v"(FW[NB]), "v"(FX[MB])); \
                        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[MB][NB]) : "v"(FW[NB]), "v"(FX[MB])); } \
        else if (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[MB][NB]) : "v"(FW[NB]), "v"(FX[MB])); \
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[MB][NB]) : "v"(FW[NB]), "v"(FX[MB])); \
    }
// the fragment set a half multiplies with stays LIVE to the end of that half: otherwise the allocator hands the registers of a fragment whose
// last MFMA has just been issued to the next inline-asm LDS read (legal in program order -- and the outputs were wrong by 2 % when it did)
#define G4_KEEP(FX, FW)                                                                                         \
    { asm volatile("" ::"v"(FX[0]), "v"(FX[1]), "v"(FX[2]), "v"(FX[3]), "v"(FX[4]), "v"(FX[5]), "v"(FX[6]), "v"(FX[7]),                      \
                   "v"(FW[0]), "v"(FW[1]), "v"(FW[2]), "v"(FW[3]), "v"(FW[4]), "v"(FW[5]), "v"(FW[6]), "v"(FW[7])); }
// request 32-row group I of the X / W operand: RQ = scalar byte offset of (tile origin row, K tile) in the operand, B = destination buffer
#define G4_DMAX(B, I, RQ)                                                                                       \
    {   if (VAR & 16) { SG[I] = __builtin_amdgcn_raw_buffer_load_b128(rsrcX, voff, (RQ) + (unsigned)(I) * k32b, 0); G4_SB(); }        \
        else if (!(VAR & 1)) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, G4_LDSP(sdst + (B) * G4_BUF + (I) * 4096), 16, voff, (RQ) + (unsigned)(I) * k32b, 0, 0); G4_SB(); } }
#define G4_DMAW(B, I, RQ)                                                                                       \
    {   if (VAR & 16) { SG[8 + (I)] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, voff, (RQ) + (unsigned)(I) * k32b, 0); G4_SB(); }  \
        else if (!(VAR & 1)) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, G4_LDSP(sdst + (B) * G4_BUF + 2 * G4_ITEM + (I) * 4096), 16, voff, (RQ) + (unsigned)(I) * k32b, 0, 0); G4_SB(); } }
// first half, row block I: 8 MFMAs on set P, the X request of the row block behind the fourth
#define G4_H1ROW(B, FIRST, I, RQX, RQW)                                                                         \
    {                                                                                                           \
        G4_MM(I, 0, PX, PW, FIRST) G4_MM(I, 1, PX, PW, FIRST) G4_MM(I, 2, PX, PW, FIRST) G4_MM(I, 3, PX, PW, FIRST) G4_SB(); \
        G4_DMAX(B, I, RQX)                                                                                      \
        G4_MM(I, 4, PX, PW, FIRST) G4_MM(I, 5, PX, PW, FIRST) G4_MM(I, 6, PX, PW, FIRST) G4_MM(I, 7, PX, PW, FIRST) G4_SB(); \
        if (!SPLIT) G4_DMAW(B, I, RQW)                                                                          \
    }
#define G4_WRS(J)                                                                                               \
    { if (VAR & 32) { asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(lds0 + (unsigned)tid * 16u), "v"(SG[J]), "n"((J) * 4096) : "memory"); G4_SB(); } \
      else if (VAR & 16) { asm volatile("" :: "v"(SG[J])); G4_SB(); } }
// second half, row block I: 8 MFMAs on the k-substep-1 set, one fragment read of K tile t + 1 behind every second, the W request behind the last
#define G4_H2ROW(I, BX, BW, R0, R1, R2, R3, B, RQW, SKIPRD)                                                             \
    {                                                                                                           \
        G4_MM(I, 0, BX, BW, false) G4_WRS(2 * (I)) G4_MM(I, 1, BX, BW, false) if (!(SKIPRD)) R0;                \
        G4_MM(I, 2, BX, BW, false) G4_MM(I, 3, BX, BW, false) if (!(SKIPRD)) R1;                                \
        G4_MM(I, 4, BX, BW, false) G4_WRS(2 * (I) + 1) G4_MM(I, 5, BX, BW, false) if (!(SKIPRD)) R2;            \
        G4_MM(I, 6, BX, BW, false) G4_MM(I, 7, BX, BW, false) if (!(SKIPRD)) R3; G4_SB();                       \
        if (SPLIT) G4_DMAW(B, I, RQW)                                                                           \
    }
#define G4_RW(F, KS, NB_, I) { if (!(VAR & 2)) F[I] = g4_read128<(I) * 2048>(wa[NB_][KS]); }
#define G4_RX(F, KS, NB_, I) { if (!(VAR & 2)) F[I] = g4_read128<(I) * 2048>(xa[NB_][KS]); }
// K tile in buffer B: (BX, BW) = its k-substep-1 set, (NX, NW) = the set that receives k-substep 1 of the following K tile (buffer B ^ 1)
#define G4_KTILE(B, FIRST, BX, BW, NX, NW, RQX, RQW, SKIPRD)                                                            \
    {                                                                                                           \
        G4_H1ROW(B, FIRST, 0, RQX, RQW) G4_H1ROW(B, FIRST, 1, RQX, RQW) G4_H1ROW(B, FIRST, 2, RQX, RQW) G4_H1ROW(B, FIRST, 3, RQX, RQW) \
        G4_H1ROW(B, FIRST, 4, RQX, RQW) G4_H1ROW(B, FIRST, 5, RQX, RQW) G4_H1ROW(B, FIRST, 6, RQX, RQW) G4_H1ROW(B, FIRST, 7, RQX, RQW) \
        G4_KEEP(PX, PW)                                                                                         \
        if (SPLIT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   /* the following K tile has landed (mine) */ \
        if (!(VAR & 4)) __builtin_amdgcn_s_barrier();                /* ... and everybody's */                  \
        G4_SB();                                                                                                \
        G4_H2ROW(0, BX, BW, G4_RW(PW, 0, (B) ^ 1, 0), G4_RW(PW, 0, (B) ^ 1, 1), G4_RW(PW, 0, (B) ^ 1, 2), G4_RW(PW, 0, (B) ^ 1, 3), B, RQW, SKIPRD) \
        G4_H2ROW(1, BX, BW, G4_RW(PW, 0, (B) ^ 1, 4), G4_RW(PW, 0, (B) ^ 1, 5), G4_RW(PW, 0, (B) ^ 1, 6), G4_RW(PW, 0, (B) ^ 1, 7), B, RQW, SKIPRD) \
        G4_H2ROW(2, BX, BW, G4_RX(PX, 0, (B) ^ 1, 0), G4_RX(PX, 0, (B) ^ 1, 1), G4_RX(PX, 0, (B) ^ 1, 2), G4_RX(PX, 0, (B) ^ 1, 3), B, RQW, SKIPRD) \
        G4_H2ROW(3, BX, BW, G4_RX(PX, 0, (B) ^ 1, 4), G4_RX(PX, 0, (B) ^ 1, 5), G4_RX(PX, 0, (B) ^ 1, 6), G4_RX(PX, 0, (B) ^ 1, 7), B, RQW, SKIPRD) \
        G4_H2ROW(4, BX, BW, G4_RW(NW, 1, (B) ^ 1, 0), G4_RW(NW, 1, (B) ^ 1, 1), G4_RW(NW, 1, (B) ^ 1, 2), G4_RW(NW, 1, (B) ^ 1, 3), B, RQW, SKIPRD) \
        G4_H2ROW(5, BX, BW, G4_RW(NW, 1, (B) ^ 1, 4), G4_RW(NW, 1, (B) ^ 1, 5), G4_RW(NW, 1, (B) ^ 1, 6), G4_RW(NW, 1, (B) ^ 1, 7), B, RQW, SKIPRD) \
        G4_H2ROW(6, BX, BW, G4_RX(NX, 1, (B) ^ 1, 0), G4_RX(NX, 1, (B) ^ 1, 1), G4_RX(NX, 1, (B) ^ 1, 2), G4_RX(NX, 1, (B) ^ 1, 3), B, RQW, SKIPRD) \
        G4_H2ROW(7, BX, BW, G4_RX(NX, 1, (B) ^ 1, 4), G4_RX(NX, 1, (B) ^ 1, 5), G4_RX(NX, 1, (B) ^ 1, 6), G4_RX(NX, 1, (B) ^ 1, 7), B, RQW, SKIPRD) \
        G4_KEEP(BX, BW)                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           /* my reads of buffer B ^ 1 are done */    \
        G4_SB();                                                                                                \
        if (!(VAR & 4)) __builtin_amdgcn_s_barrier();                /* ... and everybody's: the next K tile may request into it */ \
        G4_SB();                                                                                                \
    }

    int v = blockIdx.x, m0, n0;
    coords(v, m0, n0);
    if constexpr ((VAR & 256) != 0) {
        // experiment (timing + results valid): a start skew along the super-tile's diagonal -- workgroup (i, j) of an 8 x 4 super-tile starts
        // (i + j) x ~2 500 cycles late, so that the four workgroups that share an activation panel (and the eight that share a weight panel) do not
        // all wait for the same L2 fill but follow its first requester by one K tile
        const int w_ = (int)(blockIdx.x >> 3) & 31;                              // position inside the XCD's group of 32 concurrently running workgroups
        const int d = (w_ >> 2) + (w_ & 3);
        for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(40);
    }
    // ---- prologue of the workgroup's first tile: tail, K tiles 0 and 1, the first two fragment sets
    G4_TAIL_REQUESTS(m0, n0)
    {
        const unsigned sX = (unsigned)m0 * (unsigned)K * 2u, sW = (unsigned)n0 * (unsigned)K * 2u;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, G4_LDSP(sdst + t * G4_BUF + j * 4096), 16, voff, sX + (unsigned)j * k32b + (unsigned)t * 128u, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, G4_LDSP(sdst + t * G4_BUF + 2 * G4_ITEM + j * 4096), 16, voff, sW + (unsigned)j * k32b + (unsigned)t * 128u, 0, 0);
        }
    }
    G4_SB();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    G4_SB();
#pragma unroll
    for (int i = 0; i < 8; ++i) { PW[i] = g4_read128<0>(wa[0][0] + i * 2048); PX[i] = g4_read128<0>(xa[0][0] + i * 2048); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { QW[i] = g4_read128<0>(wa[0][1] + i * 2048); QX[i] = g4_read128<0>(xa[0][1] + i * 2048); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G4_SB();
    __builtin_amdgcn_s_barrier();
    G4_SB();

    for (;;) {
        const int vn = v + (int)gridDim.x;
        const bool has_next = vn < nblk;
        int m0n = m0, n0n = n0;                                                  // no next tile: the run-on requests re-read this tile's first K tiles (never consumed)
        if (has_next) coords(vn, m0n, n0n);
        const unsigned sX = (unsigned)m0 * (unsigned)K * 2u, sW = (unsigned)n0 * (unsigned)K * 2u;
        const unsigned sXn = (unsigned)m0n * (unsigned)K * 2u, sWn = (unsigned)n0n * (unsigned)K * 2u;
        // (nk >= 4, even: the launcher checks)
        // (ablation 64, timing only: every workgroup walks K from its own starting tile -- much SLOWER: workgroups that share an operand panel
        // must stream it together, it does not stay in the L2 between them)
        const int rot = (VAR & 64) ? (int)((blockIdx.x >> 3) * 5 + (blockIdx.x & 7) * 2) : 0;
#define kofs(t) ((VAR & 64) ? (unsigned)(((t) + rot) & (nk - 1)) * 128u : (unsigned)(t) * 128u)
        G4_KTILE(0, true, QX, QW, RX, RW, sX + kofs(2), sW + kofs(2), false)
        G4_KTILE(1, false, RX, RW, QX, QW, sX + kofs(3), sW + kofs(3), false)
        for (int t = 2; t + 2 < nk; t += 2) {
            const unsigned kb = kofs(t + 2), kb1 = kofs(t + 3);
            G4_KTILE(0, false, QX, QW, RX, RW, sX + kb, sW + kb, false)
            G4_KTILE(1, false, RX, RW, QX, QW, sX + kb1, sW + kb1, false)
        }
        G4_KTILE(0, false, QX, QW, RX, RW, sXn + kofs(0), sWn + kofs(0), false)
        G4_KTILE(1, false, RX, RW, QX, QW, sXn + kofs(1), sWn + kofs(1), LATE_FRAGS)
#undef kofs
        // the last inline-asm MFMAs' results: the compiler does not see the MFMA -> v_accvgpr_read hazard ...
        G4_SB();
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        // ... and every accumulator is re-defined BEHIND the pad (empty asm, "+a") right where the epilogue reads it: an accumulator's last MFMA is
        // issued up to 63 MFMAs before the end of the loop, and the compiler is free to read it right there, one instruction behind an MFMA that
        // takes 16 cycles.  (One redefinition of all 64 in front of the epilogue made the allocator copy half of them into VGPRs at once --
        // with the next tile's two fragment sets live that spilled six fragments, stored to scratch straight behind their asynchronous LDS reads.)
        G4_SB();
        if constexpr (EPI == G4_EPI_RESID_STATS) {
            // ---- residual epilogue with row statistics (attn.proj, mlp.lin2): out = bf16(bf16(acc + bias) + residual) -- the reference's double
            // rounding -- and the partial LayerNorm statistics of the ROUNDED output row over this tile's 256 columns, bit for bit those of
            // k_gemm256p's staged epilogue: per 16-byte chunk s = (((0 + (l0 + h0)) + (l1 + h1)) + (l2 + h2)) + (l3 + h3), q = ((q0 + q1) +
            // q2) + q3 with qi = fma(li, li, hi hi); then its balanced tree over the row's 32 chunks -- here: lane rows (fq 0, 2) and (1, 3) [one
            // v_permlane32_swap + one v_permlane16_swap per value], the four channel-block pairs in registers, the two wave columns through LDS.
            // The residual rows are 16-byte loads at the addresses the stores go to, requested one block pair (8 loads) ahead by inline asm.
            const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc((void *)g.out, 0, 0x7FFFFFFF, 0x00020000);
            const unsigned long long ra = (unsigned long long)(size_t)g.resid;
            const u32x4 rsrcR = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ra), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ra >> 32)) & 0xFFFFu,
                                 0x7FFFFFFFu, 0x00020000u};
            const unsigned ldb = (unsigned)g.ld_out * 2u;
            const unsigned ovoff = (unsigned)(wm * 128 + fr) * ldb + (unsigned)(wn * 128 + ((fq & 1) << 4) + ((fq >> 1) << 3)) * 2u;
            const unsigned so0 = (unsigned)m0 * ldb + (unsigned)n0 * 2u;
            const unsigned tb = lds0 + G4_TAIL_BIAS + (unsigned)(wn * 128 + fq * 4) * 4u;
            u32x4 RA[8], RB[8];
            float q_sm[8], q_sq[8], h_sm[8], h_sq[8];          // per row block: the previous pair's quad sum, the first half's 8-chunk sum
#define G4_RLOAD(R, NP) { _Pragma("unroll") for (int mb = 0; mb < 8; ++mb) R[mb] = g4_load128(rsrcR, ovoff, so0 + (unsigned)(mb * 16) * ldb + (unsigned)(NP) * 64u); }
#define G4_RPAIR(NP, R)                                                                                                          \
            {                                                                                                                    \
                u32x4 b0u, b1u;                                                                                                  \
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"          \
                             : "=&v"(b0u), "=&v"(b1u) : "v"(tb), "n"((2 * (NP)) * 64), "n"((2 * (NP) + 1) * 64));               \
                const f32x4 b0 = __builtin_bit_cast(f32x4, b0u), b1 = __builtin_bit_cast(f32x4, b1u);                            \
                _Pragma("unroll") for (int mb = 0; mb < 8; ++mb) {                                                               \
                    asm volatile("" : "+a"(acc[mb][2 * (NP)]), "+a"(acc[mb][2 * (NP) + 1]));                                     \
                    const f32x4 v0 = acc[mb][2 * (NP)], v1 = acc[mb][2 * (NP) + 1];                                              \
                    const f32x2_t p00 = {v0[0] + b0[0], v0[1] + b0[1]}, p01 = {v0[2] + b0[2], v0[3] + b0[3]};                    \
                    const f32x2_t p10 = {v1[0] + b1[0], v1[1] + b1[1]}, p11 = {v1[2] + b1[2], v1[3] + b1[3]};                    \
                    const unsigned a0 = __builtin_bit_cast(unsigned, __builtin_convertvector(p00, bf16x2_t)), a1 = __builtin_bit_cast(unsigned, __builtin_convertvector(p01, bf16x2_t)); \
                    const unsigned e0 = __builtin_bit_cast(unsigned, __builtin_convertvector(p10, bf16x2_t)), e1 = __builtin_bit_cast(unsigned, __builtin_convertvector(p11, bf16x2_t)); \
                    const auto r0 = __builtin_amdgcn_permlane16_swap(a0, e0, false, false);                                      \
                    const auto r1 = __builtin_amdgcn_permlane16_swap(a1, e1, false, false);                                      \
                    const unsigned first[4] = {r0[0], r1[0], r0[1], r1[1]};              /* bf16(acc + bias): 8 consecutive channels */ \
                    const u32x4 rr = R[mb];                                                                                      \
                    u32x4 o;                                                                                                     \
                    float sm = 0.f, sq = 0.f;                                                                                    \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
                        const float lo = __fadd_rn(__uint_as_float(first[i] << 16), __uint_as_float(rr[i] << 16));               \
                        const float hi = __fadd_rn(__uint_as_float(first[i] & 0xFFFF0000u), __uint_as_float(rr[i] & 0xFFFF0000u)); \
                        const f32x2_t lh = {lo, hi};                                                                             \
                        o[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(lh, bf16x2_t));                              \
                        const float l = __uint_as_float(o[i] << 16), h = __uint_as_float(o[i] & 0xFFFF0000u);                    \
                        sm = __fadd_rn(sm, __fadd_rn(l, h));                                                                     \
                        const float qi = __fmaf_rn(l, l, __fmul_rn(h, h));                                                       \
                        sq = i == 0 ? qi : __fadd_rn(sq, qi);                                                                    \
                    }                                                                                                            \
                    __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, ovoff, so0 + (unsigned)(mb * 16) * ldb + (unsigned)(NP) * 64u, 0); \
                    /* the quad of four chunks this pair holds across the lane rows: (fq 0 + fq 2) + (fq 1 + fq 3) */            \
                    const float qs = g4_add_xor16(g4_add_xor32(sm)), qq = g4_add_xor16(g4_add_xor32(sq));                        \
                    if ((NP) == 0 || (NP) == 2) { q_sm[mb] = qs; q_sq[mb] = qq; }                                                \
                    else if ((NP) == 1) { h_sm[mb] = __fadd_rn(q_sm[mb], qs); h_sq[mb] = __fadd_rn(q_sq[mb], qq); }              \
                    else {                                       /* this wave's 16 chunks: (Q0 + Q1) + (Q2 + Q3) */              \
                        const float s16 = __fadd_rn(h_sm[mb], __fadd_rn(q_sm[mb], qs)), q16 = __fadd_rn(h_sq[mb], __fadd_rn(q_sq[mb], qq)); \
                        if (fq == 0) asm volatile("ds_write_b64 %0, %1" :: "v"(lds0 + G4_TAIL + (unsigned)(((wm * 128 + mb * 16 + fr) * 2 + wn) * 8)), "v"((f32x2_t){s16, q16}) : "memory"); \
                    }                                                                                                            \
                    G4_SB();                                                                                                     \
                }                                                                                                                \
            }
            G4_RLOAD(RA, 0)
            G4_RLOAD(RB, 1)
            g4_vmwait<8>(RA);  G4_SB();
            G4_RPAIR(0, RA)
            G4_RLOAD(RA, 2)
            g4_vmwait<16>(RB); G4_SB();
            G4_RPAIR(1, RB)
            G4_RLOAD(RB, 3)
            g4_vmwait<16>(RA); G4_SB();
            G4_RPAIR(2, RA)
            g4_vmwait<8>(RB);  G4_SB();
            G4_RPAIR(3, RB)
#undef G4_RLOAD
#undef G4_RPAIR
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            G4_SB();
            __builtin_amdgcn_s_barrier();                        // both wave columns have parked their 128-column sums
            G4_SB();
            if (wn == 0) {                                       // 128 rows of this wave row, two per lane: total = (columns 128..255) + (columns 0..127)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = wm * 128 + h * 64 + lane;
                    u32x4 pr;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pr) : "v"(lds0 + G4_TAIL + (unsigned)(row * 16)));
                    const f32x4 pv = __builtin_bit_cast(f32x4, pr);               // (sum, sq) of wave column 0, (sum, sq) of wave column 1
                    float2 tot;
                    tot.x = __fadd_rn(pv[2], pv[0]); tot.y = __fadd_rn(pv[3], pv[1]);
                    *reinterpret_cast<float2 *>(g.stats_out + ((size_t)(m0 + row) * 4 + (n0 >> 8)) * 2) = tot;
                }
            }
        } else if constexpr ((VAR & 2048) != 0) {
            // (timing-only ablation, debug build: NO epilogue at all -- the accumulators are neither read nor stored.  full - this = what the
            // epilogue costs a launch with the matrix pipe idle, i.e. the most that running it under the next tile's main loop could buy)
        } else
        // ---- epilogue: [folded LayerNorm +] bias [+ GELU], one conversion per pair, lane rows swapped into 16-byte pieces, buffer stores (the
        // direct-store epilogue of k_gemm256p).  Every LDS read of the tail is inline asm: an ordinary one makes hipcc drain the request queue
        {
            const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc((void *)g.out, 0, 0x7FFFFFFF, 0x00020000);
            const unsigned ldb = (unsigned)g.ld_out * 2u;
            const unsigned ovoff = (unsigned)(wm * 128 + fr) * ldb + (unsigned)(wn * 128 + ((fq & 1) << 4) + ((fq >> 1) << 3)) * 2u;
            const unsigned so0 = (unsigned)m0 * ldb + (unsigned)n0 * 2u;
            const unsigned tb = lds0 + G4_TAIL_BIAS + (unsigned)(wn * 128 + fq * 4) * 4u;
            // this lane's 8 token rows (mb, fr): (rstd, -mean rstd) from the four partial (sum, sum of squares) slots, in exactly the operations of
            // k_gemm256p's park_tile_vectors (its ISA: two add trees, mean = inv_k sum, fma(inv_k, sq, -(mean mean)), max 0, + 1e-6, v_rsq, rstd (-mean))
            [[maybe_unused]] float ln_rs[8], ln_nm[8];
            if constexpr (LN) {
                const float inv_k = 1.0f / K;
                const unsigned ts = lds0 + G4_TAIL + (unsigned)(wm * 128 + fr) * 32u;
#pragma unroll
                for (int mb = 0; mb < 8; ++mb) {
                    u32x4 au, bu;
                    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(au), "=&v"(bu) : "v"(ts), "n"(mb * 512), "n"(mb * 512 + 16));
                    const f32x4 a = __builtin_bit_cast(f32x4, au), b = __builtin_bit_cast(f32x4, bu);
                    const float sum = __fadd_rn(__fadd_rn(a[0], a[2]), __fadd_rn(b[0], b[2])), sq = __fadd_rn(__fadd_rn(a[1], a[3]), __fadd_rn(b[1], b[3]));
                    const float mean = __fmul_rn(inv_k, sum);
                    const float var = __fmaf_rn(inv_k, sq, -__fmul_rn(mean, mean));
                    const float rstd = __builtin_amdgcn_rsqf(__fadd_rn(fmaxf(var, 0.f), 1e-6f));
                    ln_rs[mb] = rstd; ln_nm[mb] = __fmul_rn(rstd, -mean);
                }
            }
            // two channel block pairs (64 channels = the wave's 128 bytes of a token row) per pass over the row blocks, so that the two
            // 64-byte halves of every 128-byte line leave in consecutive stores (with the pairs as the OUTER loop the halves were 8 stores
            // apart and the L2 wrote part of the lines back twice: WRITE_SIZE 384 MB per mlp.lin1 launch against 268 MB of output)
#pragma unroll
            for (int nph = 0; nph < 2; ++nph) {
                f32x4 bq[2][2], cq[2][2];                                        // [pair of the pass][block of the pair]: bias, column sums
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int np = 2 * nph + h;
                    u32x4 b0u, b1u, c0u = {0u, 0u, 0u, 0u}, c1u = {0u, 0u, 0u, 0u};
                    if constexpr (LN)
                        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(b0u), "=&v"(b1u), "=&v"(c0u), "=&v"(c1u) : "v"(tb), "n"((2 * np) * 64), "n"((2 * np + 1) * 64), "n"(1024 + (2 * np) * 64), "n"(1024 + (2 * np + 1) * 64));
                    else
                        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(b0u), "=&v"(b1u) : "v"(tb), "n"((2 * np) * 64), "n"((2 * np + 1) * 64));
                    bq[h][0] = __builtin_bit_cast(f32x4, b0u); bq[h][1] = __builtin_bit_cast(f32x4, b1u);
                    cq[h][0] = __builtin_bit_cast(f32x4, c0u); cq[h][1] = __builtin_bit_cast(f32x4, c1u);
                }
#pragma unroll
                for (int mb = 0; mb < 8; ++mb)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int np = 2 * nph + h;
                        const f32x4 b0 = bq[h][0], b1 = bq[h][1];
                        [[maybe_unused]] const f32x4 c0 = cq[h][0], c1 = cq[h][1];
                        asm volatile("" : "+a"(acc[mb][2 * np]), "+a"(acc[mb][2 * np + 1]));
                        const f32x4 v0 = acc[mb][2 * np], v1 = acc[mb][2 * np + 1];
                        f32x2_t p00 = {v0[0], v0[1]}, p01 = {v0[2], v0[3]}, p10 = {v1[0], v1[1]}, p11 = {v1[2], v1[3]};
                        const f32x2_t bb00 = {b0[0], b0[1]}, bb01 = {b0[2], b0[3]}, bb10 = {b1[0], b1[1]}, bb11 = {b1[2], b1[3]};
                        if constexpr (LN) {
                            // rstd (acc - mean colsum) + bias = fma(acc, rstd, fma(-mean rstd, colsum, bias)), as in k_gemm256p
                            const f32x2_t rs = {ln_rs[mb], ln_rs[mb]}, nm = {ln_nm[mb], ln_nm[mb]};
                            p00 = __builtin_elementwise_fma(p00, rs, __builtin_elementwise_fma(nm, (f32x2_t){c0[0], c0[1]}, bb00));
                            p01 = __builtin_elementwise_fma(p01, rs, __builtin_elementwise_fma(nm, (f32x2_t){c0[2], c0[3]}, bb01));
                            p10 = __builtin_elementwise_fma(p10, rs, __builtin_elementwise_fma(nm, (f32x2_t){c1[0], c1[1]}, bb10));
                            p11 = __builtin_elementwise_fma(p11, rs, __builtin_elementwise_fma(nm, (f32x2_t){c1[2], c1[3]}, bb11));
                        } else {
                            p00 += bb00; p01 += bb01; p10 += bb10; p11 += bb11;
                        }
                        if constexpr (EPI == G4_EPI_GELU_LN) { p00 = g4_gelu2(p00); p01 = g4_gelu2(p01); p10 = g4_gelu2(p10); p11 = g4_gelu2(p11); }
                        const unsigned a0 = g4_pack2<F16>(p00), a1 = g4_pack2<F16>(p01), e0 = g4_pack2<F16>(p10), e1 = g4_pack2<F16>(p11);
                        const auto r0 = __builtin_amdgcn_permlane16_swap(a0, e0, false, false);
                        const auto r1 = __builtin_amdgcn_permlane16_swap(a1, e1, false, false);
                        const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                        __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, ovoff, so0 + (unsigned)(mb * 16) * ldb + (unsigned)np * 64u, (VAR & 512) ? 2 : (VAR & 4096) ? 0 : (VAR & 8192) ? 17 : 16);
                        // aux 16 = sc1, the store at AGENT scope (production since round 6): written through to the memory side at once instead of staying
                        // dirty in the XCD's L2 until evicted -- the 256 workgroups' epilogues burst at the same moment, 32 x 128 KB per XCD = its whole 4 MB
                        // L2, operand panels included.  mlp.lin1 126.8 -> 124.3 us in situ, same bits (profiles/r06_ab_store_scope.txt).  Debug build:
                        // VAR & 4096 = ordinary stores (the round-5 form), & 8192 = system scope (no different), & 512 = non-temporal (slower: mlp.lin2 wants the rows in the Infinity Cache)
                        G4_SB();                                                 // (keeps the scheduler from hoisting all 256 accumulator reads: the next tile's two fragment sets are live here)
                    }
            }
        }
        G4_SB();
        if constexpr (LATE_FRAGS) {
            // K tile 0 of the next tile (buffer 0) landed before the mid barrier of the last K tile
#pragma unroll
            for (int i = 0; i < 8; ++i) { PW[i] = g4_read128<0>(wa[0][0] + i * 2048); PX[i] = g4_read128<0>(xa[0][0] + i * 2048); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { QW[i] = g4_read128<0>(wa[0][1] + i * 2048); QX[i] = g4_read128<0>(xa[0][1] + i * 2048); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            G4_SB();
        }
        __builtin_amdgcn_s_barrier();                                            // everybody has read the tail [and buffer 0]: the next tile's vectors / requests may land
        G4_SB();
        if (!has_next) break;
        v = vn; m0 = m0n; n0 = n0n;
        G4_TAIL_REQUESTS(m0, n0)
        G4_SB();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // the run-on requests of the last tile: landed before the LDS is given back
#undef G4_TAIL_REQUESTS
#undef G4_MM
#undef G4_KEEP
#undef G4_DMAX
#undef G4_DMAW
#undef G4_H1ROW
#undef G4_H2ROW
#undef G4_WRS
#undef G4_RW
#undef G4_RX
#undef G4_KTILE
}

static int g4_num_cus() {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
}
// shapes this kernel takes: whole 256 x 256 tiles, an even number (>= 4) of K tiles, every operand below 2 GiB (32-bit buffer offsets)
static bool g4_shape_ok(int M, int N, int K, int ld_out) {
    if (M % 256 || N % 256 || K % 128 || K < 256) return false;
    return (size_t)M * K * 2 < ((size_t)1 << 31) && (size_t)N * K * 2 < ((size_t)1 << 31) && (size_t)M * ld_out * 2 < ((size_t)1 << 31);
}
template <int EPI, int VAR, bool F16 = false>
static void g4_launch(const Gemm4wArgs &a, hipStream_t s) {
    static CpxOncePerDevice once;
    once([] { (void)hipFuncSetAttribute((const void *)k_gemm4w<EPI, VAR, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES); });
    const int n_cu = g4_num_cus();
    hipLaunchKernelGGL((k_gemm4w<EPI, VAR, F16>), dim3(a.n_blocks < n_cu ? a.n_blocks : n_cu), dim3(G4_THREADS), G4_LDS_BYTES, s, a);
}

// mlp.lin1 of the half-precision network (bf16, or fp16 when f16 != 0): out = gelu(LayerNorm-folded(A) W^T + bias).  Returns 1 when launched, 0 when the shape is not this
// kernel's (the caller then takes k_gemm256p, which computes the same bits).
#ifdef CPX_DEBUG
static int g_gemm4w_var = 0;
extern "C" void cpx_gemm4w_set_variant(int v) { g_gemm4w_var = v; }
#endif
int cpx_gemm4w_gelu_ln(int f16, const void *A, const void *W, int M, int N, int K, const float *bias, const float *ln_stats, const float *ln_colsum,
                       void *out, int ld_out, hipStream_t s) {
    if (!g4_shape_ok(M, N, K, ld_out) || !bias || !ln_stats || !ln_colsum) return 0;
    Gemm4wArgs a;
    a.A = (const unsigned short *)A; a.W = (const unsigned short *)W; a.bias = bias; a.ln_stats = ln_stats; a.ln_colsum = ln_colsum;
    a.out = (unsigned short *)out; a.resid = nullptr; a.stats_out = nullptr;
    a.M = M; a.N = N; a.K = K; a.ld_out = ld_out; a.tiles_n = N / 256; a.n_blocks = (M / 256) * (N / 256);
#ifdef CPX_DEBUG
    if (!f16 && g_gemm4w_var == 512) { g4_launch<G4_EPI_GELU_LN, 512, false>(a, s); return 1; }     // experiment: non-temporal output stores
    if (!f16 && g_gemm4w_var == 4096) { g4_launch<G4_EPI_GELU_LN, 4096, false>(a, s); return 1; }   // A/B: ordinary output stores (the round-5 form; production stores at agent scope)
    if (!f16 && g_gemm4w_var == 8192) { g4_launch<G4_EPI_GELU_LN, 8192, false>(a, s); return 1; }   // experiment: ... at system scope
    if (!f16 && g_gemm4w_var == 2048) { g4_launch<G4_EPI_GELU_LN, 2048, false>(a, s); return 1; }   // timing only: no epilogue
    if (!f16 && g_gemm4w_var == 2049) { g4_launch<G4_EPI_BIAS, 0, false>(a, s); return 1; }          // timing only: the bias-only epilogue (no LayerNorm fold, no GELU)
#endif
    if (f16) g4_launch<G4_EPI_GELU_LN, 0, true>(a, s);
    else g4_launch<G4_EPI_GELU_LN, 0, false>(a, s);
    return 1;
}

#ifdef CPX_DEBUG
// (debug build only: measured 3 - 6 % slower than k_gemm256p, profiles/r05_ab_gemm4w_resid.txt -- the product library does not carry this instantiation)
// attn.proj / mlp.lin2 of the bf16 network: out = bf16(bf16(A W^T + bias) + resid) and the row statistics of out (N = 1024: 4 column tiles = the
// 4 slots).  Returns 1 when launched, 0 when the shape is not this kernel's.
int cpx_gemm4w_resid_stats(const void *A, const void *W, int M, int N, int K, const float *bias, const void *resid, void *out, int ld_out,
                           float *stats_out, hipStream_t s) {
    if (!g4_shape_ok(M, N, K, ld_out) || !bias || !resid || !stats_out || N != 1024) return 0;
    Gemm4wArgs a;
    a.A = (const unsigned short *)A; a.W = (const unsigned short *)W; a.bias = bias; a.ln_stats = nullptr; a.ln_colsum = nullptr;
    a.out = (unsigned short *)out; a.resid = (const unsigned short *)resid; a.stats_out = stats_out;
    a.M = M; a.N = N; a.K = K; a.ld_out = ld_out; a.tiles_n = N / 256; a.n_blocks = (M / 256) * (N / 256);
    g4_launch<G4_EPI_RESID_STATS, 0>(a, s);
    return 1;
}

// out[M][ld_out] (bf16) = A[M][K] . W[N][K]^T + bias; M, N multiples of 256, K a multiple of 128, every operand below 2 GiB
extern "C" int cpx_gemm4w(const void *A, const void *W, int M, int N, int K, const float *bias, void *out, int ld_out, void *stream) {
    if (!g4_shape_ok(M, N, K, ld_out) || !bias) return CPX_EINVAL;
    Gemm4wArgs a;
    a.A = (const unsigned short *)A; a.W = (const unsigned short *)W; a.bias = bias; a.ln_stats = nullptr; a.ln_colsum = nullptr;
    a.out = (unsigned short *)out; a.resid = nullptr; a.stats_out = nullptr;
    a.M = M; a.N = N; a.K = K; a.ld_out = ld_out; a.tiles_n = N / 256; a.n_blocks = (M / 256) * (N / 256);
    hipStream_t s = (hipStream_t)stream;
    switch (g_gemm4w_var) {
        case 1: g4_launch<G4_EPI_BIAS, 1>(a, s); break;
        case 2: g4_launch<G4_EPI_BIAS, 2>(a, s); break;
        case 3: g4_launch<G4_EPI_BIAS, 3>(a, s); break;
        case 4: g4_launch<G4_EPI_BIAS, 4>(a, s); break;
        case 7: g4_launch<G4_EPI_BIAS, 7>(a, s); break;
        case 8: g4_launch<G4_EPI_BIAS, 8>(a, s); break;
        case 128: g4_launch<G4_EPI_BIAS, 128>(a, s); break;
        case 256: g4_launch<G4_EPI_BIAS, 256>(a, s); break;
        default: g4_launch<G4_EPI_BIAS, 0>(a, s);
    }
    return hipGetLastError() == hipSuccess ? 0 : CPX_EHIP;
}
#endif  // CPX_DEBUG
