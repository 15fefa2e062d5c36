// Flash attention with the decomposed rel-pos bias (vit_sam.py:15-65, flash_forward), round-4 EXPERIMENT (debug build only,
// cpx_attention_set_variant(3)): ONE wave per SIMD, TWO image rows of queries per wave, persistent workgroups.
// Result (profiles/r04_attn2q_*.txt): float64 error equal to / slightly below the production kernel's on every test (random,
// spiked key, ramp = repeated rescale; bf16 and fp16), repeatable -- and 217-223 us against 208-215 us for k_attention4p in the
// one-process A/B, 23.47 against 23.23 ms per engine step: not faster, so k_attention4p stays the production kernel.  What the
// build measured (tools/micro/mfma_fill.hip, tools/attn2q_stamps.py) is why the shape has no more to give on this part:
//   * one wave issues a vector instruction every 4 cycles (v_exp_f32 8) and each of its own MFMAs blocks that issue for ~8:
//     per half step (32 queries x 32 keys of softmax beside 10 MFMAs) 16 v_fma + 16 v_exp + 8 v_cvt_pk + the vote = ~250
//     issue cycles + 80 = 330, beside 288 matrix-pipe cycles; the hand-placed stream runs 380 cycles stand-alone
//     (mfma_fill: "half step, no vote tail") and ~450 in the kernel (vote, offset arithmetic, hazard nops);
//   * a v_pk_fma_f32 beside MFMAs in flight costs ~30 cycles (first build: 770 cycles per half step), so the scale / offset
//     FMAs are scalar;
//   * with one workgroup per CU nothing hides an item seam (G = Q table^T, ring refill, pipeline fill: 5.4 k cycles, epilogue
//     2.9 k of 52 k per item even with the ring and the Q rows prefetched across the seam) nor the per-tile barrier + DMA /
//     LDS-read issue (~250 of ~1 150 cycles per 64 x 32 tile) -- the three co-resident workgroups of k_attention4p do.
//   * two more forms, both bitwise equal to the first and neither faster: variant 4 (GL) drops the LDS ring -- every wave loads
//     its K / V^T fragments straight from global memory one tile ahead, no LDS-DMA requests, no per-tile barrier, no ds_read, the
//     waves never synchronise -- 218.0 against 218.9 us (its stamps show the loads of one tile ahead still ~800 cycles away when
//     they are needed: L2 latency under this load exceeds a tile); variant 5 (NV) additionally drops the per-half-step overflow
//     vote (tile 0 excepted; the row sums are checked once per item and the wave re-runs the item in the exact mode if they read
//     >= 2^100 / inf / NaN, which the spiked-key test exercises) -- 217.7 us; variant 6 requests the fragments TWO tiles ahead
//     into four rotating sets that live in AGPRs (loads straight into the accumulator file, MFMA A operands read from it) so
//     that no load is waited for -- 229.7 against 226.9 (variant 4) / 228.9 (variant 3) / 223.2 us (production) on a slower
//     lease: every form of the staging lands on the same time, i.e. the staging is not what bounds this structure;
//   * the decisive measurement is the wall time of the STAMPED kernel with its streams ablated (tools/attn2q_stamps.py 0..3):
//     245.9 us as built, 196.7 without its MFMAs, 189.9 without its vector instructions, 122.4 with NEITHER -- the two pipes
//     overlap well (their removal saves 49 / 56 us), and half of the launch is the skeleton around them: staging, waits,
//     barrier, item seams (and, in that build, ~125 cycles per s_memtime stamp, four per tile).  One workgroup per CU has
//     nothing to run beside that skeleton; the production kernel's three co-resident workgroups per CU do.
//   So the bound for head dim 64 with this bias and a bf16 P is ~(2 x 330 + 150) cycles per (64 queries x 32 keys) per SIMD
//   = 0.39 of the MFMA peak at the clock held, before seams; the kernel reaches 0.26, the production kernel 0.27-0.28.
//
// Why: the round-2/3 kernel (k_attention4p, cpx_net.hip: one 32-query row per wave, three
// workgroups per CU) spends 813 SIMD cycles per (32 queries x 32 keys) while its 8 MFMAs need 256 and its 67 vector
// instructions ~270: every wave runs a strictly dependent QK^T -> softmax -> P.V chain, and three such chains per SIMD
// do not fill either pipe.  tools/micro/pk_rate.hip measured what ONE wave can issue: a vector instruction every 4
// cycles (v_exp_f32 8), v_pk_*_f32 at the same 4 (two FLOPs per lane), beside its own MFMAs in flight.  So this kernel
// gives each wave two INDEPENDENT chains (image rows 2w and 2w + 1 of the workgroup's 8) and software-pipelines them by
// hand: while the vector pipe runs the softmax of chain A (8 v_pk_fma, 16 v_exp, 8 v_cvt_pk, the overflow vote), the
// matrix pipe runs S = K Q^T + Gw of chain B and the P.V / row-sum products of chain B's previous tile, then the roles
// swap.  Both chains share every K / V^T fragment read, LDS-DMA request and barrier (half the per-score overhead of the
// old kernel), the softmax denominators come out of the matrix pipe (an all-ones A operand: no add tree), and the whole
// register file of the SIMD belongs to the one wave (S, P, K and V fragments double-buffered in registers).
//   per (64 queries x 32 keys): 20 MFMAs (640 matrix-pipe cycles) beside ~90 vector instructions (~460 issue cycles).
//
// Same decomposition of the bias, the same LDS images (swizzles, key permutation pi) and the same lazy-maximum softmax
// as k_attention4p: per key tile the probabilities are formed against a running reference m_run, bit 14 / 30 of the OR
// of the packed halves says whether some p >= 2 (or the reference is still -inf); only then the exact maximum / rescale
// path runs and parks the reference HEADROOM octaves above the true maximum.
#include "cpx_internal.h"
#include <type_traits>
#ifdef CPX_DEBUG

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool F16>
__device__ __forceinline__ f32x16 mfma32(const u32x4 &a, const u32x4 &b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    f32x2 v = {lo, hi};
    if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ int pi_perm(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// ---- the hand-placed instruction stream of the key loop: every instruction an `asm volatile`, so the order below IS the
// order in the binary (hipcc allocates the registers).  MFMAs whose result the vector pipe reads (S) are in VGPR form,
// accumulators only the matrix pipe touches (O, row sums) live in AGPRs -- a mix the builtins cannot express (the
// compiler picks one form per function and pays 32 v_accvgpr copies per score tile for the other).
// Hazards the compiler cannot see through inline asm, all kept by construction: an MFMA result is read by the vector pipe
// no earlier than half a tile after it was issued; a v_exp_f32 result is consumed >= 2 instructions later; VALU results
// that feed an MFMA (P) are written >= 4 instructions before it.
// (AF: the A operand -- a K / V^T fragment -- lives in AGPRs: the GL = 2 form loads its fragments straight into them)
template <bool F16, bool AF = false> __device__ __forceinline__ void MF_init(f32x16 &d, const u32x4 &a, const u32x4 &b, const f32x16 &c) {
    if constexpr (AF) {
        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "v"(b), "v"(c));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "v"(b), "v"(c));
    } else {
        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    }
}
template <bool F16, bool AF = false> __device__ __forceinline__ void MF_accv(f32x16 &d, const u32x4 &a, const u32x4 &b) {
    if constexpr (AF) {
        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "v"(b));
    } else {
        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    }
}
// row sums of P on the HALF-size matrix instruction: D (16 x 16, AGPRs) += SEL (16 x 32) . P^T-fragment read as a (32 x 16) B
// operand.  The fragment was built for the 32x32x16 shape (lane (r, h2): query r, keys 8 h2 .. 8 h2 + 7); read as a 16x16x32 B
// operand its k-block (lane >> 4) holds {q = n keys 0-7, q = n + 16 keys 0-7, q = n keys 8-15, q = n + 16 keys 8-15}, so SEL has
// ones in row 0 for the even k-blocks and in row 1 for the odd ones: D[0][n] = row sum of query n, D[1][n] = of query n + 16,
// i.e. lanes 0..15 hold (register 0, register 1) = the sums of queries (lane, lane + 16).  16 matrix-pipe cycles instead of 32.
template <bool F16> __device__ __forceinline__ void MF_rowsum(f32x4 &d, const u32x4 &sel, const u32x4 &p) {
    if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(d) : "v"(sel), "v"(p));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(sel), "v"(p));
}
__device__ __forceinline__ void A_zero4(f32x4 &acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { float x; asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(x)); acc[i] = x; }
}
// acc *= alpha, in place in the AGPRs (rare exact path of the softmax): the compiler never sees the accumulators outside
// the accumulator file, so it keeps no VGPR copies of them across the branch
__device__ __forceinline__ void A_scale16(f32x16 &acc, float alpha) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float x = acc[i], t;
        asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1" : "+a"(x), "=&v"(t) : "v"(alpha));
        acc[i] = x;
    }
}
template <bool F16, bool AF = false> __device__ __forceinline__ void MF_acca(f32x16 &d, const u32x4 &a, const u32x4 &b) {
    if constexpr (AF) {
        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(d) : "a"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "a"(a), "v"(b));
    } else {
        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
    }
}
// a = s * c + off on a register pair: TWO scalar FMAs -- a v_pk_fma_f32 beside MFMAs in flight costs ~30 cycles (the packed
// f32 unit and the matrix pipe get in each other's way; profiles/r04_attn2q_stamps.txt, MI355X_MICROARCH "price of one
// filler beside MFMAs")
__device__ __forceinline__ f32x2 V_pkfma(f32x2 sv, f32x2 c2, f32x2 off2) {
    f32x2 a;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a.x) : "v"(sv.x), "s"(c2.x), "v"(off2.x));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a.y) : "v"(sv.y), "s"(c2.x), "v"(off2.x));
    return a;
}
__device__ __forceinline__ float V_exp(float x) { float y; asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(x)); return y; }
template <bool F16> __device__ __forceinline__ unsigned V_cvt(float lo, float hi) {
    unsigned r;
    if constexpr (F16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ unsigned V_or3(unsigned a, unsigned b, unsigned c) {
    unsigned r; asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}

// GL variant: a K / V^T fragment straight from global memory (L2 / the CU's vector L1: the four waves of a workgroup read the
// same 8 KB per tile within a few hundred cycles of each other) into the registers the MFMAs read -- no LDS ring, no
// LDS-DMA requests, no per-tile barrier, no ds_read; the request goes out one whole tile before its first use
template <int OFF>
__device__ __forceinline__ u32x4 g_load128(const unsigned short *p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ u32x4 g_load128a(const unsigned short *p) {          // destination in AGPRs
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(v) : "v"(p), "n"(OFF));
    return v;
}
#define A2_THREADS 256
#define A2_SLOT 8192                    // K tile (32 keys x 64 d) + V^T tile (64 d x 32 keys), halves
#define A2_G_LD 68                      // padded row of the G scratch (fp16): [wave][chain][32 q][68]: 8-byte aligned rows,
                                        // and the per-tile reads (one half per query row) fall on 32 distinct banks
#define A2_LDS_BYTES (4 * A2_SLOT + 8 * 32 * A2_G_LD * 2)
template <int OFF>
__device__ __forceinline__ u32x4 a2_read128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// DBG (debug build only): s_memtime stamps per loop segment -> dbg [workgroup][4 waves][8], summed over the workgroup's items:
// 0 vmcnt wait + barrier, 1 DMA requests + fragment reads (issue + landing: the stamp itself waits lgkmcnt(0)), 2 half step A,
// 3 half step B, 4 seam (G tables, ring wait, pipeline fill), 5 whole kernel, 6 epilogue (normalise + stores), 7 items.
// The stamp values only go to `dbg`.
#define A2_STAMP(i)                                                                            \
    do {                                                                                       \
        if constexpr (DBG) {                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            unsigned long long t_;                                                             \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            seg[i] += (unsigned)(t_ - tprev);                                                  \
            tprev = t_;                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                 \
        }                                                                                      \
    } while (0)
__device__ __forceinline__ void A_zero16(f32x16 &acc) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { float x; asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(x)); acc[i] = x; }
}
// PERSISTENT: gridDim.x = min(#CUs, #items) workgroups (a multiple of 8); workgroup b lives on XCD b % 8 and walks the items
// l = b / 8, b / 8 + gridDim.x / 8, ... of that XCD's list: item l = (pair (l / 4) * 8 + b % 8, row quarter l % 4), so the 32
// workgroups of an XCD are on 8 (sub-tile, head) pairs at a time (2 MB of K / V^T in its 4 MB L2).  Across an item seam the
// K / V^T ring keeps running (the last three tiles of an item request the first three of the next), the next item's Q rows
// are requested before the epilogue's stores go out, and the rel-pos tables stay in AGPRs for the whole launch.
// GL: 0 = LDS-DMA ring; 1 = fragments straight from global memory, requested one tile ahead into VGPRs; 2 = two tiles ahead, four
// rotating fragment sets in AGPRs (the MFMA A operand may be an AGPR)
template <bool F16, bool DBG = false, int ABL = 0, int GL = 0, bool NV = false>
__global__ void __launch_bounds__(A2_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_attention2q(const unsigned short *__restrict__ qkv, const unsigned short *__restrict__ vT,
              const unsigned short *__restrict__ relh, const unsigned short *__restrict__ relw,
              unsigned short *__restrict__ out, int n_pairs, unsigned *__restrict__ dbg = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char a2_smem[];
    unsigned seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0, tstart = 0;
    if constexpr (DBG) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); tstart = tprev; }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h2 = lane >> 5;
    const int xcd = blockIdx.x & 7, l_stride = (int)gridDim.x >> 3, n_l = n_pairs >> 1;
    int l = (int)blockIdx.x >> 3;
    if (l >= n_l) return;

    // ---- the two rel-pos tables as MFMA A-operand fragments [jb][ks], in AGPRs for the whole launch
    u32x4 tfw[2][4], tfh[2][4];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (GL != 2) {      // (GL = 2 needs the AGPRs for its four fragment sets: it re-reads the tables, L2-hot, at every seam)
                tfw[jb][ks] = *reinterpret_cast<const u32x4 *>(relw + (jb * 32 + r) * 64 + 16 * ks + 8 * h2);
                tfh[jb][ks] = *reinterpret_cast<const u32x4 *>(relh + (jb * 32 + r) * 64 + 16 * ks + 8 * h2);
            }
        }

    // ---- ring requests: thread -> 16 bytes of K (key tid>>3, position tid&7) and of V^T (d tid>>2, position tid&3)
    const int kkey = tid >> 3, vd = tid >> 2;
    const int koff = kkey * 3072 + 1024 + (((tid & 7) ^ ((kkey >> 1) & 7)) * 8);
    const int voff = vd * 1024 + (((tid & 3) ^ ((vd >> 2) & 3)) * 8);
    char *dma_dst = a2_smem + wave * 1024;
    auto issue = [&](const unsigned short *ksrc, const unsigned short *vsrc, int kh) {
        char *d = dma_dst + (kh & 3) * A2_SLOT;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ksrc + (size_t)kh * 32 * 3072),
                                         (__attribute__((address_space(3))) void *)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vsrc + kh * 32),
                                         (__attribute__((address_space(3))) void *)(d + 4096), 16, 0, 0);
    };
    // item l -> (sub-tile s, head, row quarter q4) and this thread's ring source pointers
    auto coords = [&](int l_, int &s_, int &head_, int &q4_) {
        const int pair = (l_ >> 2) * 8 + xcd;
        q4_ = l_ & 3; head_ = pair & 15; s_ = pair >> 4;
    };
    int s, head, q4;
    coords(l, s, head, q4);
    const unsigned short *ksrc = qkv + (size_t)s * 1024 * 3072 + head * 64 + koff;
    const unsigned short *vsrc = vT + ((size_t)s * 16 + head) * 64 * 1024 + voff;
    constexpr bool AF = GL == 2;
    if constexpr (!GL) { issue(ksrc, vsrc, 0); issue(ksrc, vsrc, 1); issue(ksrc, vsrc, 2); }

    // ---- Q fragments (MFMA B operand) of both chains of the first item
    u32x4 qf[2][4];
    {
        const int qh0 = q4 * 8 + wave * 2;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const unsigned short *qrow = qkv + ((size_t)s * 1024 + (qh0 + c) * 32 + r) * 3072 + head * 64;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[c][ks] = *reinterpret_cast<const u32x4 *>(qrow + 16 * ks + 8 * h2);
        }
    }

    // ---- per-lane LDS byte addresses of the fragments (slot offset added as an immediate)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)a2_smem;
    const int krow = pi_perm(r);
    unsigned ka[4], va[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ka[ks] = lds0 + (unsigned)(krow * 128 + (((2 * ks + h2) ^ ((krow >> 1) & 7)) * 16));
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int d = db * 32 + r;
            va[db * 2 + st] = lds0 + 4096u + (unsigned)(d * 64 + (((2 * st + h2) ^ ((d >> 2) & 3)) * 16));
        }
    _Float16 *Gbase = reinterpret_cast<_Float16 *>(a2_smem + 4 * A2_SLOT) + wave * 2 * 32 * A2_G_LD;
    // GL: this lane's fragment origins inside a (sub-tile, head) pair: K row pi(r), d 8 h2 (+ 16 ks by immediate);
    // V^T rows r and 32 + r, key 8 h2 (+ 16 st by immediate); tile t adds 32 rows of qkv / 32 keys
    const int klane = krow * 3072 + 1024 + 8 * h2, vlane = r * 1024 + 8 * h2;
    auto kptr = [&](int s_, int head_, int t) { return qkv + (size_t)s_ * 1024 * 3072 + head_ * 64 + klane + (size_t)t * 32 * 3072; };
    auto vptr = [&](int s_, int head_, int t) { return vT + ((size_t)s_ * 16 + head_) * 64 * 1024 + vlane + t * 32; };
    auto load_k = [&](u32x4 (&dst)[4], const unsigned short *p) {
        if constexpr (AF) { dst[0] = g_load128a<0>(p); dst[1] = g_load128a<32>(p); dst[2] = g_load128a<64>(p); dst[3] = g_load128a<96>(p); }
        else { dst[0] = g_load128<0>(p); dst[1] = g_load128<32>(p); dst[2] = g_load128<64>(p); dst[3] = g_load128<96>(p); }
    };
    auto load_v = [&](u32x4 (&dst)[4], const unsigned short *p) {          // [db * 2 + st]
        const unsigned short *p1 = p + 32 * 1024;
        if constexpr (AF) { dst[0] = g_load128a<0>(p); dst[1] = g_load128a<32>(p); dst[2] = g_load128a<0>(p1); dst[3] = g_load128a<32>(p1); }
        else { dst[0] = g_load128<0>(p); dst[1] = g_load128<32>(p); dst[2] = g_load128<0>(p1); dst[3] = g_load128<32>(p1); }
    };

    const float cexp = 0.125f * 1.44269504088896340736f;
    const f32x2 cexp2 = {cexp, cexp};
    constexpr float HEADROOM = F16 ? 3.0f : 6.0f;
    const unsigned one2 = F16 ? 0x3C003C00u : 0x3F803F80u;
    // SEL fragment of the row-sum product (MF_rowsum): lane l holds SEL[m = l & 15][k-block l >> 4]
    const unsigned selv = ((lane & 15) == ((lane >> 4) & 1)) ? one2 : 0u;
    u32x4 ones = {selv, selv, selv, selv};
    asm volatile("" : "+v"(ones));                        // a VGPR tuple, materialised once
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    f32x16 S[2], O0[2], O1[2], GW[2];
    f32x4 L[2];
    float m_run[2];
    u32x4 kf[4][4], vf[4][4], pf[2][2];          // fragment sets: 2 in use (by tile parity) for GL < 2, 4 (tile & 3) for GL = 2
#pragma unroll
    for (int i = 0; i < 4; ++i) { vf[1][i] = zero4; vf[0][i] = zero4; kf[1][i] = zero4; kf[0][i] = zero4; vf[3][i] = zero4; }
    unsigned ghb[2], gaddr[2];
    f32x2 off2[2];
    using std::integral_constant;
    if constexpr (GL) { load_k(kf[0], kptr(s, head, 0)); load_k(kf[1], kptr(s, head, 1)); load_v(vf[0], vptr(s, head, 0)); }
    if constexpr (GL == 2) { load_k(kf[2], kptr(s, head, 2)); load_v(vf[1], vptr(s, head, 1)); }

    for (;;) {
        const int l_next = l + l_stride;
        const bool has_next = l_next < n_l;
        int s_n = 0, head_n = 0, q4_n = 0;
        if (has_next) coords(l_next, s_n, head_n, q4_n);
        const unsigned short *ksrc_n = qkv + (size_t)s_n * 1024 * 3072 + head_n * 64 + koff;
        const unsigned short *vsrc_n = vT + ((size_t)s_n * 16 + head_n) * 64 * 1024 + voff;
        const int qh0 = q4 * 8 + wave * 2;                     // image rows of this wave's two chains: qh0, qh0 + 1
        const size_t tok0 = (size_t)s * 1024;

        // ---- seam: G = Q table^T -> Gw (C operand of S, registers) and Gh (fp16 LDS scratch)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            _Float16 *G = Gbase + c * 32 * A2_G_LD;
            auto compute_G = [&](const u32x4 (&tf_)[2][4], const unsigned short *table) {
                f32x16 acc[2];
                if constexpr (GL == 2) {
                    u32x4 tl[2][4];
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) tl[jb][ks] = *reinterpret_cast<const u32x4 *>(table + (jb * 32 + r) * 64 + 16 * ks + 8 * h2);
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc[jb]) : "v"(tl[jb][0]), "v"(qf[c][0]));
                        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc[jb]) : "v"(tl[jb][0]), "v"(qf[c][0]));
#pragma unroll
                        for (int ks = 1; ks < 4; ++ks) {
                            if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[jb]) : "v"(tl[jb][ks]), "v"(qf[c][ks]));
                            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[jb]) : "v"(tl[jb][ks]), "v"(qf[c][ks]));
                        }
                    }
                } else {
                const u32x4 (&tf)[2][4] = tf_;
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc[jb]) : "a"(tf[jb][0]), "v"(qf[c][0]));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc[jb]) : "a"(tf[jb][0]), "v"(qf[c][0]));
#pragma unroll
                    for (int ks = 1; ks < 4; ++ks) {
                        if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[jb]) : "a"(tf[jb][ks]), "v"(qf[c][ks]));
                        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[jb]) : "a"(tf[jb][ks]), "v"(qf[c][ks]));
                    }
                }
                }
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // MFMA results -> vector pipe
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {        // registers 4 g4 .. 4 g4 + 3 = columns jb 32 + 8 g4 + 4 h2 + (0..3): 8 bytes
                        uint2 w;
                        w.x = pack2<true>(acc[jb][4 * g4 + 0], acc[jb][4 * g4 + 1]);
                        w.y = pack2<true>(acc[jb][4 * g4 + 2], acc[jb][4 * g4 + 3]);
                        *reinterpret_cast<uint2 *>(G + r * A2_G_LD + jb * 32 + 8 * g4 + 4 * h2) = w;
                    }
            };
            compute_G(tfw, relw);
#pragma unroll
            for (int i = 0; i < 16; ++i) GW[c][i] = (float)G[r * A2_G_LD + (r - pi_perm((i & 3) + 8 * (i >> 2) + 4 * h2) + 31)];
            compute_G(tfh, relh);
            // Gh[q][qh - kh + 31] of chain c: address for kh = 0, minus 2 bytes per tile
            gaddr[c] = lds0 + 4u * A2_SLOT + (unsigned)(((wave * 2 + c) * 32 * A2_G_LD + r * A2_G_LD + qh0 + c + 31) * 2);
        }
        // NV (no-vote mode, GL only): the key loop runs WITHOUT the per-half-step overflow vote (tile 0, whose reference is still
        // -inf, excepted).  That is exact as long as nothing overflows: the softmax is shift-invariant, a probability formed
        // against a reference that a later, larger score overtakes is simply a large finite number (bf16 carries the exponent
        // range of f32), and the accumulators are f32.  Only a score that beats the first tile's maximum by more than ~120
        // octaves (83 nats) can overflow; the row sums then read inf / NaN (or >= 2^100) at the end of the item and THIS WAVE
        // re-runs the item in the exact mode (vote + rescale every half step, the behaviour of the other variants).
        bool safe = !NV, redo = false;
      for (;;) {                           // passes over this item (NV: at most two)
#pragma unroll
        for (int c = 0; c < 2; ++c) { A_zero16(O0[c]); A_zero16(O1[c]); A_zero4(L[c]); }
        pf[0][0] = pf[0][1] = pf[1][0] = pf[1][1] = zero4;
        m_run[0] = m_run[1] = -1e30f;
        if constexpr (GL && NV) {
            if (redo) { load_k(kf[0], kptr(s, head, 0)); load_k(kf[1], kptr(s, head, 1)); load_v(vf[0], vptr(s, head, 0)); }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // tiles 0..2 (GL: K(0), K(1), V(0)) and this wave's G rows have landed
        if constexpr (!GL) __builtin_amdgcn_s_barrier();                 // (GL: the waves never touch each other's data)

        // pipeline fill: K(0) fragments, gh of tile 0 for both chains, S[0] = K(0) Q0^T + Gw0
        if constexpr (!GL) {
            kf[0][0] = a2_read128<0>(ka[0]); kf[0][1] = a2_read128<0>(ka[1]);
            kf[0][2] = a2_read128<0>(ka[2]); kf[0][3] = a2_read128<0>(ka[3]);
        }
        asm volatile("ds_read_u16 %0, %1" : "=v"(ghb[0]) : "v"(gaddr[0]));
        asm volatile("ds_read_u16 %0, %1" : "=v"(ghb[1]) : "v"(gaddr[1]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        MF_init<F16, AF>(S[0], kf[0][0], qf[0][0], GW[0]);
        MF_accv<F16, AF>(S[0], kf[0][1], qf[0][1]);
        MF_accv<F16, AF>(S[0], kf[0][2], qf[0][2]);
        MF_accv<F16, AF>(S[0], kf[0][3], qf[0][3]);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");       // S[0] is read by the vector pipe below
        // exponent offset of the FIRST softmax of each chain: (gh - m_run) * cexp, as a broadcast pair
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float o = ((float)__builtin_bit_cast(_Float16, (unsigned short)ghb[c]) - m_run[c]) * cexp;
            off2[c] = (f32x2){o, o};
        }
        A2_STAMP(4);

        // rare exact path of chain C (some p >= 2, or the reference still at -inf): true maximum, rescale, recompute P
        auto softmax_exact = [&](auto chain_tag, unsigned gh_bits, unsigned (&pk)[8]) {
            constexpr int C = decltype(chain_tag)::value;
            const float gh = (float)__builtin_bit_cast(_Float16, (unsigned short)gh_bits);
            float mx = __builtin_fmaxf(__builtin_fmaxf(S[C][0], S[C][1]), S[C][2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, S[C][i]), S[C][i + 1]);
            mx = __builtin_fmaxf(mx, S[C][15]);
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32)) + gh + HEADROOM / cexp;
            const float m_new = __builtin_fmaxf(m_run[C], mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run[C] - m_new) * cexp);
            m_run[C] = m_new;
            {   // row sums: lanes 0..15 hold (query lane, query lane + 16)
                const float alpha_hi = __shfl(alpha, (lane & 15) + 16);
                float x0 = L[C][0], x1 = L[C][1], t;
                asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1" : "+a"(x0), "=&v"(t) : "v"(alpha));
                asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1" : "+a"(x1), "=&v"(t) : "v"(alpha_hi));
                L[C][0] = x0; L[C][1] = x1;
            }
            A_scale16(O0[C], alpha); A_scale16(O1[C], alpha);
            const float off = (gh - m_run[C]) * cexp;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                pk[j] = pack2<F16>(__builtin_amdgcn_exp2f(__builtin_fmaf(S[C][2 * j], cexp, off)),
                                   __builtin_amdgcn_exp2f(__builtin_fmaf(S[C][2 * j + 1], cexp, off)));
            asm volatile("s_nop 7" ::: "memory");               // rescaled accumulators / P before the next MFMAs read them
        };

        // ONE half step: the vector pipe forms P of chain C from S[C] (tile kh) while the matrix pipe computes S[X] of the
        // other chain from the K fragments `kx` and adds chain X's PREVIOUS P (pf[X]) times the V fragments `vx` to O[X] / L[X].
        // 10 MFMAs, each followed by <= 28 issue cycles of vector work (v_fma 4, v_exp 8, v_cvt_pk 4).  Afterwards the
        // exponent offset of chain X's NEXT softmax is prepared from gh_next_x.
        auto half_step = [&](auto chain_tag, auto vote_tag, const u32x4 (&kx)[4], const u32x4 (&vx)[4], unsigned gh_c, unsigned gh_next_x) {
            constexpr int C = decltype(chain_tag)::value, X = 1 - C;
            constexpr bool VOTE = decltype(vote_tag)::value;
            unsigned pk[8];
            f32x2 a0, a1, a2, a3, a4, a5, a6, a7;
            float e0, e1, e2, e3;
            const f32x2 of = off2[C];
            // ABL (stamped debug build, timing only): 1 = this stream without its MFMAs, 2 = without its vector instructions
#define MF_init if constexpr (ABL != 1 && ABL != 3) MF_init
#define MF_accv if constexpr (ABL != 1 && ABL != 3) MF_accv
#define MF_acca if constexpr (ABL != 1 && ABL != 3) MF_acca
#define MF_rowsum if constexpr (ABL != 1 && ABL != 3) MF_rowsum
#define V_pkfma(A_, B_, C_) (ABL >= 2 ? (f32x2){0.f, 0.f} : V_pkfma(A_, B_, C_))
#define V_exp(A_) (ABL >= 2 ? 0.f : V_exp(A_))
#define V_or3(A_, B_, C_) (ABL >= 2 ? 0u : V_or3(A_, B_, C_))
#define V_cvtx(A_, B_) (ABL >= 2 ? 0u : V_cvt<F16>(A_, B_))
#define SV(j) ((f32x2){S[C][2 * (j)], S[C][2 * (j) + 1]})
            // (MFMAs on one accumulator sit >= 4 gaps apart, the S chain 2 gaps: a dependent 32x32x16 needs ~70 cycles)
            MF_init<F16, AF>(S[X], kx[0], qf[X][0], GW[X]);
            a0 = V_pkfma(SV(0), cexp2, of); a1 = V_pkfma(SV(1), cexp2, of); e0 = V_exp(a0.x); e1 = V_exp(a0.y);
            MF_rowsum<F16>(L[X], ones, pf[X][0]);
            a2 = V_pkfma(SV(2), cexp2, of); e2 = V_exp(a1.x); e3 = V_exp(a1.y); pk[0] = V_cvtx(e0, e1);
            MF_accv<F16, AF>(S[X], kx[1], qf[X][1]);
            a3 = V_pkfma(SV(3), cexp2, of); e0 = V_exp(a2.x); e1 = V_exp(a2.y); pk[1] = V_cvtx(e2, e3);
            MF_acca<F16, AF>(O0[X], vx[0], pf[X][0]);
            a4 = V_pkfma(SV(4), cexp2, of); e2 = V_exp(a3.x); e3 = V_exp(a3.y); pk[2] = V_cvtx(e0, e1);
            MF_accv<F16, AF>(S[X], kx[2], qf[X][2]);
            a5 = V_pkfma(SV(5), cexp2, of); e0 = V_exp(a4.x); e1 = V_exp(a4.y); pk[3] = V_cvtx(e2, e3);
            MF_acca<F16, AF>(O1[X], vx[2], pf[X][0]);
            a6 = V_pkfma(SV(6), cexp2, of); e2 = V_exp(a5.x); e3 = V_exp(a5.y); pk[4] = V_cvtx(e0, e1);
            MF_accv<F16, AF>(S[X], kx[3], qf[X][3]);
            a7 = V_pkfma(SV(7), cexp2, of); e0 = V_exp(a6.x); e1 = V_exp(a6.y); pk[5] = V_cvtx(e2, e3);
            MF_acca<F16, AF>(O0[X], vx[1], pf[X][1]);
            e2 = V_exp(a7.x); e3 = V_exp(a7.y); pk[6] = V_cvtx(e0, e1);
            unsigned u0 = 0, u1 = 0;
            if constexpr (VOTE) u0 = V_or3(pk[0], pk[1], pk[2]);
            MF_acca<F16, AF>(O1[X], vx[3], pf[X][1]);
            if constexpr (VOTE) u1 = V_or3(pk[3], pk[4], pk[5]);
            pk[7] = V_cvtx(e2, e3);
            // the overflow vote: compare BEFORE the last MFMA, branch (scalar) behind it
            unsigned long long vote = 0;
            if constexpr (VOTE) {
                u0 = V_or3(u0, u1, pk[6]);
                asm volatile("v_or_b32 %1, %1, %2\n\tv_and_b32 %1, 0x40004000, %1\n\tv_cmp_ne_u32 %0, 0, %1" : "=s"(vote), "+v"(u0) : "v"(pk[7]));
            }
            MF_rowsum<F16>(L[X], ones, pf[X][1]);
#undef SV
#undef MF_init
#undef MF_accv
#undef MF_acca
#undef MF_rowsum
#undef V_pkfma
#undef V_exp
#undef V_or3
#undef V_cvtx
            {   // offset of chain X's next softmax (its m_run changes only in its own exact path, which is behind us)
                const float o = ((float)__builtin_bit_cast(_Float16, (unsigned short)gh_next_x) - m_run[X]) * cexp;
                off2[X] = (f32x2){o, o};
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (VOTE) { if (__builtin_expect(vote != 0, 0)) softmax_exact(chain_tag, gh_c, pk); }
            pf[C][0] = (u32x4){pk[0], pk[1], pk[2], pk[3]};
            pf[C][1] = (u32x4){pk[4], pk[5], pk[6], pk[7]};
        };
        // one key tile = two half steps.  SL = ring slot of tile kh, SN = slot of tile kh + 1, PAR = kh & 1.
        auto tile = [&](const int kh, auto slot_tag, auto next_tag, auto par_tag) {
            constexpr int SL = decltype(slot_tag)::value, SN = decltype(next_tag)::value, PAR = decltype(par_tag)::value;
            if constexpr (!GL) {
            // tile kh + 1 (this thread's part) has landed; behind the barrier every part has, and every wave is done with
            // slot (kh - 1) & 3 (its K and V fragments went to registers a tile ago), which the next request overwrites
            if (kh < 30 || has_next) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            A2_STAMP(0);
            if (kh + 3 < 32) issue(ksrc, vsrc, kh + 3);
            else if (has_next) issue(ksrc_n, vsrc_n, kh + 3 - 32);      // the ring runs on into the next item
            // fragments for half step B and for the next tile (kh = 31: rows of the NEXT item's tile 0 or stale rows: never used)
            kf[PAR ^ 1][0] = a2_read128<SN * A2_SLOT>(ka[0]); kf[PAR ^ 1][1] = a2_read128<SN * A2_SLOT>(ka[1]);
            kf[PAR ^ 1][2] = a2_read128<SN * A2_SLOT>(ka[2]); kf[PAR ^ 1][3] = a2_read128<SN * A2_SLOT>(ka[3]);
            vf[PAR][0] = a2_read128<SL * A2_SLOT>(va[0]); vf[PAR][1] = a2_read128<SL * A2_SLOT>(va[1]);
            vf[PAR][2] = a2_read128<SL * A2_SLOT>(va[2]); vf[PAR][3] = a2_read128<SL * A2_SLOT>(va[3]);
            }
            unsigned ghn[2];
            asm volatile("ds_read_u16 %0, %1" : "=v"(ghn[0]) : "v"(gaddr[0] - 2u * (unsigned)(kh + 1)));
            asm volatile("ds_read_u16 %0, %1" : "=v"(ghn[1]) : "v"(gaddr[1] - 2u * (unsigned)(kh + 1)));
            __builtin_amdgcn_sched_barrier(0);
            A2_STAMP(1);
            // half step A: softmax of chain 0 (tile kh) | S[1] = K(kh) Q1^T + Gw1, P.V and row sums of chain 1's tile kh - 1
            const bool v = safe || kh == 0;        // (NV: votes only in tile 0 -- the reference starts at -inf -- and in a re-run)
            // fragment sets: K(kh), V(kh - 1) for half step A; K(kh + 1), V(kh) for half step B
            constexpr int KA = GL == 2 ? SL : PAR, VA = GL == 2 ? (SL + 3) & 3 : PAR ^ 1, KB = GL == 2 ? SN : PAR ^ 1, VB = GL == 2 ? SL : PAR;
            if (!NV || v) half_step(integral_constant<int, 0>{}, std::true_type{}, kf[KA], vf[VA], ghb[0], ghb[1]);
            else half_step(integral_constant<int, 0>{}, integral_constant<bool, !NV>{}, kf[KA], vf[VA], ghb[0], ghb[1]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the ten reads above: long since landed
            if constexpr (GL == 1) {
                // K(kh + 1) and V(kh), requested one tile ago, are what this half step's MFMAs read; then the requests for
                // K(kh + 2) (into the registers half step A just finished with) and V(kh + 1); past the item's last tile:
                // the next item's first tiles (its pipeline fill finds K(0), K(1), V(0) in place) or a harmless re-read
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int tk = kh + 2, tv = kh + 1;
                const unsigned short *pk_ = tk < 32 ? kptr(s, head, tk) : (has_next ? kptr(s_n, head_n, tk - 32) : kptr(s, head, 31));
                const unsigned short *pv_ = tv < 32 ? vptr(s, head, tv) : (has_next ? vptr(s_n, head_n, tv - 32) : vptr(s, head, 31));
                load_k(kf[PAR], pk_);
                load_v(vf[PAR ^ 1], pv_);
            }
            if constexpr (GL == 2) {
                // two tiles ahead: K(kh + 1) and V(kh) were requested two half-step-B's ago -- everything but the youngest
                // eight requests has landed -- then K(kh + 3) and V(kh + 2) go into the sets last used by tiles kh - 1 / kh - 2
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                const int tk = kh + 3, tv = kh + 2;
                const unsigned short *pk_ = tk < 32 ? kptr(s, head, tk) : (has_next ? kptr(s_n, head_n, tk - 32) : kptr(s, head, 31));
                const unsigned short *pv_ = tv < 32 ? vptr(s, head, tv) : (has_next ? vptr(s_n, head_n, tv - 32) : vptr(s, head, 31));
                load_k(kf[(SL + 3) & 3], pk_);
                load_v(vf[(SL + 2) & 3], pv_);
            }
            __builtin_amdgcn_sched_barrier(0);
            A2_STAMP(2);
            // half step B: softmax of chain 1 (tile kh) | S[0] = K(kh + 1) Q0^T + Gw0, P.V and row sums of chain 0's tile kh
            if (!NV || v) half_step(integral_constant<int, 1>{}, std::true_type{}, kf[KB], vf[VB], ghb[1], ghn[0]);
            else half_step(integral_constant<int, 1>{}, integral_constant<bool, !NV>{}, kf[KB], vf[VB], ghb[1], ghn[0]);
            ghb[0] = ghn[0]; ghb[1] = ghn[1];
            A2_STAMP(3);
        };
        for (int kh0 = 0; kh0 < 32; kh0 += 4) {
            tile(kh0 + 0, integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{});
            tile(kh0 + 1, integral_constant<int, 1>{}, integral_constant<int, 2>{}, integral_constant<int, 1>{});
            tile(kh0 + 2, integral_constant<int, 2>{}, integral_constant<int, 3>{}, integral_constant<int, 0>{});
            tile(kh0 + 3, integral_constant<int, 3>{}, integral_constant<int, 0>{}, integral_constant<int, 1>{});
        }
        // pipeline drain: P.V and row sums of chain 1's last tile (V(31) sits in vf[1])
        constexpr int VD = GL == 2 ? 3 : 1;                  // the set holding V(31)
        MF_acca<F16, AF>(O0[1], vf[VD][0], pf[1][0]);
        MF_acca<F16, AF>(O1[1], vf[VD][2], pf[1][0]);
        MF_acca<F16, AF>(O0[1], vf[VD][1], pf[1][1]);
        MF_acca<F16, AF>(O1[1], vf[VD][3], pf[1][1]);
        MF_rowsum<F16>(L[1], ones, pf[1][0]);
        MF_rowsum<F16>(L[1], ones, pf[1][1]);
        if constexpr (NV) {
            if (!safe) {
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // row sums (AGPRs) -> vector pipe
                const float l00 = L[0][0], l01 = L[0][1], l10 = L[1][0], l11 = L[1][1];
                // exponent field >= 227: >= 2^100, inf or NaN (an integer test: this file is built with -fno-honor-nans)
                const unsigned em = 0x7F800000u, lim = 0x71800000u;
                const bool bad = (__float_as_uint(l00) & em) >= lim || (__float_as_uint(l01) & em) >= lim ||
                                 (__float_as_uint(l10) & em) >= lim || (__float_as_uint(l11) & em) >= lim;
                if (__any(bad)) { safe = true; redo = true; continue; }
            }
        }
        break;
      }
        // the next item's Q rows: requested now, consumed by the next seam's G products (behind the stores below)
        if (has_next) {
            const int qh0n = q4_n * 8 + wave * 2;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const unsigned short *qrow = qkv + ((size_t)s_n * 1024 + (qh0n + c) * 32 + r) * 3072 + head_n * 64;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qf[c][ks] = *reinterpret_cast<const u32x4 *>(qrow + 16 * ks + 8 * h2);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");       // the accumulators are read by the vector pipe below
        if constexpr (DBG) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
        // normalise, round, and leave as WHOLE 128-byte rows: the lane's 4 consecutive d of query r go to this wave's (now
        // idle) G scratch as [query][64 d] with 144-byte rows, and come back as 16-byte chunks, 8 lanes per query row
        // (a per-lane 8-byte store at a 2 KB row stride touches 64 lines per instruction)
        {
            char *stg = reinterpret_cast<char *>(Gbase);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float l_lo = __shfl(L[c][0], r & 15), l_hi = __shfl(L[c][1], r & 15);
                const float inv = 1.0f / (r < 16 ? l_lo : l_hi);
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int d = db * 32 + 8 * g4 + 4 * h2;
                        const f32x16 &O = db ? O1[c] : O0[c];
                        uint2 o;
                        o.x = pack2<F16>(O[4 * g4 + 0] * inv, O[4 * g4 + 1] * inv);
                        o.y = pack2<F16>(O[4 * g4 + 2] * inv, O[4 * g4 + 3] * inv);
                        *reinterpret_cast<uint2 *>(stg + r * 144 + d * 2) = o;
                    }
                unsigned short *obase = out + (tok0 + (qh0 + c) * 32) * 1024 + head * 64;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int q = it * 8 + (lane >> 3), ch = lane & 7;
                    const uint4 v = *reinterpret_cast<const uint4 *>(stg + q * 144 + ch * 16);
                    *reinterpret_cast<uint4 *>(obase + (size_t)q * 1024 + ch * 8) = v;
                }
            }
        }
        A2_STAMP(6);
        if constexpr (DBG) seg[7] += 1;
        if (!has_next) break;
        l = l_next; s = s_n; head = head_n; q4 = q4_n; ksrc = ksrc_n; vsrc = vsrc_n;
    }
    if constexpr (DBG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long t_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        seg[5] = (unsigned)(t_ - tstart);
        if (lane == 0 && dbg) {
            unsigned *d = dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = seg[i];
        }
    }
}

// launch: persistent, min(#CUs, 64 n_subtiles) workgroups of 4 waves (one per CU; a multiple of 8)
static int a2_grid(int n_subtiles) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        n_cu = cus >= 8 ? (cus / 8) * 8 : 8;
    }
    const int items = 64 * n_subtiles;
    return items < n_cu ? items : n_cu;
}
int cpx_attention2q_launch(int dtype, const void *qkv, const void *vT, const void *rel_h, const void *rel_w,
                           int n_subtiles, void *out, int gl, hipStream_t s) {
    static CpxOncePerDevice once;
    once([] {
        (void)hipFuncSetAttribute((const void *)k_attention2q<true>, hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)k_attention2q<false>, hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)k_attention2q<true, false, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)k_attention2q<false, false, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS_BYTES);
    });
    const dim3 grid(a2_grid(n_subtiles));
    if (gl == 3) {                                  // two tiles ahead, fragment sets in AGPRs
        static CpxOncePerDevice once_g2;
        once_g2([] {
            (void)hipFuncSetAttribute((const void *)k_attention2q<true, false, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS_BYTES);
            (void)hipFuncSetAttribute((const void *)k_attention2q<false, false, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS_BYTES);
        });
        if (dtype == CPX_DT_F16)
            hipLaunchKernelGGL((k_attention2q<true, false, 0, 2>), grid, dim3(A2_THREADS), A2_LDS_BYTES, s, (const unsigned short *)qkv,
                               (const unsigned short *)vT, (const unsigned short *)rel_h, (const unsigned short *)rel_w, (unsigned short *)out,
                               16 * n_subtiles, (unsigned *)nullptr);
        else
            hipLaunchKernelGGL((k_attention2q<false, false, 0, 2>), grid, dim3(A2_THREADS), A2_LDS_BYTES, s, (const unsigned short *)qkv,
                               (const unsigned short *)vT, (const unsigned short *)rel_h, (const unsigned short *)rel_w, (unsigned short *)out,
                               16 * n_subtiles, (unsigned *)nullptr);
        CPX_CHECK_LAUNCH();
        return CPX_OK;
    }
    if (gl == 2 && dtype != CPX_DT_F16) {          // no-vote mode (bf16 only: fp16 probabilities overflow at 2^16, far too soon)
        static CpxOncePerDevice once_nv;
        once_nv([] { (void)hipFuncSetAttribute((const void *)k_attention2q<false, false, 0, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS_BYTES); });
        hipLaunchKernelGGL((k_attention2q<false, false, 0, 1, true>), grid, dim3(A2_THREADS), A2_LDS_BYTES, s, (const unsigned short *)qkv,
                           (const unsigned short *)vT, (const unsigned short *)rel_h, (const unsigned short *)rel_w, (unsigned short *)out,
                           16 * n_subtiles, (unsigned *)nullptr);
        CPX_CHECK_LAUNCH();
        return CPX_OK;
    }
    if (gl) {
        if (dtype == CPX_DT_F16)
            hipLaunchKernelGGL((k_attention2q<true, false, 0, 1>), grid, dim3(A2_THREADS), A2_LDS_BYTES, s, (const unsigned short *)qkv,
                               (const unsigned short *)vT, (const unsigned short *)rel_h, (const unsigned short *)rel_w, (unsigned short *)out,
                               16 * n_subtiles, (unsigned *)nullptr);
        else
            hipLaunchKernelGGL((k_attention2q<false, false, 0, 1>), grid, dim3(A2_THREADS), A2_LDS_BYTES, s, (const unsigned short *)qkv,
                               (const unsigned short *)vT, (const unsigned short *)rel_h, (const unsigned short *)rel_w, (unsigned short *)out,
                               16 * n_subtiles, (unsigned *)nullptr);
        CPX_CHECK_LAUNCH();
        return CPX_OK;
    }
    if (dtype == CPX_DT_F16)
        hipLaunchKernelGGL((k_attention2q<true>), grid, dim3(A2_THREADS), A2_LDS_BYTES, s, (const unsigned short *)qkv,
                           (const unsigned short *)vT, (const unsigned short *)rel_h, (const unsigned short *)rel_w, (unsigned short *)out,
                           16 * n_subtiles, (unsigned *)nullptr);
    else
        hipLaunchKernelGGL((k_attention2q<false>), grid, dim3(A2_THREADS), A2_LDS_BYTES, s, (const unsigned short *)qkv,
                           (const unsigned short *)vT, (const unsigned short *)rel_h, (const unsigned short *)rel_w, (unsigned short *)out,
                           16 * n_subtiles, (unsigned *)nullptr);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}
#ifdef CPX_DEBUG
// diagnostic build with cycle stamps: dbg [workgroups = min(#CUs, 64 n_subtiles)][4 waves][8]; vT holds V^T already
static int g_a2_abl = 0;
extern "C" void cpx_attention2q_set_ablation(int v) { g_a2_abl = v; }
extern "C" int cpx_attention2q_debug(const void *qkv, const void *vT, const void *rel_h, const void *rel_w, int n_subtiles,
                                     void *out, unsigned *dbg, void *stream) {
#define A2_DBG_LAUNCH(ABL_) do {                                                                                                  \
    (void)hipFuncSetAttribute((const void *)k_attention2q<false, true, ABL_>, hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS_BYTES); \
    hipLaunchKernelGGL((k_attention2q<false, true, ABL_>), dim3(a2_grid(n_subtiles)), dim3(A2_THREADS), A2_LDS_BYTES, (hipStream_t)stream, \
                       (const unsigned short *)qkv, (const unsigned short *)vT, (const unsigned short *)rel_h,                   \
                       (const unsigned short *)rel_w, (unsigned short *)out, 16 * n_subtiles, dbg); } while (0)
    if (g_a2_abl == 1) A2_DBG_LAUNCH(1); else if (g_a2_abl == 2) A2_DBG_LAUNCH(2); else if (g_a2_abl == 3) A2_DBG_LAUNCH(3);
    else if (g_a2_abl == 4) {        // the GL variant (fragments straight from global memory), stamped
        (void)hipFuncSetAttribute((const void *)k_attention2q<false, true, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS_BYTES);
        hipLaunchKernelGGL((k_attention2q<false, true, 0, 1>), dim3(a2_grid(n_subtiles)), dim3(A2_THREADS), A2_LDS_BYTES, (hipStream_t)stream,
                           (const unsigned short *)qkv, (const unsigned short *)vT, (const unsigned short *)rel_h,
                           (const unsigned short *)rel_w, (unsigned short *)out, 16 * n_subtiles, dbg);
    } else A2_DBG_LAUNCH(0);
#undef A2_DBG_LAUNCH
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}
#endif
#endif  // CPX_DEBUG
