// bf16 / fp16 MFMA GEMM for the ClassTransformer linear layers on gfx950.
//
//   C[M][N] = A[M][K] * W[N][K]^T  (+ fused epilogue)
//
// Both operands are K-contiguous ("B^T input"), so an MFMA fragment is 8
// contiguous elements = one 16-byte read.  The MFMA is issued with W as the
// matrix "A" operand and the activations as "B": the 16x16 accumulator then
// holds 4 CONSECUTIVE output channels of one token per lane, which makes the
// epilogue (bias / GELU / residual / positional embedding) a vector op and the
// store an 8-byte (bf16) or 16-byte (f32) write.
//
// Tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves (2 x 2), 64 x 64 per wave
// = 4 x 4 MFMA 16x16x32 accumulators.  Tiles are staged global -> LDS with
// 16-byte LDS-DMA (global_load_lds_dwordx4), double buffered; the LDS image is
// lane-linear, the 16-byte chunk position is XOR-swizzled through the SOURCE
// address (chunk ^ (row & 7)) and un-swizzled on the ds_read_b128 -> conflict
// free fragment reads.  Workgroup ids are remapped so that the
// blocks that share an XCD (id % 8) walk a contiguous band of M rows and re-use
// the activation panel from that XCD's L2.
#include "cpx_internal.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define BM 128
#define BN 128
#define BK 64
#define GEMM_THREADS 256
#define TILE_BYTES (128 * 128)          // one operand tile: 128 rows x 64 k x 2 B
#define STAGE_BYTES (2 * TILE_BYTES)

// 16-byte store of an epilogue, optionally non-temporal.  Production for the qkv projection (g.nt_out): its 268 MB of output (q | k rows and
// the V^T image) pass through the 256 MB Infinity Cache once on their way to the attention kernel and push everything else out of it -- the
// residual stream that attn.proj's epilogue adds, the weights; with the hint they do not stay.  In situ (tools/ab_stage_switch.py
// cpx_gemm_set_nt=0,1, launch medians): qkv 181.2 -> 177.5 us, attn.proj 72.6 -> 70.4, attention 181.9 -> 183.6 (its K / V^T tiles now
// come from HBM): -4 us per layer, engine step 21.34 -> 21.24 ms, same bits.  (The same hint on the attention kernel's K / V^T REQUESTS
// loses the L2 sharing of the eight query blocks of a head: 184 -> 199 us, profiles/r05_ab_nontemporal.txt.)
// (round 6) WT: the store at agent scope (sc1): the line is written through to the memory side at once instead of staying dirty in this XCD's
// 4 MB L2 until something evicts it -- the epilogues of all 256 workgroups burst at the same moment (32 x 128 KB per XCD = the whole L2)
template <bool NT_POSSIBLE, bool WT_POSSIBLE = false>
__device__ __forceinline__ void st16(void *p, uint4 v, bool nt, bool wt = false) {
    if (WT_POSSIBLE && wt) {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t w = {v.x, v.y, v.z, v.w};
        if (NT_POSSIBLE && nt) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(w) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
    } else if (NT_POSSIBLE && nt) {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t *>(p));
    } else {
        *reinterpret_cast<uint4 *>(p) = v;
    }
}
struct GemmArgs {
    const unsigned short *A;
    const unsigned short *W;
    int M, N, K;
    const float *bias;
    const void *aux;        // residual (bf16 [M][ld_out]) or pos (f32 [1024][N])
    void *out;
    int ld_out;
    int tiles_n, n_blocks;
    int l2_block;           // 1 / 2: 8 x 4 super-tile order per XCD, N-sweep / M-sweep (debug switch, default 1)
    int rev_m;              // 1: walk the M tiles from the last row block to the first (see gemm_launch)
    int dbg;                // timing-only ablations of the 256^2 epilogue (0 in production)
    int nt_out;             // qkv epilogue: bits 0 / 1 / 2 = the q / k / V^T tiles leave by non-temporal stores (7 in production; debug-build A/B)
    // LayerNorm folded into the GEMM (consumer side): out = rstd[m] * (acc - mean[m] * colsum[n]) + bias[n]
    // with W pre-multiplied by gamma, bias = b + W.beta, colsum[n] = sum_k W'[n][k]
    const float *ln_stats;   // [M][4][2] partial (sum, sum of squares) of the K = 1024 input row, or null
    const float *ln_colsum;  // [N]
    // producer side (RESID epilogue of the 256^2 kernel): partial row statistics of the OUTPUT rows
    float *stats_out;        // [M][4][2], slot = column tile; or null
    // implicit 3x3 convolution (128^2 kernel, CONV instantiation): A is the [S*1024][conv_c] token-major activation of
    // 32 x 32-token images, K = 9 * conv_c with k = tap * conv_c + c (tap = 3 (dy + 1) + (dx + 1)); 0 = plain GEMM
    int conv_c;
    int epi4;               // persistent 256^2 kernel: 1 = quarter-tile epilogue (conversion of quarter q beside the stores of q - 1)
    int pp_delay;           // ping-pong kernel: x ~8k cycles the second workgroup of a CU waits at launch (0 = no offset)
};
#define LN_SLOTS 4
__device__ __forceinline__ void ln_row_params(const float *st, int m, float inv_k, float &mean, float &rstd) {
    const float4 a = *reinterpret_cast<const float4 *>(st + (size_t)m * 8);
    const float4 b = *reinterpret_cast<const float4 *>(st + (size_t)m * 8 + 4);
    const float sum = (a.x + a.z) + (b.x + b.z), sq = (a.y + a.w) + (b.y + b.w);
    mean = sum * inv_k;
    rstd = rsqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + 1e-6f);
}

// v + (v permuted by a DPP control): one VALU instruction, no LDS traffic (ds_bpermute-based __shfl_xor
// cost 160 LDS-pipe operations per lane in the residual epilogue)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// sum over each 32-lane half of the wave; the total is valid in lanes 16..31 and 48..63
__device__ __forceinline__ float half_wave_sum(float v) {
    v = dpp_add<0xB1>(v);            // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);            // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);           // row_half_mirror
    v = dpp_add<0x140>(v);           // row_mirror: every lane of a 16-lane row holds the row total
    return dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3: + the total of the row before
}

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const uint4 &a, const uint4 &b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8 *>(&a),
                                                      *reinterpret_cast<const f16x8 *>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&a),
                                                       *reinterpret_cast<const bf16x8 *>(&b), c, 0, 0, 0);
}

// erf-GELU (nn.GELU default) in 7 VALU + 1 transcendental slots per element:
//   gelu(x) = x Phi(x) = max(x, 0) - |x| q(|x|),   q(a) = 0.5 erfc(a / sqrt 2) = 2 ^ P(a)
// (Phi = 1 - q for x >= 0, q for x < 0: no cancellation in the negative tail).  P is a degree-5 polynomial fitted to
// log2 q with the weight a q(a), i.e. for the absolute error of the product |x| q, the only place q enters
// (tools/fit_gelu.py: fit, float32 evaluation against float64 erfc on 4.8 M points of [-14, 14] and far beyond: max |error|
// 6.4e-7 -- 1.3 ulp of the result at x = 4 -- against 4.8e-7 for the Abramowitz-Stegun 7.1.26 form it replaces, which needed a
// v_rcp and five more multiply-adds; both are > 3 orders of magnitude below half-precision output rounding).  The leading
// coefficient is negative, so beyond the fitted interval P -> -inf and q -> 0 without a clamp (|x| q < 1.1e-9 there).
// The epilogue of mlp.lin1 is VALU-bound: every slot per element is ~0.5 us per 256^2 tile.
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x);                                                  // free source modifier
    float p = -4.732933965e-04f;
    p = __fmaf_rn(p, ax, 7.084452800e-03f);
    p = __fmaf_rn(p, ax, -5.182713611e-02f);
    p = __fmaf_rn(p, ax, -4.599926953e-01f);
    p = __fmaf_rn(p, ax, -1.150787739e+00f);
    p = __fmaf_rn(p, ax, -1.000037638e+00f);
    const float q = __builtin_amdgcn_exp2f(p);
    return __fmaf_rn(-ax, q, fmaxf(x, 0.0f));
}

// the form used until round 2 (A/B reference, g.dbg & 16 in the DBG instantiation): erfc by Abramowitz-Stegun 7.1.26,
// q = 0.5 poly(t) t exp(-x^2 / 2), t = 1 / (1 + (p / sqrt2) |x|): 11 VALU + 2 transcendental slots
__device__ __forceinline__ float gelu_erf_as26(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__fmaf_rn(0.23164189f, ax, 1.0f));   // 0.3275911 / sqrt(2); v_rcp_f32 (1 ulp)
    float poly = 0.5f * 1.061405429f;
    poly = __fmaf_rn(poly, t, 0.5f * -1.453152027f);
    poly = __fmaf_rn(poly, t, 0.5f * 1.421413741f);
    poly = __fmaf_rn(poly, t, 0.5f * -0.284496736f);
    poly = __fmaf_rn(poly, t, 0.5f * 0.254829592f);
    const float ex = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);   // exp(-x^2/2)
    const float q = (poly * t) * ex;
    return __fmaf_rn(-ax, q, fmaxf(x, 0.0f));
}

template <bool F16>
__device__ __forceinline__ unsigned short to_half(float f) {
    if constexpr (F16) { _Float16 h = (_Float16)f; return *reinterpret_cast<unsigned short *>(&h); }
    else return f32_to_bf16(f);
}
// two f32 -> one dword of two halves (lo = a, hi = b) in ONE instruction: the scalar to_half pair above lowers to two
// conversions with a dead second source plus a shift and an SDWA OR (4 VALU slots per dword in the epilogues' ISA).
// Same rounding as to_half (round to nearest even; the instruction the compiler itself picks for the scalar cast).
template <bool F16>
__device__ __forceinline__ unsigned pack2(float a, float b) {
    // a vector convert, NOT inline asm: one v_cvt_pk_* either way, but the hazard recogniser must see this VALU write -- v_permlane16_swap
    // needs two wait states behind a VALU write of either operand (the direct-store epilogue), and behind an inline-asm conversion the
    // compiler inserted none: one launch in ~1 500 of the light bf16 epilogue stored a stale half-row (round 4, tools/stress_gemm_variants.py)
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
    const f32x2_t v = {a, b};
    if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
template <bool F16>
__device__ __forceinline__ float from_half(unsigned short u) {
    if constexpr (F16) { return (float)*reinterpret_cast<_Float16 *>(&u); }
    else return bf16_to_f32(u);
}

// all-zero 16-byte chunk: the source of the out-of-image taps of the implicit convolution (an LDS-DMA cannot
// zero-fill, but every lane has its own source address)
__device__ uint4 g_zero_chunk;

// stage one 128x64 operand tile (rows row0.., k from k0) into LDS at `lds` (16 KB).
// CONV: the operand is never materialised -- row = token (s, y, x), the 64-wide K tile lies inside ONE tap (conv_c is
// a multiple of 64), so the chunk comes from token (s, y + dy, x + dx) of the activation or from the zero chunk
template <bool GLDS, bool CONV = false>
__device__ __forceinline__ void stage_tile(const unsigned short *__restrict__ g, int ldk, int row0, int k0,
                                           char *lds, int tid, uint4 *regs, int conv_c = 0) {
    int dy = 0, dx = 0, c0 = 0;
    if constexpr (CONV) {
        const int tap = k0 / conv_c;
        c0 = k0 - tap * conv_c;
        dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int L = q * GEMM_THREADS + tid;          // linear 16-B chunk index in the LDS image
        int row = L >> 3, pc = L & 7;
        int kc = pc ^ (row & 7);                 // source chunk that belongs at this position
        const unsigned short *src;
        if constexpr (CONV) {
            const int m = row0 + row, yy = ((m >> 5) & 31) + dy, xx = (m & 31) + dx;
            src = ((unsigned)yy < 32u && (unsigned)xx < 32u)
                      ? g + ((size_t)(m & ~1023) + yy * 32 + xx) * conv_c + c0 + kc * 8
                      : reinterpret_cast<const unsigned short *>(&g_zero_chunk);
        } else {
            src = g + (size_t)(row0 + row) * ldk + k0 + kc * 8;
        }
        if constexpr (GLDS) {
            // LDS destination = wave-uniform base + lane*16 (hardware); base of this wave's 1 KB slab
            char *base = lds + (q * GEMM_THREADS + (tid & ~63)) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)base, 16, 0, 0);
        } else {
            regs[q] = *reinterpret_cast<const uint4 *>(src);
        }
    }
}
__device__ __forceinline__ void write_tile(char *lds, int tid, const uint4 *regs) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4 *>(lds + (q * GEMM_THREADS + tid) * 16) = regs[q];
}

template <int EPI, bool F16, bool GLDS, bool CONV = false>
__global__ void __launch_bounds__(GEMM_THREADS, 2) k_gemm(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware remap: blocks with equal (id % 8) share an XCD -> give them a contiguous id range
    int bid = blockIdx.x;
    {
        const int nxcd = 8, q = g.n_blocks / nxcd, r = g.n_blocks % nxcd, x = bid % nxcd;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / nxcd;
    }
    const int tile_m = bid / g.tiles_n, tile_n = bid - tile_m * g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = g.K / BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rw[4];
    // prologue
    stage_tile<GLDS, CONV>(g.A, g.K, m0, 0, smem, tid, ra, g.conv_c);
    stage_tile<GLDS>(g.W, g.K, n0, 0, smem + TILE_BYTES, tid, rw);
    if constexpr (!GLDS) { write_tile(smem, tid, ra); write_tile(smem + TILE_BYTES, tid, rw); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    // per-lane LDS byte offsets of the fragment chunks (swizzle folded in); row&7 == fr&7
    int offA[2], offW[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        int pc = (ks * 4 + fq) ^ (fr & 7);
        offA[ks] = (wm * 64 + fr) * 128 + pc * 16;
        offW[ks] = TILE_BYTES + (wn * 64 + fr) * 128 + pc * 16;
    }
    for (int kt = 0; kt < nk; ++kt) {
        char *cur = smem + (kt & 1) * STAGE_BYTES;
        char *nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
        // phase A: every fragment of this K tile -> registers (no LDS-DMA in flight here, so
        // hipcc does not put a vmcnt(0) in front of these reads)
        uint4 fa[2][4], fw[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[ks][i] = *reinterpret_cast<const uint4 *>(cur + offA[ks] + i * 16 * 128);
                fw[ks][i] = *reinterpret_cast<const uint4 *>(cur + offW[ks] + i * 16 * 128);
            }
        __builtin_amdgcn_sched_barrier(0);
        // phase B: start the next tile's LDS-DMA; it lands while the MFMAs below run
        if (kt + 1 < nk) {
            stage_tile<GLDS, CONV>(g.A, g.K, m0, (kt + 1) * BK, nxt, tid, ra, g.conv_c);
            stage_tile<GLDS>(g.W, g.K, n0, (kt + 1) * BK, nxt + TILE_BYTES, tid, rw);
        }
        __builtin_amdgcn_sched_barrier(0);
        // phase C: 32 MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    acc[nb][mb] = mfma16<F16>(fw[ks][nb], fa[ks][mb], acc[nb][mb]);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            if constexpr (!GLDS) { write_tile(nxt, tid, ra); write_tile(nxt + TILE_BYTES, tid, rw); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }

    // epilogue: lane holds, per (nb, mb): token m, channels n..n+3
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int m = m0 + wm * 64 + mb * 16 + fr;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int n = n0 + wn * 64 + nb * 16 + fq * 4;
            f32x4 v = acc[nb][mb];
            if (g.ln_stats) {
                float mean, rstd;
                ln_row_params(g.ln_stats, m, 1.0f / g.K, mean, rstd);
                const float4 cs = *reinterpret_cast<const float4 *>(g.ln_colsum + n);
                v[0] = (v[0] - mean * cs.x) * rstd; v[1] = (v[1] - mean * cs.y) * rstd;
                v[2] = (v[2] - mean * cs.z) * rstd; v[3] = (v[3] - mean * cs.w) * rstd;
            }
            if (g.bias) {
                const float4 b = *reinterpret_cast<const float4 *>(g.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if constexpr (EPI == CPX_EPI_GELU_BF16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            } else if constexpr (EPI == CPX_EPI_RELU_BF16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if constexpr (EPI == CPX_EPI_RESID_BF16) {
                const uint2 rr = *reinterpret_cast<const uint2 *>((const unsigned short *)g.aux + (size_t)m * g.ld_out + n);
                v[0] += from_half<F16>(rr.x & 0xFFFF); v[1] += from_half<F16>(rr.x >> 16);
                v[2] += from_half<F16>(rr.y & 0xFFFF); v[3] += from_half<F16>(rr.y >> 16);
            } else if constexpr (EPI == CPX_EPI_POS_BF16) {
                const float4 p = *reinterpret_cast<const float4 *>((const float *)g.aux + (size_t)(m & 1023) * g.N + n);
                v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
            }
            if constexpr (EPI == CPX_EPI_F32) {
                *reinterpret_cast<float4 *>((float *)g.out + (size_t)m * g.ld_out + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else if (EPI == CPX_EPI_QKV_BF16 && n >= 2048) {
                unsigned short *vT = (unsigned short *)g.aux;
                const size_t s_ = (size_t)(m >> 10), t_ = (size_t)(m & 1023);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = n - 2048 + r;
                    vT[((s_ * 16 + (c >> 6)) * 64 + (c & 63)) * 1024 + t_] = to_half<F16>(v[r]);
                }
            } else {
                uint2 o;
                o.x = pack2<F16>(v[0], v[1]);
                o.y = pack2<F16>(v[2], v[3]);
                *reinterpret_cast<uint2 *>((unsigned short *)g.out + (size_t)m * g.ld_out + n) = o;
            }
        }
    }
}


// ===========================================================================
// 256 x 256 x 64 tile, 512 threads (8 waves = 2 (M) x 4 (N)), 4 phases per K tile
// ===========================================================================
// Per-wave output = 2 x 2 quadrants of (64 tokens x 32 channels): quadrant (hm, hn) covers
// tokens 128*hm + 64*wm + [0,64) and channels 128*hn + 32*wn + [0,32), so in every phase ALL
// waves read the same two 16 KB half-tiles (X_hm, W_hn) and a half-tile can be re-staged
// as soon as its phase is over.  LDS = 2 K-tile buffers x {X0, X1, W0, W1} x 16 KB = 128 KB,
// filled by 16-byte LDS-DMA with the chunk swizzle on the source address.  Schedule per K
// tile t (reads -> quadrant), one half-tile staged per phase, each >= 2 phases after the last
// read of the region it overwrites and never drained to zero in the steady state:
//   p1: X0,W0 -> (0,0)   stage X1(t+1)        p3: X1 -> (1,1)   stage X0(t+2)
//   p2: W1    -> (0,1)   stage W0(t+1)        p4: W0 -> (1,0)   stage W1(t+2), vmcnt(4)
// The two wave rows (wm) run half a phase apart (one extra s_barrier for wm == 1 at the
// start, one for wm == 0 at the end): on every SIMD one wave issues MFMAs while its partner
// issues LDS reads and DMA.  Fragment reads are inline-asm ds_read_b128 so that hipcc
// neither drains the DMA in front of them nor re-orders them across the raw barriers.
// Epilogue: accumulators -> bf16 tile in LDS (528-byte rows) -> whole 512-byte rows stored
// with 16-byte stores (+ residual add on the store side, = the reference's double rounding).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define G2_THREADS 512
#define G2_HALF 16384
#define G2_BUF 65536
#define G2_EPI_LD 528
#define G2_LN_OFF (256 * G2_EPI_LD)         // 256 x (rstd, -mean * rstd) of the tile's token rows (folded LayerNorm)
#define G2_LDS_BYTES (G2_LN_OFF + 256 * 8)  // 135168 (>= 2 * G2_BUF) + 2048

template <int OFF>
__device__ __forceinline__ u32x4 lds_read128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma16v(const u32x4 &a, const u32x4 &b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int HM>
__device__ __forceinline__ void g2_read_x(u32x4 (&fx)[4][2], unsigned b0, unsigned b1) {
    fx[0][0] = lds_read128<HM * G2_HALF + 0 * 2048>(b0); fx[0][1] = lds_read128<HM * G2_HALF + 0 * 2048>(b1);
    fx[1][0] = lds_read128<HM * G2_HALF + 1 * 2048>(b0); fx[1][1] = lds_read128<HM * G2_HALF + 1 * 2048>(b1);
    fx[2][0] = lds_read128<HM * G2_HALF + 2 * 2048>(b0); fx[2][1] = lds_read128<HM * G2_HALF + 2 * 2048>(b1);
    fx[3][0] = lds_read128<HM * G2_HALF + 3 * 2048>(b0); fx[3][1] = lds_read128<HM * G2_HALF + 3 * 2048>(b1);
}
template <int HN>
__device__ __forceinline__ void g2_read_w(u32x4 (&fw)[2][2], unsigned b0, unsigned b1) {
    fw[0][0] = lds_read128<(2 + HN) * G2_HALF + 0 * 2048>(b0); fw[0][1] = lds_read128<(2 + HN) * G2_HALF + 0 * 2048>(b1);
    fw[1][0] = lds_read128<(2 + HN) * G2_HALF + 1 * 2048>(b0); fw[1][1] = lds_read128<(2 + HN) * G2_HALF + 1 * 2048>(b1);
}
// FIRST: the quadrant's first K step of a tile -- the accumulators START as the product (C operand = the inline constant 0)
// instead of being cleared by 128 v_mov_b32 per lane in front of every tile
template <bool F16, bool FIRST = false>
__device__ __forceinline__ void g2_mma(f32x4 (&acc)[4][2], const u32x4 (&fx)[4][2], const u32x4 (&fw)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                acc[mb][nb] = mfma16v<F16>(fw[nb][ks], fx[mb][ks], (FIRST && ks == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mb][nb]);
    __builtin_amdgcn_s_setprio(0);
}
// SPLIT (experiment, round 4): fragment reads ordered k-substep 0 first, so that the first eight MFMAs of a phase can start
// behind a COUNTED lgkmcnt while the k-substep-1 fragments are still landing
template <int HM>
__device__ __forceinline__ void g2_read_x_ks(u32x4 (&fx)[4][2], unsigned b0, unsigned b1) {
    fx[0][0] = lds_read128<HM * G2_HALF + 0 * 2048>(b0); fx[1][0] = lds_read128<HM * G2_HALF + 1 * 2048>(b0);
    fx[2][0] = lds_read128<HM * G2_HALF + 2 * 2048>(b0); fx[3][0] = lds_read128<HM * G2_HALF + 3 * 2048>(b0);
    fx[0][1] = lds_read128<HM * G2_HALF + 0 * 2048>(b1); fx[1][1] = lds_read128<HM * G2_HALF + 1 * 2048>(b1);
    fx[2][1] = lds_read128<HM * G2_HALF + 2 * 2048>(b1); fx[3][1] = lds_read128<HM * G2_HALF + 3 * 2048>(b1);
}
template <int HN>
__device__ __forceinline__ void g2_read_w_ks(u32x4 (&fw)[2][2], unsigned b0, unsigned b1) {
    fw[0][0] = lds_read128<(2 + HN) * G2_HALF + 0 * 2048>(b0); fw[1][0] = lds_read128<(2 + HN) * G2_HALF + 1 * 2048>(b0);
    fw[0][1] = lds_read128<(2 + HN) * G2_HALF + 0 * 2048>(b1); fw[1][1] = lds_read128<(2 + HN) * G2_HALF + 1 * 2048>(b1);
}
template <bool F16, bool FIRST, int KS>
__device__ __forceinline__ void g2_mma_ks(f32x4 (&acc)[4][2], const u32x4 (&fx)[4][2], const u32x4 (&fw)[2][2]) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = mfma16v<F16>(fw[nb][KS], fx[mb][KS], (FIRST && KS == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mb][nb]);
}
#define G2_LGKM(N) do { asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G2_BAR() __builtin_amdgcn_s_barrier()
#define G2_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// FLAGS (compile time, so that none of these tests sits inside the unrolled epilogue loops -- the runtime
// versions cost ~250 scalar branches per lane and tile): 1 = consume a folded LayerNorm (g.ln_stats),
// 2 = emit row statistics (g.stats_out, residual epilogue only), 4 = honour the g.dbg timing ablations
#define G2F_LN 1
#define G2F_STATS 2
#define G2F_DBG 4
#define G2F_Q4 8          // persistent kernel: quarter-tile epilogue (experiment, debug build)
#define G2F_SPLIT 16      // persistent kernel: counted LDS waits inside a phase (experiment, round 4)
#define G2F_BAL 64        // persistent kernel: balanced fragment-read schedule (W0 of the next K tile pre-read in phase 4, no W0 re-read, every operand item 6 phases ahead)
#define G2F_DIRECT 32     // persistent kernel: epilogue stores straight from the accumulator registers (v_permlane16_swap -> 16-byte rows), no LDS staging
template <int EPI, bool F16, int FLAGS>
__global__ void __launch_bounds__(G2_THREADS, 2) k_gemm256(GemmArgs g) {
    constexpr bool LN_IN = (FLAGS & G2F_LN) != 0 && EPI != CPX_EPI_RESID_BF16 && EPI != CPX_EPI_POS_BF16;
    constexpr bool STATS = (FLAGS & G2F_STATS) != 0 && (EPI == CPX_EPI_RESID_BF16 || EPI == CPX_EPI_POS_BF16);
    constexpr bool DBG = (FLAGS & G2F_DBG) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    int bid = blockIdx.x;
    {
        const int nxcd = 8, q = g.n_blocks / nxcd, r = g.n_blocks % nxcd, x = bid % nxcd;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / nxcd;
    }
    // L2 blocking: the 32 workgroups an XCD runs at a time cover an 8 (M) x 4 (N) super-tile, i.e.
    // 12 distinct operand panels instead of the 18 of a 2 x 16 strip (fabric reads -1/3, measured)
    int tile_m, tile_n;
    {
        const int tiles_m = g.n_blocks / g.tiles_n;
        if (g.l2_block && (tiles_m & 7) == 0 && (g.tiles_n & 3) == 0) {
            const int grp = bid >> 5, w_ = bid & 31, cgn = g.tiles_n >> 2, rgn = tiles_m >> 3;
            // mode 1: column groups innermost (an XCD sweeps N for a fixed band of 8 row tiles); mode 2: row groups
            // innermost (an XCD keeps ONE 4-tile W panel, 2 MB, and streams the activation rows past it)
            const int rg = g.l2_block == 2 ? grp % rgn : grp / cgn, cg = g.l2_block == 2 ? grp / rgn : grp - rg * cgn;
            tile_m = rg * 8 + (w_ >> 2);
            tile_n = cg * 4 + (w_ & 3);
        } else {
            tile_m = bid / g.tiles_n;
            tile_n = bid - tile_m * g.tiles_n;
        }
    }
    if (g.rev_m) tile_m = g.n_blocks / g.tiles_n - 1 - tile_m;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const int K = g.K, nk = K / 64;

    // staging: thread -> (row = tid>>3 (+64 for the second DMA), position pc = tid&7, source chunk kc)
    const int srow = tid >> 3, kc = (tid & 7) ^ (srow & 7);
    const unsigned short *pX = g.A + (size_t)(m0 + srow) * K + kc * 8;
    const unsigned short *pW = g.W + (size_t)(n0 + srow) * K + kc * 8;
    const size_t k64 = (size_t)64 * K, k128 = (size_t)128 * K;
    char *sdst = smem + wave * 1024;
    auto stage = [&](int which, int t) {        // which: 0 X0, 1 X1, 2 W0, 3 W1 (compile-time at call sites)
        const unsigned short *p = (which < 2 ? pX : pW) + (which & 1) * k128 + (size_t)t * 64;
        char *d = sdst + (t & 1) * G2_BUF + which * G2_HALF;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                         (__attribute__((address_space(3))) void *)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + k64),
                                         (__attribute__((address_space(3))) void *)(d + 8192), 16, 0, 0);
    };

    // fragment read bases (LDS byte addresses), one per k-substep and buffer
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    unsigned xb[2][2], wb[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned sw = (unsigned)(((ks * 4 + fq) ^ (fr & 7)) << 4);
        xb[0][ks] = lds0 + (wm * 64 + fr) * 128 + sw;
        wb[0][ks] = lds0 + (wn * 32 + fr) * 128 + sw;
        xb[1][ks] = xb[0][ks] + G2_BUF;
        wb[1][ks] = wb[0][ks] + G2_BUF;
    }

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) acc[a][b][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 fx[4][2], fw[2][2];

    // folded LayerNorm: the tile's 256 token rows get their (rstd, -mean * rstd) once, computed by one lane
    // each under the prologue's DMA latency and parked in the LDS tail for the epilogue.  The statistics are
    // requested BEFORE the DMA so that waiting for them (in-order vmcnt) does not drain the staging.
    const bool ln_in = LN_IN && tid < 256;
    float4 st_a = make_float4(0.f, 0.f, 0.f, 0.f), st_b = st_a;
    if (ln_in) {
        st_a = *reinterpret_cast<const float4 *>(g.ln_stats + (size_t)(m0 + tid) * 8);
        st_b = *reinterpret_cast<const float4 *>(g.ln_stats + (size_t)(m0 + tid) * 8 + 4);
    }
    // prologue: all of tile 0, then X0 / W1 of tile 1 (the two a steady-state tile -1 would have staged)
    stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0);
    if (nk > 1) { stage(0, 1); stage(3, 1); }
    if (ln_in) {
        const float inv_k = 1.0f / K;
        const float sum = (st_a.x + st_a.z) + (st_b.x + st_b.z), sq = (st_a.y + st_a.w) + (st_b.y + st_b.w);
        const float mean = sum * inv_k;
        const float rstd = rsqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + 1e-6f);
        *reinterpret_cast<float2 *>(smem + G2_LN_OFF + tid * 8) = make_float2(rstd, -mean * rstd);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G2_BAR();
    if (wm == 1) G2_BAR();                       // stagger the second wave row by one barrier

#define G2_TILE(T, B)                                                                       \
    {                                                                                       \
        const int t_ = (T);                                                                 \
        /* p1 */                                                                            \
        g2_read_w<0>(fw, wb[B][0], wb[B][1]);                                               \
        g2_read_x<0>(fx, xb[B][0], xb[B][1]);                                               \
        if (t_ + 1 < nk) stage(1, t_ + 1);                                                  \
        G2_BAR(); G2_LGKM0();                                                               \
        g2_mma<F16>(acc[0][0], fx, fw);                                                     \
        __builtin_amdgcn_sched_barrier(0); G2_BAR();                                        \
        /* p2 */                                                                            \
        g2_read_w<1>(fw, wb[B][0], wb[B][1]);                                               \
        if (t_ + 1 < nk) stage(2, t_ + 1);                                                  \
        G2_BAR(); G2_LGKM0();                                                               \
        g2_mma<F16>(acc[0][1], fx, fw);                                                     \
        __builtin_amdgcn_sched_barrier(0); G2_BAR();                                        \
        /* p3 */                                                                            \
        g2_read_x<1>(fx, xb[B][0], xb[B][1]);                                               \
        if (t_ + 2 < nk) stage(0, t_ + 2);                                                  \
        G2_BAR(); G2_LGKM0();                                                               \
        g2_mma<F16>(acc[1][1], fx, fw);                                                     \
        __builtin_amdgcn_sched_barrier(0); G2_BAR();                                        \
        /* p4 */                                                                            \
        g2_read_w<0>(fw, wb[B][0], wb[B][1]);                                               \
        if (t_ + 2 < nk) { stage(3, t_ + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } \
        else if (t_ + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              \
        G2_BAR(); G2_LGKM0();                                                               \
        g2_mma<F16>(acc[1][0], fx, fw);                                                     \
        __builtin_amdgcn_sched_barrier(0); G2_BAR();                                        \
    }

    for (int t = 0; t + 1 < nk; t += 2) {
        G2_TILE(t, 0)
        G2_TILE(t + 1, 1)
    }
    if (nk & 1) G2_TILE(nk - 1, 0)               // odd number of K tiles (the patch embedding: K = 192): the last one sits in buffer 0
#undef G2_TILE
    if (wm == 0) G2_BAR();                       // re-balance the barrier count of the two wave rows
    __builtin_amdgcn_sched_barrier(0);

    if (DBG && (g.dbg & 4)) {   // timing-only ablation: main loop + prologue, accumulators kept live
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int d = 0; d < 2; ++d) asm volatile("" ::"v"(acc[a][b][c][d]));
        return;
    }
    // ---- epilogue: bias/activation in f32 -> half tile in LDS -> whole rows to HBM
    const bool vt_tile = (EPI == CPX_EPI_QKV_BF16) && n0 >= 2048;     // V third of the qkv projection
    const int c16 = tid & 31;                   // 16-byte chunk within a 512-byte row (store phase)
    float ln_mean[2][4], ln_rstd[2][4];         // this lane's 8 token rows (hm, mb)
    if constexpr (LN_IN) {
#pragma unroll
        for (int hm = 0; hm < 2; ++hm)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float2 p = *reinterpret_cast<const float2 *>(smem + G2_LN_OFF + (hm * 128 + wm * 64 + mb * 16 + fr) * 8);
                ln_rstd[hm][mb] = p.x; ln_mean[hm][mb] = p.y;      // ln_mean holds -mean * rstd
            }
    }
    uint2 direct_prev = make_uint2(0u, 0u);
    uint4 rres[16];
    if constexpr (EPI == CPX_EPI_RESID_BF16) {  // residual rows: issue the loads now, consume after the LDS pass
#pragma unroll
        for (int it = 0; it < 16; ++it)
            rres[it] = *reinterpret_cast<const uint4 *>((const unsigned short *)g.aux +
                                                        (size_t)(m0 + it * 16 + (tid >> 5)) * g.ld_out + n0 + c16 * 8);
    }
    // this lane's 4 column groups (hn, nb): bias and LayerNorm column sums once, not once per accumulator
    // (the LDS stores below alias every pointer for the compiler, so it would reload them 32 times)
    float4 colb[2][2], colc[2][2];
#pragma unroll
    for (int hn = 0; hn < 2; ++hn)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int nl = hn * 128 + wn * 32 + nb * 16 + fq * 4;
            colb[hn][nb] = g.bias ? *reinterpret_cast<const float4 *>(g.bias + n0 + nl) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (LN_IN) colc[hn][nb] = *reinterpret_cast<const float4 *>(g.ln_colsum + n0 + nl);
        }
    // the V third of the qkv projection stages a transposed image; resolved per tile OUTSIDE the unrolled loops
    auto convert_and_stage = [&](auto vt_tag) {
        constexpr bool VT = decltype(vt_tag)::value;
#pragma unroll
    for (int hm = 0; hm < 2; ++hm)
#pragma unroll
        for (int hn = 0; hn < 2; ++hn)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int ml = hm * 128 + wm * 64 + mb * 16 + fr;
                    const int nl = hn * 128 + wn * 32 + nb * 16 + fq * 4;
                    f32x4 v = acc[hm][hn][mb][nb];
                    const float4 b = colb[hn][nb];
                    if constexpr (LN_IN) {
                        // rstd * (acc - mean * colsum) + bias = fma(acc, rstd, fma(-mean * rstd, colsum, bias))
                        const float4 cs = colc[hn][nb];
                        const float nm = ln_mean[hm][mb], rs = ln_rstd[hm][mb];      // ln_mean holds -mean * rstd
                        v[0] = fmaf(v[0], rs, fmaf(nm, cs.x, b.x)); v[1] = fmaf(v[1], rs, fmaf(nm, cs.y, b.y));
                        v[2] = fmaf(v[2], rs, fmaf(nm, cs.z, b.z)); v[3] = fmaf(v[3], rs, fmaf(nm, cs.w, b.w));
                    } else {
                        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    }
                    if constexpr (EPI == CPX_EPI_GELU_BF16) {
                        if (!DBG || !(g.dbg & 2)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                        }
                    } else if constexpr (EPI == CPX_EPI_RELU_BF16) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    } else if constexpr (EPI == CPX_EPI_POS_BF16) {      // + pos_embed[token within its sub-tile][channel], float32
                        const float4 pe = *reinterpret_cast<const float4 *>((const float *)g.aux + (size_t)((m0 + ml) & 1023) * g.N + n0 + nl);
                        v[0] += pe.x; v[1] += pe.y; v[2] += pe.z; v[3] += pe.w;
                    }
                    if constexpr (VT) {          // transposed image [channel][token] for the V^T layout
                        const unsigned p01 = pack2<F16>(v[0], v[1]), p23 = pack2<F16>(v[2], v[3]);
                        *reinterpret_cast<unsigned short *>(smem + (nl + 0) * G2_EPI_LD + ml * 2) = (unsigned short)p01;
                        *reinterpret_cast<unsigned short *>(smem + (nl + 1) * G2_EPI_LD + ml * 2) = (unsigned short)(p01 >> 16);
                        *reinterpret_cast<unsigned short *>(smem + (nl + 2) * G2_EPI_LD + ml * 2) = (unsigned short)p23;
                        *reinterpret_cast<unsigned short *>(smem + (nl + 3) * G2_EPI_LD + ml * 2) = (unsigned short)(p23 >> 16);
                    } else {
                        uint2 o;
                        o.x = pack2<F16>(v[0], v[1]);
                        o.y = pack2<F16>(v[2], v[3]);
                        if (DBG && EPI != CPX_EPI_RESID_BF16 && (g.dbg & 8)) {
                            // EXPERIMENT (timing + correctness switch): no LDS staging.  Lanes fq and fq^1 (16 lanes
                            // apart) trade one 8-byte piece so that each holds 8 consecutive channels -- the even
                            // one of the nb = 0 block, the odd one of the nb = 1 block -- and stores 16 bytes itself
                            // (64 contiguous bytes per token row and wave instead of whole 512-byte rows).
                            if (nb == 0) { direct_prev = o; }
                            else {
                                const uint2 send = (fq & 1) ? direct_prev : o;
                                uint2 recv;
                                recv.x = __shfl_xor((int)send.x, 16); recv.y = __shfl_xor((int)send.y, 16);
                                const uint4 w = (fq & 1) ? make_uint4(recv.x, recv.y, o.x, o.y)
                                                         : make_uint4(direct_prev.x, direct_prev.y, recv.x, recv.y);
                                const int ch = hn * 128 + wn * 32 + ((fq & 1) ? 16 : 0) + (fq >> 1) * 8;
                                *reinterpret_cast<uint4 *>((unsigned short *)g.out + (size_t)(m0 + ml) * g.ld_out + n0 + ch) = w;
                            }
                        } else
                        *reinterpret_cast<uint2 *>(smem + ml * G2_EPI_LD + nl * 2) = o;
                    }
                }
    };
    if (EPI == CPX_EPI_QKV_BF16 && vt_tile) convert_and_stage(std::true_type{});
    else convert_and_stage(std::false_type{});
    if (DBG && EPI != CPX_EPI_RESID_BF16 && (g.dbg & 8) && !vt_tile) return;
    __syncthreads();
    if (vt_tile) {
        // LDS row = channel c (head = c/64, d = c%64), 256 tokens contiguous -> vT[s][head][d][t0..t0+255]
        unsigned short *vT = (unsigned short *)g.aux;
        const size_t s_ = (size_t)(m0 >> 10), t0 = (size_t)(m0 & 1023);
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = it * 16 + (tid >> 5);
            const int c = n0 - 2048 + row;
            uint4 v = *reinterpret_cast<const uint4 *>(smem + row * G2_EPI_LD + c16 * 16);
            *reinterpret_cast<uint4 *>(vT + ((s_ * 16 + (c >> 6)) * 64 + (c & 63)) * 1024 + t0 + c16 * 8) = v;
        }
    } else {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int ml = it * 16 + (tid >> 5);
            uint4 v = *reinterpret_cast<const uint4 *>(smem + ml * G2_EPI_LD + c16 * 16);
            const size_t go = (size_t)(m0 + ml) * g.ld_out + n0 + c16 * 8;
            if constexpr (EPI == CPX_EPI_RESID_BF16) {
                const uint4 rr = rres[it];
                unsigned a[4] = {v.x, v.y, v.z, v.w}, b[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float lo = from_half<F16>(a[i] & 0xFFFF) + from_half<F16>(b[i] & 0xFFFF);
                    float hi = from_half<F16>(a[i] >> 16) + from_half<F16>(b[i] >> 16);
                    a[i] = pack2<F16>(lo, hi);
                }
                v = make_uint4(a[0], a[1], a[2], a[3]);
            }
            if (!DBG || !(g.dbg & 1) || v.x == 0x12345678u) *reinterpret_cast<uint4 *>((unsigned short *)g.out + go) = v;
            if constexpr (STATS) {
                // partial LayerNorm statistics of the (rounded) output row over this block's 256 columns:
                // 32 lanes share a row; slot = column tile, written whole -> deterministic, no atomics
                unsigned a[4] = {v.x, v.y, v.z, v.w};
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float lo = from_half<F16>(a[i] & 0xFFFF), hi = from_half<F16>(a[i] >> 16);
                    sm += lo + hi; sq += lo * lo + hi * hi;
                }
                sm = half_wave_sum(sm); sq = half_wave_sum(sq);
                if (c16 == 16)
                    *reinterpret_cast<float2 *>(g.stats_out + ((size_t)(m0 + ml) * LN_SLOTS + tile_n) * 2) = make_float2(sm, sq);
            }
        }
    }
}


// ===========================================================================
// persistent form of the 256^2 kernel: one workgroup per CU walks its tiles
// ===========================================================================
// Same main loop, LDS map and phase schedule as k_gemm256.  What changes is everything around it:
//   * grid = min(#tiles, #CUs); workgroup b takes the virtual tile ids b, b + grid, b + 2 grid ... (id % 8 is
//     unchanged, so an XCD still sweeps its own contiguous tile range super-tile by super-tile);
//   * the epilogue works in two 128-row halves through a 66 KB staging area that lies in the SECOND K-tile
//     buffer, so the first K-tile buffer is free while it runs: the four half-tile DMAs of the NEXT tile's
//     K-tile 0 are issued at the start of the epilogue and land under its convert / GELU / store work
//     instead of being waited for in front of the next main loop (a 256^2 tile's prologue costs 2-4 us of
//     its ~37 us; fc1 runs 8 tiles per CU);
//   * the epilogue has no global loads any more: bias and the folded-LayerNorm column sums of the tile's 256
//     columns are parked in the LDS tail next to the LayerNorm row parameters under the prologue's latency
//     (hipcc drains every LDS-DMA in flight with vmcnt(0) at the first use of an ordinary load's result);
//     its barriers are raw s_barriers for the same reason (__syncthreads adds vmcnt(0) while a DMA is in flight).
// The residual epilogue (its residual rows ARE global loads) and the V^T tiles of the qkv projection (their
// transposed image needs the whole LDS) keep the un-overlapped order: prefetch after the epilogue.
#define G2P_STG_OFF G2_BUF                          // staging area = second K-tile buffer + 2 KB
#define G2P_TAIL G2_LN_OFF                          // 2 KB LayerNorm row params, 1 KB bias, 1 KB column sums
#define G2P_LDS_BYTES (G2_LN_OFF + 4096)
template <int EPI, bool F16, int FLAGS>
__global__ void __launch_bounds__(G2_THREADS, 2) k_gemm256p(GemmArgs g) {
    constexpr bool LN_IN = (FLAGS & G2F_LN) != 0 && EPI != CPX_EPI_RESID_BF16;
    constexpr bool STATS = (FLAGS & G2F_STATS) != 0 && EPI == CPX_EPI_RESID_BF16;
    constexpr bool DBG = (FLAGS & G2F_DBG) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int K = g.K, nk = K / 64, nblk = g.n_blocks, tiles_m = g.n_blocks / g.tiles_n;
    auto coords = [&](int v, int &m0_, int &n0_, int &tn_) {
        const int nxcd = 8, q = nblk / nxcd, r = nblk % nxcd, x = v % nxcd;
        int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + v / nxcd;
        int tile_m, tile_n;
        if (EPI == CPX_EPI_QKV_BF16 && g.l2_block == 3) {
            // balanced list for the qkv projection (launcher: 256 workgroups, tiles_m % 64 == 0, 12 column tiles): the
            // V^T tiles cost more (their transposed image needs the whole LDS, so nothing is prefetched under their
            // epilogue), hence every workgroup gets the same mix -- four q|k tiles, then two V^T tiles, per round -- and the
            // 32 workgroups of an XCD still cover one 8 x 4 super-tile at a time
            // (rounds of 6: four q|k super-tiles and the two V^T super-tiles of the same 16 row tiles; when the XCD's row
            // count is an odd multiple of 8, a last half round of 3: two q|k super-tiles and one V^T of the last 8 rows)
            const int x = v & 7, slot = (v >> 3) & 31, i = v >> 8, rx = tiles_m >> 3, full = rx >> 4;
            int rg, cg;
            if (i < 6 * full) {
                const int rnd = i / 6, j = i - rnd * 6;
                rg = 2 * rnd + (j < 4 ? j >> 1 : j - 4); cg = j < 4 ? j & 1 : 2;
            } else {
                const int j = i - 6 * full;
                rg = 2 * full; cg = j < 2 ? j : 2;
            }
            tile_m = x * rx + rg * 8 + (slot >> 2);
            tile_n = cg * 4 + (slot & 3);
        } else if (g.l2_block && (tiles_m & 7) == 0 && (g.tiles_n & 3) == 0) {
            const int grp = bid >> 5, w_ = bid & 31, cgn = g.tiles_n >> 2, rgn = tiles_m >> 3;
            // mode 1: column groups innermost (an XCD sweeps N for a fixed band of 8 row tiles); mode 2: row groups
            // innermost (an XCD keeps ONE 4-tile W panel, 2 MB, and streams the activation rows past it)
            const int rg = g.l2_block == 2 ? grp % rgn : grp / cgn, cg = g.l2_block == 2 ? grp / rgn : grp - rg * cgn;
            tile_m = rg * 8 + (w_ >> 2);
            tile_n = cg * 4 + (w_ & 3);
        } else {
            tile_m = bid / g.tiles_n;
            tile_n = bid - tile_m * g.tiles_n;
        }
        if (g.rev_m) tile_m = tiles_m - 1 - tile_m;
        m0_ = tile_m * 256; n0_ = tile_n * 256; tn_ = tile_n;
    };
    char *sdst = smem + wave * 1024;
    // LDS-DMA requests as BUFFER loads (round 4): the per-lane part of the source address is ONE 32-bit offset register per operand for
    // the whole launch (row tid >> 3 of a 64-row piece, swizzled chunk), everything that changes -- tile origin, half-tile, K tile,
    // second piece -- is a scalar offset, so a request costs no vector instruction (the flat-pointer form recomputed a 64-bit lane
    // address with 2-3 VALU instructions per request, 16 requests per wave and K-tile pair, all in the read phase the loop is bound by).
    // (A and W are < 2^31 bytes: 32 768 x 4 096 bf16 = 268 MB.)
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void *)g.A, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void *)g.W, 0, 0x7FFFFFFF, 0x00020000);
    const unsigned k64b = (unsigned)K * 128u, k128b = (unsigned)K * 256u;         // 64 / 128 rows, in bytes
    auto stage_at = [&](unsigned voff, unsigned sX, unsigned sW, int which, int t) {
        const unsigned so = (which < 2 ? sX : sW) + (which & 1) * k128b + (unsigned)t * 128u;
        char *d = sdst + (t & 1) * G2_BUF + which * G2_HALF;
        if (which < 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (__attribute__((address_space(3))) void *)d, 16, voff, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (__attribute__((address_space(3))) void *)(d + 8192), 16, voff, so + k64b, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void *)d, 16, voff, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void *)(d + 8192), 16, voff, so + k64b, 0, 0);
        }
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    int v = blockIdx.x, m0, n0, tile_n;
    coords(v, m0, n0, tile_n);
    bool prefetched = false;                     // K-tile 0 of the current tile already in flight / landed in buffer 0
    for (;;) {
        const int vn = v + (int)gridDim.x;
        const bool has_next = vn < nblk;
        int m0n = 0, n0n = 0, tile_nn = 0;
        if (has_next) coords(vn, m0n, n0n, tile_nn);
        // every lane-dependent constant is re-derived per tile from an opaque copy of the thread id: hoisted out of
        // the tile loop they would stay live through the epilogue, be spilled, and their scratch reloads (VMEM) would
        // make hipcc drain the prefetch DMA with vmcnt(0)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, fr = lane & 15, fq = lane >> 4, c16 = tid & 31;
        const int srow = tid >> 3, kc = (tid & 7) ^ (srow & 7);
        unsigned xb[2][2], wb[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned sw = (unsigned)(((ks * 4 + fq) ^ (fr & 7)) << 4);
            xb[0][ks] = lds0 + (wm * 64 + fr) * 128 + sw;
            wb[0][ks] = lds0 + (wn * 32 + fr) * 128 + sw;
            xb[1][ks] = xb[0][ks] + G2_BUF;
            wb[1][ks] = wb[0][ks] + G2_BUF;
        }
        const unsigned voff = ((unsigned)srow * (unsigned)K + (unsigned)kc * 8u) * 2u;     // this lane inside a 64-row piece (bytes)
        const unsigned sX = (DBG && (g.dbg & 512)) ? 0u : (unsigned)m0 * (unsigned)K * 2u, sW = (unsigned)n0 * (unsigned)K * 2u;   // tile origins (scalar); dbg & 512: every tile reads the FIRST 256 activation rows (cache-resident operand, timing only)
        auto stage = [&](int which, int t) { if (!(DBG && (g.dbg & 64) && t > 1)) stage_at(voff, sX, sW, which, t); };
        // per-tile vectors, requested BEFORE this tile's DMAs (in-order vmcnt): LayerNorm row statistics by
        // threads 0..255, bias and column sums of the tile's 256 columns by threads 256..511
        float4 st_a = make_float4(0.f, 0.f, 0.f, 0.f), st_b = st_a;
        float bias_v = 0.f, csum_v = 0.f;
        if (tid < 256) {
            if (LN_IN) {
                st_a = *reinterpret_cast<const float4 *>(g.ln_stats + (size_t)(m0 + tid) * 8);
                st_b = *reinterpret_cast<const float4 *>(g.ln_stats + (size_t)(m0 + tid) * 8 + 4);
            }
        } else {
            if (g.bias) bias_v = g.bias[n0 + tid - 256];
            if (LN_IN) csum_v = g.ln_colsum[n0 + tid - 256];
        }
        auto park_tile_vectors = [&]() {              // first use of the loads above (hipcc waits vmcnt(0) here while a DMA is in flight)
            if (tid < 256) {
                if (LN_IN) {
                    const float inv_k = 1.0f / K;
                    const float sum = (st_a.x + st_a.z) + (st_b.x + st_b.z), sq = (st_a.y + st_a.w) + (st_b.y + st_b.w);
                    const float mean = sum * inv_k;
                    const float rstd = rsqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + 1e-6f);
                    *reinterpret_cast<float2 *>(smem + G2P_TAIL + tid * 8) = make_float2(rstd, -mean * rstd);
                }
            } else {
                *reinterpret_cast<float *>(smem + G2P_TAIL + 2048 + (tid - 256) * 4) = bias_v;
                *reinterpret_cast<float *>(smem + G2P_TAIL + 3072 + (tid - 256) * 4) = csum_v;
            }
        };
        constexpr bool BAL = (FLAGS & G2F_BAL) != 0;          // (launcher: nk >= 4)
        if (prefetched) {
            // K-tile 0 was requested an epilogue ago: consuming the small loads first costs only their own latency,
            // and K-tile 1's DMAs are issued AFTER that wait so that they are not drained by it
            park_tile_vectors();
            if (nk > 1) { stage(0, 1); stage(3, 1); if constexpr (BAL) stage(2, 1); }
        } else {
            stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0);
            if (nk > 1) { stage(0, 1); stage(3, 1); if constexpr (BAL) stage(2, 1); }
            park_tile_vectors();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (BAL) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G2_BAR();
        f32x4 acc[2][2][4][2];                       // not cleared: K tile 0 is peeled and starts every accumulator with C = 0
        u32x4 fx[4][2], fw[2][2], fw2[2][2];
        if constexpr (BAL) g2_read_w<0>(fw, wb[0][0], wb[0][1]);    // W0 of K tile 0, in front of the stagger barrier ("phase 4 of tile -1")
        if (wm == 1) G2_BAR();                       // stagger the second wave row by one barrier

// timing-only ablations of the main loop (DBG instantiation, results are garbage): g.dbg & 64 = no LDS-DMA requests inside the loop,
// & 128 = no fragment reads (the MFMAs run on whatever the registers hold), & 256 = no barriers, & 512 = all tiles read activation rows 0..255
#define G2_BARX() do { if (!(DBG && (g.dbg & 256))) G2_BAR(); } while (0)
// KIND -1: run-time end conditions (production); 2: steady state (t + 2 < nk), 3 / 4: the last two K tiles of nk >= 4 (no scalar branch
// inside those bodies; available for experiments)
#define G2_TILE_PLAIN(T, B, FIRST, KIND)                                                                 \
    {                                                                                       \
        const int t_ = (T);                                                                 \
        const bool c1_ = (KIND) == -1 ? t_ + 1 < nk : (KIND) != 4, c2_ = (KIND) == -1 ? t_ + 2 < nk : (KIND) == 2; \
        if (!(DBG && (g.dbg & 128))) g2_read_w<0>(fw, wb[B][0], wb[B][1]);                                               \
        if (!(DBG && (g.dbg & 128))) g2_read_x<0>(fx, xb[B][0], xb[B][1]);                                               \
        if (c1_) stage(1, t_ + 1);                                                  \
        G2_BARX(); G2_LGKM0();                                                               \
        g2_mma<F16, FIRST>(acc[0][0], fx, fw);                                                     \
        __builtin_amdgcn_sched_barrier(0); G2_BARX();                                        \
        if (!(DBG && (g.dbg & 128))) g2_read_w<1>(fw, wb[B][0], wb[B][1]);                                               \
        if (c1_) stage(2, t_ + 1);                                                  \
        G2_BARX(); G2_LGKM0();                                                               \
        g2_mma<F16, FIRST>(acc[0][1], fx, fw);                                                     \
        __builtin_amdgcn_sched_barrier(0); G2_BARX();                                        \
        if (!(DBG && (g.dbg & 128))) g2_read_x<1>(fx, xb[B][0], xb[B][1]);                                               \
        if (c2_) stage(0, t_ + 2);                                                  \
        G2_BARX(); G2_LGKM0();                                                               \
        g2_mma<F16, FIRST>(acc[1][1], fx, fw);                                                     \
        __builtin_amdgcn_sched_barrier(0); G2_BARX();                                        \
        if (!(DBG && (g.dbg & 128))) g2_read_w<0>(fw, wb[B][0], wb[B][1]);                                               \
        if (c2_) { stage(3, t_ + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } \
        else if (c1_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              \
        G2_BARX(); G2_LGKM0();                                                               \
        g2_mma<F16, FIRST>(acc[1][0], fx, fw);                                                     \
        __builtin_amdgcn_sched_barrier(0); G2_BARX();                                        \
    }
#define G2_TILE_SPLIT(T, B, FIRST)                                                                 \
    {                                                                                       \
        const int t_ = (T);                                                                 \
        /* p1: W0 + X0 -> (0,0): ks 0 operands = the first 6 of the 12 reads */             \
        fw[0][0] = lds_read128<2 * G2_HALF>(wb[B][0]); fw[1][0] = lds_read128<2 * G2_HALF + 2048>(wb[B][0]); \
        fx[0][0] = lds_read128<0>(xb[B][0]); fx[1][0] = lds_read128<2048>(xb[B][0]);        \
        fx[2][0] = lds_read128<4096>(xb[B][0]); fx[3][0] = lds_read128<6144>(xb[B][0]);     \
        fw[0][1] = lds_read128<2 * G2_HALF>(wb[B][1]); fw[1][1] = lds_read128<2 * G2_HALF + 2048>(wb[B][1]); \
        fx[0][1] = lds_read128<0>(xb[B][1]); fx[1][1] = lds_read128<2048>(xb[B][1]);        \
        fx[2][1] = lds_read128<4096>(xb[B][1]); fx[3][1] = lds_read128<6144>(xb[B][1]);     \
        if (t_ + 1 < nk) stage(1, t_ + 1);                                                  \
        G2_BAR(); G2_LGKM(6);                                                               \
        __builtin_amdgcn_s_setprio(1);                                                      \
        g2_mma_ks<F16, FIRST, 0>(acc[0][0], fx, fw);                                        \
        G2_LGKM(0);                                                                         \
        g2_mma_ks<F16, FIRST, 1>(acc[0][0], fx, fw);                                        \
        __builtin_amdgcn_s_setprio(0);                                                      \
        __builtin_amdgcn_sched_barrier(0); G2_BAR();                                        \
        /* p2: W1 -> (0,1) */                                                               \
        g2_read_w_ks<1>(fw, wb[B][0], wb[B][1]);                                            \
        if (t_ + 1 < nk) stage(2, t_ + 1);                                                  \
        G2_BAR(); G2_LGKM(2);                                                               \
        __builtin_amdgcn_s_setprio(1);                                                      \
        g2_mma_ks<F16, FIRST, 0>(acc[0][1], fx, fw);                                        \
        G2_LGKM(0);                                                                         \
        g2_mma_ks<F16, FIRST, 1>(acc[0][1], fx, fw);                                        \
        __builtin_amdgcn_s_setprio(0);                                                      \
        __builtin_amdgcn_sched_barrier(0); G2_BAR();                                        \
        /* p3: X1 -> (1,1) */                                                               \
        g2_read_x_ks<1>(fx, xb[B][0], xb[B][1]);                                            \
        if (t_ + 2 < nk) stage(0, t_ + 2);                                                  \
        G2_BAR(); G2_LGKM(4);                                                               \
        __builtin_amdgcn_s_setprio(1);                                                      \
        g2_mma_ks<F16, FIRST, 0>(acc[1][1], fx, fw);                                        \
        G2_LGKM(0);                                                                         \
        g2_mma_ks<F16, FIRST, 1>(acc[1][1], fx, fw);                                        \
        __builtin_amdgcn_s_setprio(0);                                                      \
        __builtin_amdgcn_sched_barrier(0); G2_BAR();                                        \
        /* p4: W0 -> (1,0) */                                                               \
        g2_read_w_ks<0>(fw, wb[B][0], wb[B][1]);                                            \
        if (t_ + 2 < nk) { stage(3, t_ + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } \
        else if (t_ + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              \
        G2_BAR(); G2_LGKM(2);                                                               \
        __builtin_amdgcn_s_setprio(1);                                                      \
        g2_mma_ks<F16, FIRST, 0>(acc[1][0], fx, fw);                                        \
        G2_LGKM(0);                                                                         \
        g2_mma_ks<F16, FIRST, 1>(acc[1][0], fx, fw);                                        \
        __builtin_amdgcn_s_setprio(0);                                                      \
        __builtin_amdgcn_sched_barrier(0); G2_BAR();                                        \
    }
// ---- balanced schedule (G2F_BAL).  The plain schedule reads 12 fragments in phase 1 (W0 + X0: 48 KB per wave row against a 256-cycle
// MFMA slot of the partner row), re-reads W0 in phase 4 (28 reads per K tile) and requests W-lo only 3 phases before its first read.
// Here the two W fragment sets swap roles every K tile: W0 of tile t sits in FA, phase 2 reads W1 into FB, phase 4 (which multiplies with
// FA) pre-reads W0 of tile t + 1 into FB -- 8 / 4 / 8 / 4 reads, 24 per K tile, no re-read.  Requests: phase 1 X-hi(t+1), phase 2
// W-lo(t+2), phase 3 X-lo(t+2), phase 4 W-hi(t+2): every item is refilled two phases after its region's last read (the partner row reads
// it one barrier later) and 6 phases before its own; after its request every phase waits for the item of the NEXT phase with a counted
// vmcnt (10 = five younger items in the steady state) in front of its mid barrier, which both rows pass before either reads the item.
// W0 of K tile 0 is read right behind the tile-top barrier (in front of the stagger barrier), so K tile 0 runs the steady-state phases too
// (its phase-3 wait counts the three items of K tile 1 the tile top requested: vmcnt(6)).
#define G2_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define G2_TILE_BAL(T, B, FIRST, KIND, FA, FB)                                                       \
    {   /* KIND 0: K tile 0 (its W0 was read in front of the loop); 2: steady state (t + 2 < nk); 3 / 4: the last two K tiles -- no run-time condition in here */ \
        const int t_ = (T);                                                                         \
        /* p1: X0 -> (0,0) */                                                                       \
        g2_read_x<0>(fx, xb[B][0], xb[B][1]);                                                       \
        if ((KIND) != 4) stage(1, t_ + 1);                                                          \
        if ((KIND) == 4) G2_VM(2); else G2_VM(10);                                                  \
        G2_BAR(); G2_LGKM0();                                                                       \
        g2_mma<F16, FIRST>(acc[0][0], fx, FA);                                                      \
        __builtin_amdgcn_sched_barrier(0); G2_BAR();                                                \
        /* p2: W1 -> (0,1) */                                                                       \
        g2_read_w<1>(FB, wb[B][0], wb[B][1]);                                                       \
        if ((KIND) <= 2) stage(2, t_ + 2);                                                          \
        if ((KIND) == 4) G2_VM(0); else if ((KIND) == 3) G2_VM(8); else G2_VM(10);                  \
        G2_BAR(); G2_LGKM0();                                                                       \
        g2_mma<F16, FIRST>(acc[0][1], fx, FB);                                                      \
        __builtin_amdgcn_sched_barrier(0); G2_BAR();                                                \
        /* p3: X1 -> (1,1) */                                                                       \
        g2_read_x<1>(fx, xb[B][0], xb[B][1]);                                                       \
        if ((KIND) <= 2) stage(0, t_ + 2);                                                          \
        if ((KIND) == 0 || (KIND) == 3) G2_VM(6); else if ((KIND) == 2) G2_VM(10);                  \
        G2_BAR(); G2_LGKM0();                                                                       \
        g2_mma<F16, FIRST>(acc[1][1], fx, FB);                                                      \
        __builtin_amdgcn_sched_barrier(0); G2_BAR();                                                \
        /* p4: pre-read W0 of K tile t + 1 (other buffer) into the set phase 3 just finished with; (1,0) multiplies with W0 = FA */ \
        if ((KIND) != 4) g2_read_w<0>(FB, wb[(B) ^ 1][0], wb[(B) ^ 1][1]);                          \
        if ((KIND) <= 2) stage(3, t_ + 2);                                                          \
        if ((KIND) == 3) G2_VM(4); else if ((KIND) != 4) G2_VM(10);                                 \
        G2_BAR(); G2_LGKM0();                                                                       \
        g2_mma<F16, FIRST>(acc[1][0], fx, FA);                                                      \
        __builtin_amdgcn_sched_barrier(0); G2_BAR();                                                \
    }
        if constexpr (BAL) {
            G2_TILE_BAL(0, 0, true, 0, fw, fw2)
            G2_TILE_BAL(1, 1, false, 2, fw2, fw)
            for (int t = 2; t + 2 < nk; t += 2) {
                G2_TILE_BAL(t, 0, false, 2, fw, fw2)
                G2_TILE_BAL(t + 1, 1, false, 2, fw2, fw)
            }
            G2_TILE_BAL(nk - 2, 0, false, 3, fw, fw2)
            G2_TILE_BAL(nk - 1, 1, false, 4, fw2, fw)
        } else {
#define G2_TILE(T, B, FIRST, KIND) { if constexpr ((FLAGS & G2F_SPLIT) != 0) G2_TILE_SPLIT(T, B, FIRST) else G2_TILE_PLAIN(T, B, FIRST, KIND) }
        // (peeling the last two K tiles here as well -- KIND 2 / 3 / 4 -- measured no gain for this schedule: 1.000 of the run-time form)
        G2_TILE(0, 0, true, -1)
        G2_TILE(1, 1, false, -1)
        for (int t = 2; t < nk; t += 2) {
            G2_TILE(t, 0, false, -1)
            G2_TILE(t + 1, 1, false, -1)
        }
#undef G2_TILE
        }
#undef G2_TILE_BAL
#undef G2_VM
#undef G2_BARX
#undef G2_TILE_PLAIN
#undef G2_TILE_SPLIT
        if (wm == 0) G2_BAR();                       // re-balance the barrier count of the two wave rows
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue
        const bool vt_tile = (EPI == CPX_EPI_QKV_BF16) && n0 >= 2048;
        const bool main_only = DBG && (g.dbg & 4);
        bool next_issued = false;
        const unsigned sXn = (DBG && (g.dbg & 512)) ? 0u : (unsigned)m0n * (unsigned)K * 2u, sWn = (unsigned)n0n * (unsigned)K * 2u;
        auto prefetch_next = [&]() {                 // K-tile 0 of the next tile -> buffer 0 (free since K-tile nk - 2)
            stage_at(voff, sXn, sWn, 0, 0); stage_at(voff, sXn, sWn, 2, 0); stage_at(voff, sXn, sWn, 3, 0); stage_at(voff, sXn, sWn, 1, 0);
            next_issued = true;
        };
        if (main_only) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int d = 0; d < 2; ++d) asm volatile("" ::"v"(acc[a][b][c][d]));
        } else if (vt_tile) {
            // V third of the qkv projection: transposed image [channel][token] over the whole LDS, as in k_gemm256
#pragma unroll
            for (int hm = 0; hm < 2; ++hm)
#pragma unroll
                for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            const int ml = hm * 128 + wm * 64 + mb * 16 + fr;
                            const int nl = hn * 128 + wn * 32 + nb * 16 + fq * 4;
                            f32x4 vv = acc[hm][hn][mb][nb];
                            const float4 b = *reinterpret_cast<const float4 *>(smem + G2P_TAIL + 2048 + nl * 4);
                            if constexpr (LN_IN) {
                                const float4 cs = *reinterpret_cast<const float4 *>(smem + G2P_TAIL + 3072 + nl * 4);
                                const float2 pr = *reinterpret_cast<const float2 *>(smem + G2P_TAIL + ml * 8);
                                vv[0] = fmaf(vv[0], pr.x, fmaf(pr.y, cs.x, b.x)); vv[1] = fmaf(vv[1], pr.x, fmaf(pr.y, cs.y, b.y));
                                vv[2] = fmaf(vv[2], pr.x, fmaf(pr.y, cs.z, b.z)); vv[3] = fmaf(vv[3], pr.x, fmaf(pr.y, cs.w, b.w));
                            } else {
                                vv[0] += b.x; vv[1] += b.y; vv[2] += b.z; vv[3] += b.w;
                            }
                            // (tail reads above complete before the staging writes below reach the tail: rows < 256)
                            const unsigned p01 = pack2<F16>(vv[0], vv[1]), p23 = pack2<F16>(vv[2], vv[3]);
                            *reinterpret_cast<unsigned short *>(smem + (nl + 0) * G2_EPI_LD + ml * 2) = (unsigned short)p01;
                            *reinterpret_cast<unsigned short *>(smem + (nl + 1) * G2_EPI_LD + ml * 2) = (unsigned short)(p01 >> 16);
                            *reinterpret_cast<unsigned short *>(smem + (nl + 2) * G2_EPI_LD + ml * 2) = (unsigned short)p23;
                            *reinterpret_cast<unsigned short *>(smem + (nl + 3) * G2_EPI_LD + ml * 2) = (unsigned short)(p23 >> 16);
                        }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            G2_BAR();
            unsigned short *vT = (unsigned short *)g.aux;
            const size_t s_ = (size_t)(m0 >> 10), t0 = (size_t)(m0 & 1023);
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int row = it * 16 + (tid >> 5);
                const int c = n0 - 2048 + row;
                uint4 vv = *reinterpret_cast<const uint4 *>(smem + row * G2_EPI_LD + c16 * 16);
                st16<EPI == CPX_EPI_QKV_BF16, EPI == CPX_EPI_QKV_BF16>(vT + ((s_ * 16 + (c >> 6)) * 64 + (c & 63)) * 1024 + t0 + c16 * 8, vv, (g.nt_out & 4) != 0, (g.nt_out & 16) != 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            G2_BAR();
        } else if constexpr (EPI != CPX_EPI_RESID_BF16 && (FLAGS & G2F_Q4) != 0) {
            // ---- quarter-tile epilogue: the tile leaves in four 128-row x 128-column quarters through TWO 34 KB staging
            // buffers; in stage q a wave first requests its 16-byte chunks of quarter q - 1 from LDS, then converts quarter q
            // (the VALU stream: LayerNorm fold, bias, GELU, pack) into the other buffer, then stores quarter q - 1 -- the LDS
            // round trip and the store issue of one quarter run beside the vector work of the next instead of behind it.
            // Five barriers per tile instead of four; outputs bit for bit those of the two-half epilogue.
            if (has_next) prefetch_next();
            char *stg = smem + G2P_STG_OFF;
            const int srow = tid >> 4, sc16 = tid & 15;          // store side: 32 rows x 16 chunks of 16 bytes per pass
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                uint4 h0 = make_uint4(0u, 0u, 0u, 0u), h1 = h0, h2 = h0, h3 = h0;      // (named registers: an array here goes to scratch)
                const int hm = (q >> 1) & 1, hn = q & 1;
                const int phm = ((q - 1) >> 1) & 1, phn = (q - 1) & 1;
                char *buf = stg + (q & 1) * 34816, *pbuf = stg + ((q - 1) & 1) * 34816;
                if (q > 0) {
                    const char *src = pbuf + srow * 272 + sc16 * 16;
                    h0 = *reinterpret_cast<const uint4 *>(src);
                    h1 = *reinterpret_cast<const uint4 *>(src + 32 * 272);
                    h2 = *reinterpret_cast<const uint4 *>(src + 64 * 272);
                    h3 = *reinterpret_cast<const uint4 *>(src + 96 * 272);
                }
                if (q < 4) {
                    float ln_rs[4], ln_nm[4];
                    if constexpr (LN_IN) {
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) {
                            const float2 pr = *reinterpret_cast<const float2 *>(smem + G2P_TAIL + (hm * 128 + wm * 64 + mb * 16 + fr) * 8);
                            ln_rs[mb] = pr.x; ln_nm[mb] = pr.y;
                        }
                    }
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const int nq = wn * 32 + nb * 16 + fq * 4;                  // column inside the quarter
                        const int nl = hn * 128 + nq;
                        const float4 b = *reinterpret_cast<const float4 *>(smem + G2P_TAIL + 2048 + nl * 4);
                        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
                        if constexpr (LN_IN) cs = *reinterpret_cast<const float4 *>(smem + G2P_TAIL + 3072 + nl * 4);
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) {
                            const int mlh = wm * 64 + mb * 16 + fr;
                            f32x4 vv = acc[hm][hn][mb][nb];
                            if constexpr (LN_IN) {
                                const float nm = ln_nm[mb], rs = ln_rs[mb];
                                vv[0] = fmaf(vv[0], rs, fmaf(nm, cs.x, b.x)); vv[1] = fmaf(vv[1], rs, fmaf(nm, cs.y, b.y));
                                vv[2] = fmaf(vv[2], rs, fmaf(nm, cs.z, b.z)); vv[3] = fmaf(vv[3], rs, fmaf(nm, cs.w, b.w));
                            } else {
                                vv[0] += b.x; vv[1] += b.y; vv[2] += b.z; vv[3] += b.w;
                            }
                            if constexpr (EPI == CPX_EPI_GELU_BF16) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) vv[r] = gelu_erf(vv[r]);
                            } else if constexpr (EPI == CPX_EPI_RELU_BF16) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) vv[r] = fmaxf(vv[r], 0.f);
                            }
                            uint2 o;
                            o.x = pack2<F16>(vv[0], vv[1]);
                            o.y = pack2<F16>(vv[2], vv[3]);
                            *reinterpret_cast<uint2 *>(buf + mlh * 272 + nq * 2) = o;
                        }
                    }
                }
                if (q > 0) {
                    unsigned short *dst = (unsigned short *)g.out + (size_t)(m0 + phm * 128 + srow) * g.ld_out + n0 + phn * 128 + sc16 * 8;
                    const size_t r32 = (size_t)32 * g.ld_out;
                    *reinterpret_cast<uint4 *>(dst) = h0;
                    *reinterpret_cast<uint4 *>(dst + r32) = h1;
                    *reinterpret_cast<uint4 *>(dst + 2 * r32) = h2;
                    *reinterpret_cast<uint4 *>(dst + 3 * r32) = h3;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                G2_BAR();
            }
        } else if constexpr (EPI != CPX_EPI_RESID_BF16 && (FLAGS & G2F_DIRECT) != 0) {
            // ---- direct epilogue: no LDS staging, no barriers between the halves.  After packing, a lane holds 4 consecutive channels
            // (8 bytes) of token row fr for each of the wave's two 16-channel blocks nb; v_permlane16_swap (vdst = block 0, src = block 1:
            // lane rows 1 / 3 of block 0 <-> lane rows 0 / 2 of block 1) leaves every lane with 8 CONSECUTIVE channels -- lane row fq holds
            // channels (fq & 1) * 16 + (fq >> 1) * 8 ... + 7 of the wave's 32 -- so one 16-byte buffer store per (half, column half, row
            // block) writes 16 token rows x 64 contiguous bytes; the neighbouring wave column writes the other half of each 128-byte line.
            // Store addresses: one 32-bit lane offset + scalar offsets.  The stores leave along the vector stream instead of in two bursts.
            if (has_next) prefetch_next();
            const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc((void *)g.out, 0, 0x7FFFFFFF, 0x00020000);
            const unsigned ldb = (unsigned)g.ld_out * 2u;
            const unsigned ovoff = (unsigned)(wm * 64 + fr) * ldb + (unsigned)(wn * 32 + ((fq & 1) << 4) + ((fq >> 1) << 3)) * 2u;
            const unsigned so0 = (unsigned)m0 * ldb + (unsigned)n0 * 2u;
#pragma unroll
            for (int hm = 0; hm < 2; ++hm) {
                float ln_rs[4], ln_nm[4];
                if constexpr (LN_IN) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        const float2 pr = *reinterpret_cast<const float2 *>(smem + G2P_TAIL + (hm * 128 + wm * 64 + mb * 16 + fr) * 8);
                        ln_rs[mb] = pr.x; ln_nm[mb] = pr.y;
                    }
                }
#pragma unroll
                for (int hn = 0; hn < 2; ++hn) {
                    float4 b[2], cs[2];
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const int nl = hn * 128 + wn * 32 + nb * 16 + fq * 4;
                        b[nb] = *reinterpret_cast<const float4 *>(smem + G2P_TAIL + 2048 + nl * 4);
                        cs[nb] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if constexpr (LN_IN) cs[nb] = *reinterpret_cast<const float4 *>(smem + G2P_TAIL + 3072 + nl * 4);
                    }
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        unsigned pk[2][2];
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            f32x4 vv = acc[hm][hn][mb][nb];
                            if constexpr (LN_IN) {
                                const float nm = ln_nm[mb], rs = ln_rs[mb];
                                vv[0] = fmaf(vv[0], rs, fmaf(nm, cs[nb].x, b[nb].x)); vv[1] = fmaf(vv[1], rs, fmaf(nm, cs[nb].y, b[nb].y));
                                vv[2] = fmaf(vv[2], rs, fmaf(nm, cs[nb].z, b[nb].z)); vv[3] = fmaf(vv[3], rs, fmaf(nm, cs[nb].w, b[nb].w));
                            } else {
                                vv[0] += b[nb].x; vv[1] += b[nb].y; vv[2] += b[nb].z; vv[3] += b[nb].w;
                            }
                            if constexpr (EPI == CPX_EPI_GELU_BF16) {
                                if (!DBG || !(g.dbg & 2)) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) vv[r] = gelu_erf(vv[r]);
                                }
                            } else if constexpr (EPI == CPX_EPI_RELU_BF16) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) vv[r] = fmaxf(vv[r], 0.f);
                            }
                            pk[nb][0] = pack2<F16>(vv[0], vv[1]);
                            pk[nb][1] = pack2<F16>(vv[2], vv[3]);
                        }
                        const auto r0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                        const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                        if (!DBG || !(g.dbg & 1) || o[0] == 0x12345678u)
                            __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, ovoff, so0 + (unsigned)(hm * 128 + mb * 16) * ldb + (unsigned)hn * 256u, 0);
                    }
                }
            }
            // the next tile's top rewrites the LDS tail (row parameters, bias, column sums) every wave has just read from
            G2_BAR();
        } else {
            if constexpr (EPI != CPX_EPI_RESID_BF16) {
                if (has_next) prefetch_next();       // no global load result is consumed from here to the next tile top
            }
            char *stg = smem + G2P_STG_OFF;
#pragma unroll
            for (int hm = 0; hm < 2; ++hm) {
                uint4 rres[8];                       // residual rows of this half: requested now, consumed after the LDS pass
                if constexpr (EPI == CPX_EPI_RESID_BF16) {
#pragma unroll
                    for (int it = 0; it < 8; ++it)
                        rres[it] = *reinterpret_cast<const uint4 *>((const unsigned short *)g.aux +
                                                                    (size_t)(m0 + hm * 128 + it * 16 + (tid >> 5)) * g.ld_out + n0 + c16 * 8);
                }
                float ln_rs[4], ln_nm[4];
                if constexpr (LN_IN) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        const float2 pr = *reinterpret_cast<const float2 *>(smem + G2P_TAIL + (hm * 128 + wm * 64 + mb * 16 + fr) * 8);
                        ln_rs[mb] = pr.x; ln_nm[mb] = pr.y;
                    }
                }
#pragma unroll
                for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const int nl = hn * 128 + wn * 32 + nb * 16 + fq * 4;
                        const float4 b = *reinterpret_cast<const float4 *>(smem + G2P_TAIL + 2048 + nl * 4);
                        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
                        if constexpr (LN_IN) cs = *reinterpret_cast<const float4 *>(smem + G2P_TAIL + 3072 + nl * 4);
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) {
                            const int mlh = wm * 64 + mb * 16 + fr;          // row inside this half
                            f32x4 vv = acc[hm][hn][mb][nb];
                            if constexpr (LN_IN) {
                                const float nm = ln_nm[mb], rs = ln_rs[mb];
                                vv[0] = fmaf(vv[0], rs, fmaf(nm, cs.x, b.x)); vv[1] = fmaf(vv[1], rs, fmaf(nm, cs.y, b.y));
                                vv[2] = fmaf(vv[2], rs, fmaf(nm, cs.z, b.z)); vv[3] = fmaf(vv[3], rs, fmaf(nm, cs.w, b.w));
                            } else {
                                vv[0] += b.x; vv[1] += b.y; vv[2] += b.z; vv[3] += b.w;
                            }
                            if constexpr (EPI == CPX_EPI_GELU_BF16) {
                                if (DBG && (g.dbg & 16)) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) vv[r] = gelu_erf_as26(vv[r]);
                                } else if (!DBG || !(g.dbg & 2)) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) vv[r] = gelu_erf(vv[r]);
                                }
                            } else if constexpr (EPI == CPX_EPI_RELU_BF16) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) vv[r] = fmaxf(vv[r], 0.f);
                            }
                            uint2 o;
                            o.x = pack2<F16>(vv[0], vv[1]);
                            o.y = pack2<F16>(vv[2], vv[3]);
                            *reinterpret_cast<uint2 *>(stg + mlh * G2_EPI_LD + nl * 2) = o;
                        }
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                G2_BAR();
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int mlh = it * 16 + (tid >> 5);
                    uint4 vv = *reinterpret_cast<const uint4 *>(stg + mlh * G2_EPI_LD + c16 * 16);
                    const int ml = hm * 128 + mlh;
                    const size_t go = (size_t)(m0 + ml) * g.ld_out + n0 + c16 * 8;
                    if constexpr (EPI == CPX_EPI_RESID_BF16) {
                        const uint4 rr = rres[it];
                        unsigned a[4] = {vv.x, vv.y, vv.z, vv.w}, b[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float lo = from_half<F16>(a[i] & 0xFFFF) + from_half<F16>(b[i] & 0xFFFF);
                            float hi = from_half<F16>(a[i] >> 16) + from_half<F16>(b[i] >> 16);
                            a[i] = pack2<F16>(lo, hi);
                        }
                        vv = make_uint4(a[0], a[1], a[2], a[3]);
                    }
                    if (!DBG || !(g.dbg & 1) || vv.x == 0x12345678u) st16<EPI == CPX_EPI_QKV_BF16, EPI == CPX_EPI_RESID_BF16 || EPI == CPX_EPI_QKV_BF16>((unsigned short *)g.out + go, vv, (g.nt_out & (n0 < 1024 ? 1 : 2)) != 0,
                                                                                        (g.nt_out & (EPI == CPX_EPI_QKV_BF16 ? 16 : 8)) != 0);
                    if constexpr (STATS) {
                        unsigned a[4] = {vv.x, vv.y, vv.z, vv.w};
                        float sm = 0.f, sq = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float lo = from_half<F16>(a[i] & 0xFFFF), hi = from_half<F16>(a[i] >> 16);
                            sm += lo + hi; sq += lo * lo + hi * hi;
                        }
                        sm = half_wave_sum(sm); sq = half_wave_sum(sq);
                        if (c16 == 16)
                            *reinterpret_cast<float2 *>(g.stats_out + ((size_t)(m0 + ml) * LN_SLOTS + tile_n) * 2) = make_float2(sm, sq);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                G2_BAR();
            }
        }
        if (!has_next) break;
        prefetched = next_issued;
        v = vn; m0 = m0n; n0 = n0n; tile_n = tile_nn;
    }
}

#ifdef CPX_DEBUG
// ===========================================================================
// EXPERIMENT (debug build only, cpx_gemm_set_pingpong): "ping-pong" form, 256 (M) x 128 (N) x 64 tile, 256 threads = 4 waves,
// TWO workgroups per CU.  Bitwise identical to the 256^2 kernel, race-screened, and 5-8 % SLOWER on every layer shape
// (profiles/r03_gemm_pingpong_ab.txt) -- kept as the measured record of why:
//   * the two workgroups of a CU do fall into complementary phases by themselves (89 % of every epilogue runs beside the
//     partner's main loop, tools/pp_stamps.py), a forced start offset changes nothing;
//   * but a wave's OWN timeline is the limit: per K tile 64 MFMAs (1 088 cycles) + 12 LDS-DMA instructions at ~60-100 issue
//     cycles each + two LDS round trips + two barriers = ~2 700 cycles, so the main loop of a 256 x 128 tile takes 43 k cycles
//     where the MFMAs alone need 17 k: half the tile size means 1.5x the operand bytes -- and DMA instructions -- per MFMA
//     (0.19 against 0.125 KB), and ~35 B/clk/CU is at the rate one CU can pull from its L2 (~70 GB/s);
//   * beside another wave's saturating MFMA stream a wave issues one VALU instruction per ~8.5 cycles, not one per 3.3
//     (tools/micro/coexec.hip), so an overlapped epilogue is also a slower epilogue.
// ===========================================================================
// The 256^2 kernel holds a CU with one 8-wave workgroup: while its waves run the VALU-bound epilogue (LayerNorm
// fold, bias, erf-GELU, conversion, staging, stores: ~40 % of a mlp.lin1 tile) the CU's matrix cores idle, and while
// they wait at the phase barriers of the main loop nothing else can issue.  Here a workgroup is half as large
// (4 waves, one per SIMD, <= 256 registers, 80 KB of LDS) so that two of them, working on DIFFERENT tiles and
// drifting freely against each other, share every SIMD: one workgroup's epilogue VALU stream issues beside the other's
// MFMA stream (a v_mfma_f32_16x16x32 holds the SIMD's vector issue for 8 of its 16 cycles, tools/micro/coexec.hip), and
// one's barrier / LDS-latency bubbles are the other's issue slots.
// Per-wave output = 2 quadrants (hm) of 64 tokens x 64 channels: tokens 128 hm + 64 wm + [0,64), channels 64 wn + [0,64)
// -> in a phase ALL waves read the same operand items.  Operand items of 16 KB (128 rows x 64 k, lane-linear LDS-DMA
// image, 16-byte chunk c of row r at position c ^ (r & 7)): per K tile t  W(t), X0(t), X1(t) = items 3t, 3t+1, 3t+2, item
// j in slot j % 5 of a 5-slot ring.  Two phases per K tile, ONE raw barrier each, behind a counted vmcnt:
//   A(t): reads W(t), X0(t) -> 32 MFMAs into quadrant 0;  issues item 3t+4 = X0(t+1)   (slot of X1(t-1), read in B(t-1))
//   B(t): reads X1(t)       -> 32 MFMAs into quadrant 1;  issues items 3t+5, 3t+6 = X1(t+1), W(t+2)  (slots of W(t), X0(t))
// A slot is re-filled only after the barrier that follows its last reading phase; every item is requested >= 2 phases
// before the phase that reads it; waits never drain the queue in the steady state (vmcnt(8)).
#define PP_THREADS 256
#define PP_SLOT 16384
#define PP_NSLOT 5
#define PP_EPI_LD 272                                   // staging row: 128 channels x 2 B + 16
#define PP_PAR_OFF (256 * PP_EPI_LD)                    // 69632: 256 x (rstd, -mean rstd) behind the staging rows
#define PP_LDS_BYTES (PP_NSLOT * PP_SLOT)               // 81920 >= PP_PAR_OFF + 2048
__device__ __forceinline__ u32x4 pp_read128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// Which of a CU's two resident workgroups arrived second?  (arrival counters per CU, monotonically increasing: two
// consecutive arrivals on a CU always get different parities, so no reset is needed and nothing depends on it but timing)
__device__ unsigned g_pp_arrival[8 * 256];
#ifdef CPX_DEBUG
// diagnostic (g.dbg & 32): s_memtime at tile start / main-loop end / tile end of every workgroup's tiles + its CU key and
// arrival number: [workgroup < 1024][2 + 3 * 32] 64-bit words, fetched by cpx_gemm_pingpong_stamps
__device__ unsigned long long g_pp_stamps[1024 * 98];
#define PP_STAMP(slot_)                                                                                   \
    do {                                                                                                  \
        if ((g.dbg & 32) && tid == 0 && blockIdx.x < 1024 && it_ < 32) {                                  \
            unsigned long long t_;                                                                        \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
            g_pp_stamps[blockIdx.x * 98 + 2 + 3 * it_ + (slot_)] = t_;                                    \
        }                                                                                                 \
    } while (0)
#else
#define PP_STAMP(slot_) do { } while (0)
#endif
template <int EPI, bool F16, int FLAGS>
__global__ void __launch_bounds__(PP_THREADS, 2) k_gemm_pp(GemmArgs g) {
    constexpr bool LN_IN = (FLAGS & G2F_LN) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int K = g.K, nk = K / 64;
    // ---- phase offset between the two workgroups of a CU: identical tiles take identical time, so two workgroups that
    // start together stay in lockstep (both in the main loop, then both in the epilogue) and never complement each other.
    // The second arrival on a CU therefore waits g.pp_delay x ~8k cycles once; from then on one workgroup's epilogue runs
    // beside the other's main loop for the whole launch.
    if (g.pp_delay > 0 || (g.dbg & 32)) {
        if (tid == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4);      // HW_REG_HW_ID bits 15:8 = SE, SH, CU
            const unsigned xcc = __builtin_amdgcn_s_getreg((2 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID bits 2:0
            *reinterpret_cast<unsigned *>(smem) = atomicAdd(&g_pp_arrival[((xcc & 7) << 8) | (hw & 255)], 1u);
        }
        __syncthreads();
        const unsigned arrival = *reinterpret_cast<volatile unsigned *>(smem);
        __syncthreads();
#ifdef CPX_DEBUG
        if ((g.dbg & 32) && tid == 0 && blockIdx.x < 1024) {
            const unsigned hw = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4), xcc = __builtin_amdgcn_s_getreg((2 << 11) | (0 << 6) | 20);
            g_pp_stamps[blockIdx.x * 98] = ((xcc & 7) << 8) | (hw & 255);
            g_pp_stamps[blockIdx.x * 98 + 1] = arrival;
        }
#endif
        if (arrival & 1)
            for (int i = 0; i < g.pp_delay; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int tiles_m = g.n_blocks / g.tiles_n;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  int it_ = 0;
  for (int v = blockIdx.x; v < g.n_blocks; v += gridDim.x, ++it_) {
    PP_STAMP(0);
    // workgroup -> tile: ids that share an XCD (id % 8) get a contiguous range; the 64 workgroups an XCD runs at a time
    // (2 per CU) cover an 8 (M) x 8 (N) super-tile = 8 activation panels + 8 weight panels
    int tile_m, tile_n;
    {
        const int nxcd = 8, q = g.n_blocks / nxcd, r = g.n_blocks % nxcd, x = v % nxcd;
        const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + v / nxcd;
        if ((tiles_m & 7) == 0 && (g.tiles_n & 7) == 0) {
            const int grp = bid >> 6, w_ = bid & 63, cgn = g.tiles_n >> 3;
            const int rg = grp / cgn, cg = grp - rg * cgn;
            tile_m = rg * 8 + (w_ >> 3);
            tile_n = cg * 8 + (w_ & 7);
        } else {
            tile_m = bid / g.tiles_n;
            tile_n = bid - tile_m * g.tiles_n;
        }
    }
    const int m0 = tile_m * 256, n0 = tile_n * 128;

    // folded LayerNorm: this thread's token row (m0 + tid) -> (rstd, -mean rstd), parked in LDS after the main loop
    float ln_rs = 0.f, ln_nm = 0.f;
    if constexpr (LN_IN) {
        const float4 a = *reinterpret_cast<const float4 *>(g.ln_stats + (size_t)(m0 + tid) * 8);
        const float4 b = *reinterpret_cast<const float4 *>(g.ln_stats + (size_t)(m0 + tid) * 8 + 4);
        const float inv_k = 1.0f / K;
        const float sum = (a.x + a.z) + (b.x + b.z), sq = (a.y + a.w) + (b.y + b.w);
        const float mean = sum * inv_k;
        ln_rs = rsqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + 1e-6f);
        ln_nm = -mean * ln_rs;
    }

    // staging: one item = 128 rows x 64 k = 4 DMA instructions per wave; thread -> (row = 32 q + tid>>3, position tid&7)
    const int srow = tid >> 3, kc = (tid & 7) ^ (srow & 7);
    const unsigned short *pX = g.A + (size_t)(m0 + srow) * K + kc * 8;
    const unsigned short *pW = g.W + (size_t)(n0 + srow) * K + kc * 8;
    const size_t k32 = (size_t)32 * K, k128 = (size_t)128 * K;
    char *sdst = smem + wave * 1024;
    auto issue = [&](const unsigned short *p, int slot) {
        char *d = sdst + slot * PP_SLOT;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + q * k32),
                                             (__attribute__((address_space(3))) void *)(d + q * 4096), 16, 0, 0);
    };
    auto issue_w = [&](int t, int slot) { issue(pW + (size_t)t * 64, slot); };
    auto issue_x = [&](int hm, int t, int slot) { issue(pX + hm * k128 + (size_t)t * 64, slot); };

    unsigned xa[2], wa[2];                       // per-lane fragment bases (slot offset added per phase)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned sw = (unsigned)(((ks * 4 + fq) ^ (fr & 7)) << 4);
        xa[ks] = lds0 + (wm * 64 + fr) * 128 + sw;
        wa[ks] = lds0 + (wn * 64 + fr) * 128 + sw;
    }

    f32x4 acc[2][4][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][b][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 fx[4][2], fw[4][2];

    // prologue: items 0..4 = W(0), X0(0), X1(0), W(1), X0(1)
    issue_w(0, 0); issue_x(0, 0, 1); issue_x(1, 0, 2);
    if (nk > 1) { issue_w(1, 3); issue_x(0, 1, 4); }
    int sl = 0;                                  // slot of item 3t = W(t)
    auto nxt = [](int s_, int d) { int r = s_ + d; return r >= PP_NSLOT ? r - PP_NSLOT : r; };
    for (int t = 0; t < nk; ++t) {
        const int s_w = sl, s_x0 = nxt(sl, 1), s_x1 = nxt(sl, 2);
        // ---- phase A(t)
        if (t == 0) { if (nk > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else if (t < nk - 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t >= 1 && t + 1 < nk) issue_x(0, t + 1, nxt(sl, 4));          // item 3t+4 -> slot of item 3t-1
        {
            const unsigned ow = (unsigned)s_w * PP_SLOT, ox = (unsigned)s_x0 * PP_SLOT;
#pragma unroll
            for (int i = 0; i < 4; ++i) { fw[i][0] = pp_read128(wa[0] + ow + i * 2048); fw[i][1] = pp_read128(wa[1] + ow + i * 2048); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { fx[i][0] = pp_read128(xa[0] + ox + i * 2048); fx[i][1] = pp_read128(xa[1] + ox + i * 2048); }
        }
        G2_LGKM0();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[0][mb][nb] = mfma16v<F16>(fw[nb][ks], fx[mb][ks], acc[0][mb][nb]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase B(t)
        if (t < nk - 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < nk) {
            issue_x(1, t + 1, nxt(sl, 0));                                 // item 3t+5 -> slot of W(t)
            if (t + 2 < nk) issue_w(t + 2, nxt(sl, 1));                    // item 3t+6 -> slot of X0(t)
        }
        {
            const unsigned ox = (unsigned)s_x1 * PP_SLOT;
#pragma unroll
            for (int i = 0; i < 4; ++i) { fx[i][0] = pp_read128(xa[0] + ox + i * 2048); fx[i][1] = pp_read128(xa[1] + ox + i * 2048); }
        }
        G2_LGKM0();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[1][mb][nb] = mfma16v<F16>(fw[nb][ks], fx[mb][ks], acc[1][mb][nb]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        sl = nxt(sl, 3);
    }
    // (the vmcnt at the tail depends on how many items were actually issued after the awaited one: see the counts above;
    // with t + 2 >= nk the B phase issues one item, with t + 1 >= nk none)
    __builtin_amdgcn_s_barrier();                // every wave is done reading the ring: it becomes the staging area
    PP_STAMP(1);
    if constexpr (LN_IN) {
        *reinterpret_cast<float2 *>(smem + PP_PAR_OFF + tid * 8) = make_float2(ln_rs, ln_nm);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: f32 -> (LayerNorm fold, bias, activation) -> half rows in LDS -> whole 256-byte rows to HBM
    float4 colb[4], colc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int nl = wn * 64 + nb * 16 + fq * 4;
        colb[nb] = g.bias ? *reinterpret_cast<const float4 *>(g.bias + n0 + nl) : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (LN_IN) colc[nb] = *reinterpret_cast<const float4 *>(g.ln_colsum + n0 + nl);
    }
    // this lane's 8 token rows: LayerNorm parameters to registers BEFORE the staging stores (the stores alias every LDS
    // pointer for the compiler: read inside the loop, each parameter read would wait for the stores in front of it)
    float prs[2][4], pnm[2][4];
#pragma unroll
    for (int hm = 0; hm < 2; ++hm)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            prs[hm][mb] = 1.f; pnm[hm][mb] = 0.f;
            if constexpr (LN_IN) {
                const float2 pr = *reinterpret_cast<const float2 *>(smem + PP_PAR_OFF + (hm * 128 + wm * 64 + mb * 16 + fr) * 8);
                prs[hm][mb] = pr.x; pnm[hm][mb] = pr.y;
            }
        }
#pragma unroll
    for (int hm = 0; hm < 2; ++hm)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int ml = hm * 128 + wm * 64 + mb * 16 + fr;
            const float rs = prs[hm][mb], nm = pnm[hm][mb];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int nl = wn * 64 + nb * 16 + fq * 4;
                f32x4 vv = acc[hm][mb][nb];
                const float4 b = colb[nb];
                if constexpr (LN_IN) {
                    const float4 cs = colc[nb];
                    vv[0] = fmaf(vv[0], rs, fmaf(nm, cs.x, b.x)); vv[1] = fmaf(vv[1], rs, fmaf(nm, cs.y, b.y));
                    vv[2] = fmaf(vv[2], rs, fmaf(nm, cs.z, b.z)); vv[3] = fmaf(vv[3], rs, fmaf(nm, cs.w, b.w));
                } else {
                    vv[0] += b.x; vv[1] += b.y; vv[2] += b.z; vv[3] += b.w;
                }
                if constexpr (EPI == CPX_EPI_GELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[r] = gelu_erf(vv[r]);
                } else if constexpr (EPI == CPX_EPI_RELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[r] = fmaxf(vv[r], 0.f);
                }
                uint2 o;
                o.x = pack2<F16>(vv[0], vv[1]);
                o.y = pack2<F16>(vv[2], vv[3]);
                *reinterpret_cast<uint2 *>(smem + ml * PP_EPI_LD + nl * 2) = o;
            }
        }
    __syncthreads();
    const int c16 = tid & 15;                    // 16-byte chunk of a 256-byte row
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int ml = it * 16 + (tid >> 4);
        const uint4 vv = *reinterpret_cast<const uint4 *>(smem + ml * PP_EPI_LD + c16 * 16);
        *reinterpret_cast<uint4 *>((unsigned short *)g.out + (size_t)(m0 + ml) * g.ld_out + n0 + c16 * 8) = vv;
    }
    __syncthreads();                             // staging rows read: the ring may be re-filled for the next tile
    PP_STAMP(2);
  }
}

#endif  // CPX_DEBUG (ping-pong experiment)

CPX_SWITCH(g_gemm_persist, 1);      // 1 = persistent 256^2 kernel (k_gemm256p), 0 = one workgroup per tile (k_gemm256)
CPX_SWITCH(g_gemm_variant, 1);      // 1 = LDS-DMA staging, 0 = register staging (debug / A-B)
CPX_SWITCH(g_gemm_dbg, 0);          // timing-only ablations of the 256^2 epilogue
CPX_SWITCH(g_gemm_l2, 1);           // 1 = 8 x 4 super-tile order per XCD, N-sweep; 2: M-sweep; 0: row-major
CPX_SWITCH(g_gemm_persist_qkv, 1);  // balanced persistent tile list for the qkv projection
// experiment switch (default off): mlp.lin2 walks M backwards so that the most recently written rows of the
// 268 MB hidden tensor (> the 256 MB Infinity Cache) are read first.  Bitwise identical; measured 26.50 vs
// 26.55 ms per engine step in a one-process A/B (tools/ab_switch.py) -> no gain, not enabled.
CPX_SWITCH(g_gemm_rev, 0);
CPX_SWITCH(g_gemm_big, 1);          // 1 = use the 256^2 kernel when the shape allows
CPX_SWITCH(g_gemm_bal, 1);          // 1 = balanced fragment-read schedule of the persistent main loop (G2F_BAL; K >= 256) for the bf16 residual + statistics epilogue, 0 = plain
CPX_SWITCH(g_gemm_split, 0);        // 1 = counted LDS waits inside the main-loop phases (k_gemm256p<.., G2F_SPLIT>; experiment)
CPX_SWITCH(g_gemm_direct, 1);       // 1 = direct-store epilogue (G2F_DIRECT) for the GELU epilogue, 2 (debug build) = for every non-residual epilogue, 0 = staged rows
CPX_SWITCH(g_gemm_epi4, 0);         // 1 = quarter-tile epilogue of the persistent 256^2 kernel (conversion beside the previous quarter's stores)
CPX_SWITCH(g_gemm_nt, 15);          // bits 0 / 1 / 2: the q / k / V^T thirds of the qkv projection's output leave by non-temporal stores; bit 3 (round 6): the residual epilogues (attn.proj, mlp.lin2) store at agent scope (sc1, written through: proj 73.9 -> 72.4 us, lin2 105.2 -> 103.2 in situ); 15 = production, 0 = ordinary stores (A/B)
CPX_SWITCH(g_gemm_4w, 1);           // bit 0 (production): mlp.lin1 (bf16, folded LayerNorm + GELU) on the one-wave-per-SIMD kernel (cpx_gemm4w.hip), 0.98 of k_gemm256p;
                                    // bit 1 (debug build, measured and NOT shipped): attn.proj / mlp.lin2 (residual + row statistics) on it as well -- 1.06 / 1.03 of
                                    // k_gemm256p (profiles/r05_ab_gemm4w_resid.txt): that epilogue is unpacked integer / f32 vector work, which a lone wave issues at
                                    // 4 cycles per instruction against 2.9 for two waves of a SIMD.  Same bits either way.
CPX_SWITCH(g_gemm_pp, 0);           // 1 = ping-pong kernel (256 x 128 tiles, two 4-wave workgroups per CU) for the epilogues it covers
CPX_SWITCH(g_gemm_pp_persist, 1);   // ping-pong kernel: 1 = two persistent workgroups per CU walk the tiles, 0 = one workgroup per tile
CPX_SWITCH(g_gemm_pp_delay, 2);     // ping-pong kernel: start offset of a CU's second workgroup, x s_sleep 127 (~8k cycles)
#ifdef CPX_DEBUG
extern "C" void cpx_gemm_set_persistent(int on) { g_gemm_persist = on; }
extern "C" void cpx_gemm_set_variant(int glds) { g_gemm_variant = glds; }
extern "C" void cpx_gemm_set_dbg(int v) { g_gemm_dbg = v; }
extern "C" void cpx_gemm_set_persistent_qkv(int on) { g_gemm_persist_qkv = on; }
extern "C" void cpx_gemm_set_l2_block(int on) { g_gemm_l2 = on; }
extern "C" void cpx_gemm_set_reverse(int on) { g_gemm_rev = on; }
extern "C" void cpx_gemm_set_big(int on) { g_gemm_big = on; }
extern "C" void cpx_gemm_set_epi4(int on) { g_gemm_epi4 = on; }
extern "C" void cpx_gemm_set_split(int on) { g_gemm_split = on; }
extern "C" void cpx_gemm_set_direct(int on) { g_gemm_direct = on; }
extern "C" void cpx_gemm_set_balanced(int on) { g_gemm_bal = on; }
extern "C" void cpx_gemm_set_pingpong(int on) { g_gemm_pp = on; }
extern "C" void cpx_gemm_set_4w(int on) { g_gemm_4w = on; }
extern "C" void cpx_gemm_set_nt(int on) { g_gemm_nt = on; }
extern "C" void cpx_gemm_set_pingpong_opts(int persistent, int delay) { g_gemm_pp_persist = persistent; g_gemm_pp_delay = delay; }
#endif

#ifdef CPX_DEBUG
// workgroups of the ping-pong kernel the runtime admits per CU (2 is the design point: 2 x 80 KB of LDS = all 160 KB)
extern "C" int cpx_gemm_pingpong_stamps(unsigned long long *host_out, size_t n_words) {
    if (n_words > 1024 * 98) n_words = 1024 * 98;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pp_stamps), n_words * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
extern "C" int cpx_gemm_pingpong_occupancy(void) {
    int n = -1;
    (void)hipFuncSetAttribute((const void *)k_gemm_pp<CPX_EPI_GELU_BF16, false, G2F_LN>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k_gemm_pp<CPX_EPI_GELU_BF16, false, G2F_LN>, PP_THREADS, PP_LDS_BYTES) != hipSuccess) return -1;
    return n;
}
#endif

template <int EPI, bool F16, int FLAGS>
static void launch_gemm256_flags(const GemmArgs &a, hipStream_t s) {
    // the qkv projection keeps one workgroup per tile: a third of its tiles (the V^T ones) cannot overlap their
    // epilogue with the next prefetch and cost more, and a static tile list cannot balance that (measured +3 %)
    const bool qkv_balanced = EPI == CPX_EPI_QKV_BF16 && g_gemm_persist_qkv && (a.n_blocks / a.tiles_n) % 64 == 0 && a.tiles_n == 12;
    // (the persistent kernel addresses its operands through 32-bit buffer offsets)
    const bool fits32 = (size_t)a.M * (size_t)a.K * 2 < ((size_t)1 << 31) && (size_t)a.N * (size_t)a.K * 2 < ((size_t)1 << 31);
    if (g_gemm_persist && fits32 && (EPI != CPX_EPI_QKV_BF16 || qkv_balanced)) {
        static CpxOncePerDevice once_p;
        static int n_cu = 0;
        once_p([] {
            (void)hipFuncSetAttribute((const void *)k_gemm256p<EPI, F16, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, G2P_LDS_BYTES);
            int dev = 0, cus = 256;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            n_cu = cus > 0 ? cus : 256;
        });
        if (EPI == CPX_EPI_QKV_BF16) {
            if (n_cu == 256) {
                GemmArgs b = a;
                b.l2_block = 3;
                hipLaunchKernelGGL((k_gemm256p<EPI, F16, FLAGS>), dim3(256), dim3(G2_THREADS), G2P_LDS_BYTES, s, b);
                return;
            }
        } else {
            const int grid = a.n_blocks < n_cu ? a.n_blocks : n_cu;
            hipLaunchKernelGGL((k_gemm256p<EPI, F16, FLAGS>), dim3(grid), dim3(G2_THREADS), G2P_LDS_BYTES, s, a);
            return;
        }
    }
    static CpxOncePerDevice once;
    once([] { (void)hipFuncSetAttribute((const void *)k_gemm256<EPI, F16, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES); });
    hipLaunchKernelGGL((k_gemm256<EPI, F16, FLAGS>), dim3(a.n_blocks), dim3(G2_THREADS), G2_LDS_BYTES, s, a);
}

template <int EPI, bool F16>
static bool launch_gemm256(const GemmArgs &a0, hipStream_t s) {
    if constexpr (EPI == CPX_EPI_F32) return false;
    else {
        // (an odd number of K tiles only on the one-workgroup-per-tile kernel, which is what the positional epilogue takes)
        if (!g_gemm_big || a0.M % 256 || a0.N % 256 || a0.K < 128) return false;
        if ((a0.K / 64) % 2 && EPI != CPX_EPI_POS_BF16) return false;
        if ((a0.M / 256) * (a0.N / 256) < 256) return false;          // not enough tiles for 256 CUs
#ifdef CPX_DEBUG
        if constexpr (EPI == CPX_EPI_BF16 || EPI == CPX_EPI_GELU_BF16 || EPI == CPX_EPI_RELU_BF16) {
            if (g_gemm_pp) {
                GemmArgs p = a0;
                p.tiles_n = p.N / 128; p.n_blocks = (p.M / 256) * (p.N / 128);
                static CpxOncePerDevice once_pp;
                once_pp([] {
                    (void)hipFuncSetAttribute((const void *)k_gemm_pp<EPI, F16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
                    (void)hipFuncSetAttribute((const void *)k_gemm_pp<EPI, F16, G2F_LN>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
                });
                static int n_cu_pp = 0;
                if (!n_cu_pp) {
                    int dev = 0, cus = 256;
                    (void)hipGetDevice(&dev);
                    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                    n_cu_pp = cus > 0 ? cus : 256;
                }
                const int grid = g_gemm_pp_persist ? (p.n_blocks < 2 * n_cu_pp ? p.n_blocks : 2 * n_cu_pp) : p.n_blocks;
                p.pp_delay = g_gemm_pp_delay;
                if (p.ln_stats) hipLaunchKernelGGL((k_gemm_pp<EPI, F16, G2F_LN>), dim3(grid), dim3(PP_THREADS), PP_LDS_BYTES, s, p);
                else hipLaunchKernelGGL((k_gemm_pp<EPI, F16, 0>), dim3(grid), dim3(PP_THREADS), PP_LDS_BYTES, s, p);
                return true;
            }
        }
#endif
        GemmArgs a = a0;
        a.tiles_n = a.N / 256; a.n_blocks = (a.M / 256) * (a.N / 256);
        // one instantiation per (LayerNorm consumer | statistics producer) x (timing ablations, bf16 only)
        constexpr bool PRODUCER = EPI == CPX_EPI_RESID_BF16 || EPI == CPX_EPI_POS_BF16;
        constexpr int F1 = PRODUCER ? G2F_STATS : G2F_LN;
        const bool f1 = PRODUCER ? a.stats_out != nullptr : a.ln_stats != nullptr;
        if constexpr (EPI == CPX_EPI_POS_BF16) {       // one workgroup per tile (its epilogue reads the positional table with plain loads)
            static CpxOncePerDevice once_pos;
            once_pos([] {
                (void)hipFuncSetAttribute((const void *)k_gemm256<EPI, F16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES);
                (void)hipFuncSetAttribute((const void *)k_gemm256<EPI, F16, G2F_STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES);
            });
            if (f1) hipLaunchKernelGGL((k_gemm256<EPI, F16, G2F_STATS>), dim3(a.n_blocks), dim3(G2_THREADS), G2_LDS_BYTES, s, a);
            else hipLaunchKernelGGL((k_gemm256<EPI, F16, 0>), dim3(a.n_blocks), dim3(G2_THREADS), G2_LDS_BYTES, s, a);
            return true;
        } else {
#ifdef CPX_DEBUG
            if constexpr (!F16) {
                if (a.dbg) {
                    if constexpr (EPI == CPX_EPI_GELU_BF16) {        // the ablations time the epilogue form production uses
                        if (g_gemm_direct) {
                            if (f1) launch_gemm256_flags<EPI, F16, F1 | G2F_DBG | G2F_DIRECT>(a, s);
                            else launch_gemm256_flags<EPI, F16, G2F_DBG | G2F_DIRECT>(a, s);
                            return true;
                        }
                    }
                    if (f1) launch_gemm256_flags<EPI, F16, F1 | G2F_DBG>(a, s);
                    else launch_gemm256_flags<EPI, F16, G2F_DBG>(a, s);
                    return true;
                }
            }
#endif
#ifdef CPX_DEBUG
            if (g_gemm_split) {
                if (f1) launch_gemm256_flags<EPI, F16, F1 | G2F_SPLIT>(a, s);
                else launch_gemm256_flags<EPI, F16, G2F_SPLIT>(a, s);
                return true;
            }
#endif
#ifdef CPX_DEBUG
            if constexpr (EPI != CPX_EPI_RESID_BF16) {
                if (g_gemm_epi4) {
                    if (f1) launch_gemm256_flags<EPI, F16, F1 | G2F_Q4>(a, s);
                    else launch_gemm256_flags<EPI, F16, G2F_Q4>(a, s);
                    return true;
                }
            }
#endif
            // mlp.lin1 of the bf16 network (folded LayerNorm + bias + erf-GELU) on the one-wave-per-SIMD kernel: bitwise equal to the
            // k_gemm256p instantiation below, its main loop and its epilogue both faster (profiles/r05_ab_gemm4w_*.txt)
            if constexpr (EPI == CPX_EPI_GELU_BF16) {
                if ((g_gemm_4w & 1) && f1 && a.bias && a.K >= 256 && (a.K / 64) % 2 == 0 &&
                    cpx_gemm4w_gelu_ln(F16 ? 1 : 0, a.A, a.W, a.M, a.N, a.K, a.bias, a.ln_stats, a.ln_colsum, a.out, a.ld_out, s))
                    return true;
            }
#ifdef CPX_DEBUG
            // attn.proj / mlp.lin2 of the bf16 network (residual + row statistics, N = 1024) on that kernel: bitwise equal to k_gemm256p<.., F1 | G2F_BAL>
            // and 3 - 6 % SLOWER, so only the debug build's cpx_gemm_set_4w(3) takes this branch (g_gemm_4w is the constant 1 in the product)
            if constexpr (EPI == CPX_EPI_RESID_BF16 && !F16) {
                if ((g_gemm_4w & 2) && f1 && a.bias && a.aux && a.N == 1024 && a.K >= 256 && (a.K / 64) % 2 == 0 &&
                    cpx_gemm4w_resid_stats(a.A, a.W, a.M, a.N, a.K, a.bias, a.aux, a.out, a.ld_out, a.stats_out, s))
                    return true;
            }
#endif
            // direct-store epilogue (G2F_DIRECT): production for the GELU epilogue (mlp.lin1 -3.5 %, bitwise equal); the lighter epilogues
            // measured equal or slower with it (qkv +1.2 %) and keep the staged rows -- cpx_gemm_set_direct(2) in the debug build forces it
            constexpr bool DIRECT_OK = EPI == CPX_EPI_GELU_BF16
#ifdef CPX_DEBUG
                                       || EPI != CPX_EPI_RESID_BF16
#endif
                ;
            // balanced fragment-read schedule (G2F_BAL): ONLY the bf16 residual epilogue with row statistics (proj / mlp.lin2 of the engine: 0.98 of the
            // plain schedule, bitwise equal, 255 VGPRs and no scratch).  Everything else keeps the plain schedule: the fp16 and the statistics-less
            // instantiations spill an accumulator under it, qkv gains nothing, mlp.lin1 loses what its direct-store epilogue gained, and the
            // LayerNorm + direct-store + balanced combination -- built once as a debug variant, 255 VGPRs + scratch -- returned NaNs in four token rows
            // of every tile (not understood; the combination no longer exists in either build)
            constexpr bool BAL_OK = EPI == CPX_EPI_RESID_BF16 && !F16;
            const bool bal = BAL_OK && g_gemm_bal != 0 && f1 && a.K >= 256;
            if constexpr (DIRECT_OK) {
                if ((EPI == CPX_EPI_GELU_BF16 ? g_gemm_direct != 0 : g_gemm_direct == 2) && (size_t)a.M * (size_t)a.ld_out * 2 < ((size_t)1 << 31)) {
                    if (f1) launch_gemm256_flags<EPI, F16, F1 | G2F_DIRECT>(a, s);
                    else launch_gemm256_flags<EPI, F16, G2F_DIRECT>(a, s);
                    return true;
                }
            }
            if constexpr (BAL_OK) {
                if (bal) {
                    launch_gemm256_flags<EPI, F16, F1 | G2F_BAL>(a, s);
                    return true;
                }
            }
            if (f1) launch_gemm256_flags<EPI, F16, F1>(a, s);
            else launch_gemm256_flags<EPI, F16, 0>(a, s);
            return true;
        }
    }
}

template <int EPI>
static void launch_gemm(const GemmArgs &a, hipStream_t s, bool f16) {
    if (f16 ? launch_gemm256<EPI, true>(a, s) : launch_gemm256<EPI, false>(a, s)) return;
    dim3 grid(a.n_blocks), block(GEMM_THREADS);
    size_t lds = 2 * STAGE_BYTES;
#ifdef CPX_DEBUG
    if (!g_gemm_variant) {           // register-staged variant of the 128^2 kernel: A/B reference only
        if (f16) hipLaunchKernelGGL((k_gemm<EPI, true, false>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_gemm<EPI, false, false>), grid, block, lds, s, a);
        return;
    }
#endif
    if (f16) hipLaunchKernelGGL((k_gemm<EPI, true, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((k_gemm<EPI, false, true>), grid, block, lds, s, a);
}

// row statistics (sum, sum of squares) of a half-precision [rows][1024] matrix into slot 0 of
// the [rows][4][2] layout the LN-folded GEMMs read (slots 1..3 zeroed).  One wave per row.
template <bool F16>
__global__ void __launch_bounds__(256) k_row_stats(const unsigned short *__restrict__ x, int rows,
                                                   float *__restrict__ st) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const unsigned short *xr = x + (size_t)row * 1024 + lane * 16;
    const uint4 a = *reinterpret_cast<const uint4 *>(xr), b = *reinterpret_cast<const uint4 *>(xr + 8);
    const unsigned u[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float lo = from_half<F16>(u[i] & 0xFFFF), hi = from_half<F16>(u[i] >> 16);
        sm += lo + hi; sq += lo * lo + hi * hi;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o); sq += __shfl_xor(sq, o); }
    if (lane < 4) *reinterpret_cast<float2 *>(st + ((size_t)row * LN_SLOTS + lane) * 2) = lane == 0 ? make_float2(sm, sq) : make_float2(0.f, 0.f);
}
int cpx_row_stats_half(int dtype, const void *x, int rows, float *stats, void *stream) {
    CPX_REQUIRE(x && stats && rows > 0 && (dtype == CPX_DT_BF16 || dtype == CPX_DT_F16));
    if (dtype == CPX_DT_F16) hipLaunchKernelGGL(k_row_stats<true>, dim3(cpx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x, rows, stats);
    else hipLaunchKernelGGL(k_row_stats<false>, dim3(cpx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x, rows, stats);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

extern "C" int cpx_row_stats(const void *x, int rows, float *stats, void *stream) {
    return cpx_row_stats_half(CPX_DT_BF16, x, rows, stats, stream);
}

// does this shape take the 256^2 kernel (whose RESID epilogue can emit LN statistics)?
int cpx_gemm_half_uses_big_tile(int M, int N, int K, int epilogue) {
    if (epilogue == CPX_EPI_F32) return 0;
    if (!g_gemm_big || M % 256 || N % 256 || K < 128) return 0;
    if ((K / 64) % 2 && epilogue != CPX_EPI_POS_BF16) return 0;
    return (M / 256) * (N / 256) >= 256;
}

int cpx_gemm_half(int dtype, const void *A, const void *Wt, int M, int N, int K, int epilogue,
                  const float *bias, const void *aux, void *out, int ld_out,
                  const float *ln_stats, const float *ln_colsum, float *stats_out, void *stream) {
    CPX_REQUIRE(A && Wt && out && (dtype == CPX_DT_BF16 || dtype == CPX_DT_F16));
    const bool f16 = dtype == CPX_DT_F16;
    CPX_REQUIRE(M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % BK == 0);
    CPX_REQUIRE(ld_out >= N && ld_out % 4 == 0);
    CPX_REQUIRE((epilogue != CPX_EPI_RESID_BF16 && epilogue != CPX_EPI_POS_BF16 && epilogue != CPX_EPI_QKV_BF16) || aux);
    CPX_REQUIRE(epilogue != CPX_EPI_QKV_BF16 || (N == 3072 && M % 1024 == 0));
    CPX_REQUIRE(!ln_stats || (ln_colsum && epilogue != CPX_EPI_RESID_BF16));
    CPX_REQUIRE(!stats_out || ((epilogue == CPX_EPI_RESID_BF16 || epilogue == CPX_EPI_POS_BF16) && N == 1024 && cpx_gemm_half_uses_big_tile(M, N, K, epilogue)));
    GemmArgs a;
    a.A = (const unsigned short *)A; a.W = (const unsigned short *)Wt;
    a.M = M; a.N = N; a.K = K; a.bias = bias; a.aux = aux; a.out = out; a.ld_out = ld_out;
    a.tiles_n = N / BN; a.n_blocks = (M / BM) * (N / BN);
    a.ln_stats = ln_stats; a.ln_colsum = ln_colsum; a.stats_out = stats_out; a.l2_block = g_gemm_l2; a.dbg = g_gemm_dbg; a.nt_out = g_gemm_nt;
    a.rev_m = (g_gemm_rev && K >= 4096) ? 1 : 0;
    a.conv_c = 0; a.pp_delay = 0; a.epi4 = g_gemm_epi4;
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case CPX_EPI_BF16: launch_gemm<CPX_EPI_BF16>(a, s, f16); break;
        case CPX_EPI_GELU_BF16: launch_gemm<CPX_EPI_GELU_BF16>(a, s, f16); break;
        case CPX_EPI_RESID_BF16: launch_gemm<CPX_EPI_RESID_BF16>(a, s, f16); break;
        case CPX_EPI_F32: launch_gemm<CPX_EPI_F32>(a, s, f16); break;
        case CPX_EPI_POS_BF16: launch_gemm<CPX_EPI_POS_BF16>(a, s, f16); break;
        case CPX_EPI_RELU_BF16: launch_gemm<CPX_EPI_RELU_BF16>(a, s, f16); break;
        case CPX_EPI_QKV_BF16: launch_gemm<CPX_EPI_QKV_BF16>(a, s, f16); break;
        default: CPX_REQUIRE(!"unknown epilogue");
    }
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// 3x3 convolution (padding 1, no bias unless given) over 32 x 32-token images as an implicit GEMM on the 128^2 kernel:
// x [M = S*1024][C] token-major, Wt [N][9*C] with k = tap*C + c, out [M][ld_out]; epilogue CPX_EPI_BF16 or CPX_EPI_RELU_BF16.
// The im2col operand (9x the activation) is never written: the LDS-DMA of each K tile reads the shifted token's chunk,
// or a zero chunk outside the image.
int cpx_conv3_half(int dtype, const void *x, const void *Wt, int M, int N, int C, int epilogue, const float *bias,
                   void *out, int ld_out, void *stream) {
    CPX_REQUIRE(x && Wt && out && (dtype == CPX_DT_BF16 || dtype == CPX_DT_F16));
    CPX_REQUIRE(M > 0 && M % 1024 == 0 && N > 0 && N % BN == 0 && C > 0 && C % BK == 0);
    CPX_REQUIRE(ld_out >= N && ld_out % 4 == 0 && (epilogue == CPX_EPI_BF16 || epilogue == CPX_EPI_RELU_BF16));
    GemmArgs a;
    a.A = (const unsigned short *)x; a.W = (const unsigned short *)Wt;
    a.M = M; a.N = N; a.K = 9 * C; a.bias = bias; a.aux = nullptr; a.out = out; a.ld_out = ld_out;
    a.tiles_n = N / BN; a.n_blocks = (M / BM) * (N / BN);
    a.ln_stats = nullptr; a.ln_colsum = nullptr; a.stats_out = nullptr; a.l2_block = 0; a.dbg = 0; a.rev_m = 0; a.nt_out = 0;
    a.conv_c = C; a.pp_delay = 0; a.epi4 = 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(a.n_blocks), block(GEMM_THREADS);
    const size_t lds = 2 * STAGE_BYTES;
    const bool f16 = dtype == CPX_DT_F16;
    if (epilogue == CPX_EPI_BF16) {
        if (f16) hipLaunchKernelGGL((k_gemm<CPX_EPI_BF16, true, true, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_gemm<CPX_EPI_BF16, false, true, true>), grid, block, lds, s, a);
    } else {
        if (f16) hipLaunchKernelGGL((k_gemm<CPX_EPI_RELU_BF16, true, true, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_gemm<CPX_EPI_RELU_BF16, false, true, true>), grid, block, lds, s, a);
    }
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

extern "C" int cpx_conv3x3(int dtype, const void *x, const void *Wt, int M, int N, int C, int epilogue, const float *bias,
                           void *out, int ld_out, void *stream) {
    return cpx_conv3_half(dtype, x, Wt, M, N, C, epilogue, bias, out, ld_out, stream);
}

extern "C" int cpx_gemm_ln(const void *A, const void *Wt, int M, int N, int K, int epilogue,
                           const float *bias, const void *aux, void *out, int ld_out,
                           const float *ln_stats, const float *ln_colsum, float *stats_out, void *stream) {
    return cpx_gemm_half(CPX_DT_BF16, A, Wt, M, N, K, epilogue, bias, aux, out, ld_out, ln_stats, ln_colsum, stats_out, stream);
}
extern "C" int cpx_gemm_bf16(const void *A, const void *Wt, int M, int N, int K, int epilogue,
                             const float *bias, const void *aux, void *out, int ld_out, void *stream) {
    return cpx_gemm_half(CPX_DT_BF16, A, Wt, M, N, K, epilogue, bias, aux, out, ld_out, nullptr, nullptr, nullptr, stream);
}
extern "C" int cpx_gemm(int dtype, const void *A, const void *Wt, int M, int N, int K, int epilogue,
                        const float *bias, const void *aux, void *out, int ld_out, void *stream) {
    if (dtype == CPX_DT_F32)
        return cpx_gemm_f32((const float *)A, (const float *)Wt, M, N, K, epilogue, bias, (const float *)aux, (float *)out, ld_out, stream);
    return cpx_gemm_half(dtype, A, Wt, M, N, K, epilogue, bias, aux, out, ld_out, nullptr, nullptr, nullptr, stream);
}
extern "C" int cpx_gemm_uses_big_tile(int M, int N, int K, int epilogue) { return cpx_gemm_half_uses_big_tile(M, N, K, epilogue); }
