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
#include "cpx_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define BM 128
#define BN 128
#define BK 64
#define GEMM_THREADS 256
#define TILE_BYTES (128 * 128)          // one operand tile: 128 rows x 64 k x 2 B
#define STAGE_BYTES (2 * TILE_BYTES)

struct GemmArgs {
    const unsigned short *A;
    const unsigned short *W;
    int M, N, K;
    const float *bias;
    const void *aux;        // residual (bf16 [M][ld_out]) or pos (f32 [1024][N])
    void *out;
    int ld_out;
    int tiles_n, n_blocks;
};

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const uint4 &a, const uint4 &b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8 *>(&a),
                                                      *reinterpret_cast<const f16x8 *>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&a),
                                                       *reinterpret_cast<const bf16x8 *>(&b), c, 0, 0, 0);
}

// exact-erf GELU (nn.GELU default).  erf by Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7
// (below half-precision output rounding by > 3 orders of magnitude); one v_exp + one v_rcp
// instead of libm erff's ~40-instruction piecewise polynomial in the epilogue.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(1.0f + 0.3275911f * z);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float e = 1.0f - poly * t * __expf(-z * z);
    const float erfv = copysignf(e, x);
    return 0.5f * x * (1.0f + erfv);
}

template <bool F16>
__device__ __forceinline__ unsigned short to_half(float f) {
    if constexpr (F16) { _Float16 h = (_Float16)f; return *reinterpret_cast<unsigned short *>(&h); }
    else return f32_to_bf16(f);
}
template <bool F16>
__device__ __forceinline__ float from_half(unsigned short u) {
    if constexpr (F16) { return (float)*reinterpret_cast<_Float16 *>(&u); }
    else return bf16_to_f32(u);
}

// stage one 128x64 operand tile (rows row0.., k from k0) into LDS at `lds` (16 KB)
template <bool GLDS>
__device__ __forceinline__ void stage_tile(const unsigned short *__restrict__ g, int ldk, int row0, int k0,
                                           char *lds, int tid, uint4 *regs) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int L = q * GEMM_THREADS + tid;          // linear 16-B chunk index in the LDS image
        int row = L >> 3, pc = L & 7;
        int kc = pc ^ (row & 7);                 // source chunk that belongs at this position
        const unsigned short *src = g + (size_t)(row0 + row) * ldk + k0 + kc * 8;
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

template <int EPI, bool F16, bool GLDS>
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
    stage_tile<GLDS>(g.A, g.K, m0, 0, smem, tid, ra);
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
            stage_tile<GLDS>(g.A, g.K, m0, (kt + 1) * BK, nxt, tid, ra);
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
            } else {
                uint2 o;
                o.x = (unsigned)to_half<F16>(v[0]) | ((unsigned)to_half<F16>(v[1]) << 16);
                o.y = (unsigned)to_half<F16>(v[2]) | ((unsigned)to_half<F16>(v[3]) << 16);
                *reinterpret_cast<uint2 *>((unsigned short *)g.out + (size_t)m * g.ld_out + n) = o;
            }
        }
    }
}

static int g_gemm_variant = 1;     // 1 = LDS-DMA staging, 0 = register staging (debug / A-B)
static int g_gemm_f16 = 0;
extern "C" void cpx_gemm_set_variant(int glds) { g_gemm_variant = glds; }
extern "C" void cpx_set_half_dtype(int f16) { g_gemm_f16 = f16; }
extern "C" int cpx_get_half_dtype(void) { return g_gemm_f16; }

template <int EPI>
static void launch_gemm(const GemmArgs &a, hipStream_t s) {
    dim3 grid(a.n_blocks), block(GEMM_THREADS);
    size_t lds = 2 * STAGE_BYTES;
    if (g_gemm_f16) {
        if (g_gemm_variant) hipLaunchKernelGGL((k_gemm<EPI, true, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_gemm<EPI, true, false>), grid, block, lds, s, a);
    } else {
        if (g_gemm_variant) hipLaunchKernelGGL((k_gemm<EPI, false, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_gemm<EPI, false, false>), grid, block, lds, s, a);
    }
}

extern "C" int cpx_gemm_bf16(const void *A, const void *Wt, int M, int N, int K, int epilogue,
                             const float *bias, const void *aux, void *out, int ld_out, void *stream) {
    CPX_REQUIRE(A && Wt && out);
    CPX_REQUIRE(M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % BK == 0);
    CPX_REQUIRE(ld_out >= N && ld_out % 4 == 0);
    CPX_REQUIRE((epilogue != CPX_EPI_RESID_BF16 && epilogue != CPX_EPI_POS_BF16) || aux);
    GemmArgs a;
    a.A = (const unsigned short *)A; a.W = (const unsigned short *)Wt;
    a.M = M; a.N = N; a.K = K; a.bias = bias; a.aux = aux; a.out = out; a.ld_out = ld_out;
    a.tiles_n = N / BN; a.n_blocks = (M / BM) * (N / BN);
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case CPX_EPI_BF16: launch_gemm<CPX_EPI_BF16>(a, s); break;
        case CPX_EPI_GELU_BF16: launch_gemm<CPX_EPI_GELU_BF16>(a, s); break;
        case CPX_EPI_RESID_BF16: launch_gemm<CPX_EPI_RESID_BF16>(a, s); break;
        case CPX_EPI_F32: launch_gemm<CPX_EPI_F32>(a, s); break;
        case CPX_EPI_POS_BF16: launch_gemm<CPX_EPI_POS_BF16>(a, s); break;
        case CPX_EPI_RELU_BF16: launch_gemm<CPX_EPI_RELU_BF16>(a, s); break;
        default: CPX_REQUIRE(!"unknown epilogue");
    }
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}
