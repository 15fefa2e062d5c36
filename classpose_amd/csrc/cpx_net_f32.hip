// ClassTransformer forward in FLOAT32 on gfx950 (--precision fp32, the precision every reference
// integration test passes: /root/reference/tests/test_prediction_integration.py:67,113,159,199;
// resolve_precision, /root/reference/src/classpose/models.py:37-69).
//
// gfx950 has no TF32/xf32 path; its f32-input matrix instruction v_mfma_f32_32x32x2_f32 is EXACT
// float32 (products and sums are IEEE f32 FMAs) at the f32 vector rate (157 TFLOP/s dense), so the
// fp32 network keeps float32 weights, activations and accumulation end to end -- no split-bf16
// emulation, no reduced-precision shortcut.  Same token-major layouts as the half-precision path:
//   * k_gemm_f32     C[M][N] = A[M][K] W[N][K]^T, 128 x 128 x 16 tiles, 4 waves of 64 x 64 (2 x 2 MFMA
//                    blocks), register-staged double buffering, 20-float LDS rows (conflict-free b128);
//   * k_layernorm_f32 two-pass mean / variance like torch.nn.functional.layer_norm;
//   * k_attention_f32 flash attention with the decomposed rel-pos bias (flash_forward, vit_sam.py:15-65):
//                    one wave = one image row of 32 queries, key tiles = image rows, S^T = K Q^T with the
//                    Gw bias as the initial accumulator, P^T fed to the P.V MFMA straight from the
//                    accumulator registers (the k-slot of step j IS accumulator register j).
// The MFMA k index is free to be permuted identically on both operands: lane half h2 takes
// k = 8*h2 + s (GEMM) / d = 32*h2 + s (attention) so that fragments are 16-byte LDS / global reads.
#include "cpx_internal.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------
#define F_BM 128
#define F_BN 128
#define F_BK 16
#define F_LD 20                         // LDS row stride in floats: rows r, r+1 are 80 B apart -> b128 reads of
#define F_TILE (128 * F_LD)             // 16 lanes with distinct (row mod 16) cover all 64 banks exactly once
enum { FE_NONE = 0, FE_GELU = 1, FE_RESID = 2, FE_POS = 3, FE_RELU = 4 };

struct GemmF32Args {
    const float *A, *W;
    int M, N, K;
    const float *bias, *aux;
    float *out;
    int ld_out, tiles_n, n_blocks;
};

template <int EPI>
__global__ void __launch_bounds__(256, 2) k_gemm_f32(GemmF32Args g) {
    __shared__ __attribute__((aligned(16))) float sm[2][2][F_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h2 = lane >> 5;
    int bid = blockIdx.x;
    {   // blocks that share an XCD (id % 8) walk a contiguous band of tiles
        const int nxcd = 8, q = g.n_blocks / nxcd, rr = g.n_blocks % nxcd, x = bid % nxcd;
        bid = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + bid / nxcd;
    }
    const int tile_m = bid / g.tiles_n, tile_n = bid - tile_m * g.tiles_n;
    const int m0 = tile_m * F_BM, n0 = tile_n * F_BN, K = g.K, nk = K / F_BK;

    const int srow = tid >> 2, sc4 = (tid & 3) * 4;
    const float *pa0 = g.A + (size_t)(m0 + srow) * K + sc4, *pa1 = pa0 + (size_t)64 * K;
    const float *pw0 = g.W + (size_t)(n0 + srow) * K + sc4, *pw1 = pw0 + (size_t)64 * K;
    const int so0 = srow * F_LD + sc4, so1 = so0 + 64 * F_LD;
    float4 ra0, ra1, rw0, rw1;
    auto load = [&](int k0) {
        ra0 = *reinterpret_cast<const float4 *>(pa0 + k0); ra1 = *reinterpret_cast<const float4 *>(pa1 + k0);
        rw0 = *reinterpret_cast<const float4 *>(pw0 + k0); rw1 = *reinterpret_cast<const float4 *>(pw1 + k0);
    };
    auto store = [&](int buf) {
        *reinterpret_cast<float4 *>(&sm[buf][0][so0]) = ra0; *reinterpret_cast<float4 *>(&sm[buf][0][so1]) = ra1;
        *reinterpret_cast<float4 *>(&sm[buf][1][so0]) = rw0; *reinterpret_cast<float4 *>(&sm[buf][1][so1]) = rw1;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    const int aoff = (wm * 64 + r) * F_LD + 8 * h2, woff = (wn * 64 + r) * F_LD + 8 * h2;
    load(0);
    store(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) load((t + 1) * F_BK);
        const float *sa = sm[t & 1][0], *sw = sm[t & 1][1];
        float fa[2][8], fw[2][8];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float4 a0 = *reinterpret_cast<const float4 *>(sa + aoff + b * 32 * F_LD);
            const float4 a1 = *reinterpret_cast<const float4 *>(sa + aoff + b * 32 * F_LD + 4);
            const float4 w0 = *reinterpret_cast<const float4 *>(sw + woff + b * 32 * F_LD);
            const float4 w1 = *reinterpret_cast<const float4 *>(sw + woff + b * 32 * F_LD + 4);
            fa[b][0] = a0.x; fa[b][1] = a0.y; fa[b][2] = a0.z; fa[b][3] = a0.w;
            fa[b][4] = a1.x; fa[b][5] = a1.y; fa[b][6] = a1.z; fa[b][7] = a1.w;
            fw[b][0] = w0.x; fw[b][1] = w0.y; fw[b][2] = w0.z; fw[b][3] = w0.w;
            fw[b][4] = w1.x; fw[b][5] = w1.y; fw[b][6] = w1.z; fw[b][7] = w1.w;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = MFMA_F32(fa[mb][s], fw[nb][s], acc[mb][nb]);
        if (t + 1 < nk) store((t + 1) & 1);
        __syncthreads();
    }
    // accumulator register v of lane (r, h2): token row (v&3) + 8*(v>>2) + 4*h2, channel r
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int ch = n0 + wn * 64 + nb * 32 + r;
        const float bv = g.bias ? g.bias[ch] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = m0 + wm * 64 + mb * 32 + (v & 3) + 8 * (v >> 2) + 4 * h2;
                float x = acc[mb][nb][v] + bv;
                if constexpr (EPI == FE_GELU) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
                else if constexpr (EPI == FE_RELU) x = fmaxf(x, 0.f);
                else if constexpr (EPI == FE_RESID) x += g.aux[(size_t)row * g.ld_out + ch];
                else if constexpr (EPI == FE_POS) x += g.aux[(size_t)(row & 1023) * g.N + ch];
                g.out[(size_t)row * g.ld_out + ch] = x;
            }
    }
}

int cpx_gemm_f32(const float *A, const float *Wt, int M, int N, int K, int epilogue, const float *bias,
                 const float *aux, float *out, int ld_out, void *stream) {
    CPX_REQUIRE(A && Wt && out && M > 0 && N > 0 && K > 0);
    CPX_REQUIRE(M % F_BM == 0 && N % F_BN == 0 && K % F_BK == 0 && ld_out >= N);
    CPX_REQUIRE((epilogue != CPX_EPI_RESID_BF16 && epilogue != CPX_EPI_POS_BF16) || aux);
    GemmF32Args a;
    a.A = A; a.W = Wt; a.M = M; a.N = N; a.K = K; a.bias = bias; a.aux = aux; a.out = out; a.ld_out = ld_out;
    a.tiles_n = N / F_BN; a.n_blocks = (M / F_BM) * (N / F_BN);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(a.n_blocks), block(256);
    switch (epilogue) {
        case CPX_EPI_BF16: case CPX_EPI_F32: case CPX_EPI_QKV_BF16:
            hipLaunchKernelGGL(k_gemm_f32<FE_NONE>, grid, block, 0, s, a); break;
        case CPX_EPI_GELU_BF16: hipLaunchKernelGGL(k_gemm_f32<FE_GELU>, grid, block, 0, s, a); break;
        case CPX_EPI_RESID_BF16: hipLaunchKernelGGL(k_gemm_f32<FE_RESID>, grid, block, 0, s, a); break;
        case CPX_EPI_POS_BF16: hipLaunchKernelGGL(k_gemm_f32<FE_POS>, grid, block, 0, s, a); break;
        case CPX_EPI_RELU_BF16: hipLaunchKernelGGL(k_gemm_f32<FE_RELU>, grid, block, 0, s, a); break;
        default: CPX_REQUIRE(!"unknown epilogue");
    }
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// ---------------------------------------------------------------------------
// LayerNorm over the last dim (C = 1024 or 256), one wave per row, two-pass statistics
// ---------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) k_layernorm_f32(const float *__restrict__ x, const float *__restrict__ w,
                                                       const float *__restrict__ b, int rows, float eps,
                                                       float *__restrict__ out) {
    constexpr int PER = C / 64;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (size_t)row * C + lane * PER;
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; i += 4) {
        const float4 a = *reinterpret_cast<const float4 *>(xr + i);
        v[i] = a.x; v[i + 1] = a.y; v[i + 2] = a.z; v[i + 3] = a.w;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float d = v[i] - mean; q += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q * (1.0f / C) + eps);
    float *orow = out + (size_t)row * C + lane * PER;
#pragma unroll
    for (int i = 0; i < PER; i += 4) {
        const float4 ww = *reinterpret_cast<const float4 *>(w + lane * PER + i);
        const float4 bb = *reinterpret_cast<const float4 *>(b + lane * PER + i);
        float4 o;
        o.x = (v[i] - mean) * rstd * ww.x + bb.x; o.y = (v[i + 1] - mean) * rstd * ww.y + bb.y;
        o.z = (v[i + 2] - mean) * rstd * ww.z + bb.z; o.w = (v[i + 3] - mean) * rstd * ww.w + bb.w;
        *reinterpret_cast<float4 *>(orow + i) = o;
    }
}

int cpx_layernorm_f32(const float *x, const float *w, const float *b, int rows, int C, float eps, float *out,
                      void *stream) {
    CPX_REQUIRE(x && w && b && out && rows > 0 && (C == 1024 || C == 256));
    dim3 grid(cpx_cdiv(rows, 4)), block(256);
    if (C == 1024) hipLaunchKernelGGL(k_layernorm_f32<1024>, grid, block, 0, (hipStream_t)stream, x, w, b, rows, eps, out);
    else hipLaunchKernelGGL(k_layernorm_f32<256>, grid, block, 0, (hipStream_t)stream, x, w, b, rows, eps, out);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// ---------------------------------------------------------------------------
// flash attention with decomposed rel-pos bias, float32
// ---------------------------------------------------------------------------
#define AF_LD 68                       // K / V tile row stride in floats (b128 reads of 16 rows: 64 distinct banks)
#define AF_GLD 65
__global__ void __launch_bounds__(256) k_attention_f32(const float *__restrict__ qkv, const float *__restrict__ relh,
                                                       const float *__restrict__ relw, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float sK[2][32 * AF_LD];
    __shared__ __attribute__((aligned(16))) float sV[2][32 * AF_LD];
    __shared__ float sG[4][32 * AF_GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h2 = lane >> 5;
    // the 8 row groups of one (sub-tile, head) stream the same K / V rows: keep them on one XCD's L2
    const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int jj = lin >> 3, pair = (jj >> 3) * 8 + (lin & 7);
    const int rg = jj & 7, head = pair & 15, s = pair >> 4;
    const int qh = rg * 4 + wave;
    const size_t tok0 = (size_t)s * 1024;

    // Q fragment: lane (q = r, h2) holds Q[q][32*h2 + 0..31]
    float qf[32];
    {
        const float *qrow = qkv + (tok0 + qh * 32 + r) * 3072 + head * 64 + 32 * h2;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 a = *reinterpret_cast<const float4 *>(qrow + 4 * c);
            qf[4 * c] = a.x; qf[4 * c + 1] = a.y; qf[4 * c + 2] = a.z; qf[4 * c + 3] = a.w;
        }
    }
    // G = Q . table^T (tables hold rel_pos / scale, row 63 = 0) -> LDS scratch [q][j]
    float *G = sG[wave];
    auto compute_G = [&](const float *table) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            f32x16 acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
            const float *trow = table + (jb * 32 + r) * 64 + 32 * h2;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 t4 = *reinterpret_cast<const float4 *>(trow + 4 * c);
                acc = MFMA_F32(t4.x, qf[4 * c], acc); acc = MFMA_F32(t4.y, qf[4 * c + 1], acc);
                acc = MFMA_F32(t4.z, qf[4 * c + 2], acc); acc = MFMA_F32(t4.w, qf[4 * c + 3], acc);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) G[r * AF_GLD + jb * 32 + (v & 3) + 8 * (v >> 2) + 4 * h2] = acc[v];
        }
    };
    compute_G(relw);
    f32x16 GW;
#pragma unroll
    for (int v = 0; v < 16; ++v) GW[v] = G[r * AF_GLD + (r - ((v & 3) + 8 * (v >> 2) + 4 * h2) + 31)];
    compute_G(relh);

    // K / V tiles: 32 keys x 64 d, thread -> (key = tid>>4 (+16), 4 d at (tid&15)*4)
    const int skey = tid >> 4, sc4 = (tid & 15) * 4;
    const float *kbase = qkv + (tok0 + skey) * 3072 + 1024 + head * 64 + sc4;
    const float *vbase = kbase + 1024;
    const int sdst0 = skey * AF_LD + sc4, sdst1 = sdst0 + 16 * AF_LD;
    float4 k0r, k1r, v0r, v1r;
    auto load = [&](int kh) {
        const size_t o = (size_t)kh * 32 * 3072;
        k0r = *reinterpret_cast<const float4 *>(kbase + o); k1r = *reinterpret_cast<const float4 *>(kbase + o + (size_t)16 * 3072);
        v0r = *reinterpret_cast<const float4 *>(vbase + o); v1r = *reinterpret_cast<const float4 *>(vbase + o + (size_t)16 * 3072);
    };
    auto store = [&](int buf) {
        *reinterpret_cast<float4 *>(&sK[buf][sdst0]) = k0r; *reinterpret_cast<float4 *>(&sK[buf][sdst1]) = k1r;
        *reinterpret_cast<float4 *>(&sV[buf][sdst0]) = v0r; *reinterpret_cast<float4 *>(&sV[buf][sdst1]) = v1r;
    };
    load(0);
    store(0);
    __syncthreads();

    f32x16 O[2];
#pragma unroll
    for (int v = 0; v < 16; ++v) { O[0][v] = 0.f; O[1][v] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;
    const float cexp = 0.125f * 1.44269504088896340736f;      // softmax scale (64^-0.5) * log2(e)
    for (int kh = 0; kh < 32; ++kh) {
        const int buf = kh & 1;
        if (kh + 1 < 32) load(kh + 1);
        const float gh = G[r * AF_GLD + (qh - kh + 31)];
        // S'^T = K . Q^T + Gw  (lane: query r, keys (v&3) + 8*(v>>2) + 4*h2)
        f32x16 S = GW;
        const float *kp = &sK[buf][r * AF_LD + 32 * h2];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 kf = *reinterpret_cast<const float4 *>(kp + 4 * c);
            S = MFMA_F32(kf.x, qf[4 * c], S); S = MFMA_F32(kf.y, qf[4 * c + 1], S);
            S = MFMA_F32(kf.z, qf[4 * c + 2], S); S = MFMA_F32(kf.w, qf[4 * c + 3], S);
        }
        float mx = S[0];
#pragma unroll
        for (int v = 1; v < 16; ++v) mx = fmaxf(mx, S[v]);
        mx = fmaxf(mx, __shfl_xor(mx, 32)) + gh;
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cexp);
        m_run = m_new;
        const float off = (gh - m_new) * cexp;
        float p[16], ps = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) { p[v] = __builtin_amdgcn_exp2f(fmaf(S[v], cexp, off)); ps += p[v]; }
        l_run = fmaf(l_run, alpha, ps);
#pragma unroll
        for (int v = 0; v < 16; ++v) { O[0][v] *= alpha; O[1][v] *= alpha; }
        // O^T[d][q] += V^T[d][key] P^T[key][q]: the k-slot of step j is the key of accumulator register j
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float *vp = &sV[buf][((j & 3) + 8 * (j >> 2) + 4 * h2) * AF_LD + r];
            O[0] = MFMA_F32(vp[0], p[j], O[0]);
            O[1] = MFMA_F32(vp[32], p[j], O[1]);
        }
        if (kh + 1 < 32) store(buf ^ 1);
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    float *orow = out + (tok0 + qh * 32 + r) * 1024 + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = db * 32 + 8 * g4 + 4 * h2;
            *reinterpret_cast<float4 *>(orow + d) = make_float4(O[db][4 * g4] * inv, O[db][4 * g4 + 1] * inv,
                                                                O[db][4 * g4 + 2] * inv, O[db][4 * g4 + 3] * inv);
        }
}

int cpx_attention_f32(const float *qkv, const float *rel_h, const float *rel_w, int n_subtiles, float *out, void *stream) {
    CPX_REQUIRE(qkv && rel_h && rel_w && out && n_subtiles > 0);
    hipLaunchKernelGGL(k_attention_f32, dim3(8, 16, n_subtiles), dim3(256), 0, (hipStream_t)stream, qkv, rel_h, rel_w, out);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// im2col for the neck's 3x3 conv: [S*1024][256] -> [S*1024][9*256], k = tap*256 + c, 4 floats per thread
__global__ void __launch_bounds__(256) k_im2col3_f32(const float *__restrict__ x, size_t n_chunks, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_chunks) return;
    const size_t row = i / 576;
    const int rem = (int)(i - row * 576), tap = rem >> 6, c4 = rem & 63;
    const int tok = (int)(row & 1023), ph = tok >> 5, pw = tok & 31;
    const int yy = ph + tap / 3 - 1, xx = pw + tap % 3 - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)yy < 32u && (unsigned)xx < 32u)
        v = *reinterpret_cast<const float4 *>(x + ((row & ~(size_t)1023) + yy * 32 + xx) * 256 + c4 * 4);
    *reinterpret_cast<float4 *>(out + row * 2304 + tap * 256 + c4 * 4) = v;
}

// ---------------------------------------------------------------------------
// forward driver (ClassTransformer.forward, vit_sam.py:148-197)
// ---------------------------------------------------------------------------
struct NetWsF32 { size_t off_x, off_xn, off_qkv, off_ao, off_h, off_neck, off_neck2, off_col, total; };
static NetWsF32 net_ws_f32(int nS) {
    NetWsF32 w; size_t o = 0; const size_t M = (size_t)nS * 1024;
    auto take = [&](size_t b) { size_t r = o; o = cpx_align_up(o + b, 256); return r; };
    w.off_x = take(M * 1024 * 4); w.off_xn = take(M * 1024 * 4); w.off_qkv = take(M * 3072 * 4);
    w.off_ao = take(M * 1024 * 4); w.off_h = take(M * 4096 * 4); w.off_neck = take(M * 256 * 4);
    w.off_neck2 = take(M * 256 * 4); w.off_col = take(M * 2304 * 4);
    w.total = o;
    return w;
}
size_t cpx_net_f32_workspace_bytes(int n_subtiles) { return n_subtiles > 0 ? net_ws_f32(n_subtiles).total : 0; }

int cpx_net_forward_f32(const cpx_net_weights *w, const void *patches, int nS, float *head, void *workspace,
                        size_t workspace_bytes, void *stream) {
    const NetWsF32 L = net_ws_f32(nS);
    CPX_REQUIRE(workspace_bytes >= L.total && !w->fuse_ln);
    char *ws = (char *)workspace;
    const int M = nS * 1024;
    float *x = (float *)(ws + L.off_x), *xn = (float *)(ws + L.off_xn), *qkv = (float *)(ws + L.off_qkv),
          *ao = (float *)(ws + L.off_ao), *hb = (float *)(ws + L.off_h), *nk = (float *)(ws + L.off_neck),
          *nk2 = (float *)(ws + L.off_neck2), *col = (float *)(ws + L.off_col);
    CpxProf *prof = (CpxProf *)w->prof;
    hipStream_t hs = (hipStream_t)stream;
    int rc;
#define RUN(call) do { rc = (call); if (rc) return rc; } while (0)
#define F(p) ((const float *)(p))
    RUN(cpx_gemm_f32(F(patches), F(w->pe_w), M, 1024, 192, CPX_EPI_POS_BF16, w->pe_b, w->pos, x, 1024, stream));
    for (int i = 0; i < w->depth; ++i) {
        const cpx_block_weights &b = w->blocks[i];
        RUN(cpx_layernorm_f32(x, b.ln1_w, b.ln1_b, M, 1024, 1e-6f, xn, stream));
        RUN(cpx_gemm_f32(xn, F(b.qkv_w), M, 3072, 1024, CPX_EPI_F32, b.qkv_b, nullptr, qkv, 3072, stream));
        bool t = cpx_prof_begin(prof, CPX_PROF_ATTN, i, hs);
        RUN(cpx_attention_f32(qkv, F(b.rel_h), F(b.rel_w), nS, ao, stream));
        if (t) cpx_prof_end(prof, hs);
        RUN(cpx_gemm_f32(ao, F(b.proj_w), M, 1024, 1024, CPX_EPI_RESID_BF16, b.proj_b, x, x, 1024, stream));
        RUN(cpx_layernorm_f32(x, b.ln2_w, b.ln2_b, M, 1024, 1e-6f, xn, stream));
        t = cpx_prof_begin(prof, CPX_PROF_FC1, i, hs);
        RUN(cpx_gemm_f32(xn, F(b.fc1_w), M, 4096, 1024, CPX_EPI_GELU_BF16, b.fc1_b, nullptr, hb, 4096, stream));
        if (t) cpx_prof_end(prof, hs);
        RUN(cpx_gemm_f32(hb, F(b.fc2_w), M, 1024, 4096, CPX_EPI_RESID_BF16, b.fc2_b, x, x, 1024, stream));
    }
    RUN(cpx_gemm_f32(x, F(w->neck0_w), M, 256, 1024, CPX_EPI_F32, nullptr, nullptr, nk, 256, stream));
    RUN(cpx_layernorm_f32(nk, w->neck_ln1_w, w->neck_ln1_b, M, 256, 1e-6f, nk2, stream));
    {
        const size_t n_chunks = (size_t)M * 576;
        hipLaunchKernelGGL(k_im2col3_f32, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, hs, nk2, n_chunks, col);
        CPX_CHECK_LAUNCH();
    }
    RUN(cpx_gemm_f32(col, F(w->neck2_w), M, 256, 2304, CPX_EPI_F32, nullptr, nullptr, nk, 256, stream));
    RUN(cpx_layernorm_f32(nk, w->neck_ln2_w, w->neck_ln2_b, M, 256, 1e-6f, nk2, stream));
    RUN(cpx_gemm_f32(nk2, F(w->head_w), M, w->ld_head, 256, CPX_EPI_F32, w->head_b, nullptr, head, w->ld_head, stream));
    if (w->n_unet_ops > 0) {
        const size_t need = cpx_unet_ws_bytes(CPX_DT_F32, w->unet_ops, w->n_unet_ops, nS);
        CPX_REQUIRE(workspace_bytes >= L.total + need);
        RUN(cpx_unet_head_run(CPX_DT_F32, w->unet_ops, w->n_unet_ops, nk2, nS, head, w->ld_head, 192, ws + L.total, need, stream));
    }
#undef F
#undef RUN
    return CPX_OK;
}
