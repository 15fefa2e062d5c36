// a20  GrandQC tissue / artefact networks (UNet++ decoder on an EfficientNet-B0 encoder) on gfx950.
// The reference runs them in float32 through smp/timm (wsi_tissue_detection.py:86-160,
// wsi_artefact_detection.py:124-195); the class maps are argmax decisions, so the arithmetic stays
// float32 here too: dense convolutions (1x1, 3x3, stem) are implicit GEMMs on the exact-f32 matrix
// instruction v_mfma_f32_32x32x2_f32 with NHWC activations gathered tap by tap into LDS tiles,
// depthwise convolutions / squeeze-excite / argmax are per-pixel VALU kernels.  BatchNorm is folded
// into the weights by the host, nearest x2 upsampling and the UNet++ channel concatenations are
// index arithmetic in the gather (two sources: the low-resolution tensor and a channel slice of the
// level's concat buffer), the squeeze-excite gate is applied while staging the projection's input.
#include "cpx_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
#define QC_NONE ((size_t)-1)
#define QC_TP 128          // output pixels per workgroup
#define QC_CK 16           // channels per K chunk
#define QC_LDS 18          // LDS row stride in floats (16 + 2: both k-halves of a wave hit disjoint banks)

__device__ __forceinline__ float qc_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v / (1.f + __expf(-v));
    return v;
}

struct QcConvArgs {
    const float *a; const float *b; const float *gate; const float *w; const float *bias; const float *res;
    float *d;
    int ldA, Ca, upA, ldB, Cb, ldR, ldD, Cout;
    int nB, Hin, Win, Hout, Wout, k, stride, pad, act, Apad, Bpad;
};

// grid: (ceil(P / 128), CoutPad / TN); 256 threads = 4 waves, wave w owns pixels 32w..32w+31 x TN couts
template <int TN>
__global__ void __launch_bounds__(256) k_qc_conv(QcConvArgs g) {
    __shared__ float sA[QC_TP * QC_LDS];
    __shared__ float sW[TN * QC_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = g.nB * g.Hout * g.Wout;
    const int n0 = blockIdx.y * TN;
    const int Kc = g.Apad + g.Bpad, taps = g.k * g.k;

    // staging role for the activation tile: pixel pl, channel half hc (8 channels)
    const int pl = tid >> 1, hc = (tid & 1) * 8;
    const int p = blockIdx.x * QC_TP + pl;
    const bool pvalid = p < P;
    int n = 0, oy = 0, ox = 0;
    if (pvalid) { n = p / (g.Hout * g.Wout); const int r = p - n * g.Hout * g.Wout; oy = r / g.Wout; ox = r - oy * g.Wout; }
    const int HA = g.upA ? g.Hin >> 1 : g.Hin, WA = g.upA ? g.Win >> 1 : g.Win;
    // staging role for the weight tile: row wr (cout), 4 channels at wq
    const int wr = tid >> 2, wq = (tid & 3) * 4;

    f32x16 acc[TN / 32];
#pragma unroll
    for (int i = 0; i < TN / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    for (int tap = 0; tap < taps; ++tap) {
        const int ky = tap / g.k, kx = tap - ky * g.k;
        const int iy = oy * g.stride - g.pad + ky, ix = ox * g.stride - g.pad + kx;
        const bool inb = pvalid && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win;
        const float *pa = nullptr, *pb = nullptr;
        if (inb) {
            const int ya = g.upA ? iy >> 1 : iy, xa = g.upA ? ix >> 1 : ix;
            pa = g.a + ((size_t)(n * HA + ya) * WA + xa) * g.ldA;
            if (g.b) pb = g.b + ((size_t)(n * g.Hin + iy) * g.Win + ix) * g.ldB;
        }
        for (int kc = 0; kc < Kc; kc += QC_CK) {
            // ---- stage the activation chunk (zero outside the image / beyond the channel count)
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
            if (inb) {
                if (kc < g.Apad) {
                    const int c = kc + hc;
                    if (c < g.Ca) v0 = *reinterpret_cast<const float4 *>(pa + c);
                    if (c + 4 < g.Ca) v1 = *reinterpret_cast<const float4 *>(pa + c + 4);
                    if (g.gate) {
                        const float *gt = g.gate + (size_t)n * g.Ca + c;
                        if (c < g.Ca) { const float4 s = *reinterpret_cast<const float4 *>(gt); v0.x *= s.x; v0.y *= s.y; v0.z *= s.z; v0.w *= s.w; }
                        if (c + 4 < g.Ca) { const float4 s = *reinterpret_cast<const float4 *>(gt + 4); v1.x *= s.x; v1.y *= s.y; v1.z *= s.z; v1.w *= s.w; }
                    }
                } else {
                    const int c = kc - g.Apad + hc;
                    if (c < g.Cb) v0 = *reinterpret_cast<const float4 *>(pb + c);
                    if (c + 4 < g.Cb) v1 = *reinterpret_cast<const float4 *>(pb + c + 4);
                }
            }
            float *da = sA + pl * QC_LDS + hc;
            *reinterpret_cast<float2 *>(da) = make_float2(v0.x, v0.y);
            *reinterpret_cast<float2 *>(da + 2) = make_float2(v0.z, v0.w);
            *reinterpret_cast<float2 *>(da + 4) = make_float2(v1.x, v1.y);
            *reinterpret_cast<float2 *>(da + 6) = make_float2(v1.z, v1.w);
            // ---- stage the weight chunk (host layout is fully padded: no bounds checks)
            if (wr < TN) {
                const float4 wv = *reinterpret_cast<const float4 *>(g.w + ((size_t)(n0 + wr) * taps + tap) * Kc + kc + wq);
                float *dw = sW + wr * QC_LDS + wq;
                *reinterpret_cast<float2 *>(dw) = make_float2(wv.x, wv.y);
                *reinterpret_cast<float2 *>(dw + 2) = make_float2(wv.z, wv.w);
            }
            __syncthreads();
            const float *ra = sA + (wave * 32 + (lane & 31)) * QC_LDS + (lane >> 5);
            const float *rw = sW + (lane & 31) * QC_LDS + (lane >> 5);
#pragma unroll
            for (int kk = 0; kk < QC_CK / 2; ++kk) {
                const float av = ra[2 * kk];
#pragma unroll
                for (int nb = 0; nb < TN / 32; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, rw[nb * 32 * QC_LDS + 2 * kk], acc[nb], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    // ---- epilogue: accumulator register r of lane l = pixel row (r/4)*8 + (l/32)*4 + r%4, cout l%32
#pragma unroll
    for (int nb = 0; nb < TN / 32; ++nb) {
        const int co = n0 + nb * 32 + (lane & 31);
        if (co >= g.Cout) continue;
        const float bv = g.bias ? g.bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = blockIdx.x * QC_TP + wave * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
            if (q >= P) continue;
            float v = qc_act(acc[nb][r] + bv, g.act);
            if (g.res) v += g.res[(size_t)q * g.ldR + co];
            g.d[(size_t)q * g.ldD + co] = v;
        }
    }
}

// depthwise k x k convolution, NHWC, 4 channels per thread; weights [taps][C]
__global__ void k_qc_dw(const float *__restrict__ x, int ldX, const float *__restrict__ w, const float *__restrict__ bias,
                        float *__restrict__ y, int C, int nB, int Hin, int Win, int Hout, int Wout, int k,
                        int stride, int pad, int act) {
    const int c4n = C >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nB * Hout * Wout * c4n) return;
    const int c = (int)(idx % c4n) * 4;
    const long long p = idx / c4n;
    const int n = (int)(p / (Hout * Wout)), r = (int)(p - (long long)n * Hout * Wout);
    const int oy = r / Wout, ox = r - oy * Wout;
    float4 acc = *reinterpret_cast<const float4 *>(bias + c);
    for (int ky = 0; ky < k; ++ky) {
        const int iy = oy * stride - pad + ky;
        if (iy < 0 || iy >= Hin) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int ix = ox * stride - pad + kx;
            if (ix < 0 || ix >= Win) continue;
            const float4 v = *reinterpret_cast<const float4 *>(x + ((size_t)(n * Hin + iy) * Win + ix) * ldX + c);
            const float4 ww = *reinterpret_cast<const float4 *>(w + (size_t)(ky * k + kx) * C + c);
            acc.x = fmaf(v.x, ww.x, acc.x); acc.y = fmaf(v.y, ww.y, acc.y);
            acc.z = fmaf(v.z, ww.z, acc.z); acc.w = fmaf(v.w, ww.w, acc.w);
        }
    }
    acc.x = qc_act(acc.x, act); acc.y = qc_act(acc.y, act); acc.z = qc_act(acc.z, act); acc.w = qc_act(acc.w, act);
    *reinterpret_cast<float4 *>(y + (size_t)p * C + c) = acc;
}

// squeeze: per-image channel sums over HW; grid (C/64, nB, slices), 256 threads = 4 pixel lanes x 64 channels;
// every slice writes its own partial row (summed in slice order by k_qc_se: deterministic, no atomics)
__global__ void k_qc_pool(const float *__restrict__ x, int ldX, int C, int HW, float *__restrict__ sums) {
    __shared__ float sh[256];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g4 = threadIdx.x >> 6, n = blockIdx.y;
    const int per = (HW + gridDim.z - 1) / gridDim.z, p0 = blockIdx.z * per, p1 = min(HW, p0 + per);
    float s = 0.f;
    if (c < C)
        for (int p = p0 + g4; p < p1; p += 4) s += x[((size_t)n * HW + p) * ldX + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (g4 == 0 && c < C)
        sums[((size_t)blockIdx.z * gridDim.y + n) * C + c] = sh[threadIdx.x] + sh[threadIdx.x + 64] + sh[threadIdx.x + 128] + sh[threadIdx.x + 192];
}

// excite: gate[n][c] = sigmoid(W2 silu(W1 mean + b1) + b2); one block per image
__global__ void k_qc_se(const float *__restrict__ sums, int slices, float inv_hw, const float *__restrict__ w1,
                        const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2, int C,
                        int Cr, float *__restrict__ gate) {
    extern __shared__ float sm[];          // [C] means, [Cr] reduced
    float *mean = sm, *red = sm + C;
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < slices; ++z) s += sums[((size_t)z * gridDim.x + n) * C + c];
        mean[c] = s * inv_hw;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int j = wave; j < Cr; j += nw) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = fmaf(w1[(size_t)j * C + c], mean[c], s);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) { s += b1[j]; red[j] = s / (1.f + __expf(-s)); }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = b2[c];
        for (int j = 0; j < Cr; ++j) s = fmaf(w2[(size_t)c * Cr + j], red[j], s);
        gate[(size_t)n * C + c] = 1.f / (1.f + __expf(-s));
    }
}

// get_preprocessing (wsi_qc_helpers.py:104-120): ((x / 255) - mean) / std in float64, cast to float32; RGB0
__global__ void k_qc_pre(const uint8_t *__restrict__ u8, long long npix, float *__restrict__ out) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const double mean[3] = {0.485, 0.456, 0.406}, sd[3] = {0.229, 0.224, 0.225};
    float4 o;
    o.x = (float)__ddiv_rn(__dsub_rn(__ddiv_rn((double)u8[p * 3 + 0], 255.0), mean[0]), sd[0]);
    o.y = (float)__ddiv_rn(__dsub_rn(__ddiv_rn((double)u8[p * 3 + 1], 255.0), mean[1]), sd[1]);
    o.z = (float)__ddiv_rn(__dsub_rn(__ddiv_rn((double)u8[p * 3 + 2], 255.0), mean[2]), sd[2]);
    o.w = 0.f;
    *reinterpret_cast<float4 *>(out + p * 4) = o;
}

// np.argmax over the class axis (first maximum wins) -> int8 map; optional compact logits copy
__global__ void k_qc_argmax(const float *__restrict__ logits, int ld, int ncls, long long npix, int8_t *__restrict__ cls,
                            float *__restrict__ logits_out) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float *l = logits + p * ld;
    int best = 0;
    float bv = l[0];
    for (int c = 1; c < ncls; ++c) { const float v = l[c]; if (v > bv) { bv = v; best = c; } }
    cls[p] = (int8_t)best;
    if (logits_out) for (int c = 0; c < ncls; ++c) logits_out[p * ncls + c] = l[c];
}

static inline int qc_round_up(int x, int a) { return (x + a - 1) / a * a; }

extern "C" int cpx_qc_forward(const cpx_qc_op *ops, int n_ops, const uint8_t *patches_u8, int nB, int H, int W,
                              size_t input_off, size_t logits_off, int n_classes, int ld_logits,
                              int8_t *class_map, float *logits_out, void *workspace, size_t workspace_bytes,
                              void *stream) {
    CPX_REQUIRE(ops && n_ops > 0 && patches_u8 && nB > 0 && class_map && workspace);
    CPX_REQUIRE(H % 32 == 0 && W % 32 == 0 && n_classes > 0 && n_classes <= ld_logits);
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    auto at = [&](size_t off) -> float * { return off == QC_NONE ? nullptr : (float *)(ws + off); };
    const long long npix = (long long)nB * H * W;
    CPX_REQUIRE(input_off + (size_t)npix * 16 <= workspace_bytes);
    CPX_REQUIRE(logits_off + (size_t)npix * ld_logits * 4 <= workspace_bytes);
    hipLaunchKernelGGL(k_qc_pre, dim3(cpx_cdiv(npix, 256)), dim3(256), 0, s, patches_u8, npix, at(input_off));
    CPX_CHECK_LAUNCH();
    for (int i = 0; i < n_ops; ++i) {
        const cpx_qc_op &o = ops[i];
        const long long P = (long long)nB * o.h_out * o.w_out;
        CPX_REQUIRE(o.src_a != QC_NONE && o.dst != QC_NONE && o.w);
        if (o.kind == 0) {
            CPX_REQUIRE(o.c_a % 4 == 0 && o.c_b % 4 == 0 && o.ld_a % 4 == 0 && o.ld_b % 4 == 0 && o.c_out > 0);
            CPX_REQUIRE(!o.up_a || (o.h_in % 2 == 0 && o.w_in % 2 == 0));
            CPX_REQUIRE(o.dst + (size_t)P * o.ld_dst * 4 <= workspace_bytes);
            QcConvArgs g;
            g.a = at(o.src_a); g.b = at(o.src_b); g.gate = at(o.gate); g.w = o.w; g.bias = o.bias; g.res = at(o.res);
            g.d = at(o.dst);
            g.ldA = o.ld_a; g.Ca = o.c_a; g.upA = o.up_a; g.ldB = o.ld_b; g.Cb = g.b ? o.c_b : 0; g.ldR = o.ld_res;
            g.ldD = o.ld_dst; g.Cout = o.c_out; g.nB = nB; g.Hin = o.h_in; g.Win = o.w_in; g.Hout = o.h_out;
            g.Wout = o.w_out; g.k = o.k; g.stride = o.stride; g.pad = o.pad; g.act = o.act;
            g.Apad = qc_round_up(o.c_a, QC_CK); g.Bpad = g.b ? qc_round_up(o.c_b, QC_CK) : 0;
            const int tn = o.c_out <= 32 ? 32 : 64, cpad = qc_round_up(o.c_out, tn);
            dim3 grid(cpx_cdiv(P, QC_TP), cpad / tn);
            if (tn == 32) hipLaunchKernelGGL(k_qc_conv<32>, grid, dim3(256), 0, s, g);
            else hipLaunchKernelGGL(k_qc_conv<64>, grid, dim3(256), 0, s, g);
        } else if (o.kind == 1) {
            CPX_REQUIRE(o.c_a % 4 == 0 && o.ld_a % 4 == 0 && o.bias);
            CPX_REQUIRE(o.dst + (size_t)P * o.c_a * 4 <= workspace_bytes);
            hipLaunchKernelGGL(k_qc_dw, dim3(cpx_cdiv(P * (o.c_a / 4), 256)), dim3(256), 0, s, at(o.src_a), o.ld_a,
                               o.w, o.bias, at(o.dst), o.c_a, nB, o.h_in, o.w_in, o.h_out, o.w_out, o.k, o.stride,
                               o.pad, o.act);
        } else if (o.kind == 2) {
            CPX_REQUIRE(o.res != QC_NONE && o.w2 && o.bias && o.bias2 && o.c_red > 0);
            const int HW = o.h_in * o.w_in;
            const int slices = HW >= 16384 ? 16 : (HW >= 1024 ? 4 : 1);        /* scratch: 16 * nB * C floats */
            CPX_REQUIRE(o.res + (size_t)16 * nB * o.c_a * 4 <= workspace_bytes);
            hipLaunchKernelGGL(k_qc_pool, dim3(cpx_cdiv(o.c_a, 64), nB, slices), dim3(256), 0, s, at(o.src_a), o.ld_a,
                               o.c_a, HW, at(o.res));
            CPX_CHECK_LAUNCH();
            hipLaunchKernelGGL(k_qc_se, dim3(nB), dim3(256), (size_t)(o.c_a + o.c_red) * 4, s, at(o.res), slices, 1.0f / HW,
                               o.w, o.bias, o.w2, o.bias2, o.c_a, o.c_red, at(o.dst));
        } else {
            CPX_REQUIRE(!"unknown cpx_qc_op kind");
        }
        CPX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_qc_argmax, dim3(cpx_cdiv(npix, 256)), dim3(256), 0, s, at(logits_off), ld_logits, n_classes,
                       npix, class_map, logits_out);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}
