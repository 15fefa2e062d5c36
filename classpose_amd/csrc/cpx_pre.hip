// Tile ingest and network I/O reshaping on gfx950:
//   a6  percentile normalisation of uint8 tiles (exact: 256-bin histogram)
//   a7  zero pad -> overlapping sub-tiles (+TTA flips) -> patch rows (bf16)
//   a7  pixel shuffle + un-augment + taper-weighted blend + crop
// Arithmetic follows oracle/tiling.py (numpy float32 / float64 op order); built
// with -ffp-contract=off.
#include "cpx_common.h"

#define NTHR 256

// ---------------------------------------------------------------------------
// a3/f4 residual mpp rescale: cv2.resize(tile, (dw, dh), INTER_LINEAR) for 8UC3
// ---------------------------------------------------------------------------
// OpenCV's classic fixed-point bilinear (imgproc resize.cpp: resizeGeneric_ with
// HResizeLinear<uchar,int,short,2048> + VResizeLinear<uchar,int,short,FixedPtCast>):
// taps and 11-bit weights come from the float32 source coordinate
// (dx+0.5)*scale-0.5, columns beyond the last source column collapse to one tap,
// rows clamp; an exact 2x2 decimation is re-dispatched to INTER_AREA
// ((a+b+c+d+2)>>2).  One thread per output pixel, 3 channels.
__device__ __forceinline__ void cv_linear_tap(int d, double scale, int slen, bool clamp_weight,
                                              int &s0, int &s1, int &w0, int &w1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (clamp_weight) {             // horizontal: xmin / xmax handling zeroes the fraction
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= slen - 1) { f = 0.f; s = slen - 1; }
    }
    w0 = __float2int_rn((1.f - f) * 2048.f);
    w1 = __float2int_rn(f * 2048.f);
    s0 = min(max(s, 0), slen - 1);
    s1 = min(max(s + 1, 0), slen - 1);
}

__global__ void k_resize_linear_u8(const uint8_t *__restrict__ src, int sh, int sw,
                                   uint8_t *__restrict__ dst, int dh, int dw, double scale_x,
                                   double scale_y, int area2) {
    int p = blockIdx.x * NTHR + threadIdx.x;
    if (p >= dh * dw) return;
    int dy = p / dw, dx = p - dy * dw;
    const uint8_t *s = src + (size_t)blockIdx.y * sh * sw * 3;
    uint8_t *o = dst + ((size_t)blockIdx.y * dh * dw + p) * 3;
    if (area2) {
        const uint8_t *r0 = s + ((size_t)(2 * dy) * sw + 2 * dx) * 3, *r1 = r0 + (size_t)sw * 3;
        for (int c = 0; c < 3; ++c) o[c] = (uint8_t)((r0[c] + r0[3 + c] + r1[c] + r1[3 + c] + 2) >> 2);
        return;
    }
    int x0, x1, a0, a1, y0, y1, b0, b1;
    cv_linear_tap(dx, scale_x, sw, true, x0, x1, a0, a1);
    cv_linear_tap(dy, scale_y, sh, false, y0, y1, b0, b1);
    const uint8_t *r0 = s + (size_t)y0 * sw * 3, *r1 = s + (size_t)y1 * sw * 3;
    for (int c = 0; c < 3; ++c) {
        int h0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
        int h1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
        o[c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
}

extern "C" int cpx_resize_linear_u8(const uint8_t *src, int nT, int sh, int sw, uint8_t *dst, int dh,
                                    int dw, void *stream) {
    CPX_REQUIRE(src && dst && nT > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0);
    CPX_REQUIRE((long long)dh * dw < (1ll << 31) && (long long)sh * sw < (1ll << 31));
    hipStream_t s = (hipStream_t)stream;
    if (sh == dh && sw == dw) {
        hipError_t e = hipMemcpyAsync(dst, src, (size_t)nT * sh * sw * 3, hipMemcpyDeviceToDevice, s);
        CPX_REQUIRE(e == hipSuccess);
        return CPX_OK;
    }
    // cv::resize: inv_scale = dsize/ssize in double, scale = 1/inv_scale
    double scale_x = 1.0 / ((double)dw / (double)sw), scale_y = 1.0 / ((double)dh / (double)sh);
    int area2 = (sw == 2 * dw && sh == 2 * dh) ? 1 : 0;
    hipLaunchKernelGGL(k_resize_linear_u8, dim3(cpx_cdiv((long long)dh * dw, NTHR), nT), dim3(NTHR), 0, s,
                       src, sh, sw, dst, dh, dw, scale_x, scale_y, area2);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// ---------------------------------------------------------------------------
// a6 normalisation
// ---------------------------------------------------------------------------
__global__ void k_hist_u8(const uint8_t *__restrict__ tiles, int HW, uint32_t *__restrict__ hist) {
    __shared__ uint32_t sh[3 * 256];
    for (int i = threadIdx.x; i < 768; i += NTHR) sh[i] = 0;
    __syncthreads();
    const uint8_t *t = tiles + (size_t)blockIdx.y * HW * 3;
    // each thread handles 4 pixels = 12 bytes
    for (int p = (blockIdx.x * NTHR + threadIdx.x) * 4; p < HW; p += gridDim.x * NTHR * 4) {
        int n = min(4, HW - p);
        for (int k = 0; k < n; ++k) {
            const uint8_t *px = t + (size_t)(p + k) * 3;
            atomicAdd(&sh[px[0]], 1u);
            atomicAdd(&sh[256 + px[1]], 1u);
            atomicAdd(&sh[512 + px[2]], 1u);
        }
    }
    __syncthreads();
    uint32_t *h = hist + (size_t)blockIdx.y * 768;
    for (int i = threadIdx.x; i < 768; i += NTHR)
        if (sh[i]) atomicAdd(&h[i], sh[i]);
}

// np.percentile(X, q) for float32 X ('linear'): sorted[prev] lerp sorted[prev+1] by gamma,
// all in float32 (numpy _lerp).  One thread per (tile, channel).
__device__ __forceinline__ float np_lerp(float a, float b, float t) {
    float diff = b - a;
    float r = a + diff * t;
    if (t >= 0.5f) r = b - diff * (1.0f - t);
    return r;
}
// one WAVE per (tile, channel): lane l owns bins 4l .. 4l+3 (one 16-byte load), a wave scan gives the
// cumulative counts, and "the value at sorted index k" = the smallest v with cum(v) > k falls out of one
// ballot per index (the previous one-thread-per-channel loop chased 256 dependent loads: 81 us on the
// network stream for 24 channels)
__global__ void k_norm_stats(const uint32_t *__restrict__ hist, int nTC, int HW, int lo_prev,
                             float lo_g, int hi_prev, float hi_g, float *__restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int tc = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (tc >= nTC) return;                         // whole waves exit together
    const uint4 c = *reinterpret_cast<const uint4 *>(hist + (size_t)tc * 256 + 4 * lane);
    const long long s0 = c.x, s1 = s0 + c.y, s2 = s1 + c.z, s3 = s2 + c.w;
    long long incl = s3;                           // inclusive scan of the lane totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const long long excl = incl - s3;
    auto value_at = [&](int k) -> float {          // smallest v with cum(v) > k
        const unsigned long long m = __ballot(incl > (long long)k);
        const int l = __ffsll((long long)m) - 1;   // first lane whose cumulative count passes k
        int v = 0;
        if (lane == l) v = 4 * lane + (excl + s0 > k ? 0 : (excl + s1 > k ? 1 : (excl + s2 > k ? 2 : 3)));
        return (float)__shfl(v, l);
    };
    const float g0 = value_at(lo_prev), g1 = value_at(min(lo_prev + 1, HW - 1));
    const float g2 = value_at(hi_prev), g3 = value_at(min(hi_prev + 1, HW - 1));
    const unsigned long long nz = __ballot(s3 > 0);
    const int lmin = __ffsll((long long)nz) - 1, lmax = 63 - __clzll((long long)nz);
    int vlo = 0, vhi = 0;
    if (lane == lmin) vlo = 4 * lane + (c.x ? 0 : (c.y ? 1 : (c.z ? 2 : 3)));
    if (lane == lmax) vhi = 4 * lane + (c.w ? 3 : (c.z ? 2 : (c.y ? 1 : 0)));
    const int vmin = __shfl(vlo, lmin), vmax = __shfl(vhi, lmax);
    if (lane != 0) return;
    float x01 = np_lerp(g0, g1, lo_g);
    float x99 = np_lerp(g2, g3, hi_g);
    float den = x99 - x01;
    float mode;
    if (vmax == vmin) mode = 0.f;                  // np.ptp == 0: channel left untouched
    else if (den > (float)1e-3) mode = 1.f;
    else mode = 2.f;
    stats[tc * 4 + 0] = x01;
    stats[tc * 4 + 1] = den;
    stats[tc * 4 + 2] = mode;
    stats[tc * 4 + 3] = x99;
}

__device__ __forceinline__ float norm_px(uint8_t v, const float *st) {
    float x = (float)v;
    float mode = st[2];
    if (mode == 1.f) { x = x - st[0]; x = __fdiv_rn(x, st[1]); }
    else if (mode == 2.f) x = 0.f;
    return x;
}

__global__ void k_norm_apply(const uint8_t *__restrict__ tiles, const float *__restrict__ stats,
                             int HW, float *__restrict__ out) {
    int i = blockIdx.x * NTHR + threadIdx.x;      // element index in [HW*3]
    if (i >= HW * 3) return;
    size_t t = blockIdx.y;
    int c = i % 3;
    out[t * HW * 3 + i] = norm_px(tiles[t * HW * 3 + i], stats + (t * 3 + c) * 4);
}

extern "C" int cpx_normalize_stats_u8(const uint8_t *tiles, int nT, int H, int W, int lo_prev,
                                      float lo_gamma, int hi_prev, float hi_gamma, float *stats,
                                      uint32_t *hist_ws, void *stream) {
    CPX_REQUIRE(tiles && stats && hist_ws && nT > 0 && H > 0 && W > 0);
    CPX_REQUIRE(lo_prev >= 0 && hi_prev >= lo_prev && hi_prev < H * W);
    hipStream_t s = (hipStream_t)stream;
    CPX_HIP(hipMemsetAsync(hist_ws, 0, sizeof(uint32_t) * 768 * (size_t)nT, s));
    int nb = min(64, cpx_cdiv((long long)H * W, NTHR * 4));
    hipLaunchKernelGGL(k_hist_u8, dim3(nb, nT), dim3(NTHR), 0, s, tiles, H * W, hist_ws);
    hipLaunchKernelGGL(k_norm_stats, dim3(cpx_cdiv(nT * 3, 4)), dim3(256), 0, s, hist_ws, nT * 3,
                       H * W, lo_prev, lo_gamma, hi_prev, hi_gamma, stats);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

extern "C" int cpx_normalize_apply_u8(const uint8_t *tiles, const float *stats, int nT, int H, int W,
                                      float *out, void *stream) {
    CPX_REQUIRE(tiles && stats && out && nT > 0 && H > 0 && W > 0);
    hipLaunchKernelGGL(k_norm_apply, dim3(cpx_cdiv((long long)H * W * 3, NTHR), nT), dim3(NTHR), 0,
                       (hipStream_t)stream, tiles, stats, H * W, out);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// ---------------------------------------------------------------------------
// a7 sub-tiles
// ---------------------------------------------------------------------------
struct TilingDev {
    int H, W, ypad1, xpad1, Ly, Lx, ny, nx, b, augment;
    int ys[16], xs[16];
};
static int tiling_to_dev(const cpx_tiling *t, TilingDev *d) {
    CPX_REQUIRE(t && t->ny >= 1 && t->nx >= 1 && t->ny <= 16 && t->nx <= 16);
    CPX_REQUIRE(t->bsize > 0 && t->bsize % 8 == 0);
    d->H = t->H; d->W = t->W; d->ypad1 = t->ypad1; d->xpad1 = t->xpad1; d->Ly = t->Ly; d->Lx = t->Lx;
    d->ny = t->ny; d->nx = t->nx; d->b = t->bsize; d->augment = t->augment;
    for (int i = 0; i < 16; ++i) { d->ys[i] = t->ystart[i]; d->xs[i] = t->xstart[i]; }
    return CPX_OK;
}

// normalised value of padded-image pixel (Yp, Xp), channel c, tile t  (0 in the pad)
__device__ __forceinline__ float padded_px(const uint8_t *tiles, const float *stats, const TilingDev &g,
                                           size_t t, int Yp, int Xp, int c) {
    int Y = Yp - g.ypad1, X = Xp - g.xpad1;
    if ((unsigned)Y >= (unsigned)g.H || (unsigned)X >= (unsigned)g.W) return 0.f;
    return norm_px(tiles[(t * g.H * g.W + (size_t)Y * g.W + X) * 3 + c], stats + (t * 3 + c) * 4);
}

// one thread = one (sub-tile, token, c, ii) = 8 consecutive k (jj = 0..7)
__device__ __forceinline__ unsigned short f32_to_f16(float f) {
    _Float16 h = (_Float16)f;
    return *reinterpret_cast<unsigned short *>(&h);
}
// DT: 0 bf16, 1 fp16, 2 float32 patch rows
template <int DT>
__global__ void k_make_patches(const uint8_t *__restrict__ tiles, const float *__restrict__ stats,
                               TilingDev g, void *__restrict__ patches_v) {
    const int tok_per = (g.b / 8) * (g.b / 8);
    const int per_sub = tok_per * 24;                         // 3 channels * 8 rows
    int i = blockIdx.x * NTHR + threadIdx.x;
    if (i >= per_sub) return;
    int sub = blockIdx.y;                                     // t*ny*nx + j*nx + ii
    int t = sub / (g.ny * g.nx), ji = sub - t * g.ny * g.nx;
    int j = ji / g.nx, ii_t = ji - j * g.nx;
    int tok = i / 24, r = i - tok * 24;
    int c = r >> 3, ii = r & 7;
    int ph = tok / (g.b / 8), pw = tok - ph * (g.b / 8);
    bool fy = g.augment && (ii_t & 1) && !(j & 1);            // j even, i odd -> flip Y
    bool fx = g.augment && (j & 1) && !(ii_t & 1);            // j odd, i even -> flip X
    if (g.augment && (j & 1) && (ii_t & 1)) { fy = true; fx = true; }
    int y = 8 * ph + ii;
    int sy = fy ? g.b - 1 - y : y;
    union { unsigned short h[8]; uint4 v; } o;
    float f[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        int x = 8 * pw + jj;
        int sx = fx ? g.b - 1 - x : x;
        float v = padded_px(tiles, stats, g, t, g.ys[j] + sy, g.xs[ii_t] + sx, c);
        f[jj] = v;
        o.h[jj] = DT == 1 ? f32_to_f16(v) : f32_to_bf16(v);
    }
    const size_t e = ((size_t)sub * tok_per + tok) * 192 + c * 64 + ii * 8;
    if constexpr (DT == 2) {
        float *patches = (float *)patches_v;
        *reinterpret_cast<float4 *>(patches + e) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4 *>(patches + e + 4) = make_float4(f[4], f[5], f[6], f[7]);
    } else {
        *reinterpret_cast<uint4 *>((unsigned short *)patches_v + e) = o.v;
    }
}

__global__ void k_make_subtiles_f32(const uint8_t *__restrict__ tiles, const float *__restrict__ stats,
                                    TilingDev g, float *__restrict__ out) {
    int i = blockIdx.x * NTHR + threadIdx.x;
    int per = 3 * g.b * g.b;
    if (i >= per) return;
    int sub = blockIdx.y;
    int t = sub / (g.ny * g.nx), ji = sub - t * g.ny * g.nx;
    int j = ji / g.nx, ii_t = ji - j * g.nx;
    int c = i / (g.b * g.b), r = i - c * g.b * g.b;
    int y = r / g.b, x = r - y * g.b;
    bool fy = g.augment && (ii_t & 1) && !(j & 1);
    bool fx = g.augment && (j & 1) && !(ii_t & 1);
    if (g.augment && (j & 1) && (ii_t & 1)) { fy = true; fx = true; }
    int sy = fy ? g.b - 1 - y : y, sx = fx ? g.b - 1 - x : x;
    out[(size_t)sub * per + i] = padded_px(tiles, stats, g, t, g.ys[j] + sy, g.xs[ii_t] + sx, c);
}

extern "C" int cpx_make_patches(const uint8_t *tiles, const float *stats, int nT,
                                const cpx_tiling *tiling, int dtype, void *patches, void *stream) {
    TilingDev g; int rc = tiling_to_dev(tiling, &g); if (rc) return rc;
    CPX_REQUIRE(tiles && stats && patches && nT > 0 && dtype >= 0 && dtype <= 2);
    int per_sub = (g.b / 8) * (g.b / 8) * 24;
    const dim3 grid(cpx_cdiv(per_sub, NTHR), nT * g.ny * g.nx);
    if (dtype == 2) hipLaunchKernelGGL(k_make_patches<2>, grid, dim3(NTHR), 0, (hipStream_t)stream, tiles, stats, g, patches);
    else if (dtype == 1) hipLaunchKernelGGL(k_make_patches<1>, grid, dim3(NTHR), 0, (hipStream_t)stream, tiles, stats, g, patches);
    else hipLaunchKernelGGL(k_make_patches<0>, grid, dim3(NTHR), 0, (hipStream_t)stream, tiles, stats, g, patches);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}
extern "C" int cpx_make_subtiles(const uint8_t *tiles, const float *stats, int nT,
                                 const cpx_tiling *tiling, void *patches, void *stream) {
    return cpx_make_patches(tiles, stats, nT, tiling, 0, patches, stream);
}

extern "C" int cpx_make_subtiles_f32(const uint8_t *tiles, const float *stats, int nT,
                                     const cpx_tiling *tiling, float *subtiles, void *stream) {
    TilingDev g; int rc = tiling_to_dev(tiling, &g); if (rc) return rc;
    CPX_REQUIRE(tiles && stats && subtiles && nT > 0);
    hipLaunchKernelGGL(k_make_subtiles_f32, dim3(cpx_cdiv(3 * g.b * g.b, NTHR), nT * g.ny * g.nx),
                       dim3(NTHR), 0, (hipStream_t)stream, tiles, stats, g, subtiles);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// ---------------------------------------------------------------------------
// a7 blend: un-augment + taper-weighted average + crop (+ pixel shuffle)
// ---------------------------------------------------------------------------
// TOKEN_MAJOR: src = head [sub][token][ld] (col = ch*64 + ii*8 + jj, class block at +192)
// else       : src = y [sub][3][b][b], src2 = y_class [sub][ncls][b][b]
template <bool TOKEN_MAJOR>
__global__ void k_blend(const float *__restrict__ src, const float *__restrict__ src2, int ld, int ncls,
                        TilingDev g, const double *__restrict__ taper, float *__restrict__ dP,
                        float *__restrict__ cellprob, float *__restrict__ logits) {
    int idx = blockIdx.x * NTHR + threadIdx.x;
    const int HW = g.H * g.W;
    if (idx >= HW) return;
    const int ch = blockIdx.z;                 // 0..2 flow block, 3.. class channels
    size_t t = blockIdx.y;
    int Y = idx / g.W, X = idx - Y * g.W;
    int Yp = Y + g.ypad1, Xp = X + g.xpad1;
    const int tb = g.b / 8;
    float acc = 0.f;
    double navg = 0.0;
    for (int j = 0; j < g.ny; ++j) {
        int y = Yp - g.ys[j];
        if ((unsigned)y >= (unsigned)g.b) continue;
        for (int i = 0; i < g.nx; ++i) {
            int x = Xp - g.xs[i];
            if ((unsigned)x >= (unsigned)g.b) continue;
            bool fy = g.augment && (i & 1) && !(j & 1);
            bool fx = g.augment && (j & 1) && !(i & 1);
            if (g.augment && (j & 1) && (i & 1)) { fy = true; fx = true; }
            int sy = fy ? g.b - 1 - y : y, sx = fx ? g.b - 1 - x : x;
            size_t sub = (t * g.ny + j) * g.nx + i;
            float v;
            if (TOKEN_MAJOR) {
                int tok = (sy >> 3) * tb + (sx >> 3);
                int col = (ch < 3 ? ch * 64 : 192 + (ch - 3) * 64) + (sy & 7) * 8 + (sx & 7);
                v = src[(sub * tb * tb + tok) * (size_t)ld + col];
            } else {
                v = ch < 3 ? src[((sub * 3 + ch) * g.b + sy) * (size_t)g.b + sx]
                           : src2[((sub * ncls + (ch - 3)) * g.b + sy) * (size_t)g.b + sx];
            }
            if ((ch == 0 && fy) || (ch == 1 && fx)) v = -v;          // unaugment_tiles sign flip
            double m = taper[x] * taper[y];
            acc = (float)((double)acc + (double)v * m);
            navg = navg + m;
        }
    }
    float r = (float)((double)acc / navg);
    if (ch < 2) dP[(t * 2 + ch) * HW + idx] = r;
    else if (ch == 2) cellprob[t * HW + idx] = r;
    else logits[(t * ncls + (ch - 3)) * HW + idx] = r;
}

extern "C" int cpx_blend_subtiles(const float *head, int ld_head, int ncls, int nT,
                                  const cpx_tiling *tiling, const double *taper1d, float *dP,
                                  float *cellprob, float *logits, void *stream) {
    TilingDev g; int rc = tiling_to_dev(tiling, &g); if (rc) return rc;
    CPX_REQUIRE(head && taper1d && dP && cellprob && nT > 0 && ncls >= 0 && (ncls == 0 || logits));
    CPX_REQUIRE(ld_head >= 192 + ncls * 64);
    hipLaunchKernelGGL(k_blend<true>, dim3(cpx_cdiv(g.H * g.W, NTHR), nT, 3 + ncls), dim3(NTHR), 0,
                       (hipStream_t)stream, head, (const float *)nullptr, ld_head, ncls, g, taper1d,
                       dP, cellprob, logits);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

extern "C" int cpx_blend_subtiles_nchw(const float *y, const float *y_class, int ncls, int nT,
                                       const cpx_tiling *tiling, const double *taper1d, float *dP,
                                       float *cellprob, float *logits, void *stream) {
    TilingDev g; int rc = tiling_to_dev(tiling, &g); if (rc) return rc;
    CPX_REQUIRE(y && taper1d && dP && cellprob && nT > 0 && ncls >= 0 && (ncls == 0 || (logits && y_class)));
    hipLaunchKernelGGL(k_blend<false>, dim3(cpx_cdiv(g.H * g.W, NTHR), nT, 3 + ncls), dim3(NTHR), 0,
                       (hipStream_t)stream, y, y_class, 0, ncls, g, taper1d, dP, cellprob, logits);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}
