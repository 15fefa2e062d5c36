/* Oracle (TEST INFRASTRUCTURE, never shipped or measured as the product):
 * plain-C restatement of cellpose==4.0.8 dynamics.steps_interp (the Euler
 * integration behind dynamics.follow_flows; call site
 * /root/reference/src/classpose/models.py:149-159).
 *
 * Why C and not numpy: the torch CPU kernel this has to equal bit for bit
 * (torch.nn.functional.grid_sample, bilinear / zeros / align_corners=False) is
 * built with FMA contraction, and numpy has no single-rounding fmaf.  The exact
 * op order below was determined against torch 2.10 CPU in this image and is
 * pinned by tests/test_oracle_pins.py::test_c_follow_flows_equals_torch:
 *     ix  = fmaf(gx + 1, W/2, -0.5)            (unnormalize, align_corners=False)
 *     w   = ix - floor(ix); e = 1 - w; (same n, s in y)
 *     nw = s*e, ne = s*w, sw = n*e, se = n*w   (plain products)
 *     out = fmaf(v_se, se, fmaf(v_sw, sw, fmaf(v_ne, ne, v_nw * nw)))
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float tap(const float *im, int H, int W, int y, int x) {
    return (y >= 0 && y < H && x >= 0 && x < W) ? im[(size_t)y * W + x] : 0.0f;
}

/* dP: [2][H][W] float32, ALREADY masked by (cellprob > thr) and divided by 5
 * (dynamics.compute_masks does that in numpy before follow_flows).
 * iy, ix: N start pixels.  out: [2][N] float32 rows (y, x).                 */
void oracle_follow_flows(const float *dP, int H, int W, const int32_t *iy,
                         const int32_t *ix, int64_t N, int niter, float *out)
{
    const float shx = (float)(W - 1), shy = (float)(H - 1);
    /* im[:,0] = dX * (2/(W-1)), im[:,1] = dY * (2/(H-1));  python float -> fp32 scalar */
    const float kx = (float)(2.0 / (double)(W - 1));
    const float ky = (float)(2.0 / (double)(H - 1));
    float *imx = (float *)malloc(sizeof(float) * (size_t)H * W);
    float *imy = (float *)malloc(sizeof(float) * (size_t)H * W);
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        imx[i] = dP[(size_t)H * W + i] * kx;
        imy[i] = dP[i] * ky;
    }
    const float hw = (float)W / 2.0f, hh = (float)H / 2.0f;   /* scaling_factor = size / 2 */
    for (int64_t n = 0; n < N; ++n) {
        float px = (float)ix[n] / shx;     /* pt[...,0] /= W-1   (true division) */
        float py = (float)iy[n] / shy;
        px = px * 2.0f; px = px - 1.0f;    /* pt *= 2 ; pt -= 1 */
        py = py * 2.0f; py = py - 1.0f;
        for (int t = 0; t < niter; ++t) {
            float fx = fmaf(px + 1.0f, hw, -0.5f);
            float fy = fmaf(py + 1.0f, hh, -0.5f);
            float x_w = floorf(fx), y_n = floorf(fy);
            float w = fx - x_w, e = 1.0f - w, nn = fy - y_n, s = 1.0f - nn;
            float nw = s * e, ne = s * w, sw = nn * e, se = nn * w;
            int x0 = (int)x_w, y0 = (int)y_n;
            float dx = fmaf(tap(imx, H, W, y0 + 1, x0 + 1), se,
                       fmaf(tap(imx, H, W, y0 + 1, x0), sw,
                       fmaf(tap(imx, H, W, y0, x0 + 1), ne, tap(imx, H, W, y0, x0) * nw)));
            float dy = fmaf(tap(imy, H, W, y0 + 1, x0 + 1), se,
                       fmaf(tap(imy, H, W, y0 + 1, x0), sw,
                       fmaf(tap(imy, H, W, y0, x0 + 1), ne, tap(imy, H, W, y0, x0) * nw)));
            px = px + dx; px = px < -1.0f ? -1.0f : (px > 1.0f ? 1.0f : px);
            py = py + dy; py = py < -1.0f ? -1.0f : (py > 1.0f ? 1.0f : py);
        }
        px = px + 1.0f; px = px * 0.5f; px = px * shx;   /* pt += 1; pt *= 0.5; pt[...,k] *= shape[k] */
        py = py + 1.0f; py = py * 0.5f; py = py * shy;
        out[n] = py;
        out[N + n] = px;
    }
    free(imx);
    free(imy);
}
