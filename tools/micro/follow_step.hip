// What one Euler step of k_follow (csrc/cpx_postproc.hip) costs a lone wave, by part: the production step (two 16-byte gathers from
// memory), the same with the taps from LDS, with no loads at all (the dependent arithmetic alone), and each without the orbit-closure
// tests.  A 256 x 256 field of cells of radius 7 on a 16-pixel grid (every chain converges onto its centre, as on the bench's fields);
// s_memtime around the whole 200-step loop of each wave, minimum / median over the waves; 1 wave per SIMD (4 per workgroup, one
// workgroup per CU) and 2, 4 per SIMD.
// Build + run: hipcc --offload-arch=gfx950 -O3 follow_step.hip -o follow_step && ./follow_step
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
typedef float flow4 __attribute__((ext_vector_type(4), aligned(8)));
constexpr int H = 256, W = 256, TW = W + 2, TH = H + 2;

// MODE 0: memory taps, 1: LDS taps (the workgroup's 66 x 66 window), 2: no taps (a = b = position-derived constants)
// EARLY 0: no orbit tests, 1: per lane and per step (round 4), 2: once per four steps, for the wave as a whole (round 5)
template <int MODE, int EARLY>
__global__ void __launch_bounds__(256) k(const float *__restrict__ im, unsigned long long *out, float *sink, int niter, int early_exit) {
    __shared__ float2 win[66 * 66];
    // workgroup -> a 32 x 32 block of the field (blockIdx.x % 64), thread -> a pixel of it (4 waves = rows 0..7 of the block: 256 pixels)
    const int blk = blockIdx.x % 64, by = blk / 8, bx = blk % 8;
    const int wx0 = std::max(bx * 32 - 16, 0), wy0 = std::max(by * 32 - 16, 0);
    const int ww = std::min(bx * 32 + 32 + 16 + 2, TW) - wx0, wh = std::min(by * 32 + 32 + 16 + 2, TH) - wy0;
    if (MODE == 1) {
        for (int i = threadIdx.x; i < 66 * 66; i += 256) {
            int r = i / 66, c = i - r * 66;
            if (r < wh && c < ww) win[i] = reinterpret_cast<const float2 *>(im)[(size_t)(wy0 + r) * TW + wx0 + c];
        }
        __syncthreads();
    }
    const int x = bx * 32 + (threadIdx.x & 31), y = by * 32 + 12 + (threadIdx.x >> 5);
    const float shx = W - 1, shy = H - 1, hw = W / 2.0f, hh = H / 2.0f;
    float px = __fdiv_rn((float)x, shx), py = __fdiv_rn((float)y, shy);
    px = px * 2.0f; px = px - 1.0f; py = py * 2.0f; py = py - 1.0f;
    float qx = __int_as_float(0x7fc00000), qy = qx;
    unsigned long long t0, t1;
    STAMP(t0);
    float h1x = qx, h1y = qx;         // (EARLY == 2) the two positions before the current one
    for (int it = 0; it < niter; ++it) {
        float fx = __fmaf_rn(px + 1.0f, hw, -0.5f);
        float fy = __fmaf_rn(py + 1.0f, hh, -0.5f);
        float x_w = floorf(fx), y_n = floorf(fy);
        float w = fx - x_w, e = 1.0f - w, n = fy - y_n, s = 1.0f - n;
        float nw = s * e, ne = s * w, sw = n * e, se = n * w;
        flow4 a, b;
        if (MODE == 0) {
            int x0 = min(max((int)x_w, -1), W - 1), y0 = min(max((int)y_n, -1), H - 1);
            const float *r0 = im + ((size_t)(y0 + 1) * TW + (x0 + 1)) * 2;
            a = *reinterpret_cast<const flow4 *>(r0);
            b = *reinterpret_cast<const flow4 *>(r0 + (size_t)TW * 2);
        } else if (MODE == 1) {
            const int lx = (int)x_w + 1 - wx0, ly = (int)y_n + 1 - wy0;
            typedef const volatile flow4 __attribute__((address_space(3))) *lds_taps;
            a = *(lds_taps)((const __attribute__((address_space(3))) float2 *)win + ly * 66 + lx);
            b = *(lds_taps)((const __attribute__((address_space(3))) float2 *)win + (ly + 1) * 66 + lx);
        } else {
            const float v = (float)((int)x_w + (int)y_n) * 1e-9f;        // keeps the address arithmetic's conversions in the chain
            a = flow4{v, v, v, v}; b = a;
        }
        float dx = __fmaf_rn(b[2], se, __fmaf_rn(b[0], sw, __fmaf_rn(a[2], ne, a[0] * nw)));
        float dy = __fmaf_rn(b[3], se, __fmaf_rn(b[1], sw, __fmaf_rn(a[3], ne, a[1] * nw)));
        float nx = px + dx; nx = nx < -1.0f ? -1.0f : (nx > 1.0f ? 1.0f : nx);
        float ny = py + dy; ny = ny < -1.0f ? -1.0f : (ny > 1.0f ? 1.0f : ny);
        if (EARLY == 2) {
            if ((it & 3) == 3) {
                const bool closed = (nx == px && ny == py) || (nx == h1x && ny == h1y);
                if (early_exit && __builtin_amdgcn_ballot_w64(!closed) == 0) { if ((niter - (it + 1)) & 1) { nx = px; ny = py; } px = nx; py = ny; break; }
            }
            h1x = px; h1y = py;
        }
        if (EARLY == 1) {
            if (early_exit && nx == px && ny == py) break;
            if (early_exit && nx == qx && ny == qy) {
                if (((niter - (it + 1)) & 1) == 0) { px = nx; py = ny; }
                break;
            }
        }
        qx = px; qy = py; px = nx; py = ny;
    }
    STAMP(t1);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = px + py;
}

template <int MODE, int EARLY>
void run(const char *name, const float *im, unsigned long long *d, float *sink, int early) {
    for (int wgs : {256, 512, 1024}) {          // 256 CUs: 1 / 2 / 4 workgroups of 4 waves per CU = 1 / 2 / 4 waves per SIMD
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, EARLY>), dim3(wgs), dim3(256), 0, 0, im, d, sink, 200, early);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MODE, EARLY>), dim3(wgs), dim3(256), 0, 0, im, d, sink, 200, early);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(wgs * 4);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("%-44s early_exit %d  %d waves/SIMD: s_memtime ticks per step min %7.1f median %7.1f max %7.1f   kernel %6.1f us\n", name, early, wgs / 256,
               h[0] / 200.0, h[h.size() / 2] / 200.0, h.back() / 200.0, ms * 1e3);
    }
}

int main() {
    std::vector<float> im((size_t)TH * TW * 2, 0.f);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float cx = (x / 16) * 16 + 8.f, cy = (y / 16) * 16 + 8.f, dx = cx - x, dy = cy - y, r = std::sqrt(dx * dx + dy * dy);
            if (r < 7.f && r > 0.f) {
                im[((size_t)(y + 1) * TW + x + 1) * 2 + 0] = dx / r * 5.f / 5.f * (2.0f / (W - 1));
                im[((size_t)(y + 1) * TW + x + 1) * 2 + 1] = dy / r * 5.f / 5.f * (2.0f / (H - 1));
            }
        }
    float *dim, *sink; unsigned long long *d;
    hipMalloc(&dim, im.size() * 4); hipMemcpy(dim, im.data(), im.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&d, 4096 * 8 * 4); hipMalloc(&sink, 1024 * 256 * 4);
    // calibrate the s_memtime tick against the wall clock of a long kernel
    run<0, 1>("memory taps, per-lane per-step tests (r4)", dim, d, sink, 1);
    run<0, 1>("memory taps, per-lane per-step tests (r4)", dim, d, sink, 0);
    run<0, 0>("memory taps, no orbit tests", dim, d, sink, 0);
    run<0, 2>("memory taps, wave test per 4 steps", dim, d, sink, 0);
    run<1, 1>("LDS taps, per-lane per-step tests", dim, d, sink, 1);
    run<1, 1>("LDS taps, per-lane per-step tests", dim, d, sink, 0);
    run<1, 0>("LDS taps, no orbit tests", dim, d, sink, 0);
    run<1, 2>("LDS taps, wave test per 4 steps (r5)", dim, d, sink, 0);
    run<1, 2>("LDS taps, wave test per 4 steps (r5)", dim, d, sink, 1);
    run<2, 1>("no taps (arithmetic chain + r4 tests)", dim, d, sink, 0);
    run<2, 0>("no taps, no orbit tests", dim, d, sink, 0);
    return 0;
}
