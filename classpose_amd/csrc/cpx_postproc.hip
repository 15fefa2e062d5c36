// Per-pixel post-processing of the classpose WSI tile path on gfx950:
// flows -> Euler integration -> instance ids -> flow-error filter -> hole fill /
// size filter -> class vote -> border removal -> per-instance records.
//
// Every kernel follows the CPU oracle (oracle/dynamics.py, oracle/classmask.py)
// operation for operation; integer results are bit-exact, the float paths use the
// same rounding sequence (this file is built with -ffp-contract=off and spells
// out every fused multiply-add the reference's torch CPU kernels perform).
//
// Layout: all kernels are batched over tiles with blockIdx.y = tile and read /
// write planar [tile][...][H][W] arrays, one thread per pixel along x so loads
// and stores are coalesced; small per-label tables live in the caller's
// workspace (no hidden allocation).
#include "cpx_common.h"
#include "cpx_internal.h"

#define RPAD 20
#define NTHR 256

// Every launch of this file is counted per calling thread (a diagnostic for bench.py's dispatch-bound post-processing
// stage: launches per batch next to its device and host times); the counter is read-only for callers and touches no
// compute state.
static thread_local unsigned long long g_pp_launches = 0;
#define PP_LAUNCH(...) do { ++g_pp_launches; hipLaunchKernelGGL(__VA_ARGS__); } while (0)
extern "C" unsigned long long cpx_postproc_launch_count(void) { return g_pp_launches; }

// ---------------------------------------------------------------------------
// workspace layout
// ---------------------------------------------------------------------------
struct PPLayout {
    int H, W, HW, Hp, Wp, HWp, L, TW, TH, THW, nbx, nseg;
    size_t off_h1, off_M1, off_tmp, off_im, off_fg, off_fgcnt, off_rankbits, off_T, off_e, off_seed_pos, off_seed_cnt,
        off_rank, off_cnt, off_first, off_remap, off_flag, off_bbox, off_sumy, off_sumx,
        off_d2, off_center, off_err, off_cls, off_scal, per_tile, tab0, tab_bytes;
};
#define PP_NSETS 6
#define PP_NSCAL 16
#define SC_NSEEDS 0
#define SC_NLAB 1
#define SC_NITER 2
#define SC_CONFLICT 3
#define SC_HASBIG 4
#define SC_HASBG 5
#define SC_VMAX 6
#define SC_HASMID 7
#define PP_MAXCLS 32

#define FG_BLOCK 1024       // k_prep_flow workgroup = one segment of the foreground list
#define FG_SIDE 32          // ... which is a FG_SIDE x FG_SIDE block of cells of the bordered flow field (k_follow stages its neighbourhood in LDS)
static PPLayout pp_layout(int H, int W) {
    PPLayout p;
    p.H = H; p.W = W; p.HW = H * W; p.Hp = H + 2 * RPAD; p.Wp = W + 2 * RPAD;
    p.HWp = p.Hp * p.Wp; p.L = p.HW / 11 + 2; p.TH = H + 2; p.TW = W + 2; p.THW = p.TH * p.TW;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = cpx_align_up(o + bytes, 256); return r; };
    p.off_h1 = take(sizeof(int) * p.HWp);
    p.off_M1 = take(sizeof(int) * p.HWp);
    p.off_tmp = take(sizeof(int) * p.HW);
    p.off_im = take(sizeof(float) * 2 * p.THW);      // zero-bordered interleaved flow field [H+2][W+2][dx, dy]
    p.nbx = cpx_cdiv(p.TW, FG_SIDE); p.nseg = p.nbx * cpx_cdiv(p.TH, FG_SIDE);   // (>= cdiv(THW, FG_BLOCK), the linear segmentation's count)
    p.off_fg = take(sizeof(int) * FG_BLOCK * p.nseg);   // foreground pixels, one segment per k_prep_flow block
    p.off_fgcnt = take(sizeof(int) * p.nseg);
    p.off_rankbits = take(sizeof(unsigned) * 2 * cpx_cdiv(p.HW, 32));   // tail_rank: one bit per pixel + the words' prefix counts
    p.off_T = take(sizeof(double) * 2 * p.THW);
    p.off_e = take(sizeof(double) * 2 * p.HW);
    p.off_cls = take(sizeof(int) * (size_t)p.L * PP_MAXCLS);
    // ---- the per-label tables + scalars: one contiguous block, PP_NSETS copies of it (the fused chain of
    // cpx_compute_masks gives every stage its own copy, all of them cleared by ONE pass at the start of the chain;
    // the stage-wise entry points use copy 0)
    p.tab0 = o;
    p.off_seed_pos = take(sizeof(int) * p.L);
    p.off_seed_cnt = take(sizeof(int) * p.L);
    p.off_rank = take(sizeof(int) * p.L);
    p.off_cnt = take(sizeof(int) * p.L);
    p.off_first = take(sizeof(int) * p.L);
    p.off_remap = take(sizeof(int) * p.L);
    p.off_flag = take(sizeof(int) * (p.L > 65536 ? p.L : 65536));   // border flags index raw uint16 ids
    p.off_bbox = take(sizeof(int) * 4 * p.L);
    p.off_sumy = take(sizeof(unsigned long long) * p.L);
    p.off_sumx = take(sizeof(unsigned long long) * p.L);
    p.off_d2 = take(sizeof(unsigned long long) * p.L);
    p.off_center = take(sizeof(int) * p.L);
    p.off_err = take(sizeof(double) * p.L);
    p.off_scal = take(sizeof(int) * PP_NSCAL);
    p.tab_bytes = o - p.tab0;
    o = p.tab0 + p.tab_bytes * PP_NSETS;
    p.per_tile = o;
    return p;
}

// the layout with its table block moved to copy k
static PPLayout lay_set(PPLayout p, int k) {
    const size_t d = (size_t)k * p.tab_bytes;
    p.off_seed_pos += d; p.off_seed_cnt += d; p.off_rank += d; p.off_cnt += d; p.off_first += d; p.off_remap += d;
    p.off_flag += d; p.off_bbox += d; p.off_sumy += d; p.off_sumx += d; p.off_d2 += d; p.off_center += d; p.off_err += d;
    p.off_scal += d;
    return p;
}

// workspace = [dense: masks32 nT*HW | p_final nT*HW] + nT per-tile blocks
static size_t pp_dense_bytes(int nT, int H, int W) {
    return cpx_align_up(sizeof(int) * (size_t)nT * H * W, 256) * 2;
}
static void *pp_tiles(void *workspace, int nT, int H, int W) {
    return (char *)workspace + pp_dense_bytes(nT, H, W);
}
extern "C" size_t cpx_postproc_workspace_bytes(int nT, int H, int W) {
    if (nT <= 0 || H <= 0 || W <= 0) return 0;
    return pp_dense_bytes(nT, H, W) + pp_layout(H, W).per_tile * (size_t)nT;
}
extern "C" int cpx_postproc_max_labels(int H, int W) { return H * W / 11 + 2; }

#define WS(T, field) ((T *)((char *)ws + (size_t)blockIdx.y * lay.per_tile + lay.field))

// ---------------------------------------------------------------------------
// a11  follow_flows
// ---------------------------------------------------------------------------
// im = (dP * (cellprob > thr) / 5) * (2 / (size-1))   [numpy fp32 ops, then torch fp32 scalar mul]
// flow field of the Euler loop: (dX, dY) * mask / 5 * 2/(size-1), interleaved and zero-bordered by one pixel,
// so that a bilinear sample is two 16-byte loads (one per row) with no bounds tests -- the loop is bound by
// the texture-addresser rate of its gathers (8 scalar taps per step before)
// The same pass writes p = -1 for background pixels and compacts the foreground pixels (cellprob > thr) of each
// 1024-cell block into that block's segment of the foreground list (ballot + a 16-entry LDS prefix, no atomics, so
// the list order is fixed): the Euler loop then runs on dense waves instead of one thread per pixel with ~70 % of the
// lanes idle for 200 dependent steps (it is VALU-issue bound: every resident wave issues the whole loop).
// INIT (fused chain of cpx_compute_masks): the same launch clears everything the 15 launches behind it accumulate into --
// the padded histogram / seed-label frames, the two fp64 diffusion planes, the class-vote table and ALL PP_NSETS copies of
// the per-label tables and scalars (every stage has its own copy) -- instead of one k_pp_init launch per stage.
__device__ __forceinline__ void pp_init_stats_entry(char *tb, const PPLayout &lay, size_t d, int v) {
    int *bb = (int *)(tb + lay.off_bbox + d) + 4 * v;
    bb[0] = 0x7FFFFFFF; bb[1] = 0x7FFFFFFF; bb[2] = -1; bb[3] = -1;
    ((int *)(tb + lay.off_cnt + d))[v] = 0;
    ((unsigned long long *)(tb + lay.off_sumy + d))[v] = 0;
    ((unsigned long long *)(tb + lay.off_sumx + d))[v] = 0;
    ((unsigned long long *)(tb + lay.off_d2 + d))[v] = 0xFFFFFFFFFFFFFFFFull;
    ((int *)(tb + lay.off_center + d))[v] = 0x7FFFFFFF;
    ((int *)(tb + lay.off_flag + d))[v] = 0;
    ((int *)(tb + lay.off_first + d))[v] = 0x7FFFFFFF;
}
// B2D: the workgroup's 1024 cells are a 32 x 32 block of the bordered field (segment = block; k_follow<.., true> stages the block's
// neighbourhood in LDS) instead of 1024 consecutive cells of its row-major order.
template <bool INIT, bool B2D>
__global__ void __launch_bounds__(FG_BLOCK) k_prep_flow(const float *__restrict__ dP, const float *__restrict__ cp,
                                                        float thr, float kx, float ky, int32_t *__restrict__ p_final,
                                                        float *__restrict__ p_float, PPLayout lay, void *ws) {
    __shared__ int sW[FG_BLOCK / 64];
    if constexpr (INIT) {
        char *tb = (char *)ws + (size_t)blockIdx.y * lay.per_tile;
        const int n_cls = lay.L * PP_MAXCLS, n_tab = lay.L * PP_NSETS;
        int nmax = lay.HWp > 2 * lay.THW ? lay.HWp : 2 * lay.THW;
        nmax = nmax > n_cls ? nmax : n_cls;
        nmax = nmax > n_tab ? nmax : n_tab;
        for (int i = blockIdx.x * FG_BLOCK + threadIdx.x; i < nmax; i += gridDim.x * FG_BLOCK) {
            if (i < lay.HWp) { ((int *)(tb + lay.off_h1))[i] = 0; ((int *)(tb + lay.off_M1))[i] = 0; }
            if (i < 2 * lay.THW) ((double *)(tb + lay.off_T))[i] = 0.0;
            if (i < n_cls) ((int *)(tb + lay.off_cls))[i] = 0;
            if (i < n_tab) { const int k = i / lay.L; pp_init_stats_entry(tb, lay, (size_t)k * lay.tab_bytes, i - k * lay.L); }
            if (i < PP_NSCAL * PP_NSETS) ((int *)(tb + lay.off_scal + (size_t)(i / PP_NSCAL) * lay.tab_bytes))[i % PP_NSCAL] = 0;
        }
    }
    int c, py, px;
    bool cell;
    if constexpr (B2D) {
        const int by = blockIdx.x / lay.nbx, bx = blockIdx.x - by * lay.nbx;
        py = by * FG_SIDE + (threadIdx.x >> 5); px = bx * FG_SIDE + (threadIdx.x & 31);
        c = py * lay.TW + px;
        cell = py < lay.TH && px < lay.TW;
    } else {
        c = blockIdx.x * FG_BLOCK + threadIdx.x;
        py = c / lay.TW; px = c - py * lay.TW;
        cell = c < lay.THW;
    }
    size_t t = blockIdx.y;
    float2 v = make_float2(0.f, 0.f);
    bool fg = false;
    int idx = 0;
    if (cell && py >= 1 && py <= lay.H && px >= 1 && px <= lay.W) {
        idx = (py - 1) * lay.W + (px - 1);
        fg = cp[t * lay.HW + idx] > thr;
        float m = fg ? 1.0f : 0.0f;
        float dy = dP[(t * 2 + 0) * lay.HW + idx];
        float dx = dP[(t * 2 + 1) * lay.HW + idx];
        v.x = __fdiv_rn(dx * m, 5.0f) * kx;               // im[0] = dX
        v.y = __fdiv_rn(dy * m, 5.0f) * ky;               // im[1] = dY
        if (!fg) {
            p_final[t * lay.HW + idx] = -1;
            if (p_float) { p_float[(t * 2) * lay.HW + idx] = -1.f; p_float[(t * 2 + 1) * lay.HW + idx] = -1.f; }
        }
    }
    if (cell) reinterpret_cast<float2 *>(WS(float, off_im))[c] = v;
    const unsigned long long bal = __ballot(fg);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sW[wave] = __popcll(bal);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < FG_BLOCK / 64; ++w) { int n = sW[w]; base += w < wave ? n : 0; total += n; }
    if (fg) WS(int, off_fg)[blockIdx.x * FG_BLOCK + base + __popcll(bal & ((1ull << lane) - 1ull))] = idx;
    if (threadIdx.x == 0) WS(int, off_fgcnt)[blockIdx.x] = total;
}

typedef float flow4 __attribute__((ext_vector_type(4), aligned(8)));

// HIST (fused chain): the final position goes straight into the padded histogram of get_masks (k_hist's pass)
// What the Euler loop costs and why (round 5; profiles/r05_micro_follow_step.txt, r05_follow_niter_scan.txt, r05_ab_follow_segments_window_grouped.txt):
// the kernel lasted 0.53 us per step on the bench's fields although a lone wave needs 534 cycles for one (0.22 us), for three reasons found in
// this order.  (1) Row-major foreground segments (1024 consecutive cells = 4 rows of the field) put the pixels of ~20 different cells into one
// wave, so each of its two gathers per step touches up to 64 different cache lines and the CU's one texture addresser serialises them: 32 x 32
// -cell segments give a wave the pixels of two rows of ONE block, which converge onto a handful of centres (84.6 against 118.4 us for a11).
// (2) The per-lane, per-step orbit tests were a third of the lone wave's step (GROUPED below: 67.1 us).  (3) The workgroups that had anything
// to do were the multiples of four, i.e. all on two of the eight XCDs (see the index arithmetic at the top of the kernel): worth 3 % while the
// gathers bound every wave wherever it ran, and 113.6 -> 65.4 us for the kernel as it is now.  An earlier attempt
// at the same loop -- TWO chains per lane in half as many waves, as the round-4 review suggested for a latency-bound loop -- made it slower
// (194-207 against 120-123 us, profiles/r05_ab_follow_two_chains.txt): with eight waves to a SIMD on the two busy XCDs the loop was bound by
// instruction issue there, not by any wave's latency.
// LDSWIN (production, with k_prep_flow<.., true>'s 32 x 32-cell segments): the workgroup first copies the segment's neighbourhood of the
// flow field -- the block and FW_HALO cells around it, clipped to the bordered field: at most 66 x 66 cells, 34 KB -- into LDS, and a step
// of a wave whose taps ALL lie inside it reads them there (two ds_read2_b64 per lane: ~70 cycles in the dependent chain of a step instead of
// the ~160 of two 16-byte gathers, and nothing for the CU's one texture addresser to serialise).  The values are the same values and so is
// every operation on them; a wave of which some lane has left the window (a pixel further than FW_HALO cells from its segment: a long cell,
// a noise field) takes the memory path for that step, all lanes together -- the decision is one ballot, the loop has no divergent branch.
// GROUPED (production): the orbit-closure test is made once per four steps and for the wave as a whole.  tools/micro/follow_step.hip: of the
// 534 cycles a lone wave spent per step, 190 were the dependent arithmetic, 158 the gathers and 186 the per-lane, per-step orbit tests
// (four compares whose results travel vector -> scalar -> exec mask and back, two divergent breaks and the register copies of a
// loop-carried (q, p) pair).  A lane whose orbit has closed recomputes its fixed point or 2-cycle at no cost to anyone until the last lane of
// its wave has closed too -- the wave lasted that long before as well.
#define FW_HALO 16
#define FW_SIDE (FG_SIDE + 2 * FW_HALO + 2)
struct FollowCtx {
    const float *im; const float2 *win;
    float hw, hh;
    int W, H, TW, wx0, wy0, ww1, wh1;
};
// one Euler step: (px, py) -> (nx, ny)
template <int LDSWIN>
__device__ __forceinline__ void follow_step(const FollowCtx &c, float px, float py, float &nx, float &ny) {
    float fx = __fmaf_rn(px + 1.0f, c.hw, -0.5f);
    float fy = __fmaf_rn(py + 1.0f, c.hh, -0.5f);
    float x_w = floorf(fx), y_n = floorf(fy);
    float w = fx - x_w, e = 1.0f - w, n = fy - y_n, s = 1.0f - n;
    float nw = s * e, ne = s * w, sw = n * e, se = n * w;
    flow4 a, b;                                                            // (dx, dy) at x0, (dx, dy) at x0+1; the row below
    const int lx = (int)x_w + 1 - c.wx0, ly = (int)y_n + 1 - c.wy0;        // the north-west tap relative to the window
    bool inside = LDSWIN && (unsigned)lx < (unsigned)c.ww1 && (unsigned)ly < (unsigned)c.wh1;
    if constexpr (LDSWIN == 1) inside = __builtin_amdgcn_ballot_w64(!inside) == 0;        // the whole wave one way
    if (inside) {
        // inside the window is inside the bordered field: the clamps of the memory path are the identity here
        // (volatile: without it the two paths' loads are merged into ONE flat load through a selected generic pointer)
        typedef const volatile flow4 __attribute__((address_space(3))) *lds_taps;
        a = *(lds_taps)((const __attribute__((address_space(3))) float2 *)c.win + ly * FW_SIDE + lx);
        b = *(lds_taps)((const __attribute__((address_space(3))) float2 *)c.win + (ly + 1) * FW_SIDE + lx);
    } else {
        // positions are clamped to [-1, 1], so x0 is in [-1, W-1] and the 2x2 taps lie inside the bordered
        // field (the min/max only keeps a NaN position from addressing outside the buffer)
        int x0 = min(max((int)x_w, -1), c.W - 1), y0 = min(max((int)y_n, -1), c.H - 1);
        const float *r0 = c.im + ((size_t)(y0 + 1) * c.TW + (x0 + 1)) * 2;
        a = *reinterpret_cast<const flow4 *>(r0);
        b = *reinterpret_cast<const flow4 *>(r0 + (size_t)c.TW * 2);
    }
    float dx = __fmaf_rn(b[2], se, __fmaf_rn(b[0], sw, __fmaf_rn(a[2], ne, a[0] * nw)));
    float dy = __fmaf_rn(b[3], se, __fmaf_rn(b[1], sw, __fmaf_rn(a[3], ne, a[1] * nw)));
    nx = px + dx; nx = nx < -1.0f ? -1.0f : (nx > 1.0f ? 1.0f : nx);
    ny = py + dy; ny = ny < -1.0f ? -1.0f : (ny > 1.0f ? 1.0f : ny);
}
// LDSWIN: 0 = taps from memory, 1 = LDS window, wave-uniform choice per step (production), 2 = LDS window, per-lane choice (A/B)
template <bool HIST, int LDSWIN, bool GROUPED>
__global__ void __launch_bounds__(NTHR) k_follow(int niter, float shx, float shy,
                         float hw, float hh, int32_t *__restrict__ p_final,
                         float *__restrict__ p_float, PPLayout lay, void *ws, int early_exit) {
    // workgroup b handles entries [256 q, 256 q + 256) of foreground segment b % nsegs, q = b / nsegs.  (Until round 5 it was segment b >> 2,
    // q = b & 3.  With a fifth of the pixels in the foreground nearly every segment has fewer than 256 entries, i.e. the workgroups that run
    // the 200 steps were the multiples of 4 -- and workgroups go round the 8 XCDs by their index: all of them sat on XCDs 0 and 4, eight
    // waves to a SIMD, while six XCDs ran the ones that return at once.)
    const int nsegs = gridDim.x / (FG_BLOCK / NTHR), quarter = blockIdx.x / nsegs;
    const int seg = blockIdx.x - quarter * nsegs, ent = quarter * NTHR + threadIdx.x;
    const int cnt = WS(int, off_fgcnt)[seg];
    __shared__ float2 win[LDSWIN ? FW_SIDE * FW_SIDE : 1];
    FollowCtx c;
    c.im = WS(float, off_im); c.win = win; c.hw = hw; c.hh = hh; c.W = lay.W; c.H = lay.H; c.TW = lay.TW;
    c.wx0 = 0; c.wy0 = 0; c.ww1 = 0; c.wh1 = 0;      // window origin (bordered-field cell) and its extent minus one
    if constexpr (LDSWIN != 0) {
        if (quarter * NTHR >= cnt) return;                  // (the whole workgroup: no barrier is skipped by a part of it)
        const int by = seg / lay.nbx, bx = seg - by * lay.nbx;
        c.wx0 = max(bx * FG_SIDE - FW_HALO, 0); c.wy0 = max(by * FG_SIDE - FW_HALO, 0);
        const int ww = min(bx * FG_SIDE + FG_SIDE + FW_HALO + 2, c.TW) - c.wx0, wh = min(by * FG_SIDE + FG_SIDE + FW_HALO + 2, lay.TH) - c.wy0;
        c.ww1 = ww - 1; c.wh1 = wh - 1;
        const float2 *src = reinterpret_cast<const float2 *>(c.im) + (size_t)c.wy0 * c.TW + c.wx0;
        // all of a thread's loads are issued before the first is stored: the field was written by the launch before this one, on whatever
        // XCD, so every load is a trip to the Infinity Cache -- 18 of them one behind the other cost more than the window saves
        constexpr int FW_LOADS = (FW_SIDE * FW_SIDE + NTHR - 1) / NTHR;
        float2 v[FW_LOADS];
#pragma unroll
        for (int k = 0; k < FW_LOADS; ++k) {
            const int i = threadIdx.x + k * NTHR, r = i / FW_SIDE, cc = i - r * FW_SIDE;
            v[k] = src[(size_t)min(r, c.wh1) * c.TW + min(cc, c.ww1)];
        }
#pragma unroll
        for (int k = 0; k < FW_LOADS; ++k) {
            const int i = threadIdx.x + k * NTHR, r = i / FW_SIDE, cc = i - r * FW_SIDE;
            if (i < FW_SIDE * FW_SIDE && r < wh && cc < ww) win[i] = v[k];
        }
        __syncthreads();
    }
    if (ent >= cnt) return;
    const int idx = WS(int, off_fg)[seg * FG_BLOCK + ent];
    size_t t = blockIdx.y;
    const int W = lay.W;
    int y = idx / W, x = idx - y * W;
    float px = __fdiv_rn((float)x, shx);
    float py = __fdiv_rn((float)y, shy);
    px = px * 2.0f; px = px - 1.0f;
    py = py * 2.0f; py = py - 1.0f;
    // The step is a pure function of the position, so the remaining iterations can be skipped EXACTLY
    // once the orbit closes: a fixed point repeats forever, a 2-cycle alternates (+0 / -0 compare equal
    // and behave identically: the next step starts with p + 1).  Anything else runs all niter steps.
    if constexpr (GROUPED) {
        int it = 0;
        for (; it + 4 <= niter; it += 4) {
            float x1, y1, x2, y2, x3, y3, x4, y4;
            follow_step<LDSWIN>(c, px, py, x1, y1);
            follow_step<LDSWIN>(c, x1, y1, x2, y2);
            follow_step<LDSWIN>(c, x2, y2, x3, y3);
            follow_step<LDSWIN>(c, x3, y3, x4, y4);
            px = x4; py = y4;
            // s[it+4] == s[it+3] (fixed point) or == s[it+2] (2-cycle), in EVERY lane of the wave: the state after the niter - (it + 4)
            // steps that remain is s[it+4] if that many is even, s[it+3] if odd (the same value for a fixed point)
            const bool closed = (x4 == x3 && y4 == y3) || (x4 == x2 && y4 == y2);
            if (early_exit && __builtin_amdgcn_ballot_w64(!closed) == 0) {
                if ((niter - (it + 4)) & 1) { px = x3; py = y3; }
                it = niter;
                break;
            }
        }
        for (; it < niter; ++it) {                  // niter % 4 steps
            float nx, ny;
            follow_step<LDSWIN>(c, px, py, nx, ny);
            px = nx; py = ny;
        }
    } else {
        float qx = __int_as_float(0x7fc00000), qy = qx;      // position two steps back (NaN: never equal)
        for (int it = 0; it < niter; ++it) {
            float nx, ny;
            follow_step<LDSWIN>(c, px, py, nx, ny);
            if (early_exit && nx == px && ny == py) break;
            if (early_exit && nx == qx && ny == qy) {                       // s[it+1] == s[it-1]
                if (((niter - (it + 1)) & 1) == 0) { px = nx; py = ny; }
                break;
            }
            qx = px; qy = py; px = nx; py = ny;
        }
    }
    px = px + 1.0f; px = px * 0.5f; px = px * shx;
    py = py + 1.0f; py = py * 0.5f; py = py * shy;
    if (p_float) { p_float[(t * 2) * lay.HW + idx] = py; p_float[(t * 2 + 1) * lay.HW + idx] = px; }
    int iy = (int)py, ix = (int)px;            // .int(): truncation toward zero
    p_final[t * lay.HW + idx] = (iy << 16) | (ix & 0xFFFF);
    if constexpr (HIST) {                      // exactly k_hist on the value just written
        int hy = iy + RPAD, hx = (int)(short)(ix & 0xFFFF) + RPAD;
        hy = max(hy, 0); hx = max(hx, 0);
        hy = min(hy, lay.H + RPAD - 1); hx = min(hx, lay.W + RPAD - 1);
        atomicAdd(&WS(int, off_h1)[hy * lay.Wp + hx], 1);
    }
}

// ---------------------------------------------------------------------------
// a12  get_masks
// ---------------------------------------------------------------------------
__global__ void k_zero_i32(int *p, size_t per_tile_stride_bytes, size_t off, int n) {
    int idx = blockIdx.x * NTHR + threadIdx.x;
    if (idx >= n) return;
    ((int *)((char *)p + (size_t)blockIdx.y * per_tile_stride_bytes + off))[idx] = 0;
}

__global__ void k_hist(const int32_t *__restrict__ p_final, PPLayout lay, void *ws) {
    int idx = blockIdx.x * NTHR + threadIdx.x;
    if (idx >= lay.HW) return;
    int p = p_final[(size_t)blockIdx.y * lay.HW + idx];
    if (p == -1) return;
    int py = (p >> 16) + RPAD, px = (int)(short)(p & 0xFFFF) + RPAD;
    py = max(py, 0); px = max(px, 0);
    py = min(py, lay.H + RPAD - 1); px = min(px, lay.W + RPAD - 1);
    atomicAdd(&WS(int, off_h1)[py * lay.Wp + px], 1);
}

// seeds = (h == 5x5 max) & (h > 10); collected unordered, ranked afterwards
__global__ void k_seeds(PPLayout lay, void *ws) {
    int idx = blockIdx.x * NTHR + threadIdx.x;
    if (idx >= lay.HWp) return;
    const int *h1 = WS(int, off_h1);
    int h = h1[idx];
    if (h <= 10) return;
    int y = idx / lay.Wp, x = idx - y * lay.Wp;
    int mx = 0;
    for (int dy = -2; dy <= 2; ++dy) {
        int yy = y + dy;
        if (yy < 0 || yy >= lay.Hp) continue;
        for (int dx = -2; dx <= 2; ++dx) {
            int xx = x + dx;
            if (xx < 0 || xx >= lay.Wp) continue;
            mx = max(mx, h1[yy * lay.Wp + xx]);
        }
    }
    if (h == mx) {
        int slot = atomicAdd(&WS(int, off_scal)[SC_NSEEDS], 1);
        if (slot < lay.L) {
            WS(int, off_seed_pos)[slot] = idx;
            WS(int, off_seed_cnt)[slot] = h;
        }
    }
}

// 11x11 window, 5 x (3x3 dilation & (h > 2)); label = rank+1, later (larger) wins; rank = the seed's position in the
// stable ascending sort by (count, raster index), counted by the wave over all seeds (was a kernel of its own).
// One WAVE per seed: the lanes fetch the 121 window counts two each, two ballots give the 11 "allowed" row masks to
// every lane (the five dilations are then wave-uniform bit operations), and lane l writes window positions l and l + 64
// (one thread per seed read its 121 counts one after the other: 35-40 us for ~650 seeds per batch).
#define SEED_WGS 64                                         // x 4 waves: seeds k, k + 256, ... per wave
__global__ void __launch_bounds__(NTHR) k_seed_grow(PPLayout lay, void *ws) {
    const int n = min(WS(int, off_scal)[SC_NSEEDS], lay.L);
    const int lane = threadIdx.x & 63;
    const int *h1 = WS(int, off_h1);
    int *M1 = WS(int, off_M1);
    for (int k = blockIdx.x * (NTHR / 64) + (threadIdx.x >> 6); k < n; k += gridDim.x * (NTHR / 64)) {
        const int p = WS(int, off_seed_pos)[k];
        int label = 1;
        {
            const int *pos = WS(int, off_seed_pos), *cnt = WS(int, off_seed_cnt);
            const int c = cnt[k];
            for (int j0 = 0; j0 < n; j0 += 64) {
                const int j = j0 + lane;
                bool before = false;
                if (j < n) { const int cj = cnt[j], pj = pos[j]; before = (cj < c) || (cj == c && pj < p); }
                label += (int)__popcll(__ballot(before));
            }
        }
        const int sy = p / lay.Wp, sx = p - sy * lay.Wp;
        bool a0 = false, a1 = false;
        {
            const int r = lane / 11, c = lane - r * 11, yy = sy - 5 + r, xx = sx - 5 + c;
            a0 = (unsigned)yy < (unsigned)lay.Hp && (unsigned)xx < (unsigned)lay.Wp && h1[yy * lay.Wp + xx] > 2;
            const int q = lane + 64;
            if (q < 121) {
                const int r1 = q / 11, c1 = q - r1 * 11, y1 = sy - 5 + r1, x1 = sx - 5 + c1;
                a1 = (unsigned)y1 < (unsigned)lay.Hp && (unsigned)x1 < (unsigned)lay.Wp && h1[y1 * lay.Wp + x1] > 2;
            }
        }
        const unsigned long long b0 = __ballot(a0), b1 = __ballot(a1);      // window positions 0..63 and 64..120
        unsigned allowed[11], m[11];
#pragma unroll
        for (int r = 0; r < 11; ++r) {
            const int lo = 11 * r;                                           // bit position of (r, 0)
            unsigned long long v;
            if (lo + 11 <= 64) v = b0 >> lo;
            else if (lo >= 64) v = b1 >> (lo - 64);
            else v = (b0 >> lo) | (b1 << (64 - lo));
            allowed[r] = (unsigned)v & 0x7FFu;
            m[r] = 0;
        }
        m[5] = 1u << 5;
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            unsigned hrow[11];
#pragma unroll
            for (int r = 0; r < 11; ++r) hrow[r] = (m[r] | (m[r] << 1) | (m[r] >> 1)) & 0x7FFu;
#pragma unroll
            for (int r = 0; r < 11; ++r) {
                unsigned v = hrow[r];
                if (r > 0) v |= hrow[r - 1];
                if (r < 10) v |= hrow[r + 1];
                m[r] = v & allowed[r];
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int q = lane + 64 * half;
            if (q < 121) {
                const int r = q / 11, c = q - r * 11;
                unsigned row = 0;
#pragma unroll
                for (int rr = 0; rr < 11; ++rr) row = rr == r ? m[rr] : row;
                if ((row >> c) & 1u) atomicMax(&M1[(sy - 5 + r) * lay.Wp + (sx - 5 + c)], label);
            }
        }
    }
}

// Label statistics visit RUNS, not pixels: a thread owns 8 consecutive pixels and calls f(label, first
// linear index, run length, y, x of the first pixel) once per maximal run of one positive label on one image
// row (instances are ~15 px wide, so this issues ~8x fewer atomics than one call per pixel; all updates are
// integer min / max / add, so the results are identical).  Returns true if the thread saw a background pixel.
#define RUN_PX 8
#define GRID_RUN(lay, nT) dim3(cpx_cdiv(cpx_cdiv((lay).HW, RUN_PX), NTHR), (nT))
template <typename F>
__device__ __forceinline__ bool label_runs(const int (&lab)[RUN_PX], int base, int HW, int W, int limit, F &&f) {
    bool bg = false;
    int y = base / W, x = base - y * W;
    int cur = 0, start = 0, sx = 0, sy = 0, n = 0;
#pragma unroll
    for (int i = 0; i < RUN_PX; ++i) {
        int l = lab[i];
        if (base + i < HW && l <= 0) bg = true;
        if (l <= 0 || l >= limit) l = 0;
        if (l != cur || x == 0) {                       // label change or a new image row: close the run
            if (cur > 0) f(cur, start, n, sy, sx);
            cur = l; start = base + i; sx = x; sy = y; n = 0;
        }
        ++n;
        if (++x == W) { x = 0; ++y; }
    }
    if (cur > 0) f(cur, start, n, sy, sx);
    return bg;
}
template <typename T, typename F>
__device__ __forceinline__ bool for_label_runs(const T *__restrict__ m, int HW, int W, int limit, F &&f,
                                               int32_t *__restrict__ copy_to = nullptr) {
    const int base = (blockIdx.x * NTHR + threadIdx.x) * RUN_PX;
    if (base >= HW) return false;
    int lab[RUN_PX];
    if (base + RUN_PX <= HW && sizeof(T) == 4 && (reinterpret_cast<size_t>(m + base) & 15) == 0) {
        const int4 a = *reinterpret_cast<const int4 *>(m + base), b = *reinterpret_cast<const int4 *>(m + base + 4);
        lab[0] = a.x; lab[1] = a.y; lab[2] = a.z; lab[3] = a.w; lab[4] = b.x; lab[5] = b.y; lab[6] = b.z; lab[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < RUN_PX; ++i) lab[i] = base + i < HW ? (int)m[base + i] : 0;
    }
    if (copy_to) {                                       // a verbatim copy of the map on the way (fill-holes scratch)
#pragma unroll
        for (int i = 0; i < RUN_PX; ++i)
            if (base + i < HW) copy_to[base + i] = lab[i];
    }
    return label_runs(lab, base, HW, W, limit, f);
}

// Per-workgroup aggregation of per-label updates.  A label's table entries are 4-16 bytes apart, so the ~50 row runs of
// each of a tile's ~80 labels all hit the same three or four cache lines, and the L2 serialises them (k_first: 13 us for a
// pass that reads 2 MB).  A 256-slot LDS hash keyed by the label collects a workgroup's updates first (2 048 pixels =
// 8 image rows see a few dozen labels); one global atomic per (label, workgroup) follows.  Integer min / max / add only,
// so the tables end up identical.  lh_slot returns -1 when 8 probes find no slot: the caller then updates global memory.
#define LH_SLOTS NTHR
__device__ __forceinline__ int lh_slot(int *key, int lab) {
    const unsigned h = ((unsigned)lab * 2654435761u) >> 24;
#pragma unroll 1
    for (int p = 0; p < 8; ++p) {
        const int s = (int)((h + p) & (LH_SLOTS - 1));
        const int old = atomicCAS(&key[s], 0, lab);
        if (old == 0 || old == lab) return s;
    }
    return -1;
}

// label gather + per-label pixel count (label 0 not counted)
__global__ void k_gather(const int32_t *__restrict__ p_final, int32_t *__restrict__ masks,
                         PPLayout lay, void *ws) {
    __shared__ int key[LH_SLOTS], vadd[LH_SLOTS];
    key[threadIdx.x] = 0; vadd[threadIdx.x] = 0;
    __syncthreads();
    const int base = (blockIdx.x * NTHR + threadIdx.x) * RUN_PX;
    const size_t t = blockIdx.y;
    int *cnt = WS(int, off_cnt);
    if (base < lay.HW) {
        const int *M1 = WS(int, off_M1);
        int lab[RUN_PX];
#pragma unroll
        for (int i = 0; i < RUN_PX; ++i) {
            lab[i] = 0;
            if (base + i < lay.HW) {
                const int p = p_final[t * lay.HW + base + i];
                if (p != -1) {
                    int py = (p >> 16) + RPAD, px = (int)(short)(p & 0xFFFF) + RPAD;
                    py = min(max(py, 0), lay.H + RPAD - 1); px = min(max(px, 0), lay.W + RPAD - 1);
                    lab[i] = M1[py * lay.Wp + px];
                }
                masks[t * lay.HW + base + i] = lab[i];
            }
        }
        label_runs(lab, base, lay.HW, lay.W, 0x7FFFFFFF, [&](int l, int, int n, int, int) {
            const int sl = lh_slot(key, l);
            if (sl >= 0) atomicAdd(&vadd[sl], n); else atomicAdd(&cnt[l], n);
        });
    }
    __syncthreads();
    if (key[threadIdx.x]) atomicAdd(&cnt[key[threadIdx.x]], vadd[threadIdx.x]);
}

// big-mask removal (count > H*W*fraction) + first raster index per surviving label
__global__ void k_big_first(int32_t *__restrict__ masks, double big, PPLayout lay, void *ws) {
    __shared__ int key[LH_SLOTS], vmin[LH_SLOTS];
    key[threadIdx.x] = 0; vmin[threadIdx.x] = 0x7FFFFFFF;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {       // labels are seed ranks 1..nseeds; clear the count for the rank kernel
        WS(int, off_scal)[SC_VMAX] = WS(int, off_scal)[SC_NSEEDS];
        WS(int, off_scal)[SC_NLAB] = 0;
    }
    const int *cnt = WS(int, off_cnt);
    int *first = WS(int, off_first);
    int32_t *m = masks + (size_t)blockIdx.y * lay.HW;
    for_label_runs(m, lay.HW, lay.W, 0x7FFFFFFF, [&](int lab, int idx, int n, int, int) {
        if ((double)cnt[lab] > big) { for (int i = 0; i < n; ++i) m[idx + i] = 0; }
        else {
            const int sl = lh_slot(key, lab);
            if (sl >= 0) atomicMin(&vmin[sl], idx); else atomicMin(&first[lab], idx);
        }
    });
    __syncthreads();
    if (key[threadIdx.x]) atomicMin(&first[key[threadIdx.x]], vmin[threadIdx.x]);
}

// generic: first raster index per label (labels < L); also clears the SC_NLAB scalar for the rank kernel that follows.
// APPLY_FLAGS: labels whose flag is set are zeroed in the map on the way (the size filter's removal pass, one launch
// instead of k_zero_flagged + k_first).
template <bool APPLY_FLAGS>
__global__ void k_first(int32_t *__restrict__ masks, PPLayout lay, void *ws) {
    __shared__ int key[LH_SLOTS], vmin[LH_SLOTS];
    key[threadIdx.x] = 0; vmin[threadIdx.x] = 0x7FFFFFFF;
    if (blockIdx.x == 0 && threadIdx.x == 0) WS(int, off_scal)[SC_NLAB] = 0;
    __syncthreads();
    int *first = WS(int, off_first);
    int32_t *m = masks + (size_t)blockIdx.y * lay.HW;
    const int base = (blockIdx.x * NTHR + threadIdx.x) * RUN_PX;
    if (base < lay.HW) {
        int lab[RUN_PX];
        bool changed = false;
        const int *flag = WS(int, off_flag);
#pragma unroll
        for (int i = 0; i < RUN_PX; ++i) {
            lab[i] = base + i < lay.HW ? m[base + i] : 0;
            if (APPLY_FLAGS && lab[i] > 0 && flag[lab[i]]) { lab[i] = 0; changed = true; }
        }
        if (APPLY_FLAGS && changed) {
#pragma unroll
            for (int i = 0; i < RUN_PX; ++i)
                if (base + i < lay.HW) m[base + i] = lab[i];
        }
        label_runs(lab, base, lay.HW, lay.W, 0x7FFFFFFF, [&](int l, int idx, int, int, int) {
            const int sl = lh_slot(key, l);
            if (sl >= 0) atomicMin(&vmin[sl], idx); else atomicMin(&first[l], idx);
        });
    }
    __syncthreads();
    if (key[threadIdx.x]) atomicMin(&first[key[threadIdx.x]], vmin[threadIdx.x]);
}

__global__ void k_fill_i32(size_t off, int n, int value, PPLayout lay, void *ws) {
    int idx = blockIdx.x * NTHR + threadIdx.x;
    if (idx >= n) return;
    ((int *)((char *)ws + (size_t)blockIdx.y * lay.per_tile + off))[idx] = value;
}

// fastremap.renumber: new id = 1 + #labels whose first appearance is earlier.
// vmax_slot: scalar holding the largest label value that can occur (exclusive bound = value+1)
__global__ void k_renumber_rank(int vmax_slot, PPLayout lay, void *ws) {
    int vmax = min(WS(int, off_scal)[vmax_slot], lay.L - 1);
    int v = blockIdx.x * NTHR + threadIdx.x + 1;
    if (v > vmax) return;
    const int *first = WS(int, off_first);
    int f = first[v];
    int *remap = WS(int, off_remap);
    if (f == 0x7FFFFFFF) { remap[v] = 0; return; }
    int r = 1;
    for (int j = 1; j <= vmax; ++j) r += first[j] < f;
    remap[v] = r;
    atomicMax(&WS(int, off_scal)[SC_NLAB], r);
}

__global__ void k_relabel(int32_t *__restrict__ masks, int32_t *__restrict__ nlabels_out, PPLayout lay, void *ws) {
    int idx = blockIdx.x * NTHR + threadIdx.x;
    if (idx == 0 && nlabels_out) nlabels_out[blockIdx.y] = WS(int, off_scal)[SC_NLAB];   // final after k_renumber_rank
    if (idx >= lay.HW) return;
    size_t t = blockIdx.y;
    int lab = masks[t * lay.HW + idx];
    if (lab > 0) masks[t * lay.HW + idx] = WS(int, off_remap)[lab];
}

__global__ void k_store_scalar(int slot, int32_t *__restrict__ out, PPLayout lay, void *ws) {
    if (threadIdx.x == 0) out[blockIdx.y] = WS(int, off_scal)[slot];
}
__global__ void k_copy_scalar(int dst, int src, int zero_slot, PPLayout lay, void *ws) {
    if (threadIdx.x == 0) {
        WS(int, off_scal)[dst] = WS(int, off_scal)[src];
        if (zero_slot >= 0) WS(int, off_scal)[zero_slot] = 0;
    }
}

// ---------------------------------------------------------------------------
// a13  flow-error filter
// ---------------------------------------------------------------------------
// per-label bbox / count / coordinate sums (+ first raster index and the largest label for the records pass), aggregated
// per workgroup in LDS (lh_slot).  A workgroup covers 2 048 pixels, so its partial coordinate sums fit 32 bits
// (H, W <= 16 384).
struct LabStatsLds {
    int key[LH_SLOTS], ymin[LH_SLOTS], xmin[LH_SLOTS], ymax[LH_SLOTS], xmax[LH_SLOTS], cnt[LH_SLOTS], first[LH_SLOTS];
    unsigned sy[LH_SLOTS], sx[LH_SLOTS];
    int vmax;
};
template <typename T, bool REC>
__device__ __forceinline__ void lab_stats_pass(const T *__restrict__ m, int limit, LabStatsLds &L, const PPLayout &lay, void *ws,
                                               int32_t *copy_to = nullptr) {
    const int i = threadIdx.x;
    L.key[i] = 0; L.ymin[i] = 0x7FFFFFFF; L.xmin[i] = 0x7FFFFFFF; L.ymax[i] = -1; L.xmax[i] = -1; L.cnt[i] = 0;
    L.first[i] = 0x7FFFFFFF; L.sy[i] = 0; L.sx[i] = 0;
    if (i == 0) L.vmax = 0;
    __syncthreads();
    int *bbox = WS(int, off_bbox), *cnt = WS(int, off_cnt), *first = WS(int, off_first), *scal = WS(int, off_scal);
    unsigned long long *sumy = WS(unsigned long long, off_sumy), *sumx = WS(unsigned long long, off_sumx);
    int vmax = 0;
    for_label_runs(m, lay.HW, lay.W, limit, [&](int lab, int idx, int n, int y, int x) {
        const unsigned long long ay = (unsigned long long)n * y;
        const unsigned long long ax = (unsigned long long)n * x + (unsigned long long)(n * (n - 1) / 2);
        const int sl = lh_slot(L.key, lab);
        if (sl >= 0) {
            atomicMin(&L.ymin[sl], y); atomicMin(&L.xmin[sl], x); atomicMax(&L.ymax[sl], y); atomicMax(&L.xmax[sl], x + n - 1);
            atomicAdd(&L.cnt[sl], n);
            atomicAdd(&L.sy[sl], (unsigned)ay); atomicAdd(&L.sx[sl], (unsigned)ax);
            if (REC) atomicMin(&L.first[sl], idx);
        } else {
            int *bb = bbox + 4 * lab;
            atomicMin(&bb[0], y); atomicMin(&bb[1], x); atomicMax(&bb[2], y); atomicMax(&bb[3], x + n - 1);
            atomicAdd(&cnt[lab], n);
            atomicAdd(&sumy[lab], ay); atomicAdd(&sumx[lab], ax);
            if (REC) atomicMin(&first[lab], idx);
        }
        vmax = max(vmax, lab);
    }, copy_to);
    if (REC && vmax > 0) atomicMax(&L.vmax, vmax);
    __syncthreads();
    const int lab = L.key[i];
    if (lab) {
        int *bb = bbox + 4 * lab;
        atomicMin(&bb[0], L.ymin[i]); atomicMin(&bb[1], L.xmin[i]); atomicMax(&bb[2], L.ymax[i]); atomicMax(&bb[3], L.xmax[i]);
        atomicAdd(&cnt[lab], L.cnt[i]);
        atomicAdd(&sumy[lab], (unsigned long long)L.sy[i]); atomicAdd(&sumx[lab], (unsigned long long)L.sx[i]);
        if (REC) atomicMin(&first[lab], L.first[i]);
    }
    if (REC && i == 0 && L.vmax > 0) atomicMax(&scal[SC_VMAX], L.vmax);
}
// copy_to_tmp: also copy the map into the per-tile scratch plane the hole fill reads (was a kernel of its own)
__global__ void k_lab_stats(const int32_t *__restrict__ masks, int copy_to_tmp, PPLayout lay, void *ws) {
    __shared__ LabStatsLds L;
    lab_stats_pass<int32_t, false>(masks + (size_t)blockIdx.y * lay.HW, 0x7FFFFFFF, L, lay, ws,
                                   copy_to_tmp ? WS(int32_t, off_tmp) : nullptr);
}

__global__ void k_init_stats(PPLayout lay, void *ws) {
    int v = blockIdx.x * NTHR + threadIdx.x;
    if (v >= lay.L) return;
    int *bb = WS(int, off_bbox) + 4 * v;
    bb[0] = 0x7FFFFFFF; bb[1] = 0x7FFFFFFF; bb[2] = -1; bb[3] = -1;
    WS(int, off_cnt)[v] = 0;
    WS(unsigned long long, off_sumy)[v] = 0;
    WS(unsigned long long, off_sumx)[v] = 0;
    WS(unsigned long long, off_d2)[v] = 0xFFFFFFFFFFFFFFFFull;
    WS(int, off_center)[v] = 0x7FFFFFFF;
    WS(int, off_flag)[v] = 0;
    WS(int, off_first)[v] = 0x7FFFFFFF;
}

// one launch for everything a stage needs initialised (was 3-5 k_fill_i32 / k_init_stats / k_zero_f64 launches):
//   PPI_SCAL  the PP_NSCAL scalars = 0          PPI_STATS  the per-label tables of k_init_stats
//   PPI_PAD   h1 and M1 (padded frames) = 0     PPI_T      the two fp64 diffusion planes = 0
//   PPI_CLS   the class-vote table = 0          PPI_NLAB   scalar SC_NLAB = 0 only
#define PPI_SCAL 1
#define PPI_STATS 2
#define PPI_PAD 4
#define PPI_T 8
#define PPI_CLS 16
#define PPI_NLAB 32
__global__ void k_pp_init(unsigned what, PPLayout lay, void *ws) {
    const int i = blockIdx.x * NTHR + threadIdx.x;
    if ((what & PPI_SCAL) && i < PP_NSCAL) WS(int, off_scal)[i] = 0;
    if ((what & PPI_NLAB) && i == 0) WS(int, off_scal)[SC_NLAB] = 0;
    if ((what & PPI_STATS) && i < lay.L) {
        int *bb = WS(int, off_bbox) + 4 * i;
        bb[0] = 0x7FFFFFFF; bb[1] = 0x7FFFFFFF; bb[2] = -1; bb[3] = -1;
        WS(int, off_cnt)[i] = 0;
        WS(unsigned long long, off_sumy)[i] = 0;
        WS(unsigned long long, off_sumx)[i] = 0;
        WS(unsigned long long, off_d2)[i] = 0xFFFFFFFFFFFFFFFFull;
        WS(int, off_center)[i] = 0x7FFFFFFF;
        WS(int, off_flag)[i] = 0;
        WS(int, off_first)[i] = 0x7FFFFFFF;
    }
    if ((what & PPI_PAD) && i < lay.HWp) { WS(int, off_h1)[i] = 0; WS(int, off_M1)[i] = 0; }
    if ((what & PPI_T) && i < 2 * lay.THW) WS(double, off_T)[i] = 0.0;
    if ((what & PPI_CLS) && i < lay.L * PP_MAXCLS) WS(int, off_cls)[i] = 0;
}
static void pp_init(unsigned what, int nT, const PPLayout &lay, void *ws, hipStream_t s) {
    int n = 1;
    if (what & PPI_SCAL) n = n > PP_NSCAL ? n : PP_NSCAL;
    if (what & PPI_STATS) n = n > lay.L ? n : lay.L;
    if (what & PPI_PAD) n = n > lay.HWp ? n : lay.HWp;
    if (what & PPI_T) n = n > 2 * lay.THW ? n : 2 * lay.THW;
    if (what & PPI_CLS) n = n > lay.L * PP_MAXCLS ? n : lay.L * PP_MAXCLS;
    PP_LAUNCH(k_pp_init, dim3(cpx_cdiv(n, NTHR), nT), dim3(NTHR), 0, s, what, lay, ws);
}

__device__ __forceinline__ double center_d2(int y, int x, const int *bb, int n,
                                            unsigned long long sy, unsigned long long sx) {
    // yi, xi are bbox-relative; ymed = yi.mean(), xmed = xi.mean() (exact int sums / n)
    double ymed = (double)((long long)sy - (long long)n * bb[0]) / (double)n;
    double xmed = (double)((long long)sx - (long long)n * bb[1]) / (double)n;
    double dx = (double)(x - bb[1]) - xmed;
    double dy = (double)(y - bb[0]) - ymed;
    double a = dx * dx;
    double b = dy * dy;
    return a + b;
}

__global__ void k_center_d2(const int32_t *__restrict__ masks, PPLayout lay, void *ws) {
    __shared__ int key[LH_SLOTS];
    __shared__ unsigned long long vmin[LH_SLOTS];
    key[threadIdx.x] = 0; vmin[threadIdx.x] = 0xFFFFFFFFFFFFFFFFull;
    __syncthreads();
    const int *bbox = WS(int, off_bbox), *cnt = WS(int, off_cnt);
    {   // per label (the grid has more threads than labels): n_iter = 2 * max(bbox extent sum), largest occupied label
        const int v = blockIdx.x * NTHR + threadIdx.x + 1;
        if (v < lay.L && cnt[v] > 0) {
            const int *bb = bbox + 4 * v;
            const int ext = (bb[2] - bb[0] + 1) + (bb[3] - bb[1] + 1) + 2;
            atomicMax(&WS(int, off_scal)[SC_NITER], 2 * ext);
            atomicMax(&WS(int, off_scal)[SC_VMAX], v);
        }
    }
    const unsigned long long *sumy = WS(unsigned long long, off_sumy), *sumx = WS(unsigned long long, off_sumx);
    unsigned long long *d2min = WS(unsigned long long, off_d2);
    for_label_runs(masks + (size_t)blockIdx.y * lay.HW, lay.HW, lay.W, 0x7FFFFFFF,
                   [&](int lab, int, int n, int y, int x) {
                       unsigned long long best = 0xFFFFFFFFFFFFFFFFull;      // d2 >= 0: the bit pattern orders like the value
                       for (int i = 0; i < n; ++i) {
                           const double d2 = center_d2(y, x + i, bbox + 4 * lab, cnt[lab], sumy[lab], sumx[lab]);
                           best = min(best, (unsigned long long)__double_as_longlong(d2));
                       }
                       const int sl = lh_slot(key, lab);
                       if (sl >= 0) atomicMin(&vmin[sl], best); else atomicMin(&d2min[lab], best);
                   });
    __syncthreads();
    if (key[threadIdx.x]) atomicMin(&d2min[key[threadIdx.x]], vmin[threadIdx.x]);
}

__global__ void k_center_pick(const int32_t *__restrict__ masks, PPLayout lay, void *ws) {
    int idx = blockIdx.x * NTHR + threadIdx.x;
    if (idx >= lay.HW) return;
    int lab = masks[(size_t)blockIdx.y * lay.HW + idx];
    if (lab <= 0) return;
    int y = idx / lay.W, x = idx - y * lay.W;
    int *bb = WS(int, off_bbox) + 4 * lab;
    double d2 = center_d2(y, x, bb, WS(int, off_cnt)[lab],
                          WS(unsigned long long, off_sumy)[lab], WS(unsigned long long, off_sumx)[lab]);
    if ((unsigned long long)__double_as_longlong(d2) == WS(unsigned long long, off_d2)[lab])
        atomicMin(&WS(int, off_center)[lab], idx);
}

// heat diffusion from the centre inside one label (fp64 Jacobi, 9-neighbour mean
// summed in neighbour order 0..8 then / 9).  One workgroup per (label, tile).
#define DIFF_LDS_CELLS 2944     // LDS of the first launch: T ping-pong (46 KB) + the cell list (6 KB) = 52 KB, i.e. THREE workgroups of 256 threads per CU.  The size is
                                // an occupancy choice: 3584 cells (64.5 KB, two per CU = 512 slots for the ~650 labels of the bench's 8-tile batch) meant a second round,
                                // 43.6 us; 2944: 32.9 us; 2048 (36 KB, four per CU): 42.5 us -- the dispatcher packs a CU to its limit before it takes the next one, so
                                // 650 workgroups at four per CU leave a third of the chip idle and share 160 CUs' LDS pipes.
#define DIFF_SMALL_CELLS 2048   // (bh+2)*(bw+2) <= this -> first launch; larger boxes go to the second launch (k_diffuse<true>: 1024 threads per label, which is also
                                // faster for them: nine 52 x 52 labels per tile took 723 us in the first launch, 327 us in the second).  43 x 43 pixels: above the
                                // boxes of ordinary nuclei, which need the first launch's many slots.
                                // (A cell list in raster order instead of the order an LDS atomic hands out slots changed nothing: 34.2 us.)
// s / 9.0 correctly rounded without the IEEE division sequence (v_div_scale / v_rcp_f64 / v_div_fmas / v_div_fixup):
// q = RN(s * c) with c = RN(1/9) is within 1 ulp of s / 9, r = s - 9 q is exact in one fma, and RN(q + r * c) is the
// correctly rounded quotient (Markstein's FMA division finish; a quotient of two doubles is never a rounding tie).
// The values here are >= 0 and far from the subnormal range (9^-n_iter).  Bit-identical to torch's mean over the
// 9 neighbours (sum / 9), which the stage-wise parity tests check.
__device__ __forceinline__ double div9(double s) {
    const double c = 1.0 / 9.0;
    const double q = s * c;
    const double r = fma(-9.0, q, s);
    return fma(r, c, q);
}

// BIG: the second launch of the stage, for the labels the first one leaves out (padded box above DIFF_SMALL_CELLS): 1024 threads and
// 147 KB of LDS per workgroup (one per CU) take boxes of up to DIFF_BIG_CELLS cells (88 x 88) through the same LDS path, boxes of up to
// DIFF_HUGE_CELLS (133 x 133) through a one-plane form of it; only what is larger still walks the global planes.  Until round 5 EVERY box above the small limit did, at ~30 us per iteration: nine labels of
// 53 x 53 pixels in a tile cost 5.3 ms where nine of 52 x 52 cost 0.72 (tools/diffuse_size_scan.py) -- a cliff in front of exactly the
// merged, oversized instances the flow-error filter exists to remove.
#define DIFF_BIG_CELLS 8192
#define DIFF_BIG_THREADS 1024
#define DIFF_HUGE_PER_THREAD 18
#define DIFF_HUGE_CELLS (DIFF_HUGE_PER_THREAD * DIFF_BIG_THREADS - 96)    // 18 336 cells (a 133 x 133 box): one plane of doubles = 143 KB
#define DIFF_BIG_LDS ((DIFF_HUGE_CELLS * 8 > DIFF_BIG_CELLS * 18 ? DIFF_HUGE_CELLS * 8 : DIFF_BIG_CELLS * 18) + 16)
template <bool BIG>
__global__ void __launch_bounds__(BIG ? DIFF_BIG_THREADS : NTHR) k_diffuse(const int32_t *__restrict__ masks, PPLayout lay, void *ws) {
    constexpr int CAP = BIG ? DIFF_BIG_CELLS : DIFF_LDS_CELLS;
    extern __shared__ __attribute__((aligned(16))) char diff_dyn[];
    __shared__ __attribute__((aligned(16))) char diff_static[BIG ? 16 : DIFF_LDS_CELLS * 18 + 16];
    char *raw = BIG ? diff_dyn : diff_static;
    double *sT = reinterpret_cast<double *>(raw);                                   // two planes of CAP cells
    unsigned short *sIdx = reinterpret_cast<unsigned short *>(raw + (size_t)CAP * 16);
    int &sN = *reinterpret_cast<int *>(raw + (size_t)CAP * 18);
    const int nthr = blockDim.x;
    // persistent over labels: a fixed grid of workgroups strides through 1..vmax (k_center_d2 left the largest
    // occupied label in SC_VMAX) instead of one 60 KB-LDS workgroup per POSSIBLE label (L-1 = H*W/11 of them,
    // ~99 % of which exited at once but still had to be dispatched and kept the GEMM workgroups off their CUs)
    const int vmax = min(WS(int, off_scal)[SC_VMAX], lay.L - 1);
  for (int lab = blockIdx.x + 1; lab <= vmax; lab += gridDim.x) {
    __syncthreads();                                     // the previous label of this workgroup is done with sT / sL
    int n = WS(int, off_cnt)[lab];
    if (n <= 0) continue;
    const int *bb = WS(int, off_bbox) + 4 * lab;
    const int y0 = bb[0], x0 = bb[1];
    const int bh = bb[2] - y0 + 1, bw = bb[3] - x0 + 1;
    const int ph = bh + 2, pw = bw + 2, cells = ph * pw;
    const int niter = WS(int, off_scal)[SC_NITER];
    const int32_t *m = masks + (size_t)blockIdx.y * lay.HW;
    const int cidx = WS(int, off_center)[lab];
    const int cy = cidx / lay.W - y0 + 1, cx = cidx % lay.W - x0 + 1;
    double *Tg = WS(double, off_T);                      // [THW] final T (shared by all labels)
    const int off9[9] = {0, -pw, pw, -1, 1, -pw - 1, -pw + 1, pw - 1, pw + 1};
    if (BIG ? cells <= DIFF_SMALL_CELLS : cells > DIFF_SMALL_CELLS) continue;      // the other launch's label (uniform: no barrier is skipped by a part of the workgroup)
    if (cells <= CAP) {
        // Cells outside the label hold 0.0 in both planes for the whole run (only label cells are ever written), and
        // s + 0.0 == s exactly, so the 9-term sum needs no neighbour flags: the label's cells are compacted once into
        // sIdx and every iteration is 9 LDS reads + 8 adds + the division per label cell, with ONE barrier (the
        // reference's `T[centre] += 1` in front of iteration it+1 is applied by the centre's owner when it writes
        // iteration it's value -- the same addition on the same value).
        if (threadIdx.x == 0) sN = 0;
        for (int c = threadIdx.x; c < cells; c += nthr) {
            sT[c] = 0.0;
            sT[CAP + c] = 0.0;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cells; c += nthr) {
            int ly = c / pw, lx = c - ly * pw;
            int gy = y0 + ly - 1, gx = x0 + lx - 1;
            if (ly >= 1 && ly <= bh && lx >= 1 && lx <= bw && m[gy * lay.W + gx] == lab)
                sIdx[atomicAdd(&sN, 1)] = (unsigned short)c;
        }
        const int cc = cy * pw + cx;
        if (threadIdx.x == 0 && niter > 0) sT[cc] = 1.0;
        __syncthreads();
        const int nl = sN;
        const int c0 = (int)threadIdx.x < nl ? (int)sIdx[threadIdx.x] : -1;
        int cur = 0;
        for (int it = 0; it < niter; ++it) {
            const double *To = sT + cur * CAP;
            double *Tn = sT + (cur ^ 1) * CAP;
            const bool more = it + 1 < niter;
            for (int j = threadIdx.x; j < nl; j += nthr) {
                const int c = j == (int)threadIdx.x ? c0 : (int)sIdx[j];
                double s = To[c];
#pragma unroll
                for (int k = 1; k < 9; ++k) s = s + To[c + off9[k]];
                double v = div9(s);
                if (c == cc && more) v += 1.0;
                Tn[c] = v;
            }
            __syncthreads();
            cur ^= 1;
        }
        const double *Tf = sT + cur * CAP;
        for (int j = threadIdx.x; j < nl; j += nthr) {
            const int c = sIdx[j];
            int ly = c / pw, lx = c - ly * pw;
            Tg[(y0 + ly) * lay.TW + (x0 + lx)] = Tf[c];     // padded coords: (gy+1, gx+1)
        }
    } else if (BIG && cells <= DIFF_HUGE_CELLS) {
        // ONE plane of doubles in the workgroup's 160 KB (boxes of up to ~140 x 140 pixels): a thread keeps the membership bits and the new values of its
        // <= DIFF_HUGE_PER_THREAD cells in registers across a barrier, then writes them -- two barriers per Jacobi iteration instead of one, the same
        // update on the same values.  (Round 5; these labels walked the global planes before: four 97 x 97 labels per tile 9 ms per 8-tile batch.)
        for (int c = threadIdx.x; c < cells; c += nthr) sT[c] = 0.0;
        __syncthreads();
        unsigned mine = 0;                                      // bit k: cell threadIdx.x + k * nthr belongs to the label
#pragma unroll
        for (int k = 0; k < DIFF_HUGE_PER_THREAD; ++k) {
            const int c = threadIdx.x + k * nthr;
            if (c < cells) {
                const int ly = c / pw, lx = c - ly * pw;
                if (ly >= 1 && ly <= bh && lx >= 1 && lx <= bw && m[(y0 + ly - 1) * lay.W + (x0 + lx - 1)] == lab) mine |= 1u << k;
            }
        }
        const int cc = cy * pw + cx;
        if (threadIdx.x == 0 && niter > 0) sT[cc] = 1.0;
        __syncthreads();
        for (int it = 0; it < niter; ++it) {
            const bool more = it + 1 < niter;
            double v[DIFF_HUGE_PER_THREAD];
#pragma unroll
            for (int k = 0; k < DIFF_HUGE_PER_THREAD; ++k) {
                v[k] = 0.0;
                if (mine & (1u << k)) {
                    int c = threadIdx.x + k * nthr;
                    // (opaque to the optimiser: otherwise the 8 x 20 neighbour addresses are hoisted out of the iteration loop as invariants and the
                    // kernel spills 119 of the 128 registers a 1024-thread workgroup leaves it)
                    asm volatile("" : "+v"(c));
                    double s_ = sT[c];
#pragma unroll
                    for (int q = 1; q < 9; ++q) s_ = s_ + sT[c + off9[q]];
                    v[k] = div9(s_);
                    if (c == cc && more) v[k] += 1.0;
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < DIFF_HUGE_PER_THREAD; ++k)
                if (mine & (1u << k)) sT[threadIdx.x + k * nthr] = v[k];
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < DIFF_HUGE_PER_THREAD; ++k)
            if (mine & (1u << k)) {
                const int c = threadIdx.x + k * nthr, ly = c / pw, lx = c - ly * pw;
                Tg[(y0 + ly) * lay.TW + (x0 + lx)] = sT[c];
            }
    } else {
        // big bbox: ping-pong directly in the global padded T planes (own pixels only)
        double *T0 = Tg, *T1 = Tg + lay.THW;
        int cur = 0;
        for (int it = 0; it < niter; ++it) {
            double *To = cur ? T1 : T0, *Tn = cur ? T0 : T1;
            if (threadIdx.x == 0) To[(y0 + cy) * lay.TW + (x0 + cx)] += 1.0;
            __syncthreads();
            for (int c = threadIdx.x; c < bh * bw; c += nthr) {
                int ly = c / bw, lx = c - ly * bw;
                int gy = y0 + ly, gx = x0 + lx;
                if (m[gy * lay.W + gx] != lab) continue;
                const int dyv[9] = {0, -1, 1, 0, 0, -1, -1, 1, 1};
                const int dxv[9] = {0, 0, 0, -1, 1, -1, 1, -1, 1};
                double s = To[(gy + 1) * lay.TW + gx + 1];
#pragma unroll
                for (int k = 1; k < 9; ++k) {
                    int yy = gy + dyv[k], xx = gx + dxv[k];
                    if ((unsigned)yy < (unsigned)lay.H && (unsigned)xx < (unsigned)lay.W &&
                        m[yy * lay.W + xx] == lab)
                        s = s + To[(yy + 1) * lay.TW + xx + 1];
                }
                Tn[(gy + 1) * lay.TW + gx + 1] = div9(s);
            }
            __threadfence_block();
            __syncthreads();
            cur ^= 1;
        }
        if (cur == 1) {     // final values sit in T1: copy own pixels to plane 0
            for (int c = threadIdx.x; c < bh * bw; c += nthr) {
                int ly = c / bw, lx = c - ly * bw;
                int gy = y0 + ly, gx = x0 + lx;
                if (m[gy * lay.W + gx] == lab) T0[(gy + 1) * lay.TW + gx + 1] = T1[(gy + 1) * lay.TW + gx + 1];
            }
        }
    }
  }
}

// a13's diffusion: the labels whose padded box fits 52 KB of LDS (three workgroups per CU), then the larger ones
static void pp_launch_diffuse(const int32_t *masks, int nT, const PPLayout &lay, void *ws, hipStream_t s) {
    static CpxOncePerDevice once;
    once([] { (void)hipFuncSetAttribute((const void *)k_diffuse<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DIFF_BIG_LDS); });
    PP_LAUNCH(k_diffuse<false>, dim3(lay.L - 1 < 128 ? lay.L - 1 : 128, nT), dim3(NTHR), 0, s, masks, lay, ws);
    // one 1024-thread workgroup per CU: 256 of them over the batch (32 per tile at 8 tiles, 128 at 2 tiles of 1024 px)
    const int per_tile = nT >= 8 ? 32 : (nT <= 2 ? 128 : 256 / nT);
    PP_LAUNCH(k_diffuse<true>, dim3(lay.L - 1 < per_tile ? lay.L - 1 : per_tile, nT), dim3(DIFF_BIG_THREADS), DIFF_BIG_LDS, s, masks, lay, ws);
}

// scipy.ndimage.mean: per-label sums accumulated in raster order (np.bincount), / count
// The per-pixel errors are the squared differences between the unit flows derived from the diffusion result T (central
// differences, / (1e-60 + norm)) and the network flows / 5.
// One WAVE per label: the lanes take 64 bbox pixels at a time (coalesced), compute (e_y, e_x) -- or (0, 0) for pixels
// of other labels: the errors are squares, and s + 0.0 == s exactly -- into the wave's LDS slab; then every lane runs
// np.bincount's sequential float64 summation over the slab (all lanes read the same address: a broadcast, and the two
// dependent add chains are the only cost).  History: one thread per label chasing three dependent global loads per
// pixel 148 us; lane-by-lane v_readlane broadcast 21 us.
#define FE_SLAB 512                                         // pixels per wave slab (16 B each)
__global__ void __launch_bounds__(NTHR) k_flow_err_label(const int32_t *__restrict__ masks, const float *__restrict__ dP,
                                                         double thr, double *__restrict__ errs_out, PPLayout lay, void *ws) {
    __shared__ double2 slab[NTHR / 64][FE_SLAB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int v = (blockIdx.x * NTHR + threadIdx.x) / 64 + 1;          // wave-uniform label
    if (v >= lay.L) return;
    const int n = WS(int, off_cnt)[v];
    if (n <= 0) { if (errs_out && lane == 0) errs_out[(size_t)blockIdx.y * lay.L + v - 1] = 0.0; return; }
    const int *bb = WS(int, off_bbox) + 4 * v;
    const int y0 = bb[0], x0 = bb[1], bw = bb[3] - x0 + 1, npx = (bb[2] - y0 + 1) * bw;
    const int32_t *m = masks + (size_t)blockIdx.y * lay.HW;
    const double *T = WS(double, off_T);
    const size_t t = blockIdx.y;
    double2 *sl = slab[wv];
    double sy = 0.0, sx = 0.0;
    for (int q0 = 0; q0 < npx; q0 += FE_SLAB) {
        const int cnt = min(FE_SLAB, npx - q0);
        for (int j = lane; j < cnt; j += 64) {
            const int q = q0 + j, ly = q / bw, p = (y0 + ly) * lay.W + x0 + (q - ly * bw);
            double2 val = make_double2(0.0, 0.0);
            if (m[p] == v) {
                const int c = (y0 + ly + 1) * lay.TW + (x0 + (q - ly * bw) + 1);
                const double dy = T[c + lay.TW] - T[c - lay.TW];
                const double dx = T[c + 1] - T[c - 1];
                const double a = dy * dy;
                const double b = dx * dx;
                const double den = 1e-60 + sqrt(a + b);
                const double muy = dy / den, mux = dx / den;
                const double vy = (double)__fdiv_rn(dP[(t * 2 + 0) * lay.HW + p], 5.0f);
                const double vx = (double)__fdiv_rn(dP[(t * 2 + 1) * lay.HW + p], 5.0f);
                const double ey = muy - vy, ex = mux - vx;
                val = make_double2(ey * ey, ex * ex);
            }
            sl[j] = val;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the slab is private to this wave
#pragma unroll 8
        for (int j = 0; j < cnt; ++j) {
            const double2 val = sl[j];
            sy = sy + val.x;
            sx = sx + val.y;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane != 0) return;
    double err = 0.0;
    err = err + sy / (double)n;
    err = err + sx / (double)n;
    WS(double, off_err)[v] = err;
    if (errs_out) errs_out[(size_t)blockIdx.y * lay.L + v - 1] = err;
    WS(int, off_flag)[v] = err > thr ? 1 : 0;
}

__global__ void k_zero_flagged(int32_t *__restrict__ masks, PPLayout lay, void *ws) {
    int idx = blockIdx.x * NTHR + threadIdx.x;
    if (idx >= lay.HW) return;
    size_t t = blockIdx.y;
    int lab = masks[t * lay.HW + idx];
    if (lab > 0 && WS(int, off_flag)[lab]) masks[t * lay.HW + idx] = 0;
}

__global__ void k_zero_f64(size_t off, int n, PPLayout lay, void *ws) {
    int idx = blockIdx.x * NTHR + threadIdx.x;
    if (idx >= n) return;
    ((double *)((char *)ws + (size_t)blockIdx.y * lay.per_tile + off))[idx] = 0.0;
}

// ---------------------------------------------------------------------------
// a14  fill_holes_and_remove_small_masks
// ---------------------------------------------------------------------------
// err_thr_on: labels whose flow error (off_err, written by k_flow_err_label) exceeds err_thr are removed from the map on
// the way -- the flow-error filter's removal pass when cpx_compute_masks chains the stages (one launch fewer).
__global__ void k_count_labels(int32_t *__restrict__ masks, int err_thr_on, double err_thr, PPLayout lay, void *ws) {
    __shared__ int key[LH_SLOTS], vadd[LH_SLOTS], s_max, s_bg;
    key[threadIdx.x] = 0; vadd[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_max = 0; s_bg = 0; }
    __syncthreads();
    int *cnt = WS(int, off_cnt), *scal = WS(int, off_scal);
    int32_t *m = masks + (size_t)blockIdx.y * lay.HW;
    const int base = (blockIdx.x * NTHR + threadIdx.x) * RUN_PX;
    int vmax = 0;
    bool bg = false;
    if (base < lay.HW) {
        int lab[RUN_PX];
        bool changed = false;
        const double *err = WS(double, off_err);
#pragma unroll
        for (int i = 0; i < RUN_PX; ++i) {
            lab[i] = base + i < lay.HW ? m[base + i] : 0;
            if (err_thr_on && lab[i] > 0 && err[lab[i]] > err_thr) { lab[i] = 0; changed = true; }
        }
        if (changed) {
#pragma unroll
            for (int i = 0; i < RUN_PX; ++i)
                if (base + i < lay.HW) m[base + i] = lab[i];
        }
        bg = label_runs(lab, base, lay.HW, lay.W, 0x7FFFFFFF, [&](int l, int, int n, int, int) {
            const int s = lh_slot(key, l);
            if (s >= 0) atomicAdd(&vadd[s], n); else atomicAdd(&cnt[l], n);
            vmax = max(vmax, l);
        });
    }
    if (vmax > 0) atomicMax(&s_max, vmax);
    if (bg) s_bg = 1;
    __syncthreads();
    if (key[threadIdx.x]) atomicAdd(&cnt[key[threadIdx.x]], vadd[threadIdx.x]);
    if (threadIdx.x == 0) {
        if (s_max > 0) atomicMax(&scal[SC_VMAX], s_max);
        if (s_bg) scal[SC_HASBG] = 1;
    }
}

// counts = unique(masks, return_counts=True)[1][1:]; remove label VALUE (i+1) where
// counts[i] < min_size  -- positional indexing, reference quirk kept (see oracle).
// One workgroup per tile; ranks by a parallel prefix over per-thread label slices.
__global__ void __launch_bounds__(1024) k_size_filter(int min_size, PPLayout lay, void *ws) {
    __shared__ int s_part[1024];
    const int vmax = min(WS(int, off_scal)[SC_VMAX], lay.L - 1);
    const int hasbg = WS(int, off_scal)[SC_HASBG];
    const int *cnt = WS(int, off_cnt);
    int *flag = WS(int, off_flag);
    const int per = (vmax + 1024) / 1024;          // labels per thread
    const int lo = 1 + threadIdx.x * per, hi = min(lo + per - 1, vmax);
    int c = 0;
    for (int v = lo; v <= hi; ++v) c += cnt[v] > 0;
    s_part[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {          // inclusive Hillis-Steele scan over the 1024 partial counts
        const int t = threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
        __syncthreads();
        s_part[threadIdx.x] += t;
        __syncthreads();
    }
    int rank = s_part[threadIdx.x] - c;           // exclusive: #present nonzero labels below lo
    for (int v = lo; v <= hi; ++v) {
        if (cnt[v] <= 0) continue;
        ++rank;                                    // 1-based rank among present nonzero labels
        int pos = hasbg ? rank - 1 : rank - 2;     // index into counts[1:]
        if (pos >= 0 && cnt[v] < min_size) atomicExch(&flag[pos + 1], 1);
    }
}

// 64-bit Kogge-Stone flood along a row: spread set bits of o through runs of `fr`
__device__ __forceinline__ unsigned long long hfill(unsigned long long o, unsigned long long fr) {
    unsigned long long x = o, f = fr;
    x |= f & (x << 1); f &= (f << 1);
    x |= f & (x << 2); f &= (f << 2);
    x |= f & (x << 4); f &= (f << 4);
    x |= f & (x << 8); f &= (f << 8);
    x |= f & (x << 16); f &= (f << 16);
    x |= f & (x << 32);
    unsigned long long y = o | x; f = fr;
    y |= f & (y >> 1); f &= (f >> 1);
    y |= f & (y >> 2); f &= (f >> 2);
    y |= f & (y >> 4); f &= (f >> 4);
    y |= f & (y >> 8); f &= (f >> 8);
    y |= f & (y >> 16); f &= (f >> 16);
    y |= f & (y >> 32);
    return y;
}

// The same flood for a label whose box is wider or taller than 64 pixels, up to 256 x 256 (round 5): the wave keeps FOUR rows per lane (row
// lane + 64 j) of FOUR 64-bit words each in registers; the horizontal fill carries across the words of a row (one sweep left to right, one right to
// left), the vertical step takes the rows above / below from the neighbouring lane (lane 0 / 63: from the adjacent row block).  Until round 5
// ONE such label in a tile sent the whole tile to k_fill_serial (every label again, one after the other, from global memory): nine discs of
// radius 32 per tile cost 1.07 ms where nine of radius 28 cost 15 us (tools/chain_size_scan.py).
__device__ __forceinline__ void fill_label_big(int lab, int lane, int y0, int x0, int bh, int bw, const int32_t *__restrict__ src,
                                               int32_t *__restrict__ dst, const PPLayout &lay, int *conflict) {
    typedef unsigned long long u64;
    const int nr = (bh + 63) >> 6, nw = (bw + 63) >> 6;
    u64 fr[4][4], ot[4][4], o[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            fr[j][w] = 0; ot[j][w] = 0; o[j][w] = 0;
            const int r = lane + 64 * j;
            if (j < nr && w < nw && r < bh) {
                const int32_t *row = src + (size_t)(y0 + r) * lay.W + x0 + 64 * w;
                const int n = min(64, bw - 64 * w);
                u64 m = 0, oth = 0;
                for (int c = 0; c < n; ++c) {
                    const int v = row[c];
                    m |= (u64)(v == lab) << c;
                    oth |= (u64)(v != lab && v != 0) << c;
                }
                const u64 wmask = n == 64 ? ~0ull : ((1ull << n) - 1);
                fr[j][w] = ~m & wmask; ot[j][w] = oth;
                // seeds: every free cell of the first / last row, the first / last column of the others
                u64 seed = 0;
                if (r == 0 || r == bh - 1) seed = fr[j][w];
                else {
                    if (w == 0) seed |= 1ull;
                    if (w == nw - 1) seed |= 1ull << ((bw - 1) & 63);
                    seed &= fr[j][w];
                }
                o[j][w] = seed;
            }
        }
    // horizontal fill of one row (4 words): left-to-right sweep with the carry out of bit 63, then right-to-left with the carry out of bit 0
    auto row_fill = [&](u64 (&x)[4], const u64 (&f)[4]) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w > 0) x[w] |= (x[w - 1] >> 63) & f[w] & 1ull;
            x[w] = hfill(x[w], f[w]);
        }
#pragma unroll
        for (int w = 2; w >= 0; --w) {
            const u64 carry = (x[w + 1] & 1ull) << 63;
            if (carry & f[w] & ~x[w]) x[w] = hfill(x[w] | carry, f[w]);
        }
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) row_fill(o[j], fr[j]);
    while (true) {
        bool ch = false;
        u64 nn[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                u64 up = __shfl_up(o[j][w], 1), dn = __shfl_down(o[j][w], 1);
                const u64 up_blk = j > 0 ? __shfl(o[j > 0 ? j - 1 : 0][w], 63) : 0ull;     // row 64 j - 1 lives in lane 63 of the block above
                const u64 dn_blk = j < 3 ? __shfl(o[j < 3 ? j + 1 : 3][w], 0) : 0ull;      // row 64 j + 64 in lane 0 of the block below
                if (lane == 0) up = up_blk;
                if (lane == 63) dn = dn_blk;
                nn[j][w] = o[j][w] | ((up | dn) & fr[j][w]);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            row_fill(nn[j], fr[j]);
#pragma unroll
            for (int w = 0; w < 4; ++w) { ch |= nn[j][w] != o[j][w]; o[j][w] = nn[j][w]; }
        }
        if (!__any(ch)) break;
    }
    bool cf = false;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            u64 holes = fr[j][w] & ~o[j][w];
            cf |= (holes & ot[j][w]) != 0;
            const int r = lane + 64 * j;
            if (r < bh) {
                int32_t *row = dst + (size_t)(y0 + r) * lay.W + x0 + 64 * w;
                while (holes) {
                    const int c = __ffsll((long long)holes) - 1;
                    holes &= holes - 1;
                    row[c] = lab;
                }
            }
        }
    if (cf) *conflict = 1;
}

// Parallel hole fill: one wave per label, lanes = bbox rows, 64-bit row masks (boxes up to 64 x 64; fill_label_big up to 256 x 256).
// Reads `src` (pre-fill copy), writes holes into `masks`.  Exact w.r.t. the
// sequential reference loop iff no hole contains a pixel of another label; that
// case (and bboxes > 256) is flagged and redone sequentially by k_fill_serial.
__global__ void __launch_bounds__(NTHR) k_fill_parallel(int32_t *__restrict__ masks, PPLayout lay,
                                                        void *ws) {
    const int lane = threadIdx.x & 63;
    const int lab = blockIdx.x * (NTHR / 64) + (threadIdx.x >> 6) + 1;
    const int nlab = WS(int, off_scal)[SC_NLAB];
    if (lab > nlab) return;
    const int *bb = WS(int, off_bbox) + 4 * lab;
    if (WS(int, off_cnt)[lab] <= 0) return;
    const int y0 = bb[0], x0 = bb[1], bh = bb[2] - y0 + 1, bw = bb[3] - x0 + 1;
    if (bh > 256 || bw > 256) { if (lane == 0) WS(int, off_scal)[SC_HASBIG] = 1; return; }
    if (bh > 64 || bw > 64) { if (lane == 0) WS(int, off_scal)[SC_HASMID] = 1; return; }     // k_fill_serial's waves take these (fill_label_big: 500 registers)
    const int32_t *src = WS(int32_t, off_tmp);
    int32_t *dst = masks + (size_t)blockIdx.y * lay.HW;
    unsigned long long m = 0, other = 0;
    const unsigned long long wmask = bw == 64 ? ~0ull : ((1ull << bw) - 1);
    if (lane < bh) {
        const int32_t *row = src + (y0 + lane) * lay.W + x0;
        for (int c = 0; c < bw; ++c) {
            int v = row[c];
            m |= (unsigned long long)(v == lab) << c;
            other |= (unsigned long long)(v != lab && v != 0) << c;
        }
    }
    unsigned long long fr = lane < bh ? (~m & wmask) : 0;
    unsigned long long o = 0;
    if (lane < bh) {
        o = (lane == 0 || lane == bh - 1) ? fr : (fr & (1ull | (1ull << (bw - 1))));
        o = hfill(o, fr);
    }
    while (true) {
        unsigned long long up = __shfl_up(o, 1), dn = __shfl_down(o, 1);
        if (lane == 0) up = 0;
        if (lane >= bh - 1) dn = 0;
        unsigned long long nn = hfill(o | ((up | dn) & fr), fr);
        bool ch = nn != o;
        o = nn;
        if (!__any(ch)) break;
    }
    unsigned long long holes = fr & ~o;
    if (holes & other) WS(int, off_scal)[SC_CONFLICT] = 1;
    if (lane < bh) {
        int32_t *row = dst + (y0 + lane) * lay.W + x0;
        while (holes) {
            int c = __ffsll((long long)holes) - 1;
            holes &= holes - 1;
            row[c] = lab;
        }
    }
}

// Sequential exact fallback (one workgroup per tile), only when flagged.
// scratch `out`side map lives in h1 (int per pixel of the padded frame, >= HW ints).
__device__ void fill_serial_body(int32_t *__restrict__ masks, const PPLayout &lay, void *ws) {
    __shared__ int s_changed;
    const int nlab = WS(int, off_scal)[SC_NLAB];
    int32_t *m = masks + (size_t)blockIdx.y * lay.HW;
    const int32_t *src = WS(int32_t, off_tmp);
    int *outm = WS(int, off_h1);
    for (int i = threadIdx.x; i < lay.HW; i += NTHR) m[i] = src[i];   // restart from pre-fill map
    __threadfence_block();
    __syncthreads();
    for (int lab = 1; lab <= nlab; ++lab) {
        const int *bb = WS(int, off_bbox) + 4 * lab;      // find_objects() taken before the loop
        if (WS(int, off_cnt)[lab] <= 0) continue;
        const int y0 = bb[0], x0 = bb[1], bh = bb[2] - y0 + 1, bw = bb[3] - x0 + 1;
        for (int c = threadIdx.x; c < bh * bw; c += NTHR) {
            int ly = c / bw, lx = c - ly * bw;
            bool isl = m[(y0 + ly) * lay.W + x0 + lx] == lab;
            bool edge = ly == 0 || lx == 0 || ly == bh - 1 || lx == bw - 1;
            outm[c] = (!isl && edge) ? 1 : 0;
        }
        __threadfence_block();
        __syncthreads();
        while (true) {
            if (threadIdx.x == 0) s_changed = 0;
            __syncthreads();
            int changed = 0;
            for (int r = threadIdx.x; r < bh; r += NTHR) {       // row sweeps, both directions
                const int32_t *mr = m + (y0 + r) * lay.W + x0;
                int *orow = outm + r * bw;
                int prev = 0;
                for (int c = 0; c < bw; ++c) {
                    bool fr = mr[c] != lab;
                    if (fr && !orow[c] && prev) { orow[c] = 1; changed = 1; }
                    prev = fr ? orow[c] : 0;
                }
                prev = 0;
                for (int c = bw - 1; c >= 0; --c) {
                    bool fr = mr[c] != lab;
                    if (fr && !orow[c] && prev) { orow[c] = 1; changed = 1; }
                    prev = fr ? orow[c] : 0;
                }
            }
            __threadfence_block();
            __syncthreads();
            for (int c = threadIdx.x; c < bw; c += NTHR) {       // column sweeps
                int prev = 0;
                for (int r = 0; r < bh; ++r) {
                    bool fr = m[(y0 + r) * lay.W + x0 + c] != lab;
                    int *o = outm + r * bw + c;
                    if (fr && !*o && prev) { *o = 1; changed = 1; }
                    prev = fr ? *o : 0;
                }
                prev = 0;
                for (int r = bh - 1; r >= 0; --r) {
                    bool fr = m[(y0 + r) * lay.W + x0 + c] != lab;
                    int *o = outm + r * bw + c;
                    if (fr && !*o && prev) { *o = 1; changed = 1; }
                    prev = fr ? *o : 0;
                }
            }
            if (changed) s_changed = 1;
            __threadfence_block();
            __syncthreads();
            int again = s_changed;
            __syncthreads();
            if (!again) break;
        }
        for (int c = threadIdx.x; c < bh * bw; c += NTHR) {
            int ly = c / bw, lx = c - ly * bw;
            int32_t *px = m + (y0 + ly) * lay.W + x0 + lx;
            if (*px != lab && !outm[c]) *px = lab;           // hole -> label
        }
        __threadfence_block();
        __syncthreads();
    }
}
// One workgroup per tile, behind k_fill_parallel: (1) the labels with boxes of 65 .. 256 pixels, four at a time (one per wave, fill_label_big) --
// unless the tile goes to the sequential loop anyway; (2) the sequential loop over ALL labels when a hole of any label holds another label's
// pixels (the one case in which the reference's label-by-label order shows) or a box exceeds 256 pixels.
__global__ void __launch_bounds__(NTHR) k_fill_serial(int32_t *__restrict__ masks, PPLayout lay,
                                                      void *ws) {
    int *scal = WS(int, off_scal);
    const int hasmid = scal[SC_HASMID];
    if (!(scal[SC_CONFLICT] || scal[SC_HASBIG] || hasmid)) return;
    if (hasmid && !(scal[SC_CONFLICT] || scal[SC_HASBIG])) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nlab = scal[SC_NLAB];
        const int32_t *src = WS(int32_t, off_tmp);
        int32_t *dst = masks + (size_t)blockIdx.y * lay.HW;
        int k = 0;
        for (int lab = 1; lab <= nlab; ++lab) {
            if (WS(int, off_cnt)[lab] <= 0) continue;
            const int *bb = WS(int, off_bbox) + 4 * lab;
            const int y0 = bb[0], x0 = bb[1], bh = bb[2] - y0 + 1, bw = bb[3] - x0 + 1;
            if (bh <= 64 && bw <= 64) continue;
            if ((k++ & (NTHR / 64 - 1)) != wave) continue;
            fill_label_big(lab, lane, y0, x0, bh, bw, src, dst, lay, &scal[SC_CONFLICT]);
        }
        __threadfence_block();
    }
    __syncthreads();
    if (!(__hip_atomic_load(&scal[SC_CONFLICT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || scal[SC_HASBIG])) return;
    fill_serial_body(masks, lay, ws);
}

// ---------------------------------------------------------------------------
// a15  class vote     a16  border removal     records
// ---------------------------------------------------------------------------
__global__ void k_class_count(const int32_t *__restrict__ masks, const float *__restrict__ logits,
                              int ncls, PPLayout lay, void *ws) {
    __shared__ int key[LH_SLOTS], vadd[LH_SLOTS];           // key = label * PP_MAXCLS + class + 1
    key[threadIdx.x] = 0; vadd[threadIdx.x] = 0;
    __syncthreads();
    int idx = blockIdx.x * NTHR + threadIdx.x;
    size_t t = blockIdx.y;
    int *cls = WS(int, off_cls);
    const int lab = idx < lay.HW ? masks[t * lay.HW + idx] : 0;
    if (lab > 0) {
        const float *lg = logits + t * ncls * lay.HW + idx;
        float best = lg[0];
        int bi = 0;
        for (int c = 1; c < ncls; ++c) {
            float v = lg[(size_t)c * lay.HW];
            if (v > best) { best = v; bi = c; }        // first maximum wins (np.argmax)
        }
        const int k = lab * PP_MAXCLS + bi;
        const int sl = lh_slot(key, k + 1);
        if (sl >= 0) atomicAdd(&vadd[sl], 1); else atomicAdd(&cls[k], 1);
    }
    __syncthreads();
    if (key[threadIdx.x]) atomicAdd(&cls[key[threadIdx.x] - 1], vadd[threadIdx.x]);
}

__global__ void k_class_pick(int ncls, PPLayout lay, void *ws) {
    int v = blockIdx.x * NTHR + threadIdx.x;
    if (v >= lay.L) return;
    const int *c = WS(int, off_cls) + v * PP_MAXCLS;
    int best = c[0], bi = 0;
    for (int k = 1; k < ncls; ++k) if (c[k] > best) { best = c[k]; bi = k; }
    WS(int, off_remap)[v] = v == 0 ? 0 : bi;
}

__global__ void k_class_write(const int32_t *__restrict__ masks, uint8_t *__restrict__ cm,
                              uint16_t *__restrict__ masks_u16, PPLayout lay, void *ws) {
    int idx = blockIdx.x * NTHR + threadIdx.x;
    if (idx >= lay.HW) return;
    size_t t = blockIdx.y;
    int lab = masks[t * lay.HW + idx];
    cm[t * lay.HW + idx] = lab > 0 ? (uint8_t)WS(int, off_remap)[lab] : 0;
    if (masks_u16) masks_u16[t * lay.HW + idx] = (uint16_t)lab;      // the uint16 id map of cpx_compute_masks, same pass
}

__global__ void k_border_flag(const int32_t *__restrict__ masks, PPLayout lay, void *ws) {
    int i = blockIdx.x * NTHR + threadIdx.x;
    const int H = lay.H, W = lay.W;
    if (i >= 2 * (H + W)) return;
    int y, x;
    if (i < W) { y = 0; x = i; }
    else if (i < 2 * W) { y = H - 1; x = i - W; }
    else if (i < 2 * W + H) { y = i - 2 * W; x = 0; }
    else { y = i - 2 * W - H; x = W - 1; }
    int lab = masks[(size_t)blockIdx.y * lay.HW + y * W + x];
    if (lab > 0 && lab < 65536) WS(int, off_flag)[lab] = 1;
}

__global__ void k_border_zero(int32_t *__restrict__ masks, uint8_t *__restrict__ cm, PPLayout lay,
                              void *ws) {
    int idx = blockIdx.x * NTHR + threadIdx.x;
    if (idx >= lay.HW) return;
    size_t t = blockIdx.y;
    int lab = masks[t * lay.HW + idx];
    if (lab > 0 && lab < 65536 && WS(int, off_flag)[lab]) {
        masks[t * lay.HW + idx] = 0;
        if (cm) cm[t * lay.HW + idx] = 0;
    }
}

__global__ void k_to_u16(const int32_t *__restrict__ masks, uint16_t *__restrict__ out, size_t n) {
    size_t idx = (size_t)blockIdx.x * NTHR + threadIdx.x;
    if (idx < n) out[idx] = (uint16_t)masks[idx];
}

__global__ void k_rec_stats(const uint16_t *__restrict__ masks, int32_t *__restrict__ counts, PPLayout lay, void *ws) {
    __shared__ LabStatsLds L;
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[blockIdx.y] = 0;      // k_rec_write raises it with atomicMax
    lab_stats_pass<uint16_t, true>(masks + (size_t)blockIdx.y * lay.HW, lay.L, L, lay, ws);
}

__global__ void k_rec_write(const uint8_t *__restrict__ cm, int max_rec, cpx_record *__restrict__ rec,
                            int32_t *__restrict__ counts, PPLayout lay, void *ws) {
    int v = blockIdx.x * NTHR + threadIdx.x + 1;
    if (v >= lay.L) return;
    int n = WS(int, off_cnt)[v];
    if (n <= 0) return;
    // records are emitted at slot label-1 (labels are contiguous after renumber);
    // counts[tile] = max label
    if (v - 1 < max_rec) {
        cpx_record r;
        const int *bb = WS(int, off_bbox) + 4 * v;
        r.tile = blockIdx.y; r.label = v;
        r.cls = cm ? cm[(size_t)blockIdx.y * lay.HW + WS(int, off_first)[v]] : 0;
        r.area = n; r.y0 = bb[0]; r.x0 = bb[1]; r.y1 = bb[2] + 1; r.x1 = bb[3] + 1;
        r.sum_y = (int64_t)WS(unsigned long long, off_sumy)[v];
        r.sum_x = (int64_t)WS(unsigned long long, off_sumx)[v];
        rec[(size_t)blockIdx.y * max_rec + v - 1] = r;
    }
    atomicMax(&counts[blockIdx.y], v);
}

// ===========================================================================
// the fused chain of cpx_compute_masks (round 4): 23 launches (22 until round 5 gave the diffusion a second launch for large boxes) where the stage-wise sequence needs 39
// ===========================================================================
// What made the chain long was never the pixel work (a 2 MB pass is ~5 us of dispatch) but its per-tile global dependencies:
// a stage's per-label tables must be complete before its "light" step (rank the labels by first appearance, apply the size
// filter, write the records) and that step before the next pixel pass -- each a launch of its own, each stage preceded by
// an initialisation launch and followed by a relabel pass.  Here:
//   * ONE initialisation: every stage has its own copy of the per-label tables (lay_set), all cleared by k_prep_flow<INIT>;
//   * no relabel passes: labels are mapped through the previous stage's table on load by the NEXT stage's first pixel pass and
//     written back; removals (too-big labels, size-filtered labels) are entries 0 of that table, so k_big_first and
//     k_first<true> are gone too (the counting pass collects the first raster index itself);
//   * the Euler loop feeds the histogram (k_hist); the size filter's flags and the ranks are one light launch, not two;
//   * the class map, the uint16 id map and the per-cell statistics of cpx_instance_records leave in one final pass.
// Results are bit-identical to the stage-wise entry points (same arithmetic, same tie rules; tests/test_gpu_postproc.py).
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The light per-tile step behind a pixel pass ("tail": rank the labels, apply the size filter, write the records) is the
// SAME kernel launched a second time with tail = 1 on a (1, nT) grid.  A first version let the last workgroup of the pixel
// pass run it (a ticket behind __threadfence): bit-identical, 16 launches -- and 1 181 us per batch against 347 us, because an
// agent-scope release on this part writes back the XCD's whole L2 (the eight L2s are not coherent with each other) and every
// workgroup of every pass paid for one: k_fill 689 us, the class vote 139 us (profiles/r04_post_fused_first_build.csv).
// A launch boundary is the cheap device-wide release here (~5 us); what the fused chain keeps is everything else.
#define TAIL_LDS 1024        // up to this many labels the tail ranks by comparing every label with every other through an LDS copy (the bench's tiles: ~80)
// remap[v] = 1 + #{kept j : first[j] < first[v]} for kept v in 1..vmax, 0 otherwise (fastremap.renumber on the map with the
// removed labels zeroed); *nlab = number of kept labels.  One workgroup (NTHR threads); `removed(v)` decides per label.
// More labels than TAIL_LDS (dense 512 / 1024-px tiles) are ranked by position, not by comparison: first[] are distinct pixel indices, so a
// bitmap of the tile with the kept labels' first pixels set, the prefix counts of its words, and rank = prefix + popcount below the bit --
// O(HW / 32 + n) for the workgroup instead of O(n^2).  (Until round 5: every label against every other, through LDS chunks above 6 144
// labels: 8 649 labels in a 1024-px tile cost 3.7 ms per call, six calls per chain = 22 of the chain's 24.5 ms, tools/chain_size_scan.py.)
// `bits`: 2 x cdiv(hw, 32) words of the tile's workspace (off_rankbits); `scratch`: the stage's unused rank table.
template <typename R>
__device__ __forceinline__ void tail_rank(int vmax, const int *first, int *remap, int *nlab_slot, int *scratch, unsigned *bits, int hw, R &&removed) {
    __shared__ int s_first[TAIL_LDS + 1];
    __shared__ int s_part[NTHR];
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    int kept = 0;
    if (vmax <= TAIL_LDS) {
        for (int v = 1 + (int)threadIdx.x; v <= vmax; v += NTHR) {
            int f = ld_agent(&first[v]);
            if (f != 0x7FFFFFFF && removed(v)) f = 0x7FFFFFFF;
            s_first[v] = f;
        }
        __syncthreads();
        for (int v = 1 + (int)threadIdx.x; v <= vmax; v += NTHR) {
            const int f = s_first[v];
            if (f == 0x7FFFFFFF) { remap[v] = 0; continue; }
            int r = 1;
            for (int j = 1; j <= vmax; ++j) r += s_first[j] < f;
            remap[v] = r;
            ++kept;
        }
    } else {
        const int nw = (hw + 31) >> 5, cw = (nw + NTHR - 1) / NTHR;        // words; words per thread (one contiguous chunk each)
        unsigned *pref = bits + nw;
        for (int w = threadIdx.x; w < nw; w += NTHR) bits[w] = 0u;
        __threadfence_block();
        __syncthreads();
        for (int v = 1 + (int)threadIdx.x; v <= vmax; v += NTHR) {
            int f = ld_agent(&first[v]);
            if (f != 0x7FFFFFFF && removed(v)) f = 0x7FFFFFFF;
            scratch[v] = f;                              // read back by this thread only
            if (f != 0x7FFFFFFF) atomicOr(&bits[f >> 5], 1u << (f & 31));
        }
        __threadfence_block();
        __syncthreads();
        int run = 0;
        for (int w = threadIdx.x * cw; w < min((int)(threadIdx.x + 1) * cw, nw); ++w) {
            pref[w] = (unsigned)run;                     // kept first-pixels in this thread's chunk before word w
            run += __popc(__hip_atomic_load(&bits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        s_part[threadIdx.x] = run;
        __threadfence_block();
        __syncthreads();
        for (int o = 1; o < NTHR; o <<= 1) {             // inclusive scan of the chunk totals
            const int tv = (int)threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
            __syncthreads();
            s_part[threadIdx.x] += tv;
            __syncthreads();
        }
        for (int v = 1 + (int)threadIdx.x; v <= vmax; v += NTHR) {
            const int f = scratch[v];
            if (f == 0x7FFFFFFF) { remap[v] = 0; continue; }
            const int w = f >> 5, t = w / cw;
            const unsigned below = __hip_atomic_load(&bits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ((1u << (f & 31)) - 1u);
            remap[v] = 1 + (t > 0 ? s_part[t - 1] : 0) + (int)__hip_atomic_load(&pref[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __popc(below);
            ++kept;
        }
    }
    if (kept) atomicAdd(&s_n, kept);
    __syncthreads();
    if (threadIdx.x == 0) *nlab_slot = s_n;
}

// a12 tail: label gather + per-label count + first raster index; the last workgroup removes the too-big labels and ranks
// (k_gather + k_big_first + k_renumber_rank; the relabel rides on k_lab_stats_r)
__global__ void __launch_bounds__(NTHR) k_gather_f(const int32_t *__restrict__ p_final, int32_t *__restrict__ masks, double big,
                                                   int tail, PPLayout lay, void *ws) {
    __shared__ int key[LH_SLOTS], vadd[LH_SLOTS], vmin[LH_SLOTS];
    if (tail) {
        int *scal = WS(int, off_scal), *cnt = WS(int, off_cnt);
        const int vmax = min(scal[SC_NSEEDS], lay.L - 1);          // labels are the seed ranks 1..nseeds
        tail_rank(vmax, WS(int, off_first), WS(int, off_remap), &scal[SC_NLAB], WS(int, off_rank), WS(unsigned, off_rankbits), lay.HW,
                  [&](int v) { return (double)cnt[v] > big; });
        return;
    }
    key[threadIdx.x] = 0; vadd[threadIdx.x] = 0; vmin[threadIdx.x] = 0x7FFFFFFF;
    __syncthreads();
    const int base = (blockIdx.x * NTHR + threadIdx.x) * RUN_PX;
    const size_t t = blockIdx.y;
    int *cnt = WS(int, off_cnt), *first = WS(int, off_first);
    if (base < lay.HW) {
        const int *M1 = WS(int, off_M1);
        int lab[RUN_PX];
#pragma unroll
        for (int i = 0; i < RUN_PX; ++i) {
            lab[i] = 0;
            if (base + i < lay.HW) {
                const int p = p_final[t * lay.HW + base + i];
                if (p != -1) {
                    int py = (p >> 16) + RPAD, px = (int)(short)(p & 0xFFFF) + RPAD;
                    py = min(max(py, 0), lay.H + RPAD - 1); px = min(max(px, 0), lay.W + RPAD - 1);
                    lab[i] = M1[py * lay.Wp + px];
                }
                masks[t * lay.HW + base + i] = lab[i];
            }
        }
        label_runs(lab, base, lay.HW, lay.W, 0x7FFFFFFF, [&](int l, int idx, int n, int, int) {
            const int sl = lh_slot(key, l);
            if (sl >= 0) { atomicAdd(&vadd[sl], n); atomicMin(&vmin[sl], idx); }
            else { atomicAdd(&cnt[l], n); atomicMin(&first[l], idx); }
        });
    }
    __syncthreads();
    if (key[threadIdx.x]) { atomicAdd(&cnt[key[threadIdx.x]], vadd[threadIdx.x]); atomicMin(&first[key[threadIdx.x]], vmin[threadIdx.x]); }
}

// loads RUN_PX labels of the map; with `remap` the labels are first mapped through the previous stage's table and written back
__device__ __forceinline__ void load_labels(int32_t *__restrict__ m, int base, int HW, const int *__restrict__ remap,
                                            int32_t *__restrict__ copy_to, int (&lab)[RUN_PX]) {
    if (base + RUN_PX <= HW && (reinterpret_cast<size_t>(m + base) & 15) == 0) {
        const int4 a = *reinterpret_cast<const int4 *>(m + base), b = *reinterpret_cast<const int4 *>(m + base + 4);
        lab[0] = a.x; lab[1] = a.y; lab[2] = a.z; lab[3] = a.w; lab[4] = b.x; lab[5] = b.y; lab[6] = b.z; lab[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < RUN_PX; ++i) lab[i] = base + i < HW ? m[base + i] : 0;
    }
    if (remap) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < RUN_PX; ++i)
            if (lab[i] > 0) { const int n = remap[lab[i]]; changed |= n != lab[i]; lab[i] = n; }
        if (changed) {
#pragma unroll
            for (int i = 0; i < RUN_PX; ++i)
                if (base + i < HW) m[base + i] = lab[i];
        }
    }
    if (copy_to) {
#pragma unroll
        for (int i = 0; i < RUN_PX; ++i)
            if (base + i < HW) copy_to[base + i] = lab[i];
    }
}

// k_relabel of the previous stage (through `prev`'s remap table) + k_lab_stats of this one (+ the pre-fill copy)
__global__ void __launch_bounds__(NTHR) k_lab_stats_r(int32_t *__restrict__ masks, int copy_to_tmp, PPLayout prev, PPLayout lay, void *ws) {
    __shared__ LabStatsLds L;
    const int i = threadIdx.x;
    L.key[i] = 0; L.ymin[i] = 0x7FFFFFFF; L.xmin[i] = 0x7FFFFFFF; L.ymax[i] = -1; L.xmax[i] = -1; L.cnt[i] = 0;
    L.first[i] = 0x7FFFFFFF; L.sy[i] = 0; L.sx[i] = 0;
    __syncthreads();
    char *tb = (char *)ws + (size_t)blockIdx.y * lay.per_tile;
    if (blockIdx.x == 0 && i == 0) WS(int, off_scal)[SC_NLAB] = ((const int *)(tb + prev.off_scal))[SC_NLAB];
    int *bbox = WS(int, off_bbox), *cnt = WS(int, off_cnt);
    unsigned long long *sumy = WS(unsigned long long, off_sumy), *sumx = WS(unsigned long long, off_sumx);
    int32_t *m = masks + (size_t)blockIdx.y * lay.HW;
    const int base = (blockIdx.x * NTHR + i) * RUN_PX;
    if (base < lay.HW) {
        int lab[RUN_PX];
        load_labels(m, base, lay.HW, (const int *)(tb + prev.off_remap), copy_to_tmp ? WS(int32_t, off_tmp) : nullptr, lab);
        label_runs(lab, base, lay.HW, lay.W, 0x7FFFFFFF, [&](int l, int, int n, int y, int x) {
            const unsigned long long ay = (unsigned long long)n * y;
            const unsigned long long ax = (unsigned long long)n * x + (unsigned long long)(n * (n - 1) / 2);
            const int sl = lh_slot(L.key, l);
            if (sl >= 0) {
                atomicMin(&L.ymin[sl], y); atomicMin(&L.xmin[sl], x); atomicMax(&L.ymax[sl], y); atomicMax(&L.xmax[sl], x + n - 1);
                atomicAdd(&L.cnt[sl], n);
                atomicAdd(&L.sy[sl], (unsigned)ay); atomicAdd(&L.sx[sl], (unsigned)ax);
            } else {
                int *bb = bbox + 4 * l;
                atomicMin(&bb[0], y); atomicMin(&bb[1], x); atomicMax(&bb[2], y); atomicMax(&bb[3], x + n - 1);
                atomicAdd(&cnt[l], n);
                atomicAdd(&sumy[l], ay); atomicAdd(&sumx[l], ax);
            }
        });
    }
    __syncthreads();
    const int l = L.key[i];
    if (l) {
        int *bb = bbox + 4 * l;
        atomicMin(&bb[0], L.ymin[i]); atomicMin(&bb[1], L.xmin[i]); atomicMax(&bb[2], L.ymax[i]); atomicMax(&bb[3], L.xmax[i]);
        atomicAdd(&cnt[l], L.cnt[i]);
        atomicAdd(&sumy[l], (unsigned long long)L.sy[i]); atomicAdd(&sumx[l], (unsigned long long)L.sx[i]);
    }
}

// the size filter in one launch: count + first index per label (with the flow-error filter's removals applied on the way:
// `errs` = that stage's error table), then in the last workgroup the positional size filter (k_size_filter) and the
// first-appearance ranks with the flagged labels removed (k_first<true> + k_renumber_rank); nlabels_out: the label count
__global__ void __launch_bounds__(NTHR) k_count_labels_f(int32_t *__restrict__ masks, PPLayout errs, int err_thr_on, double err_thr,
                                                         int min_size, int32_t *__restrict__ nlabels_out, int tail, PPLayout lay, void *ws) {
    __shared__ int key[LH_SLOTS], vadd[LH_SLOTS], vmin[LH_SLOTS], s_max, s_bg;
    if (!tail) {
    key[threadIdx.x] = 0; vadd[threadIdx.x] = 0; vmin[threadIdx.x] = 0x7FFFFFFF;
    if (threadIdx.x == 0) { s_max = 0; s_bg = 0; }
    __syncthreads();
    char *tb = (char *)ws + (size_t)blockIdx.y * lay.per_tile;
    int *cnt = WS(int, off_cnt), *first = WS(int, off_first), *scal = WS(int, off_scal);
    int32_t *m = masks + (size_t)blockIdx.y * lay.HW;
    const int base = (blockIdx.x * NTHR + threadIdx.x) * RUN_PX;
    int vmax = 0;
    bool bg = false;
    if (base < lay.HW) {
        int lab[RUN_PX];
        bool changed = false;
        const double *err = (const double *)(tb + errs.off_err);
#pragma unroll
        for (int i = 0; i < RUN_PX; ++i) {
            lab[i] = base + i < lay.HW ? m[base + i] : 0;
            if (err_thr_on && lab[i] > 0 && err[lab[i]] > err_thr) { lab[i] = 0; changed = true; }
        }
        if (changed) {
#pragma unroll
            for (int i = 0; i < RUN_PX; ++i)
                if (base + i < lay.HW) m[base + i] = lab[i];
        }
        bg = label_runs(lab, base, lay.HW, lay.W, 0x7FFFFFFF, [&](int l, int idx, int n, int, int) {
            const int s = lh_slot(key, l);
            if (s >= 0) { atomicAdd(&vadd[s], n); atomicMin(&vmin[s], idx); }
            else { atomicAdd(&cnt[l], n); atomicMin(&first[l], idx); }
            vmax = max(vmax, l);
        });
    }
    if (vmax > 0) atomicMax(&s_max, vmax);
    if (bg) s_bg = 1;
    __syncthreads();
    if (key[threadIdx.x]) { atomicAdd(&cnt[key[threadIdx.x]], vadd[threadIdx.x]); atomicMin(&first[key[threadIdx.x]], vmin[threadIdx.x]); }
    if (threadIdx.x == 0) {
        if (s_max > 0) atomicMax(&scal[SC_VMAX], s_max);
        if (s_bg) atomicExch(&scal[SC_HASBG], 1);
    }
    return;
    }
    int *cnt = WS(int, off_cnt), *first = WS(int, off_first), *scal = WS(int, off_scal);
    // ---- counts = unique(masks, return_counts=True)[1][1:]; remove label VALUE (i + 1) where counts[i] < min_size: positional
    // indexing, the reference's quirk (oracle/dynamics.py).  Ranks by a prefix over per-thread label slices.
    __shared__ int s_part[NTHR];
    const int vm = min(ld_agent(&scal[SC_VMAX]), lay.L - 1);
    const int hasbg = ld_agent(&scal[SC_HASBG]);
    int *flag = WS(int, off_flag);
    const int per = (vm + NTHR) / NTHR;
    const int lo = 1 + (int)threadIdx.x * per, hi = min(lo + per - 1, vm);
    int c = 0;
    for (int v = lo; v <= hi; ++v) c += ld_agent(&cnt[v]) > 0;
    s_part[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < NTHR; o <<= 1) {
        const int tv = (int)threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
        __syncthreads();
        s_part[threadIdx.x] += tv;
        __syncthreads();
    }
    int rank = s_part[threadIdx.x] - c;
    for (int v = lo; v <= hi; ++v) {
        const int n = ld_agent(&cnt[v]);
        if (n <= 0) continue;
        ++rank;
        const int pos = hasbg ? rank - 1 : rank - 2;
        if (pos >= 0 && n < min_size) atomicExch(&flag[pos + 1], 1);
    }
    __threadfence();
    __syncthreads();
    tail_rank(vm, first, WS(int, off_remap), &scal[SC_NLAB], WS(int, off_rank), WS(unsigned, off_rankbits), lay.HW, [&](int v) { return ld_agent(&flag[v]) != 0; });
    if (nlabels_out && threadIdx.x == 0) nlabels_out[blockIdx.y] = scal[SC_NLAB];
}

// class vote with the size filter's relabel on the way (labels -> `prev`'s remap, written back), then in the last workgroup
// the per-label arg-max (k_relabel + k_class_count + k_class_pick)
__global__ void __launch_bounds__(NTHR) k_class_count_f(int32_t *__restrict__ masks, const float *__restrict__ logits, int ncls,
                                                        PPLayout prev, PPLayout lay, void *ws) {
    __shared__ int key[LH_SLOTS], vadd[LH_SLOTS];           // key = label * PP_MAXCLS + class + 1
    key[threadIdx.x] = 0; vadd[threadIdx.x] = 0;
    __syncthreads();
    char *tb = (char *)ws + (size_t)blockIdx.y * lay.per_tile;
    const int idx = blockIdx.x * NTHR + threadIdx.x;
    const size_t t = blockIdx.y;
    int *cls = WS(int, off_cls);
    int lab = 0;
    if (idx < lay.HW) {
        lab = masks[t * lay.HW + idx];
        if (lab > 0) {
            const int n = ((const int *)(tb + prev.off_remap))[lab];
            if (n != lab) masks[t * lay.HW + idx] = n;
            lab = n;
        }
    }
    if (lab > 0 && logits) {
        const float *lg = logits + t * ncls * lay.HW + idx;
        float best = lg[0];
        int bi = 0;
        for (int c = 1; c < ncls; ++c) {
            float v = lg[(size_t)c * lay.HW];
            if (v > best) { best = v; bi = c; }        // first maximum wins (np.argmax)
        }
        const int k = lab * PP_MAXCLS + bi;
        const int sl = lh_slot(key, k + 1);
        if (sl >= 0) atomicAdd(&vadd[sl], 1); else atomicAdd(&cls[k], 1);
    }
    __syncthreads();
    if (key[threadIdx.x]) atomicAdd(&cls[key[threadIdx.x] - 1], vadd[threadIdx.x]);
}

// the chain's last pass: class map + uint16 id map per pixel (k_class_write) and, with `rec`, the per-label statistics of
// cpx_instance_records (k_rec_stats) whose records the last workgroup writes (k_rec_write).  `pick` = the stage table whose
// remap holds the class of every label (null: no vote -> class 0); with REMAP the labels still go through `prev`'s table.
__global__ void __launch_bounds__(NTHR) k_finish_f(int32_t *__restrict__ masks, uint8_t *__restrict__ cm, uint16_t *__restrict__ masks_u16,
                                                   int has_pick, PPLayout pick, int has_prev, PPLayout prev, int max_rec,
                                                   cpx_record *__restrict__ rec, int32_t *__restrict__ counts, int tail, PPLayout lay, void *ws) {
    __shared__ LabStatsLds L;
    const int i = threadIdx.x;
    if (tail) {
        // records at slot label - 1 (labels are contiguous after the renumbering); counts[tile] = the largest label
        char *tb = (char *)ws + (size_t)blockIdx.y * lay.per_tile;
        const int *clsmap = has_pick ? (const int *)(tb + pick.off_remap) : nullptr;
        const int *bbox = WS(int, off_bbox), *cnt = WS(int, off_cnt);
        const unsigned long long *sumy = WS(unsigned long long, off_sumy), *sumx = WS(unsigned long long, off_sumx);
        const int vm = min(WS(int, off_scal)[SC_VMAX], lay.L - 1);
        for (int v = 1 + i; v <= vm; v += NTHR) {
            const int n = cnt[v];
            if (n <= 0 || v - 1 >= max_rec) continue;
            cpx_record r;
            r.tile = blockIdx.y; r.label = v;
            r.cls = (cm && clsmap) ? (int)(uint8_t)clsmap[v] : 0;
            r.area = n;
            r.y0 = bbox[4 * v]; r.x0 = bbox[4 * v + 1]; r.y1 = bbox[4 * v + 2] + 1; r.x1 = bbox[4 * v + 3] + 1;
            r.sum_y = (int64_t)sumy[v];
            r.sum_x = (int64_t)sumx[v];
            rec[(size_t)blockIdx.y * max_rec + v - 1] = r;
        }
        if (i == 0) counts[blockIdx.y] = vm;
        return;
    }
    L.key[i] = 0; L.ymin[i] = 0x7FFFFFFF; L.xmin[i] = 0x7FFFFFFF; L.ymax[i] = -1; L.xmax[i] = -1; L.cnt[i] = 0;
    L.first[i] = 0x7FFFFFFF; L.sy[i] = 0; L.sx[i] = 0;
    if (i == 0) L.vmax = 0;
    __syncthreads();
    char *tb = (char *)ws + (size_t)blockIdx.y * lay.per_tile;
    const int *clsmap = has_pick ? (const int *)(tb + pick.off_remap) : nullptr;
    int *bbox = WS(int, off_bbox), *cnt = WS(int, off_cnt), *scal = WS(int, off_scal);
    unsigned long long *sumy = WS(unsigned long long, off_sumy), *sumx = WS(unsigned long long, off_sumx);
    const size_t t = blockIdx.y;
    int32_t *m = masks + t * lay.HW;
    const int base = (blockIdx.x * NTHR + i) * RUN_PX;
    int vmax = 0;
    if (base < lay.HW) {
        int lab[RUN_PX];
        load_labels(m, base, lay.HW, has_prev ? (const int *)(tb + prev.off_remap) : nullptr, nullptr, lab);
#pragma unroll
        for (int k = 0; k < RUN_PX; ++k)
            if (base + k < lay.HW) {
                if (cm) cm[t * lay.HW + base + k] = (lab[k] > 0 && clsmap) ? (uint8_t)clsmap[lab[k]] : 0;
                masks_u16[t * lay.HW + base + k] = (uint16_t)lab[k];
            }
        if (rec) {
            label_runs(lab, base, lay.HW, lay.W, lay.L, [&](int l, int, int n, int y, int x) {
                const unsigned long long ay = (unsigned long long)n * y;
                const unsigned long long ax = (unsigned long long)n * x + (unsigned long long)(n * (n - 1) / 2);
                const int sl = lh_slot(L.key, l);
                if (sl >= 0) {
                    atomicMin(&L.ymin[sl], y); atomicMin(&L.xmin[sl], x); atomicMax(&L.ymax[sl], y); atomicMax(&L.xmax[sl], x + n - 1);
                    atomicAdd(&L.cnt[sl], n);
                    atomicAdd(&L.sy[sl], (unsigned)ay); atomicAdd(&L.sx[sl], (unsigned)ax);
                } else {
                    int *bb = bbox + 4 * l;
                    atomicMin(&bb[0], y); atomicMin(&bb[1], x); atomicMax(&bb[2], y); atomicMax(&bb[3], x + n - 1);
                    atomicAdd(&cnt[l], n);
                    atomicAdd(&sumy[l], ay); atomicAdd(&sumx[l], ax);
                }
                vmax = max(vmax, l);
            });
        }
    }
    if (!rec) return;
    if (vmax > 0) atomicMax(&L.vmax, vmax);
    __syncthreads();
    const int l = L.key[i];
    if (l) {
        int *bb = bbox + 4 * l;
        atomicMin(&bb[0], L.ymin[i]); atomicMin(&bb[1], L.xmin[i]); atomicMax(&bb[2], L.ymax[i]); atomicMax(&bb[3], L.xmax[i]);
        atomicAdd(&cnt[l], L.cnt[i]);
        atomicAdd(&sumy[l], (unsigned long long)L.sy[i]); atomicAdd(&sumx[l], (unsigned long long)L.sx[i]);
    }
    if (i == 0 && L.vmax > 0) atomicMax(&scal[SC_VMAX], L.vmax);
}

// ---------------------------------------------------------------------------
// host-side launch sequences (C ABI)
// ---------------------------------------------------------------------------
extern "C" int cpx_instance_records(const uint16_t *masks_u16, const uint8_t *class_masks, int nT, int H, int W, int max_rec,
                                    cpx_record *records, int32_t *counts, void *ws, void *stream);
#define GRID_PIX(lay, nT) dim3(cpx_cdiv((lay).HW, NTHR), nT)
#define GRID_PAD(lay, nT) dim3(cpx_cdiv((lay).HWp, NTHR), nT)
#define GRID_LAB(lay, nT) dim3(cpx_cdiv((lay).L, NTHR), nT)

static int pp_check(int nT, int H, int W) {
    CPX_REQUIRE(nT > 0 && H >= 2 && W >= 2 && H <= 16384 && W <= 16384);
    return CPX_OK;
}

CPX_SWITCH(g_follow_early, 1);      // exact orbit-closure early exit of the Euler loop (debug / A-B switch)
CPX_SWITCH(g_follow_lds, 3);        // bit 0 = 32 x 32-cell segments + the LDS window, bit 1 = the orbit test once per four steps and per wave; 3 = production, 0 = round 4 (A/B)
#ifdef CPX_DEBUG
extern "C" void cpx_follow_set_early_exit(int on) { g_follow_early = on; }
extern "C" void cpx_follow_set_lds_window(int on) { g_follow_lds = on; }
#endif
// a11's two launches (INIT / HIST: the fused chain's forms)
template <bool FUSED>
static void pp_launch_follow(const float *dP, const float *cellprob, int nT, int H, int W, float thr, int niter, int32_t *p_final,
                             float *p_float, const PPLayout &lay, void *ws, hipStream_t s) {
    const float kx = (float)(2.0 / (double)(W - 1)), ky = (float)(2.0 / (double)(H - 1));
    const float shx = (float)(W - 1), shy = (float)(H - 1), hw = (float)W / 2.0f, hh = (float)H / 2.0f;
    const dim3 g2(lay.nseg, nT), g2f(lay.nseg * (FG_BLOCK / NTHR), nT);
#define FOLLOW_ARGS niter, shx, shy, hw, hh, p_final, p_float, lay, ws, g_follow_early
    if (g_follow_lds == 3) {                // production: LDS window (wave-uniform choice), grouped orbit test
        PP_LAUNCH((k_prep_flow<FUSED, true>), g2, dim3(FG_BLOCK), 0, s, dP, cellprob, thr, kx, ky, p_final, p_float, lay, ws);
        PP_LAUNCH((k_follow<FUSED, 1, true>), g2f, dim3(NTHR), 0, s, FOLLOW_ARGS);
        return;
    }
#ifdef CPX_DEBUG
    const int nlin = cpx_cdiv(lay.THW, FG_BLOCK);
    const dim3 g1(nlin, nT), g1f(nlin * (FG_BLOCK / NTHR), nT);
    if (g_follow_lds & 1) PP_LAUNCH((k_prep_flow<FUSED, true>), g2, dim3(FG_BLOCK), 0, s, dP, cellprob, thr, kx, ky, p_final, p_float, lay, ws);
    else PP_LAUNCH((k_prep_flow<FUSED, false>), g1, dim3(FG_BLOCK), 0, s, dP, cellprob, thr, kx, ky, p_final, p_float, lay, ws);
    switch (g_follow_lds) {
    case 0: PP_LAUNCH((k_follow<FUSED, 0, false>), g1f, dim3(NTHR), 0, s, FOLLOW_ARGS); break;      // round 4
    case 1: PP_LAUNCH((k_follow<FUSED, 1, false>), g2f, dim3(NTHR), 0, s, FOLLOW_ARGS); break;      // the window alone
    case 2: PP_LAUNCH((k_follow<FUSED, 0, true>), g1f, dim3(NTHR), 0, s, FOLLOW_ARGS); break;       // the grouped test alone
    case 9: PP_LAUNCH((k_follow<FUSED, 0, false>), g2f, dim3(NTHR), 0, s, FOLLOW_ARGS); break;      // 32 x 32 segments, taps from memory
    case 11: PP_LAUNCH((k_follow<FUSED, 0, true>), g2f, dim3(NTHR), 0, s, FOLLOW_ARGS); break;      // 32 x 32 segments, taps from memory, grouped test
    case 5: PP_LAUNCH((k_follow<FUSED, 2, false>), g2f, dim3(NTHR), 0, s, FOLLOW_ARGS); break;      // window, per-lane choice
    case 7: PP_LAUNCH((k_follow<FUSED, 2, true>), g2f, dim3(NTHR), 0, s, FOLLOW_ARGS); break;       // window, per-lane choice, grouped test
    default: PP_LAUNCH((k_follow<FUSED, 1, true>), g2f, dim3(NTHR), 0, s, FOLLOW_ARGS); break;
    }
#endif
#undef FOLLOW_ARGS
}

extern "C" int cpx_follow_flows(const float *dP, const float *cellprob, int nT, int H, int W,
                                float thr, int niter, int32_t *p_final, float *p_float, void *ws,
                                void *stream) {
    int rc = pp_check(nT, H, W); if (rc) return rc;
    CPX_REQUIRE(dP && cellprob && p_final && ws && niter >= 0);
    hipStream_t s = (hipStream_t)stream;
    PPLayout lay = pp_layout(H, W);
    ws = pp_tiles(ws, nT, H, W);
    pp_launch_follow<false>(dP, cellprob, nT, H, W, thr, niter, p_final, p_float, lay, ws, s);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// zero labels flagged? no: shared tail "renumber by first appearance"
// mode 0: `first` is already final and SC_NLAB already cleared (get_masks: k_big_first + k_copy_scalar);
// mode 1: plain k_first; mode 2: k_first that also zeroes the labels flagged by the size filter.
// nlabels_out (nullable) receives the label count of every tile.
static int pp_renumber(int32_t *masks, int nT, const PPLayout &lay, void *ws, hipStream_t s, int mode,
                       int32_t *nlabels_out) {
    // (`first` was set to INT_MAX by the stage's PPI_STATS init and nothing has written it since)
    if (mode == 1) PP_LAUNCH(k_first<false>, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, lay, ws);
    else if (mode == 2) PP_LAUNCH(k_first<true>, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, lay, ws);
    PP_LAUNCH(k_renumber_rank, GRID_LAB(lay, nT), dim3(NTHR), 0, s, SC_VMAX, lay, ws);
    PP_LAUNCH(k_relabel, GRID_PIX(lay, nT), dim3(NTHR), 0, s, masks, nlabels_out, lay, ws);
    return CPX_OK;
}

extern "C" int cpx_get_masks(const int32_t *p_final, int nT, int H, int W, double max_size_fraction,
                             int32_t *masks, int32_t *nlabels, void *ws, void *stream) {
    int rc = pp_check(nT, H, W); if (rc) return rc;
    CPX_REQUIRE(p_final && masks && ws);
    hipStream_t s = (hipStream_t)stream;
    PPLayout lay = pp_layout(H, W);
    ws = pp_tiles(ws, nT, H, W);
    pp_init(PPI_PAD | PPI_SCAL | PPI_STATS, nT, lay, ws, s);
    PP_LAUNCH(k_hist, GRID_PIX(lay, nT), dim3(NTHR), 0, s, p_final, lay, ws);
    PP_LAUNCH(k_seeds, GRID_PAD(lay, nT), dim3(NTHR), 0, s, lay, ws);
    PP_LAUNCH(k_seed_grow, dim3(SEED_WGS, nT), dim3(NTHR), 0, s, lay, ws);
    PP_LAUNCH(k_gather, GRID_RUN(lay, nT), dim3(NTHR), 0, s, p_final, masks, lay, ws);
    double big = (double)((long long)H * W) * max_size_fraction;
    PP_LAUNCH(k_big_first, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, big, lay, ws);
    pp_renumber(masks, nT, lay, ws, s, 0, nlabels);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// defer_zero: leave the flagged labels in the map; the caller's next pass (k_count_labels with apply_flags) removes them
static int bad_flow_impl(int32_t *masks, const float *dP, int nT, int H, int W, double threshold,
                         double *flow_errors, void *ws, void *stream, bool defer_zero) {
    int rc = pp_check(nT, H, W); if (rc) return rc;
    CPX_REQUIRE(masks && dP && ws);
    hipStream_t s = (hipStream_t)stream;
    PPLayout lay = pp_layout(H, W);
    ws = pp_tiles(ws, nT, H, W);
    pp_init(PPI_SCAL | PPI_STATS | PPI_T, nT, lay, ws, s);
    PP_LAUNCH(k_lab_stats, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, 0, lay, ws);
    PP_LAUNCH(k_center_d2, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, lay, ws);
    PP_LAUNCH(k_center_pick, GRID_PIX(lay, nT), dim3(NTHR), 0, s, masks, lay, ws);
    pp_launch_diffuse(masks, nT, lay, ws, s);
    PP_LAUNCH(k_flow_err_label, dim3(cpx_cdiv(lay.L, NTHR / 64), nT), dim3(NTHR), 0, s, masks, dP, threshold, flow_errors, lay, ws);
    if (!defer_zero) PP_LAUNCH(k_zero_flagged, GRID_PIX(lay, nT), dim3(NTHR), 0, s, masks, lay, ws);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}
extern "C" int cpx_remove_bad_flow_masks(int32_t *masks, const float *dP, int nT, int H, int W,
                                         double threshold, double *flow_errors, void *ws,
                                         void *stream) {
    return bad_flow_impl(masks, dP, nT, H, W, threshold, flow_errors, ws, stream, false);
}

static void pp_size_filter(int32_t *masks, int nT, int min_size, const PPLayout &lay, void *ws,
                           hipStream_t s, unsigned extra_init, int32_t *nlabels_out, double err_thr = 0.0) {
    pp_init(PPI_SCAL | PPI_STATS | extra_init, nT, lay, ws, s);
    PP_LAUNCH(k_count_labels, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, err_thr > 0 ? 1 : 0, err_thr, lay, ws);
    PP_LAUNCH(k_size_filter, dim3(1, nT), dim3(1024), 0, s, min_size, lay, ws);
    pp_renumber(masks, nT, lay, ws, s, 2, nlabels_out);        // k_first<true> removes the flagged labels on the way
}

// extra_init: PPI_* bits of the NEXT stage folded into this stage's last initialisation launch (cpx_compute_masks)
// pending_err_thr > 0: the flow-error filter left its flagged labels in the map (bad_flow_impl, defer_zero)
static int fill_holes_impl(int32_t *masks, int nT, int H, int W, int min_size, int32_t *nlabels, void *ws,
                           void *stream, unsigned extra_init, double pending_err_thr) {
    int rc = pp_check(nT, H, W); if (rc) return rc;
    CPX_REQUIRE(masks && ws);
    hipStream_t s = (hipStream_t)stream;
    PPLayout lay = pp_layout(H, W);
    ws = pp_tiles(ws, nT, H, W);
    if (min_size > 0) pp_size_filter(masks, nT, min_size, lay, ws, s, 0, nullptr, pending_err_thr);
    else {   // labels may be non-contiguous: bbox loop below handles absent labels (slc None)
        pp_init(PPI_SCAL | PPI_STATS, nT, lay, ws, s);
        PP_LAUNCH(k_count_labels, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, pending_err_thr > 0 ? 1 : 0, pending_err_thr, lay, ws);
        PP_LAUNCH(k_copy_scalar, dim3(1, nT), dim3(64), 0, s, SC_NLAB, SC_VMAX, -1, lay, ws);
    }
    // find_objects(masks): bbox per label, then fill
    int nlab_saved_slot = SC_NLAB;
    (void)nlab_saved_slot;
    pp_init(PPI_STATS, nT, lay, ws, s);
    PP_LAUNCH(k_lab_stats, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, 1, lay, ws);
    PP_LAUNCH(k_fill_parallel, dim3(cpx_cdiv(lay.L, NTHR / 64), nT), dim3(NTHR), 0, s, masks, lay, ws);
    PP_LAUNCH(k_fill_serial, dim3(1, nT), dim3(NTHR), 0, s, masks, lay, ws);
    if (min_size > 0) pp_size_filter(masks, nT, min_size, lay, ws, s, extra_init, nlabels);
    else if (nlabels) PP_LAUNCH(k_store_scalar, dim3(1, nT), dim3(64), 0, s, SC_NLAB, nlabels, lay, ws);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

extern "C" int cpx_fill_holes_and_remove_small_masks(int32_t *masks, int nT, int H, int W,
                                                     int min_size, int32_t *nlabels, void *ws,
                                                     void *stream) {
    return fill_holes_impl(masks, nT, H, W, min_size, nlabels, ws, stream, 0, 0.0);
}

static int class_masks_impl(const int32_t *masks, const float *logits, int nT, int ncls, int H, int W,
                            uint8_t *class_masks, uint16_t *masks_u16, bool init, void *ws, void *stream) {
    int rc = pp_check(nT, H, W); if (rc) return rc;
    CPX_REQUIRE(masks && logits && class_masks && ws && ncls >= 1 && ncls <= PP_MAXCLS);
    hipStream_t s = (hipStream_t)stream;
    PPLayout lay = pp_layout(H, W);
    ws = pp_tiles(ws, nT, H, W);
    if (init) pp_init(PPI_CLS, nT, lay, ws, s);
    PP_LAUNCH(k_class_count, GRID_PIX(lay, nT), dim3(NTHR), 0, s, masks, logits, ncls, lay, ws);
    PP_LAUNCH(k_class_pick, GRID_LAB(lay, nT), dim3(NTHR), 0, s, ncls, lay, ws);
    PP_LAUNCH(k_class_write, GRID_PIX(lay, nT), dim3(NTHR), 0, s, masks, class_masks, masks_u16, lay, ws);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}
extern "C" int cpx_compute_class_masks(const int32_t *masks, const float *logits, int nT, int ncls,
                                       int H, int W, uint8_t *class_masks, void *ws, void *stream) {
    return class_masks_impl(masks, logits, nT, ncls, H, W, class_masks, nullptr, true, ws, stream);
}

extern "C" int cpx_remove_border_instances(int32_t *masks, uint8_t *class_masks, int nT, int H,
                                           int W, void *ws, void *stream) {
    int rc = pp_check(nT, H, W); if (rc) return rc;
    CPX_REQUIRE(masks && ws);
    hipStream_t s = (hipStream_t)stream;
    PPLayout lay = pp_layout(H, W);
    ws = pp_tiles(ws, nT, H, W);
    PP_LAUNCH(k_fill_i32, dim3(cpx_cdiv(65536, NTHR), nT), dim3(NTHR), 0, s, lay.off_flag, 65536, 0, lay, ws);
    PP_LAUNCH(k_border_flag, dim3(cpx_cdiv(2 * (H + W), NTHR), nT), dim3(NTHR), 0, s, masks, lay, ws);
    PP_LAUNCH(k_border_zero, GRID_PIX(lay, nT), dim3(NTHR), 0, s, masks, class_masks, lay, ws);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

CPX_SWITCH(g_pp_fused, 1);          // 1 = the 23-launch fused chain (production; the 16-launch ticket form was measured and abandoned, see above), 0 = the stage-wise sequence (38 launches; A/B)
#ifdef CPX_DEBUG
extern "C" void cpx_postproc_set_fused(int on) { g_pp_fused = on; }
#endif

// stage-wise sequence: the entry points of a11-a15 one after the other (what cpx_compute_masks was until round 3)
static int compute_masks_staged(const float *dP, const float *cellprob, const float *logits, int nT,
                                int ncls, int H, int W, float cellprob_threshold,
                                double flow_threshold, int niter, int min_size,
                                double max_size_fraction, uint16_t *masks_u16,
                                uint8_t *class_masks, int32_t *nlabels, void *ws, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)nT * H * W;
    int32_t *masks = (int32_t *)ws;
    int32_t *p_final = (int32_t *)((char *)ws + cpx_align_up(sizeof(int) * n, 256));
    int rc = cpx_follow_flows(dP, cellprob, nT, H, W, cellprob_threshold, niter, p_final, nullptr, ws, stream);
    if (rc) return rc;
    rc = cpx_get_masks(p_final, nT, H, W, max_size_fraction, masks, nullptr, ws, stream);
    if (rc) return rc;
    if (flow_threshold > 0) {      // the flagged labels are removed by the size filter's counting pass below
        rc = bad_flow_impl(masks, dP, nT, H, W, flow_threshold, nullptr, ws, stream, true);
        if (rc) return rc;
    }
    const bool vote = class_masks && logits && ncls > 1;
    // the class-vote table is cleared by the size filter's last initialisation launch (one launch fewer)
    rc = fill_holes_impl(masks, nT, H, W, min_size, nlabels, ws, stream, vote ? PPI_CLS : 0, flow_threshold > 0 ? flow_threshold : 0.0);
    if (rc) return rc;
    if (vote) {       // k_class_write also emits the uint16 id map
        rc = class_masks_impl(masks, logits, nT, ncls, H, W, class_masks, masks_u16, false, ws, stream);
        if (rc) return rc;
    } else {
        if (class_masks) CPX_HIP(hipMemsetAsync(class_masks, 0, n, s));
        PP_LAUNCH(k_to_u16, dim3(cpx_cdiv((long long)n, NTHR)), dim3(NTHR), 0, s, masks, masks_u16, n);
    }
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// the fused chain (see "the fused chain of cpx_compute_masks" above): 23 launches with the class vote and the records (the
// stage-wise sequence: 39), 22 without records, 20 without the vote, 5 fewer without the flow-error filter (round 5: + the diffusion's
// second launch for labels with large boxes)
static int compute_masks_fused(const float *dP, const float *cellprob, const float *logits, int nT,
                               int ncls, int H, int W, float thr, double flow_threshold, int niter, int min_size,
                               double max_size_fraction, uint16_t *masks_u16, uint8_t *class_masks, int32_t *nlabels,
                               int max_rec, cpx_record *records, int32_t *rec_counts, void *ws0, hipStream_t s) {
    const size_t n = (size_t)nT * H * W;
    int32_t *masks = (int32_t *)ws0;
    int32_t *p_final = (int32_t *)((char *)ws0 + cpx_align_up(sizeof(int) * n, 256));
    const PPLayout lay = pp_layout(H, W);
    void *ws = pp_tiles(ws0, nT, H, W);
    const PPLayout L0 = lay, L1 = lay_set(lay, 1), L2 = lay_set(lay, 2), L3 = lay_set(lay, 3), L4 = lay_set(lay, 4), L5 = lay_set(lay, 5);
    // a11: flow field + foreground list + the chain's one initialisation pass; Euler loop + histogram
    pp_launch_follow<true>(dP, cellprob, nT, H, W, thr, niter, p_final, (float *)nullptr, lay, ws, s);
    // a12: seeds, growth, gather (+ big-label removal and ranks in its last workgroup)
    PP_LAUNCH(k_seeds, GRID_PAD(lay, nT), dim3(NTHR), 0, s, L0, ws);
    PP_LAUNCH(k_seed_grow, dim3(SEED_WGS, nT), dim3(NTHR), 0, s, L0, ws);
    const double big = (double)((long long)H * W) * max_size_fraction;
    PP_LAUNCH(k_gather_f, GRID_RUN(lay, nT), dim3(NTHR), 0, s, p_final, masks, big, 0, L0, ws);
    PP_LAUNCH(k_gather_f, dim3(1, nT), dim3(NTHR), 0, s, p_final, masks, big, 1, L0, ws);             // too-big labels out, ranks
    // a13: flow-error filter on the relabelled map (the flagged labels are removed by the size filter's counting pass)
    PP_LAUNCH(k_lab_stats_r, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, 0, L0, L1, ws);              // (relabel through set 0 on the way)
    if (flow_threshold > 0) {
        PP_LAUNCH(k_center_d2, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, L1, ws);
        PP_LAUNCH(k_center_pick, GRID_PIX(lay, nT), dim3(NTHR), 0, s, masks, L1, ws);
        pp_launch_diffuse(masks, nT, L1, ws, s);
        PP_LAUNCH(k_flow_err_label, dim3(cpx_cdiv(lay.L, NTHR / 64), nT), dim3(NTHR), 0, s, masks, dP, flow_threshold, (double *)nullptr, L1, ws);
    }
    // a14: size filter, hole fill, size filter
    PP_LAUNCH(k_count_labels_f, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, L1, flow_threshold > 0 ? 1 : 0, flow_threshold, min_size, (int32_t *)nullptr, 0, L2, ws);
    PP_LAUNCH(k_count_labels_f, dim3(1, nT), dim3(NTHR), 0, s, masks, L1, 0, 0.0, min_size, (int32_t *)nullptr, 1, L2, ws);
    PP_LAUNCH(k_lab_stats_r, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, 1, L2, L3, ws);
    PP_LAUNCH(k_fill_parallel, dim3(cpx_cdiv(lay.L, NTHR / 64), nT), dim3(NTHR), 0, s, masks, L3, ws);
    PP_LAUNCH(k_fill_serial, dim3(1, nT), dim3(NTHR), 0, s, masks, L3, ws);
    PP_LAUNCH(k_count_labels_f, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, L3, 0, 0.0, min_size, (int32_t *)nullptr, 0, L4, ws);
    PP_LAUNCH(k_count_labels_f, dim3(1, nT), dim3(NTHR), 0, s, masks, L3, 0, 0.0, min_size, nlabels, 1, L4, ws);
    // a15 + records
    const bool vote = class_masks && logits && ncls > 1;
    if (vote) {
        PP_LAUNCH(k_class_count_f, GRID_PIX(lay, nT), dim3(NTHR), 0, s, masks, logits, ncls, L4, L5, ws);
        PP_LAUNCH(k_class_pick, GRID_LAB(lay, nT), dim3(NTHR), 0, s, ncls, L5, ws);
        PP_LAUNCH(k_finish_f, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, class_masks, masks_u16, 1, L5, 0, L5, max_rec, records, rec_counts, 0, L5, ws);
        if (records) PP_LAUNCH(k_finish_f, dim3(1, nT), dim3(NTHR), 0, s, masks, class_masks, masks_u16, 1, L5, 0, L5, max_rec, records, rec_counts, 1, L5, ws);
    } else {
        PP_LAUNCH(k_finish_f, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks, class_masks, masks_u16, 0, L5, 1, L4, max_rec, records, rec_counts, 0, L5, ws);
        if (records) PP_LAUNCH(k_finish_f, dim3(1, nT), dim3(NTHR), 0, s, masks, class_masks, masks_u16, 0, L5, 1, L4, max_rec, records, rec_counts, 1, L5, ws);
    }
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

extern "C" int cpx_compute_masks_records(const float *dP, const float *cellprob, const float *logits, int nT,
                                         int ncls, int H, int W, float cellprob_threshold,
                                         double flow_threshold, int niter, int min_size,
                                         double max_size_fraction, uint16_t *masks_u16,
                                         uint8_t *class_masks, int32_t *nlabels, int max_rec, cpx_record *records,
                                         int32_t *rec_counts, void *ws, void *stream) {
    int rc = pp_check(nT, H, W); if (rc) return rc;
    CPX_REQUIRE(dP && cellprob && masks_u16 && ws);
    CPX_REQUIRE(min_size > 0);
    CPX_REQUIRE((records == nullptr) == (rec_counts == nullptr) && (!records || max_rec > 0));
    if (!g_pp_fused) {
        rc = compute_masks_staged(dP, cellprob, logits, nT, ncls, H, W, cellprob_threshold, flow_threshold, niter, min_size,
                                  max_size_fraction, masks_u16, class_masks, nlabels, ws, stream);
        if (rc || !records) return rc;
        return cpx_instance_records(masks_u16, class_masks, nT, H, W, max_rec, records, rec_counts, ws, stream);
    }
    return compute_masks_fused(dP, cellprob, logits, nT, ncls, H, W, cellprob_threshold, flow_threshold, niter, min_size,
                               max_size_fraction, masks_u16, class_masks, nlabels, max_rec, records, rec_counts, ws, (hipStream_t)stream);
}

extern "C" int cpx_compute_masks(const float *dP, const float *cellprob, const float *logits, int nT,
                                 int ncls, int H, int W, float cellprob_threshold,
                                 double flow_threshold, int niter, int min_size,
                                 double max_size_fraction, uint16_t *masks_u16,
                                 uint8_t *class_masks, int32_t *nlabels, void *ws, void *stream) {
    return cpx_compute_masks_records(dP, cellprob, logits, nT, ncls, H, W, cellprob_threshold, flow_threshold, niter, min_size,
                                     max_size_fraction, masks_u16, class_masks, nlabels, 0, nullptr, nullptr, ws, stream);
}

extern "C" int cpx_instance_records(const uint16_t *masks_u16, const uint8_t *class_masks, int nT,
                                    int H, int W, int max_rec, cpx_record *records,
                                    int32_t *counts, void *ws, void *stream) {
    int rc = pp_check(nT, H, W); if (rc) return rc;
    CPX_REQUIRE(masks_u16 && records && counts && ws && max_rec > 0);
    hipStream_t s = (hipStream_t)stream;
    PPLayout lay = pp_layout(H, W);
    ws = pp_tiles(ws, nT, H, W);
    pp_init(PPI_SCAL | PPI_STATS, nT, lay, ws, s);
    PP_LAUNCH(k_rec_stats, GRID_RUN(lay, nT), dim3(NTHR), 0, s, masks_u16, counts, lay, ws);
    PP_LAUNCH(k_rec_write, GRID_LAB(lay, nT), dim3(NTHR), 0, s, class_masks, max_rec, records, counts, lay, ws);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}
