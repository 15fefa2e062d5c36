// f2: the radius search of the cross-tile de-duplication on the device.
//
// The reference removes cells detected twice in tile overlaps with
//     tree = KDTree(centers); neighbours = tree.query_pairs(15 / 2)
// followed by a greedy grouping over the pair set in Python-set order
// (/root/reference/src/classpose/entrypoints/predict_wsi.py:896-965).  The expensive part is the
// fixed-radius neighbour search over every cell of the slide (2.6 M centroids at 40k^2, ~10^7 at
// 80k^2); it has a unique answer -- the set {(i, j), i < j : (xi-xj)^2 + (yi-yj)^2 <= r^2} in
// double arithmetic, which is what scipy's query_pairs returns (p = 2: squared distances summed in
// axis order against r^2, box pruning is monotone in floating point) -- so it is computed here
// exactly; the order-dependent grouping stays on the host over this pair list (geojson.py).
//
// Uniform grid with cell edge >= r: points are bucketed by a deterministic counting sort
// (atomic histogram -> exclusive scan -> scatter -> per-cell insertion sort by point index), then
// one thread per point scans the 3 x 3 neighbouring cells twice (count, write) with an exclusive
// scan between the passes, so the output is compact, sorted lexicographically by (i, j) and identical
// from run to run.  HBM-bound integer / double work: every pass is a coalesced sweep over the point arrays,
// the cell table (4 B per 8 x 8 px cell) is touched once per pass.
#include "cpx_common.h"

#define DD_THR 256
#define SCAN_PER 2048            // elements per scan block (8 per thread)

// ---------------------------------------------------------------------------
// exclusive scan of uint32 (multi-block, recursive on the block totals)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(DD_THR) k_scan_block(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                       size_t n, uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t sh[DD_THR];
    const size_t base = (size_t)blockIdx.x * SCAN_PER + (size_t)threadIdx.x * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < DD_THR; o <<= 1) {            // Hillis-Steele over the 256 thread sums
        uint32_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - s;               // exclusive prefix of this thread's chunk
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == DD_THR - 1 && block_sums) block_sums[blockIdx.x] = sh[DD_THR - 1];
}
__global__ void __launch_bounds__(DD_THR) k_scan_add(uint32_t *__restrict__ out, size_t n, const uint32_t *__restrict__ block_off) {
    const size_t i = (size_t)blockIdx.x * SCAN_PER + threadIdx.x;
    const uint32_t add = block_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t j = i + (size_t)k * DD_THR;
        if (j < n) out[j] += add;
    }
}
static size_t scan_ws_elems(size_t n) {
    size_t tot = 0;
    while (n > SCAN_PER) { n = (n + SCAN_PER - 1) / SCAN_PER; tot += cpx_align_up(n, 64); }
    return tot + 64;
}
// out may alias in.  ws: scan_ws_elems(n) uint32.
static int scan_exclusive(const uint32_t *in, uint32_t *out, size_t n, uint32_t *ws, hipStream_t s) {
    const size_t nb = (n + SCAN_PER - 1) / SCAN_PER;
    if (nb <= 1) {
        hipLaunchKernelGGL(k_scan_block, dim3(1), dim3(DD_THR), 0, s, in, out, n, (uint32_t *)nullptr);
        CPX_CHECK_LAUNCH();
        return CPX_OK;
    }
    hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nb), dim3(DD_THR), 0, s, in, out, n, ws);
    CPX_CHECK_LAUNCH();
    int rc = scan_exclusive(ws, ws, nb, ws + cpx_align_up(nb, 64), s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(DD_THR), 0, s, out, n, ws);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// ---------------------------------------------------------------------------
// bucketing
// ---------------------------------------------------------------------------
struct DdGrid { double x0, y0, inv_cell; int gw, gh; };
__device__ __forceinline__ int dd_cell(const DdGrid &g, double x, double y, int &cx, int &cy) {
    cx = (int)floor((x - g.x0) * g.inv_cell); cy = (int)floor((y - g.y0) * g.inv_cell);
    cx = min(max(cx, 0), g.gw - 1); cy = min(max(cy, 0), g.gh - 1);
    return cy * g.gw + cx;
}
__global__ void __launch_bounds__(DD_THR) k_dd_hist(const double *__restrict__ xy, int n, DdGrid g, uint32_t *__restrict__ cell_cnt,
                                                    uint32_t *__restrict__ pt_cell) {
    const int i = blockIdx.x * DD_THR + threadIdx.x;
    if (i >= n) return;
    int cx, cy;
    const int c = dd_cell(g, xy[2 * (size_t)i], xy[2 * (size_t)i + 1], cx, cy);
    pt_cell[i] = (uint32_t)c;
    atomicAdd(&cell_cnt[c], 1u);
}
__global__ void __launch_bounds__(DD_THR) k_dd_scatter(const uint32_t *__restrict__ pt_cell, int n, const uint32_t *__restrict__ cell_off,
                                                       uint32_t *__restrict__ cell_fill, uint32_t *__restrict__ order) {
    const int i = blockIdx.x * DD_THR + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = pt_cell[i];
    order[cell_off[c] + atomicAdd(&cell_fill[c], 1u)] = (uint32_t)i;
}
// every cell's slice sorted by point index: the bucket order no longer depends on the atomics' arrival order
__global__ void __launch_bounds__(DD_THR) k_dd_cell_sort(const uint32_t *__restrict__ cell_off, const uint32_t *__restrict__ cell_cnt,
                                                         size_t n_cells, uint32_t *__restrict__ order) {
    const size_t c = (size_t)blockIdx.x * DD_THR + threadIdx.x;
    if (c >= n_cells) return;
    const uint32_t m = cell_cnt[c];
    if (m < 2) return;
    uint32_t *a = order + cell_off[c];
    for (uint32_t i = 1; i < m; ++i) {
        const uint32_t v = a[i];
        uint32_t j = i;
        while (j > 0 && a[j - 1] > v) { a[j] = a[j - 1]; --j; }
        a[j] = v;
    }
}
// WRITE = false: per-point number of partners j > i within max_dist; WRITE = true: the pairs at pair_off[i]
template <bool WRITE>
__global__ void __launch_bounds__(DD_THR) k_dd_pairs(const double *__restrict__ xy, int n, DdGrid g, double r2,
                                                     const uint32_t *__restrict__ cell_off, const uint32_t *__restrict__ cell_cnt,
                                                     const uint32_t *__restrict__ order, uint32_t *__restrict__ pair_cnt,
                                                     const uint32_t *__restrict__ pair_off, int32_t *__restrict__ pairs,
                                                     long long max_pairs) {
    const int i = blockIdx.x * DD_THR + threadIdx.x;
    if (i >= n) return;
    const double xi = xy[2 * (size_t)i], yi = xy[2 * (size_t)i + 1];
    int cx, cy;
    dd_cell(g, xi, yi, cx, cy);
    uint32_t cnt = 0;
    const long long w0 = WRITE ? (long long)pair_off[i] : 0;
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = cy + dy;
        if (yy < 0 || yy >= g.gh) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = cx + dx;
            if (xx < 0 || xx >= g.gw) continue;
            const size_t c = (size_t)yy * g.gw + xx;
            const uint32_t m = cell_cnt[c];
            if (!m) continue;
            const uint32_t *a = order + cell_off[c];
            for (uint32_t k = 0; k < m; ++k) {
                const int j = (int)a[k];
                if (j <= i) continue;
                const double ddx = xi - xy[2 * (size_t)j], ddy = yi - xy[2 * (size_t)j + 1];
                const double d2 = ddx * ddx + ddy * ddy;         // -ffp-contract=off: two products, one sum, like scipy
                if (d2 <= r2) {
                    if (WRITE && w0 + cnt < max_pairs) { pairs[2 * (w0 + cnt)] = i; pairs[2 * (w0 + cnt) + 1] = j; }
                    ++cnt;
                }
            }
        }
    }
    if (!WRITE) pair_cnt[i] = cnt;
    else {      // this point's partners sorted by j: the list is then in lexicographic (i, j) order (a handful per point)
        const long long m = min((long long)cnt, max(max_pairs - w0, 0LL));
        for (long long a = 1; a < m; ++a) {
            const int v = pairs[2 * (w0 + a) + 1];
            long long b = a;
            while (b > 0 && pairs[2 * (w0 + b - 1) + 1] > v) { pairs[2 * (w0 + b) + 1] = pairs[2 * (w0 + b - 1) + 1]; --b; }
            pairs[2 * (w0 + b) + 1] = v;
        }
    }
}
__global__ void k_dd_total(const uint32_t *__restrict__ pair_cnt, const uint32_t *__restrict__ pair_off, int n, long long *__restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *total = n > 0 ? (long long)pair_off[n - 1] + pair_cnt[n - 1] : 0;
}

struct DdWs { size_t cell_cnt, cell_off, cell_fill, pt_cell, order, pair_cnt, pair_off, scan, total; };
static DdWs dd_ws(int n, size_t n_cells) {
    DdWs w; size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o = cpx_align_up(o + b, 256); return r; };
    w.cell_cnt = take(n_cells * 4); w.cell_off = take(n_cells * 4); w.cell_fill = take(n_cells * 4);
    w.pt_cell = take((size_t)n * 4); w.order = take((size_t)n * 4); w.pair_cnt = take((size_t)n * 4); w.pair_off = take((size_t)n * 4);
    w.scan = take(scan_ws_elems(n_cells > (size_t)n ? n_cells : (size_t)n) * 4);
    w.total = o;
    return w;
}
extern "C" size_t cpx_dedup_pairs_workspace_bytes(int n_points, int grid_w, int grid_h) {
    if (n_points <= 0 || grid_w <= 0 || grid_h <= 0) return 0;
    return dd_ws(n_points, (size_t)grid_w * grid_h).total;
}

// Pass 1 (pairs == NULL): buckets the points and counts the pairs -> *n_pairs (device, int64).
// Pass 2 (pairs != NULL, same workspace, untouched in between): writes min(n_pairs, max_pairs) pairs (i < j),
// sorted by i.  centers_xy [n][2] double (x, y); the grid covers [x0, x0 + grid_w * cell) x [y0, ...), cell >= max_dist.
extern "C" int cpx_dedup_pairs(const double *centers_xy, int n, double x0, double y0, double cell, int grid_w, int grid_h,
                               double max_dist, int32_t *pairs, long long max_pairs, long long *n_pairs,
                               void *workspace, size_t workspace_bytes, void *stream) {
    CPX_REQUIRE(centers_xy && n > 0 && grid_w > 0 && grid_h > 0 && n_pairs && workspace);
    CPX_REQUIRE(cell >= max_dist && max_dist > 0 && (size_t)grid_w * grid_h < ((size_t)1 << 31));
    const size_t n_cells = (size_t)grid_w * grid_h;
    const DdWs L = dd_ws(n, n_cells);
    CPX_REQUIRE(workspace_bytes >= L.total);
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    uint32_t *cell_cnt = (uint32_t *)(ws + L.cell_cnt), *cell_off = (uint32_t *)(ws + L.cell_off),
             *cell_fill = (uint32_t *)(ws + L.cell_fill), *pt_cell = (uint32_t *)(ws + L.pt_cell),
             *order = (uint32_t *)(ws + L.order), *pair_cnt = (uint32_t *)(ws + L.pair_cnt),
             *pair_off = (uint32_t *)(ws + L.pair_off), *scan = (uint32_t *)(ws + L.scan);
    const DdGrid g{x0, y0, 1.0 / cell, grid_w, grid_h};
    const double r2 = max_dist * max_dist;
    const dim3 gp(cpx_cdiv(n, DD_THR)), blk(DD_THR);
    int rc;
    if (!pairs) {
        CPX_HIP(hipMemsetAsync(cell_cnt, 0, n_cells * 4, s));
        CPX_HIP(hipMemsetAsync(cell_fill, 0, n_cells * 4, s));
        hipLaunchKernelGGL(k_dd_hist, gp, blk, 0, s, centers_xy, n, g, cell_cnt, pt_cell);
        CPX_CHECK_LAUNCH();
        if ((rc = scan_exclusive(cell_cnt, cell_off, n_cells, scan, s))) return rc;
        hipLaunchKernelGGL(k_dd_scatter, gp, blk, 0, s, pt_cell, n, cell_off, cell_fill, order);
        hipLaunchKernelGGL(k_dd_cell_sort, dim3((unsigned)((n_cells + DD_THR - 1) / DD_THR)), blk, 0, s, cell_off, cell_cnt, n_cells, order);
        hipLaunchKernelGGL(k_dd_pairs<false>, gp, blk, 0, s, centers_xy, n, g, r2, cell_off, cell_cnt, order, pair_cnt,
                           (const uint32_t *)nullptr, (int32_t *)nullptr, 0LL);
        CPX_CHECK_LAUNCH();
        if ((rc = scan_exclusive(pair_cnt, pair_off, (size_t)n, scan, s))) return rc;
        hipLaunchKernelGGL(k_dd_total, dim3(1), dim3(64), 0, s, pair_cnt, pair_off, n, n_pairs);
    } else {
        CPX_REQUIRE(max_pairs > 0);
        hipLaunchKernelGGL(k_dd_pairs<true>, gp, blk, 0, s, centers_xy, n, g, r2, cell_off, cell_cnt, order, pair_cnt,
                           pair_off, pairs, max_pairs);
    }
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}
