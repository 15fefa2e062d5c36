// f1  Device polygonisation of instance maps: the per-cell
//   cv2.findContours(cell_mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)[0] + shapely area / length /
//   centroid / is_valid loop of PostProcessor.__call__
//   (/root/reference/src/classpose/entrypoints/predict_wsi.py:601-652)
// on the GPU, so that only compact vertex lists and per-cell rows leave the device instead of the
// 2 B/pixel id maps.  Same arithmetic as the host restatement cpx_polygonize_host
// (cpx_host_polygons.cpp; bit-identical outputs, tests/test_gpu_polygons.py).  One WAVE per
// instance (round 5; one thread until then): the border walk itself is one lane's pointer chase,
// but the search for a component's start pixel over the bounding box and the ring's validity test
// (every edge against every other) are spread over the 64 lanes -- with one thread per instance
// four discs of radius 60 per tile cost 4.4 + 8.5 ms, twenty-five of radius 20 0.75 + 1.55 ms
// (tools/poly_size_scan.py).  Runs on the post-processing stream under the next batch's network.
// Precondition: instances are hole-free (cpx_fill_holes_and_remove_small_masks guarantees it);
// then every 8-connected component's raster-first pixel lies on its outer border, so the
// components are enumerated by tracing from unvisited "local tops" in raster order, and OpenCV's
// contours[0] (the component found LAST) is the last such start.
// Built with -ffp-contract=off (the double arithmetic is spelled out).
#include "cpx_common.h"
#include "cpx_ring_metrics.h"

#define PG_WG 256           // four waves per workgroup, one instance per wave at a time
#define PG_TILE_WGS 128      // workgroups per tile: 512 waves stride through the tile's records (8 tiles: 4 waves per SIMD)

__constant__ int PG_DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
__constant__ int PG_DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

struct PgMap {
    const uint16_t *m; int H, W; uint16_t lab;
    __device__ __forceinline__ bool at(int x, int y) const {
        return x >= 0 && y >= 0 && x < W && y < H && m[(size_t)y * W + x] == lab;
    }
};

// icvFetchContour (outer border) with the CHAIN_APPROX_SIMPLE vertex rule.  MODE 0: mark visited border
// pixels; MODE 1: count vertices; MODE 2: write vertices (level-0 coordinates).
template <int MODE>
__device__ int pg_trace(const PgMap &g, int sx, int sy, uint8_t *visited, double *xy, double scale, double ox,
                        double oy) {
    int n = 0;
    auto emit = [&](int x, int y) {
        if (MODE == 2) { xy[2 * n] = (double)x * scale + ox; xy[2 * n + 1] = (double)y * scale + oy; }
        ++n;
    };
    int s_end = 4, s = 4, x1 = 0, y1 = 0;
    do {
        s = (s - 1) & 7;
        x1 = sx + PG_DX[s]; y1 = sy + PG_DY[s];
    } while (!g.at(x1, y1) && s != s_end);
    if (MODE == 0) visited[(size_t)sy * g.W + sx] = 1;
    if (s == s_end) { emit(sx, sy); return n; }                   // single pixel
    int px = sx, py = sy, x3 = sx, y3 = sy, prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        int x4 = 0, y4 = 0;
        for (;;) {
            ++s;
            x4 = x3 + PG_DX[s & 7]; y4 = y3 + PG_DY[s & 7];
            if (g.at(x4, y4)) break;
            if (s >= 15) break;
        }
        s &= 7;
        if (MODE == 0) visited[(size_t)y3 * g.W + x3] = 1;
        if (s != prev_s) { emit(px, py); prev_s = s; }
        px += PG_DX[s]; py += PG_DY[s];
        if (x4 == sx && y4 == sy && x3 == x1 && y3 == y1) break;
        x3 = x4; y3 = y4;
        s = (s + 4) & 7;
    }
    return n;
}

// pass 1: per record, the start pixel of the last component and its vertex count.  A fixed grid of waves strides through the tile's
// records; the lanes test 64 pixels of a bounding-box row at a time for "local top" candidates (a label pixel without a label pixel to
// its W, NW, N, NE), lane 0 visits the candidates in raster order -- marks the outer border of a component not seen before -- exactly as
// the one-thread loop did.
#define PG_WAVES (PG_WG / 64)
__global__ void __launch_bounds__(PG_WG) k_poly_count(const uint16_t *__restrict__ masks, const cpx_record *__restrict__ recs,
                             const int32_t *__restrict__ counts, int H, int W, int max_rec, uint8_t *visited,
                             int32_t *start, int32_t *npts) {
    const int t = blockIdx.y, lane = threadIdx.x & 63;
    const int cnt = min(counts[t], max_rec);
    uint8_t *vis = visited + (size_t)t * H * W;
    for (int c = blockIdx.x * PG_WAVES + (threadIdx.x >> 6); c < cnt; c += gridDim.x * PG_WAVES) {
        const size_t id = (size_t)t * max_rec + c;
        const cpx_record r = recs[id];
        PgMap g{masks + (size_t)t * H * W, H, W, (uint16_t)r.label};
        int lx = -1, ly = -1;
        if (!(r.x1 <= r.x0 || r.y1 <= r.y0 || r.x0 < 0 || r.y0 < 0 || r.x1 > W || r.y1 > H)) {
            for (int y = r.y0; y < r.y1; ++y)
                for (int xb = r.x0; xb < r.x1; xb += 64) {
                    const int x = xb + lane;
                    const bool cand = x < r.x1 && g.at(x, y) && !(g.at(x - 1, y) || g.at(x - 1, y - 1) || g.at(x, y - 1) || g.at(x + 1, y - 1));
                    unsigned long long m = __ballot(cand);
                    if (lane == 0)
                        while (m) {
                            const int xx = xb + __ffsll((long long)m) - 1;
                            m &= m - 1;
                            if (vis[(size_t)y * W + xx]) continue;
                            pg_trace<0>(g, xx, y, vis, nullptr, 0, 0, 0);       // a new component: mark its outer border
                            lx = xx; ly = y;
                        }
                }
        }
        if (lane == 0) {
            start[id] = lx < 0 ? -1 : ly * W + lx;
            npts[id] = lx < 0 ? 0 : pg_trace<1>(g, lx, ly, nullptr, nullptr, 0, 0, 0);
        }
    }
}

// exclusive scan of npts in (tile, record) order -> offsets; total to out[0]; single block.  Only a tile's first counts[t] slots hold
// anything (the rest of the max_rec-slot table is zeros that nobody's offset is read for): 8 chunks instead of 47 for the bench's batch
// (94 -> ~20 us; the scan was the largest constant of the polygon stage).
__global__ void k_poly_scan(const int32_t *__restrict__ npts, const int32_t *__restrict__ counts, int nT, int max_rec,
                            int32_t *__restrict__ offs, int32_t *total) {
    __shared__ int sh[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int t = 0; t < nT; ++t) {
        const int n = min(counts[t], max_rec);
        for (int base = 0; base < n; base += 1024) {
            const int i = base + threadIdx.x;
            const int v = i < n ? npts[(size_t)t * max_rec + i] : 0;
            sh[threadIdx.x] = v;
            __syncthreads();
            for (int o = 1; o < 1024; o <<= 1) {
                const int a = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
                __syncthreads();
                sh[threadIdx.x] += a;
                __syncthreads();
            }
            if (i < n) offs[(size_t)t * max_rec + i] = carry + sh[threadIdx.x] - v;
            __syncthreads();
            if (threadIdx.x == 1023) carry += sh[1023];
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) total[0] = carry;
}

__device__ __forceinline__ double pg_cross(double ax, double ay, double bx, double by) { return ax * by - ay * bx; }
__device__ __forceinline__ int pg_sgn(double v) { return (v > 0) - (v < 0); }
__device__ bool pg_on(const double *a, const double *b, const double *c, double d) {
    return d == 0 && fmin(a[0], b[0]) <= c[0] && c[0] <= fmax(a[0], b[0]) && fmin(a[1], b[1]) <= c[1] &&
           c[1] <= fmax(a[1], b[1]);
}
__device__ bool pg_seg_intersect(const double *p, const double *q, const double *r, const double *s) {
    const double d1 = pg_cross(q[0] - p[0], q[1] - p[1], r[0] - p[0], r[1] - p[1]);
    const double d2 = pg_cross(q[0] - p[0], q[1] - p[1], s[0] - p[0], s[1] - p[1]);
    const double d3 = pg_cross(s[0] - r[0], s[1] - r[1], p[0] - r[0], p[1] - r[1]);
    const double d4 = pg_cross(s[0] - r[0], s[1] - r[1], q[0] - r[0], q[1] - r[1]);
    if (pg_sgn(d1) * pg_sgn(d2) < 0 && pg_sgn(d3) * pg_sgn(d4) < 0) return true;
    return pg_on(p, q, r, d1) || pg_on(p, q, s, d2) || pg_on(r, s, p, d3) || pg_on(r, s, q, d4);
}
// shapely's is_valid for the hole-free ring: no two non-adjacent edges touch, no two adjacent edges fold back onto each other.  The wave's lanes
// take the partners j = i + 1 + lane, + 64, ... of edge i (one thread walked all n (n - 1) / 2 pairs until round 5); the verdict is a vote.
__device__ bool pg_ring_is_valid(const double *xy, int n, int lane) {
    if (n < 3) return false;
    bool bad = false;
    for (int i = 0; i < n; ++i) {
        const double *a0 = xy + 2 * i, *a1 = xy + 2 * ((i + 1) % n);
        if (a0[0] == a1[0] && a0[1] == a1[1]) continue;
        for (int j = i + 1 + lane; j < n; j += 64) {
            const double *b0 = xy + 2 * j, *b1 = xy + 2 * ((j + 1) % n);
            if (b0[0] == b1[0] && b0[1] == b1[1]) continue;
            const bool adjacent = (j == i + 1) || (i == 0 && j == n - 1);
            if (!adjacent) {
                if (pg_seg_intersect(a0, a1, b0, b1)) bad = true;
            } else {
                const double *sh = (j == i + 1) ? a1 : a0;
                const double *pa = (j == i + 1) ? a0 : a1;
                const double *pb = (j == i + 1) ? b1 : b0;
                const double c = pg_cross(pa[0] - sh[0], pa[1] - sh[1], pb[0] - sh[0], pb[1] - sh[1]);
                const double d = (pa[0] - sh[0]) * (pb[0] - sh[0]) + (pa[1] - sh[1]) * (pb[1] - sh[1]);
                if (c == 0 && d > 0) bad = true;
            }
        }
        if ((i & 15) == 15 && __any(bad)) return false;
    }
    return !__any(bad);
}

// pass 2: vertices + shoelace area / perimeter / centroid / validity.  Lane 0 walks the border and sums the metrics (their order of
// operations is GEOS's); the validity test is the wave's.
__global__ void __launch_bounds__(PG_WG) k_poly_write(const uint16_t *__restrict__ masks, const cpx_record *__restrict__ recs,
                             const int32_t *__restrict__ counts, int H, int W, int max_rec, double scale,
                             const double *__restrict__ origins, const int32_t *__restrict__ start,
                             const int32_t *__restrict__ npts, const int32_t *__restrict__ offs, double *xy_pool,
                             int max_pts, cpx_cell *cells) {
    const int t = blockIdx.y, lane = threadIdx.x & 63;
    const int cnt = min(counts[t], max_rec);
    for (int c = blockIdx.x * PG_WAVES + (threadIdx.x >> 6); c < cnt; c += gridDim.x * PG_WAVES) {
        const size_t id = (size_t)t * max_rec + c;
        const cpx_record r = recs[id];
        cpx_cell o;
        o.area = 0; o.perimeter = 0; o.cx = 0; o.cy = 0; o.n_pts = 0; o.offset = offs[id]; o.valid = 0; o.cls = r.cls;
        const int np = npts[id];
        if (start[id] >= 0 && offs[id] + np <= max_pts) {                   // (uniform over the wave)
            PgMap g{masks + (size_t)t * H * W, H, W, (uint16_t)r.label};
            double *xy = xy_pool + 2 * (size_t)offs[id];
            double a2 = 0;
            if (lane == 0) {
                pg_trace<2>(g, start[id] % W, start[id] / W, nullptr, xy, scale, origins[2 * t], origins[2 * t + 1]);
                if (np >= 4) a2 = cpx_ring_metrics(xy, np, &o.area, &o.perimeter, &o.cx, &o.cy);
            }
            o.n_pts = np;
            if (np >= 4) {
                __threadfence_block();                                       // lane 0's vertices, read by the whole wave below
                a2 = __shfl(a2, 0);
                const bool ok = a2 != 0 && pg_ring_is_valid(xy, np, lane);
                o.valid = ok ? 1 : 0;
            }
        }
        if (lane == 0) cells[id] = o;
    }
}

extern "C" size_t cpx_polygonize_workspace_bytes(int nT, int H, int W, int max_rec) {
    return cpx_align_up((size_t)nT * H * W, 256) + 3 * cpx_align_up((size_t)nT * max_rec * 4, 256);
}

extern "C" int cpx_polygonize_device(const uint16_t *masks_u16, const cpx_record *records, const int32_t *rec_counts,
                                     int nT, int H, int W, int max_rec, double scale, const double *origins,
                                     double *xy_pool, int max_pts, cpx_cell *cells, int32_t *n_pts_total,
                                     void *workspace, void *stream) {
    CPX_REQUIRE(masks_u16 && records && rec_counts && origins && xy_pool && cells && n_pts_total && workspace);
    CPX_REQUIRE(nT > 0 && H > 0 && W > 0 && max_rec > 0 && max_pts > 0);
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    uint8_t *visited = (uint8_t *)ws;
    size_t o = cpx_align_up((size_t)nT * H * W, 256);
    const size_t per = cpx_align_up((size_t)nT * max_rec * 4, 256);
    int32_t *start = (int32_t *)(ws + o), *npts = (int32_t *)(ws + o + per), *offs = (int32_t *)(ws + o + 2 * per);
    CPX_HIP(hipMemsetAsync(visited, 0, (size_t)nT * H * W, s));
    dim3 grid(PG_TILE_WGS, nT);
    hipLaunchKernelGGL(k_poly_count, grid, dim3(PG_WG), 0, s, masks_u16, records, rec_counts, H, W, max_rec, visited,
                       start, npts);
    CPX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_poly_scan, dim3(1), dim3(1024), 0, s, npts, rec_counts, nT, max_rec, offs, n_pts_total);
    CPX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_poly_write, grid, dim3(PG_WG), 0, s, masks_u16, records, rec_counts, H, W, max_rec, scale,
                       origins, start, npts, offs, xy_pool, max_pts, cells);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}
