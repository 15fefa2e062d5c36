// Host-side polygonisation of instance maps: replaces the per-cell
//   cv2.findContours(np.uint8(cell_mask), RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)[0]  + shapely
// loop of PostProcessor.__call__ (/root/reference/src/classpose/entrypoints/predict_wsi.py:601-652).
// Input: one tile's uint16 id map (already on the host) and the compact cpx_record list the
// device produced (label, bbox).  Output per cell: the external contour of the instance
// (Suzuki-Abe border following with OpenCV's 8-direction conventions and its
// CHAIN_APPROX_SIMPLE vertex rule), scaled and shifted to level-0 slide coordinates, plus the
// shoelace area, perimeter, area-weighted centroid and an OGC-style simplicity test
// (what shapely's Polygon.area / .length / .centroid / .is_valid return for a hole-free ring).
// Pure host code: no HIP calls, thread-safe, callable from several Python threads (ctypes
// releases the GIL).  opencv-python-headless==4.13.0.92 and shapely==2.1.2 are absent from the
// build image, so this restatement is parity-unpinned (properties are tested instead).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/classpose_hip.h"
#include "cpx_ring_metrics.h"

namespace {
const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

struct Pt { int x, y; };

// icvFetchContour (outer border, method = CHAIN_APPROX_SIMPLE) on a zero-padded binary image
void trace_outer(const std::vector<uint8_t> &img, int W, int sx, int sy, std::vector<Pt> &out) {
    auto at = [&](int x, int y) -> uint8_t { return img[(size_t)y * W + x]; };
    int s_end = 4, s = 4;
    int x1 = 0, y1 = 0;
    do {
        s = (s - 1) & 7;
        x1 = sx + DX[s]; y1 = sy + DY[s];
    } while (at(x1, y1) == 0 && s != s_end);
    Pt pt{sx, sy};
    if (s == s_end) { out.push_back(pt); return; }       // single pixel
    int x3 = sx, y3 = sy, prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        int x4 = 0, y4 = 0;
        for (;;) {
            ++s;
            x4 = x3 + DX[s & 7]; y4 = y3 + DY[s & 7];
            if (at(x4, y4) != 0) break;
            if (s >= 15) break;                           // defensive: cannot happen on a traced border
        }
        s &= 7;
        if (s != prev_s) { out.push_back(pt); prev_s = s; }
        pt.x += DX[s]; pt.y += DY[s];
        if (x4 == sx && y4 == sy && x3 == x1 && y3 == y1) break;
        x3 = x4; y3 = y4;
        s = (s + 4) & 7;
    }
}

inline double cross(double ax, double ay, double bx, double by) { return ax * by - ay * bx; }
inline int sgn(double v) { return (v > 0) - (v < 0); }

// closed segment intersection test (touching counts)
bool seg_intersect(const double *p, const double *q, const double *r, const double *s) {
    double d1 = cross(q[0] - p[0], q[1] - p[1], r[0] - p[0], r[1] - p[1]);
    double d2 = cross(q[0] - p[0], q[1] - p[1], s[0] - p[0], s[1] - p[1]);
    double d3 = cross(s[0] - r[0], s[1] - r[1], p[0] - r[0], p[1] - r[1]);
    double d4 = cross(s[0] - r[0], s[1] - r[1], q[0] - r[0], q[1] - r[1]);
    if (sgn(d1) * sgn(d2) < 0 && sgn(d3) * sgn(d4) < 0) return true;
    auto on = [](const double *a, const double *b, const double *c, double d) {
        return d == 0 && std::fmin(a[0], b[0]) <= c[0] && c[0] <= std::fmax(a[0], b[0]) &&
               std::fmin(a[1], b[1]) <= c[1] && c[1] <= std::fmax(a[1], b[1]);
    };
    return on(p, q, r, d1) || on(p, q, s, d2) || on(r, s, p, d3) || on(r, s, q, d4);
}

// ring given as n distinct vertices (closing edge implicit).  OGC simple-ring test.
bool ring_is_valid(const double *xy, int n) {
    if (n < 3) return false;
    for (int i = 0; i < n; ++i) {
        const double *a0 = xy + 2 * i, *a1 = xy + 2 * ((i + 1) % n);
        if (a0[0] == a1[0] && a0[1] == a1[1]) continue;   // repeated point: allowed
        for (int j = i + 1; j < n; ++j) {
            const double *b0 = xy + 2 * j, *b1 = xy + 2 * ((j + 1) % n);
            if (b0[0] == b1[0] && b0[1] == b1[1]) continue;
            bool adjacent = (j == i + 1) || (i == 0 && j == n - 1);
            if (!adjacent) {
                if (seg_intersect(a0, a1, b0, b1)) return false;
            } else {
                // adjacent edges share one vertex; they may not fold back onto each other
                const double *sh = (j == i + 1) ? a1 : a0;
                const double *pa = (j == i + 1) ? a0 : a1;
                const double *pb = (j == i + 1) ? b1 : b0;
                double c = cross(pa[0] - sh[0], pa[1] - sh[1], pb[0] - sh[0], pb[1] - sh[1]);
                double d = (pa[0] - sh[0]) * (pb[0] - sh[0]) + (pa[1] - sh[1]) * (pb[1] - sh[1]);
                if (c == 0 && d > 0) return false;
            }
        }
    }
    return true;
}
}  // namespace

// masks_host [H][W] uint16 (ONE tile), recs_host[n] for that tile.  Vertices are written to
// xy_pool (x, y doubles), cells[i].offset / n_pts index it.  Returns the number of vertices
// written, or a negative code (pool too small: CPX_ENOMEM).
extern "C" int cpx_polygonize_host(const uint16_t *masks_host, int H, int W, const cpx_record *recs_host,
                                   int n, double scale, double off_x, double off_y, double *xy_pool,
                                   int max_pts, cpx_cell *cells) {
    if (!masks_host || !recs_host || !xy_pool || !cells || H <= 0 || W <= 0 || n < 0) return CPX_EINVAL;
    int used = 0;
    std::vector<uint8_t> img;
    std::vector<int> stack;
    std::vector<Pt> pts;
    for (int c = 0; c < n; ++c) {
        const cpx_record &r = recs_host[c];
        cpx_cell &o = cells[c];
        std::memset(&o, 0, sizeof(o));
        o.cls = r.cls;
        o.offset = used;
        const int bw = r.x1 - r.x0, bh = r.y1 - r.y0;
        if (bw <= 0 || bh <= 0 || r.x0 < 0 || r.y0 < 0 || r.x1 > W || r.y1 > H) continue;
        const int PW = bw + 2, PH = bh + 2;
        img.assign((size_t)PW * PH, 0);
        for (int y = 0; y < bh; ++y)
            for (int x = 0; x < bw; ++x)
                img[(size_t)(y + 1) * PW + x + 1] = masks_host[(size_t)(r.y0 + y) * W + r.x0 + x] == (uint16_t)r.label;
        // connected components (8-conn): keep the one discovered LAST in raster order
        // (legacy cvFindContours links each new contour in front: contours[0] = last found)
        int lsx = -1, lsy = -1, ncomp = 0;
        std::vector<uint8_t> lab(img.size(), 0);
        for (int y = 1; y <= bh; ++y)
            for (int x = 1; x <= bw; ++x) {
                size_t i0 = (size_t)y * PW + x;
                if (!img[i0] || lab[i0]) continue;
                ++ncomp; lsx = x; lsy = y;
                stack.clear(); stack.push_back((int)i0); lab[i0] = (uint8_t)(1 + (ncomp & 1));
                while (!stack.empty()) {
                    int i = stack.back(); stack.pop_back();
                    int yy = i / PW, xx = i % PW;
                    for (int k = 0; k < 8; ++k) {
                        size_t j = (size_t)(yy + DY[k]) * PW + xx + DX[k];
                        if (img[j] && !lab[j]) { lab[j] = lab[i0]; stack.push_back((int)j); }
                    }
                }
            }
        if (ncomp == 0) continue;
        if (ncomp > 1) {                                   // isolate the chosen component
            std::fill(lab.begin(), lab.end(), 0);
            stack.clear(); size_t i0 = (size_t)lsy * PW + lsx;
            stack.push_back((int)i0); lab[i0] = 1;
            while (!stack.empty()) {
                int i = stack.back(); stack.pop_back();
                int yy = i / PW, xx = i % PW;
                for (int k = 0; k < 8; ++k) {
                    size_t j = (size_t)(yy + DY[k]) * PW + xx + DX[k];
                    if (img[j] && !lab[j]) { lab[j] = 1; stack.push_back((int)j); }
                }
            }
            for (size_t i = 0; i < img.size(); ++i) img[i] = lab[i];
        }
        pts.clear();
        trace_outer(img, PW, lsx, lsy, pts);
        const int np = (int)pts.size();
        if (used + np > max_pts) return CPX_ENOMEM;
        double *xy = xy_pool + 2 * (size_t)used;
        for (int i = 0; i < np; ++i) {
            // (contour + bbox origin) * prediction_to_slide_scale + tile origin   (predict_wsi.py:617-620)
            xy[2 * i] = (double)(pts[i].x - 1 + r.x0) * scale + off_x;
            xy[2 * i + 1] = (double)(pts[i].y - 1 + r.y0) * scale + off_y;
        }
        o.n_pts = np;
        used += np;
        if (np < 4) continue;                              // "curr_coords.shape[0] < 4" -> invalid cell
        // GEOS-order area / length / centroid (cpx_ring_metrics.h)
        const double a2 = cpx_ring_metrics(xy, np, &o.area, &o.perimeter, &o.cx, &o.cy);
        o.valid = (a2 != 0 && ring_is_valid(xy, np)) ? 1 : 0;
    }
    return used;
}

// ---------------------------------------------------------------------------------------------
// cv2.findContours(mask, RETR_CCOMP, CHAIN_APPROX_SIMPLE) for the GrandQC tissue / artefact maps
// (/root/reference/src/classpose/grandqc/wsi_tissue_detection.py:209-213,
//  wsi_artefact_detection.py:262-265): Suzuki-Abe raster scan with border marking (outer border
// where a 1 follows a 0, hole border where a 0 follows a positive pixel), every border followed
// once with the 8-direction rule above and stored with the CHAIN_APPROX_SIMPLE vertex rule.
// Two-level hierarchy: parent[i] = index of the outer border of the component a hole border
// belongs to, -1 for outer borders.  Contours come out in discovery (raster) order; OpenCV's own
// ordering of the list is not reproduced (the reference only iterates over it).
// ---------------------------------------------------------------------------------------------
namespace {
void trace_marked(std::vector<int32_t> &f, int PW, int sx, int sy, bool is_hole, int32_t nbd, std::vector<Pt> &out) {
    auto at = [&](int x, int y) -> int32_t & { return f[(size_t)y * PW + x]; };
    int s_end = is_hole ? 0 : 4, s = s_end;
    int x1 = 0, y1 = 0;
    do {
        s = (s - 1) & 7;
        x1 = sx + DX[s]; y1 = sy + DY[s];
    } while (at(x1, y1) == 0 && s != s_end);
    Pt pt{sx, sy};
    if (s == s_end) { at(sx, sy) = -nbd; out.push_back(pt); return; }   // isolated pixel
    int x3 = sx, y3 = sy, prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        int x4 = 0, y4 = 0;
        for (;;) {
            ++s;
            x4 = x3 + DX[s & 7]; y4 = y3 + DY[s & 7];
            if (at(x4, y4) != 0) break;
            if (s >= 15) break;
        }
        s &= 7;
        // the east neighbour was examined and found empty on this visit -> right edge of this border
        if ((unsigned)(s - 1) < (unsigned)s_end) at(x3, y3) = -nbd;
        else if (at(x3, y3) == 1) at(x3, y3) = nbd;
        if (s != prev_s) { out.push_back(pt); prev_s = s; }
        pt.x += DX[s]; pt.y += DY[s];
        if (x4 == sx && y4 == sy && x3 == x1 && y3 == y1) break;
        x3 = x4; y3 = y4;
        s = (s + 4) & 7;
    }
}
}  // namespace

// mask [H][W] uint8 (non-zero = foreground).  Vertices (x, y int32 pairs, image coordinates) go to
// xy_pool, contour c is xy_pool[offsets[c] .. offsets[c] + n_pts[c]).  Returns the number of contours
// (>= 0), CPX_ENOMEM if max_pts / max_contours are too small.
extern "C" int cpx_find_contours_ccomp_host(const uint8_t *mask, int H, int W, int32_t *xy_pool, int max_pts,
                                            int32_t *offsets, int32_t *n_pts, int32_t *parent, int max_contours) {
    if (!mask || !xy_pool || !offsets || !n_pts || !parent || H <= 0 || W <= 0) return CPX_EINVAL;
    const int PW = W + 2, PH = H + 2;
    std::vector<int32_t> f((size_t)PW * PH, 0);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) f[(size_t)(y + 1) * PW + x + 1] = mask[(size_t)y * W + x] ? 1 : 0;
    // 8-connected components -> the outer border that owns each component
    std::vector<int32_t> comp(f.size(), 0);
    std::vector<int> stack;
    int ncomp = 0;
    for (int y = 1; y <= H; ++y)
        for (int x = 1; x <= W; ++x) {
            const size_t i0 = (size_t)y * PW + x;
            if (!f[i0] || comp[i0]) continue;
            ++ncomp;
            comp[i0] = ncomp; stack.clear(); stack.push_back((int)i0);
            while (!stack.empty()) {
                const int i = stack.back(); stack.pop_back();
                const int yy = i / PW, xx = i % PW;
                for (int k = 0; k < 8; ++k) {
                    const size_t j = (size_t)(yy + DY[k]) * PW + xx + DX[k];
                    if (f[j] && !comp[j]) { comp[j] = ncomp; stack.push_back((int)j); }
                }
            }
        }
    std::vector<int32_t> outer_of(ncomp + 1, -1);
    std::vector<Pt> pts;
    int nc = 0, used = 0;
    int32_t nbd = 1;
    for (int y = 1; y <= H; ++y) {
        for (int x = 1; x <= W; ++x) {
            const int32_t p = f[(size_t)y * PW + x];
            if (p == 0) continue;
            const bool outer = (p == 1 && f[(size_t)y * PW + x - 1] == 0);
            const bool hole = (!outer && p >= 1 && f[(size_t)y * PW + x + 1] == 0);
            if (!outer && !hole) continue;
            if (nc >= max_contours) return CPX_ENOMEM;
            ++nbd;
            pts.clear();
            trace_marked(f, PW, x, y, hole, nbd, pts);
            if (used + (int)pts.size() > max_pts) return CPX_ENOMEM;
            offsets[nc] = used; n_pts[nc] = (int32_t)pts.size();
            for (const Pt &q : pts) { xy_pool[2 * (size_t)used] = q.x - 1; xy_pool[2 * (size_t)used + 1] = q.y - 1; ++used; }
            const int32_t c = comp[(size_t)y * PW + x];
            if (outer) { parent[nc] = -1; outer_of[c] = nc; }
            else parent[nc] = outer_of[c];
            ++nc;
        }
    }
    return nc;
}
