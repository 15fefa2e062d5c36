// Ring metrics with the operation order of GEOS 3.13 (what shapely 2.1.2's Polygon.area / .length /
// .centroid return for the hole-free ring of predict_wsi.py:625-648), shared by the host (g++) and
// the device (hipcc) polygonisers; both translation units are compiled with -ffp-contract=off.
//   Area::ofRingSigned   sum_i (x_i - x_0) * (y_{i-1} - y_{i+1}) / 2 over the closed ring
//   Length::ofLine       sum_i sqrt(dx*dx + dy*dy) on the raw coordinates, closing edge included
//   Centroid::addShell   triangle fan from vertex 0: cg3 += sign * area2 * (p0 + p_i + p_{i+1}),
//                        centroid = cg3 / 3 / areasum2, sign = +1 for a clockwise (positive-area) ring
#pragma once
#ifdef __HIPCC__
#define CPX_HD __host__ __device__
#else
#define CPX_HD
#endif
#include <math.h>

// xy: np OPEN vertices (x, y); the closing edge is implicit.  Returns the signed double area of the fan
// (zero for degenerate rings); area / perimeter / cx / cy as GEOS reports them.
CPX_HD static inline double cpx_ring_metrics(const double *xy, int np, double *area, double *perimeter, double *cx, double *cy) {
    const double x0 = xy[0], y0 = xy[1];
    double sum = 0.0, per = 0.0;
    for (int i = 1; i < np; ++i) {                       // closed ring indices 1 .. rlen - 2 with rlen = np + 1
        const double x = xy[2 * i] - x0;
        const double y1 = (i + 1 < np) ? xy[2 * (i + 1) + 1] : y0;
        const double y2 = xy[2 * (i - 1) + 1];
        sum += x * (y2 - y1);
    }
    for (int i = 0; i < np; ++i) {
        const int j = (i + 1 < np) ? i + 1 : 0;
        const double dx = xy[2 * j] - xy[2 * i], dy = xy[2 * j + 1] - xy[2 * i + 1];
        per += sqrt(dx * dx + dy * dy);
    }
    const double a_signed = sum / 2.0;
    const double sign = a_signed > 0 ? 1.0 : -1.0;
    double cg3x = 0.0, cg3y = 0.0, areasum2 = 0.0;
    for (int i = 0; i < np; ++i) {
        const int j = (i + 1 < np) ? i + 1 : 0;
        const double p1x = xy[2 * i], p1y = xy[2 * i + 1], p2x = xy[2 * j], p2y = xy[2 * j + 1];
        const double a2 = (p1x - x0) * (p2y - y0) - (p2x - x0) * (p1y - y0);
        cg3x += sign * a2 * (x0 + p1x + p2x);
        cg3y += sign * a2 * (y0 + p1y + p2y);
        areasum2 += sign * a2;
    }
    *area = fabs(a_signed);
    *perimeter = per;
    if (areasum2 != 0.0) { *cx = cg3x / 3 / areasum2; *cy = cg3y / 3 / areasum2; }
    else { *cx = 0.0; *cy = 0.0; }
    return areasum2;
}
