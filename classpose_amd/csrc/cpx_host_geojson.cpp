// Host-side streaming GeoJSON writer: the two FeatureCollections that
//   json.dump({"type": "FeatureCollection", "features": [...]}, f)
// writes for the cells of a slide (/root/reference/src/classpose/entrypoints/predict_wsi.py:1772-1785) with features
// built by to_geojson_polygon (:813-854), apply_bounds_offset_to_feature (:857-893) and polygons_to_centroids
// (:1336-1374) -- byte for byte what CPython's json encoder emits (", " / ": " separators, float.__repr__ numbers),
// apart from the random uuid4 ids.  The reference builds ~2 M dicts and dumps them in one call; at 40k x 40k this file
// pair is 2.6 GB and the Python loop that streamed it was the serial tail of the CLI (28 s against a 12 s tile loop
// on 8 GPUs).  Here the kept cells are split into chunks of 4 096, worker threads format chunks into memory, and the chunks are
// written in order.  Pure host code: no HIP calls; ctypes releases the GIL around it.
#include <atomic>
#include <charconv>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/classpose_hip.h"

extern thread_local char cpx_err_buf[256];       // cpx_api.hip; read through cpx_last_error()
#define CPX_REQUIRE(cond)                                                                                       \
    do {                                                                                                        \
        if (!(cond)) {                                                                                          \
            std::snprintf(cpx_err_buf, sizeof(cpx_err_buf), "%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond); \
            return CPX_EINVAL;                                                                                  \
        }                                                                                                       \
    } while (0)

namespace {

// float.__repr__: shortest digits that round-trip, fixed notation for 1e-4 <= |v| < 1e16 (always with a fractional
// part), scientific otherwise ("1e+16", "1.5e-05": at least two exponent digits) -- std::to_chars produces the same
// shortest digit string, only the choice of notation differs.  nan / inf: `json` says how the encoder spells them
// (NaN / Infinity) as opposed to repr (nan / inf).
inline void put_double(std::string &out, double v, bool json) {
    if (std::isnan(v)) { out += json ? "NaN" : "nan"; return; }
    if (std::isinf(v)) { out += v < 0 ? (json ? "-Infinity" : "-inf") : (json ? "Infinity" : "inf"); return; }
    char buf[40];
    {   // fast path for the bulk of a slide's numbers: contour vertices are multiples of 0.5 (pixel corners times the
        // level-0 scale), whose shortest representation is the integer part followed by ".0" or ".5"
        const double a = std::fabs(v), t = a * 2.0;
        if (t < 2e15 && t == std::floor(t)) {
            uint64_t h = (uint64_t)t, ip = h >> 1;
            char *q = buf + sizeof buf;
            *--q = (h & 1) ? '5' : '0';
            *--q = '.';
            do { *--q = (char)('0' + ip % 10); ip /= 10; } while (ip);
            if (std::signbit(v)) *--q = '-';
            out.append(q, (size_t)(buf + sizeof buf - q));
            return;
        }
    }
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    // buf = [-]d[.ddd]e[+-]XX
    char *p = buf, *end = r.ptr;
    if (*p == '-') { out += '-'; ++p; }
    char *e = p;
    while (*e != 'e') ++e;
    int ex = 0;
    {
        const bool neg = e[1] == '-';
        for (char *q = e + 2; q < end; ++q) ex = ex * 10 + (*q - '0');
        if (neg) ex = -ex;
    }
    char digits[24];
    int nd = 0;
    for (char *q = p; q < e; ++q)
        if (*q != '.') digits[nd++] = *q;
    if (nd == 1 && digits[0] == '0') { out += "0.0"; return; }            // +-0.0
    if (ex < -4 || ex >= 16) { out.append(p, end); return; }             // repr's scientific form == to_chars'
    if (ex < 0) {
        out += "0.";
        out.append((size_t)(-ex - 1), '0');
        out.append(digits, nd);
    } else if (nd <= ex + 1) {
        out.append(digits, nd);
        out.append((size_t)(ex + 1 - nd), '0');
        out += ".0";
    } else {
        out.append(digits, ex + 1);
        out += '.';
        out.append(digits + ex + 1, nd - ex - 1);
    }
}

struct Rng {                                  // xoshiro256**, one per worker, seeded from the OS
    uint64_t s[4];
    explicit Rng(uint64_t salt) {
        std::random_device rd;
        for (auto &w : s) w = ((uint64_t)rd() << 32) ^ rd() ^ (salt * 0x9E3779B97F4A7C15ull);
        if (!(s[0] | s[1] | s[2] | s[3])) s[0] = 1;
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
};

inline void put_uuid4(std::string &out, Rng &rng) {
    uint64_t a = rng.next(), b = rng.next();
    a = (a & 0xFFFFFFFFFFFF0FFFull) | 0x0000000000004000ull;     // version 4
    b = (b & 0x3FFFFFFFFFFFFFFFull) | 0x8000000000000000ull;     // RFC 4122 variant
    static const char hex[] = "0123456789abcdef";
    char u[36];
    int k = 0;
    for (int i = 0; i < 16; ++i) {
        const unsigned byte = (unsigned)((i < 8 ? a >> (56 - 8 * i) : b >> (56 - 8 * (i - 8))) & 0xFF);
        if (i == 4 || i == 6 || i == 8 || i == 10) u[k++] = '-';
        u[k++] = hex[byte >> 4]; u[k++] = hex[byte & 15];
    }
    out.append(u, 36);
}

struct Row { double area, perimeter, cx, cy; int64_t n_pts, cls; };      // numpy CELL_ROW (predict_wsi.py of this package)

struct Job {
    const Row *cells; const double *cen; const double *xy; const int64_t *offs; const int64_t *keep;
    const char *const *class_json; int n_class; double bx, by; bool shift;
};

void format_chunk(const Job &j, int64_t lo, int64_t hi, bool first_chunk, Rng &rng, std::string &fc, std::string &fp) {
    std::string meas;
    for (int64_t k = lo; k < hi; ++k) {
        const int64_t i = j.keep[k];
        const Row &c = j.cells[i];
        const char *cls = j.class_json[c.cls >= 0 && c.cls < j.n_class ? c.cls : 0];
        double cx = j.cen[2 * i], cy = j.cen[2 * i + 1];
        if (j.shift) { cx -= j.bx; cy -= j.by; }
        meas.clear();
        meas += "[{\"name\": \"area\", \"value\": "; put_double(meas, c.area, true);
        meas += "}, {\"name\": \"perimeter\", \"value\": "; put_double(meas, c.perimeter, true);
        meas += "}, {\"name\": \"centroidX\", \"value\": "; put_double(meas, cx, true);
        meas += "}, {\"name\": \"centroidY\", \"value\": "; put_double(meas, cy, true);
        meas += "}]";
        const char *sep = (first_chunk && k == lo) ? "" : ", ";
        // polygon
        fc += sep; fc += "{\"type\": \"Feature\", \"id\": \""; put_uuid4(fc, rng);
        fc += "\", \"geometry\": {\"type\": \"Polygon\", \"coordinates\": [[";
        const double *ring = j.xy + 2 * j.offs[i];
        const int64_t n = j.offs[i + 1] - j.offs[i];
        for (int64_t v = 0; v <= n; ++v) {                       // the ring is closed with a copy of its first vertex
            const double *pt = ring + 2 * (v == n ? 0 : v);
            double x = pt[0], y = pt[1];
            if (j.shift) { x -= j.bx; y -= j.by; }
            fc += v ? ", [" : "[";
            put_double(fc, x, true); fc += ", "; put_double(fc, y, true); fc += ']';
        }
        fc += "]]}, \"properties\": {\"objectType\": \"annotation\", \"isLocked\": false, \"classification\": ";
        fc += cls; fc += ", \"measurements\": "; fc += meas; fc += "}}";
        // centroid
        fp += sep; fp += "{\"type\": \"Feature\", \"id\": \""; put_uuid4(fp, rng);
        fp += "\", \"geometry\": {\"type\": \"Point\", \"coordinates\": [";
        put_double(fp, cx, true); fp += ", "; put_double(fp, cy, true);
        fp += "]}, \"properties\": {\"objectType\": \"annotation\", \"isLocked\": false, \"classification\": ";
        fp += cls; fp += ", \"measurements\": "; fp += meas; fp += "}}";
    }
}

}  // namespace

extern "C" int cpx_write_geojson(const char *contours_path, const char *centroids_path, const void *cells,
                                 int64_t n_cells, const double *centroids_xy, const double *xy_pool,
                                 const int64_t *offsets, const int64_t *keep, int64_t n_keep,
                                 const char *const *class_json, int n_class_json, double bounds_x, double bounds_y,
                                 int n_threads) {
    CPX_REQUIRE(contours_path && centroids_path && n_cells >= 0 && n_keep >= 0 && class_json && n_class_json > 0);
    CPX_REQUIRE(n_keep == 0 || (cells && centroids_xy && xy_pool && offsets && keep));
    for (int64_t k = 0; k < n_keep; ++k) CPX_REQUIRE(keep[k] >= 0 && keep[k] < n_cells);
    FILE *f1 = std::fopen(contours_path, "wb");
    FILE *f2 = f1 ? std::fopen(centroids_path, "wb") : nullptr;
    if (!f1 || !f2) {
        if (f1) std::fclose(f1);
        CPX_REQUIRE(!"cannot open the GeoJSON output files");
    }
    static const char head[] = "{\"type\": \"FeatureCollection\", \"features\": [";
    bool ok = std::fwrite(head, 1, sizeof head - 1, f1) == sizeof head - 1 &&
              std::fwrite(head, 1, sizeof head - 1, f2) == sizeof head - 1;
    Job job{(const Row *)cells, centroids_xy, xy_pool, offsets, keep, class_json, n_class_json,
            bounds_x, bounds_y, bounds_x != 0 || bounds_y != 0};
    const int64_t CH = 4096;                   // cells per chunk (~4 MB of contour text)
    const int64_t n_chunks = (n_keep + CH - 1) / CH;
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 16) nt = 16;
    if ((int64_t)nt > n_chunks) nt = (int)(n_chunks > 0 ? n_chunks : 1);
    // workers take chunks in order and hand their text to the writer (this thread) through a bounded window,
    // so at most `window` chunks are in memory whatever the size of the slide
    const int64_t window = 2 * nt + 2;         // <= 34 chunks in flight: bounded memory when the disk is the slow side
    std::vector<std::string> out1((size_t)window), out2((size_t)window);
    std::vector<char> ready((size_t)window, 0);
    std::mutex mu;
    std::condition_variable cv_ready, cv_free;
    std::atomic<int64_t> next{0};
    int64_t written = 0;                       // chunks the writer has consumed (guarded by mu)
    auto worker = [&](int id) {
        Rng rng((uint64_t)id + 1);
        for (;;) {
            const int64_t c = next.fetch_add(1);
            if (c >= n_chunks) return;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_free.wait(lk, [&] { return c < written + window; });
            }
            std::string a, b;
            a.reserve(1 << 20); b.reserve(1 << 19);
            const int64_t lo = c * CH, hi = lo + CH < n_keep ? lo + CH : n_keep;
            format_chunk(job, lo, hi, c == 0, rng, a, b);
            {
                std::lock_guard<std::mutex> lk(mu);
                out1[(size_t)(c % window)].swap(a); out2[(size_t)(c % window)].swap(b);
                ready[(size_t)(c % window)] = 1;
            }
            cv_ready.notify_all();
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < nt && n_chunks > 0; ++t) pool.emplace_back(worker, t);
    for (int64_t c = 0; c < n_chunks; ++c) {
        std::string a, b;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_ready.wait(lk, [&] { return ready[(size_t)(c % window)] != 0; });
            a.swap(out1[(size_t)(c % window)]); b.swap(out2[(size_t)(c % window)]);
            ready[(size_t)(c % window)] = 0;
            written = c + 1;
        }
        cv_free.notify_all();
        if (ok) ok = std::fwrite(a.data(), 1, a.size(), f1) == a.size() && std::fwrite(b.data(), 1, b.size(), f2) == b.size();
    }
    for (auto &t : pool) t.join();
    if (ok) ok = std::fwrite("]}", 1, 2, f1) == 2 && std::fwrite("]}", 1, 2, f2) == 2;
    ok = (std::fclose(f1) == 0) & ok;
    ok = (std::fclose(f2) == 0) & ok;
    CPX_REQUIRE(ok && "short write on a GeoJSON output file");
    return CPX_OK;
}
