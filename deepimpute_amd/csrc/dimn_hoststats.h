// dimn_hoststats.h -- host-side per-gene statistics of the drop-in's planning (reference deepimpute/multinet.py:191:
// `raw.var() / (1 + raw.mean())` picks and orders the genes to impute, so these numbers must be pandas' TO THE BIT).
//
// pandas (nanops.nanmean / nanvar, no bottleneck) on a float64 frame without NaN computes, per column j,
//     mean_j = (sum_i a[i][j]) / n                      the frame's block is the TRANSPOSED view of the C-ordered matrix, so numpy
//                                                       reduces it row by row: out[j] += a[i][j], a sequential sum over i
//     var_j  = (sum_i (avg_j - a[i][j])^2) / (n - 1)    nanvar first makes a C-ordered COPY of the block (values.copy() for its NaN
//              avg_j = (sum_i a[i][j]) / n              mask), so both of ITS sums run along the contiguous axis: numpy's pairwise
//                                                       summation (blocks of <= 128 with eight running partial sums, halves cut at
//                                                       multiples of 8) applied to chunks of 8192 elements (numpy's buffer size)
//                                                       added in order -- avg_j is not bit-equal to mean_j in general
// -- two different summation orders, both restated here exactly (tests/test_shell.py compares to the bit).  Every thread owns a
// stripe of 128 columns (a 1 KB contiguous piece per row), two sweeps over the rows; the first pass also yields the matrix maximum predict() needs
// (multinet.py:292).  A NaN anywhere makes the call report it and the Python side falls back to pandas (skipna semantics).
// Measured at 50k x 20k: 0.8 s for pandas on column blocks from a thread pool -> see DESIGN.md section 6c.  Host-only.
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#define HOSTSTATS_W 128      // columns per stripe (one thread): 1 KB contiguous per row
#pragma clang fp contract(off)
// res[j] = numpy's pairwise sum over rows i0 .. i0+n-1 of a[i][j0+j] (SQDEV: of (avg[j] - a[i][j0+j])^2), j < w  (numpy/_core/src/umath/loops_utils.h.src,
// @TYPE@_pairwise_sum: n < 8 a plain loop from 0; n <= 128 eight partial sums r[k] over i = k mod 8, combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail; larger n split at n/2 rounded down to a multiple of 8)
struct HostSide { double* s; double mx; bool nan; double *cmin, *cmax; };      // what the first pass gathers on the way: sequential sums, maximum, NaN seen, (optional) per-column extremes
template <bool SQDEV>
static void hoststats_pairwise(const double* a, int64_t ld, int64_t j0, int w, const double* avg, int64_t i0, int64_t n, double* res, HostSide* side) {
    // rows are visited in increasing order (left half before right half, a leaf front to back), so the plain running sums of
    // the first pass (side->s: DataFrame.mean()'s order) come out of the same sweep as its pairwise sums
    auto f = [&](int64_t i, int j) {
        const double x = a[(i0 + i) * ld + j0 + j];
        if (!SQDEV) {
            side->s[j] += x;
            side->nan |= (x != x);
            side->mx = x > side->mx ? x : side->mx;
            if (side->cmin) { side->cmin[j] = x < side->cmin[j] ? x : side->cmin[j]; side->cmax[j] = x > side->cmax[j] ? x : side->cmax[j]; }
            return x;
        }
        const double d = avg[j] - x;
        return d * d;
    };
    if (n < 8) {
        for (int j = 0; j < w; ++j) res[j] = 0.0;
        for (int64_t i = 0; i < n; ++i)
            for (int j = 0; j < w; ++j) res[j] += f(i, j);
    } else if (n <= 128) {
        double r[8][HOSTSTATS_W];
        for (int k = 0; k < 8; ++k)
            for (int j = 0; j < w; ++j) r[k][j] = f(k, j);
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k)
                for (int j = 0; j < w; ++j) r[k][j] += f(i + k, j);
        for (int j = 0; j < w; ++j) res[j] = ((r[0][j] + r[1][j]) + (r[2][j] + r[3][j])) + ((r[4][j] + r[5][j]) + (r[6][j] + r[7][j]));
        for (; i < n; ++i)
            for (int j = 0; j < w; ++j) res[j] += f(i, j);
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        double hi[HOSTSTATS_W];
        hoststats_pairwise<SQDEV>(a, ld, j0, w, avg, i0, n2, res, side);
        hoststats_pairwise<SQDEV>(a, ld, j0, w, avg, i0 + n2, n - n2, hi, side);
        for (int j = 0; j < w; ++j) res[j] += hi[j];
    }
}

// which = 1: the first sweep (mean, nanvar's own average, column extremes, maximum, NaN); 2: the second sweep (var from the given
// averages); 3: both
static void hoststats_stripe(const double* a, int64_t n, int64_t ld, int64_t j0, int64_t j1, double* mean, double* avgpw, double* var, double* cmin, double* cmax,
                             double* vmax, int* has_nan, int which) {
    const int w = (int)(j1 - j0);
    double s[HOSTSTATS_W], q[HOSTSTATS_W], part[HOSTSTATS_W], avg[HOSTSTATS_W], lo[HOSTSTATS_W], hi[HOSTSTATS_W];
    HostSide side{s, -INFINITY, false, cmin ? lo : nullptr, cmin ? hi : nullptr};
    for (int j = 0; j < w; ++j) { s[j] = 0.0; lo[j] = INFINITY; hi[j] = -INFINITY; }
    if (!(which & 1)) for (int j = 0; j < w; ++j) avg[j] = avgpw[j0 + j];
    // nanvar works on a C-ordered COPY of the block (values.copy() for the NaN mask), so BOTH of its sums run along the contiguous
    // axis: numpy hands the reduction loop at most `bufsize` = 8192 elements at a time -- out = 0; out += pairwise(chunk) per chunk
    for (int pass = (which & 1) ? 0 : 1; pass < ((which & 2) ? 2 : 1); ++pass) {
        for (int j = 0; j < w; ++j) q[j] = 0.0;
        for (int64_t c0 = 0; c0 < n; c0 += 8192) {
            if (pass == 0) hoststats_pairwise<false>(a, ld, j0, w, avg, c0, std::min<int64_t>(8192, n - c0), part, &side);
            else hoststats_pairwise<true>(a, ld, j0, w, avg, c0, std::min<int64_t>(8192, n - c0), part, &side);
            for (int j = 0; j < w; ++j) q[j] += part[j];
        }
        if (pass == 0)
            for (int j = 0; j < w; ++j) {
                avg[j] = q[j] / (double)n; mean[j0 + j] = s[j] / (double)n;
                if (avgpw) avgpw[j0 + j] = avg[j];
                if (cmin) { cmin[j0 + j] = lo[j]; cmax[j0 + j] = hi[j]; }
            }
    }
    if (which & 2) for (int j = 0; j < w; ++j) var[j0 + j] = q[j] / (double)(n - 1);
    *vmax = side.mx;
    *has_nan = side.nan ? 1 : 0;
}

// mean[g], var[g] (ddof = 1; var may be NULL), *vmax = max over the matrix, *has_nan; a[i * ld + j], i < n, j < g
static int hoststats_run(const double* a, int64_t n, int64_t g, int64_t ld, double* mean, double* var, double* vmax, int* has_nan, int threads,
                         int which = 3, double* avgpw = nullptr, double* cmin = nullptr, double* cmax = nullptr) {
    if (!var) which &= 1;
    const int64_t stripes = (g + HOSTSTATS_W - 1) / HOSTSTATS_W;
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, stripes));
    std::vector<double> mx((size_t)stripes, -INFINITY);
    std::vector<int> nn((size_t)stripes, 0);
    std::atomic<int64_t> next(0);
    auto work = [&]() {
        for (;;) {
            const int64_t st = next.fetch_add(1);
            if (st >= stripes) return;
            hoststats_stripe(a, n, ld, st * HOSTSTATS_W, std::min<int64_t>(g, st * HOSTSTATS_W + HOSTSTATS_W), mean, avgpw, var, cmin, cmax, &mx[(size_t)st], &nn[(size_t)st], which);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    double m = -INFINITY; int any = 0;
    for (int64_t st = 0; st < stripes; ++st) { m = mx[(size_t)st] > m ? mx[(size_t)st] : m; any |= nn[(size_t)st]; }
    *vmax = m; *has_nan = any;
    return 0;
}
