// DIRECT (DIviding RECTangles; Jones, Perttunen, Stuckman 1993), batched: the global phase of the reference's DEFAULT
// maximiser branch (src/acquisition-function.cpp:155-165, nlopt::GN_DIRECT with max_evals = num_global_search_iters) and of
// GaussianProcessRegressor::PerformMapEstimation (src/gaussian-process-regressor.cpp:294).
//
// NLopt is not available, so this is the published algorithm, not NLopt's code (iterates are not comparable; the
// sampling pattern and the selection rule are): the box is scaled to the unit cube; every rectangle keeps its centre, its
// value and, per dimension, how often it has been trisected.  One iteration
//   1. picks the potentially optimal rectangles -- the lower-right convex hull of (size, value) over the best rectangle
//      of every size class, with Jones' epsilon = 0 as NLopt's GN_DIRECT uses (size = half diagonal);
//   2. samples centre +- side/3 along every longest side of each picked rectangle -- ALL picked rectangles' samples of
//      an iteration go to the objective in ONE batch (one sls_acq_eval call on the device);
//   3. trisects along those sides, best sample pair first, so the best values sit in the largest children.
// The evaluation budget is a cap: an iteration only starts rectangles whose samples still fit.
#include <algorithm>
#include <cmath>
#include <limits>
#include <map>
#include <numeric>

#include "device.hpp"

namespace sequential_line_search
{
    namespace optim
    {
        namespace
        {
            struct Rect
            {
                std::vector<double>        c;       // centre in the unit cube
                std::vector<unsigned char> level;   // trisections per dimension: side_i = 3^-level_i
                double                     g;       // value to MINIMISE (= -f)
            };

            // 3^(-2 l), tabulated once: the selection step asks for every rectangle's size in every iteration (1600 rectangles x 32
            // dimensions x 11 iterations in sequential_line_search_nd: std::pow was most of the 1.9 ms DIRECT spent on the host)
            const double* Pow3m2()
            {
                static const std::vector<double> t = [] {
                    std::vector<double> v(256);
                    for (int l = 0; l < 256; ++l) v[l] = std::pow(3.0, -2.0 * l);
                    return v;
                }();
                return t.data();
            }
            double HalfDiagonal(const Rect& r)
            {
                const double* t = Pow3m2();
                double        s = 0.0;
                for (unsigned char l : r.level) s += t[l];
                return 0.5 * std::sqrt(s);
            }

            // keys: size classes are identified by the sorted multiset of levels; the half diagonal is a function of it.
            // Rounded to 12 significant digits so that equal multisets compare equal whatever the summation order.
            long long SizeKey(double d) { return std::llround(std::log(d) * 1e9); }
        } // namespace

        std::vector<double> DirectMaximize(const BatchObjective& f, const std::vector<double>& lower, const std::vector<double>& upper,
                                           int max_evals, double* best_value, int* evals_used)
        {
            const size_t n = lower.size();
            auto to_box = [&](const std::vector<double>& u) {
                std::vector<double> x(n);
                for (size_t i = 0; i < n; ++i) x[i] = lower[i] + u[i] * (upper[i] - lower[i]);
                return x;
            };
            std::vector<Rect> rects;
            int               evals = 0;
            {
                Rect r0;
                r0.c.assign(n, 0.5);
                r0.level.assign(n, 0);
                std::vector<double> v;
                f({to_box(r0.c)}, v);
                r0.g = -v[0];
                if (!std::isfinite(r0.g)) r0.g = std::numeric_limits<double>::max();
                rects.push_back(r0);
                evals = 1;
            }
            size_t best = 0;
            while (evals < max_evals)
            {
                // --- potentially optimal rectangles ---
                std::map<long long, size_t> cls;   // size class -> index of its best rectangle (first one on ties)
                for (size_t i = 0; i < rects.size(); ++i)
                {
                    const long long k  = SizeKey(HalfDiagonal(rects[i]));
                    auto            it = cls.find(k);
                    if (it == cls.end()) cls[k] = i;
                    else if (rects[i].g < rects[it->second].g) it->second = i;
                }
                std::vector<std::pair<double, size_t>> pts;   // (size, rect), ascending size
                for (const auto& kv : cls) pts.emplace_back(HalfDiagonal(rects[kv.second]), kv.second);
                std::sort(pts.begin(), pts.end());
                // start at the class holding the overall best value (largest such class), hull towards larger sizes
                size_t start = 0;
                for (size_t k = 0; k < pts.size(); ++k)
                    if (rects[pts[k].second].g <= rects[pts[start].second].g) start = k;
                std::vector<size_t> hull;   // indices into pts
                for (size_t k = start; k < pts.size(); ++k)
                {
                    while (hull.size() >= 2)
                    {
                        const auto& a = pts[hull[hull.size() - 2]];
                        const auto& b = pts[hull.back()];
                        const auto& c = pts[k];
                        // b is above the chord a-c  ->  not on the lower hull
                        const double cross = (b.first - a.first) * (rects[c.second].g - rects[a.second].g) -
                                             (rects[b.second].g - rects[a.second].g) * (c.first - a.first);
                        if (cross <= 0.0) hull.pop_back();
                        else break;
                    }
                    // the hull must go DOWN-RIGHT to UP-RIGHT only through points that beat their left neighbour's slope; a
                    // point with a larger value than the previous hull point is still potentially optimal (large K)
                    hull.push_back(k);
                }
                // --- samples of this iteration (one batch) ---
                struct Job { size_t rect; std::vector<size_t> dims; size_t first; };
                std::vector<Job>                 jobs;
                std::vector<std::vector<double>> xs;
                for (size_t hk : hull)
                {
                    const size_t ri = pts[hk].second;
                    const Rect&  r  = rects[ri];
                    unsigned char lmin = 255;
                    for (unsigned char l : r.level) lmin = std::min(lmin, l);
                    if (lmin >= 30) continue;   // side 3^-30: nothing left to resolve
                    Job job;
                    job.rect  = ri;
                    job.first = xs.size();
                    for (size_t i = 0; i < n; ++i)
                        if (r.level[i] == lmin) job.dims.push_back(i);
                    const int need = 2 * static_cast<int>(job.dims.size());
                    if (evals + static_cast<int>(xs.size()) + need > max_evals) break;   // the budget is a cap
                    const double delta = std::pow(3.0, -(lmin + 1.0));
                    for (size_t i : job.dims)
                    {
                        std::vector<double> up = r.c, dn = r.c;
                        up[i] += delta;
                        dn[i] -= delta;
                        xs.push_back(to_box(up));
                        xs.push_back(to_box(dn));
                    }
                    jobs.push_back(job);
                }
                if (jobs.empty()) break;
                std::vector<double> vals;
                f(xs, vals);
                evals += static_cast<int>(xs.size());
                // --- trisect ---
                for (const Job& job : jobs)
                {
                    const size_t        m = job.dims.size();
                    std::vector<double> gp(m), gm(m), w(m);
                    for (size_t k = 0; k < m; ++k)
                    {
                        gp[k] = -vals[job.first + 2 * k];
                        gm[k] = -vals[job.first + 2 * k + 1];
                        if (!std::isfinite(gp[k])) gp[k] = std::numeric_limits<double>::max();
                        if (!std::isfinite(gm[k])) gm[k] = std::numeric_limits<double>::max();
                        w[k] = std::min(gp[k], gm[k]);
                    }
                    std::vector<size_t> order(m);
                    std::iota(order.begin(), order.end(), 0);
                    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return w[a] < w[b]; });
                    const unsigned char lmin  = rects[job.rect].level[job.dims[0]];
                    const double        delta = std::pow(3.0, -(lmin + 1.0));
                    for (size_t k : order)
                    {
                        const size_t i = job.dims[k];
                        rects[job.rect].level[i] += 1;
                        Rect up = rects[job.rect], dn = rects[job.rect];
                        up.c[i] += delta; up.g = gp[k];
                        dn.c[i] -= delta; dn.g = gm[k];
                        rects.push_back(up);
                        rects.push_back(dn);
                    }
                }
                for (size_t i = 0; i < rects.size(); ++i)
                    if (rects[i].g < rects[best].g) best = i;
            }
            if (best_value) *best_value = -rects[best].g;
            if (evals_used) *evals_used = evals;
            return to_box(rects[best].c);
        }
    } // namespace optim
} // namespace sequential_line_search
