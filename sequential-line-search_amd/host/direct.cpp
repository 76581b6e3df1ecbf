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
            // 3^(-2 l) and 3^-(l + 1), tabulated once with the calls the untabulated code made (same values): the selection step
            // needs every rectangle's size (1600 rectangles x 32 dimensions in sequential_line_search_nd: std::pow was most of the
            // 1.9 ms DIRECT once spent on the host)
            struct Pow3
            {
                double m2[256], m1[256];
                Pow3()
                {
                    for (int l = 0; l < 256; ++l)
                    {
                        m2[l] = std::pow(3.0, -2.0 * l);
                        m1[l] = std::pow(3.0, -(l + 1.0));
                    }
                }
            };
            const Pow3& Tables()
            {
                static const Pow3 t;
                return t;
            }

            // keys: size classes are identified by the sorted multiset of levels; the half diagonal is a function of it.
            // Rounded to 12 significant digits so that equal multisets compare equal whatever the summation order.
            long long SizeKey(double d) { return std::llround(std::log(d) * 1e9); }

            // The rectangles, structure-of-arrays (round 4: one Rect object with two heap vectors per rectangle, sizes recomputed and
            // looked up in a std::map for every rectangle in every iteration, fresh vectors for every sample: 0.76 ms of host time
            // per 1600-evaluation run at D = 32 where the device needs 0.13 ms for the evaluations; same trajectory, see
            // tools/probes/direct_bench.cpp).  A rectangle's half diagonal and size class are computed when its levels change.
            struct Rects
            {
                size_t                     n = 0;
                std::vector<double>        c;       // [rect][n] centre in the unit cube
                std::vector<unsigned char> level;   // [rect][n] trisections per dimension: side_i = 3^-level_i
                std::vector<double>        g;       // value to MINIMISE (= -f)
                std::vector<double>        diag;    // half diagonal
                std::vector<int>           cls;     // size class id
                std::map<long long, int>   cls_of_key;
                size_t                     size() const { return g.size(); }
                void                       Resize(size_t r) { diag.resize(r); cls.resize(r); }
                void                       Classify(size_t r)
                {
                    const double*        t = Tables().m2;
                    const unsigned char* l = level.data() + r * n;
                    double               s = 0.0;
                    for (size_t i = 0; i < n; ++i) s += t[l[i]];
                    const double d = 0.5 * std::sqrt(s);
                    diag[r]        = d;
                    const auto ins = cls_of_key.emplace(SizeKey(d), static_cast<int>(cls_of_key.size()));
                    cls[r]         = ins.first->second;
                }
                // copy of rectangle `src` (as it is now) with centre coordinate i moved by `shift` and value `value`; like = none:
                // classify it, else: it has the levels, and therefore the size and class, of rectangle `like`
                size_t Child(size_t src, size_t i, double shift, double value, size_t like)
                {
                    const size_t r = size();
                    // grow first, then copy by index: inserting a range of the vector into itself is undefined behaviour
                    c.resize((r + 1) * n);
                    level.resize((r + 1) * n);
                    std::copy_n(c.data() + src * n, n, c.data() + r * n);
                    std::copy_n(level.data() + src * n, n, level.data() + r * n);
                    c[r * n + i] += shift;
                    g.push_back(value);
                    Resize(r + 1);
                    if (like == static_cast<size_t>(-1)) Classify(r);
                    else
                    {
                        diag[r] = diag[like];
                        cls[r]  = cls[like];
                    }
                    return r;
                }
                void Clear(size_t dims)
                {
                    n = dims;
                    c.clear(); level.clear(); g.clear(); diag.clear(); cls.clear(); cls_of_key.clear();
                }
            };
        } // namespace

        std::vector<double> DirectMaximize(const BatchObjective& f, const std::vector<double>& lower, const std::vector<double>& upper,
                                           int max_evals, double* best_value, int* evals_used)
        {
            const size_t n      = lower.size();
            const Pow3&  tables = Tables();
            auto         to_box = [&](const double* u, std::vector<double>& x) {
                x.resize(n);
                for (size_t i = 0; i < n; ++i) x[i] = lower[i] + u[i] * (upper[i] - lower[i]);
            };
            // storage kept between calls (per thread; a run touches ~0.5 MB at D = 32, 1600 evaluations); an objective that itself
            // runs DIRECT on this thread gets storage of its own
            thread_local Rects                            kept_rects;
            thread_local std::vector<std::vector<double>> kept_xs, kept_spare;
            thread_local bool                             kept_in_use = false;
            const bool                                    own = kept_in_use;
            Rects                                         own_rects;
            std::vector<std::vector<double>>              own_xs, own_spare;
            struct Release
            {
                bool* flag;
                ~Release() { if (flag) *flag = false; }
            } release{own ? nullptr : &kept_in_use};
            if (!own) kept_in_use = true;
            Rects& rects = own ? own_rects : kept_rects;
            rects.Clear(n);
            {
                const size_t cap = static_cast<size_t>(std::max(max_evals, 1)) + 1;
                rects.c.reserve(cap * n); rects.level.reserve(cap * n); rects.g.reserve(cap); rects.diag.reserve(cap); rects.cls.reserve(cap);
            }
            // sample points of an iteration: the vectors are kept and refilled (the batch interface wants a vector of vectors)
            std::vector<std::vector<double>>& xs    = own ? own_xs : kept_xs;
            std::vector<std::vector<double>>& spare = own ? own_spare : kept_spare;
            auto                              next_x = [&]() -> std::vector<double>& {
                if (spare.empty()) xs.emplace_back();
                else
                {
                    xs.push_back(std::move(spare.back()));
                    spare.pop_back();
                }
                return xs.back();
            };
            auto recycle_xs = [&]() {
                while (!xs.empty())
                {
                    spare.push_back(std::move(xs.back()));
                    xs.pop_back();
                }
            };
            std::vector<double> vals;
            int                 evals = 0;
            recycle_xs();
            {
                rects.c.assign(n, 0.5);
                rects.level.assign(n, 0);
                to_box(rects.c.data(), next_x());
                f(xs, vals);
                double g0 = -vals[0];
                if (!std::isfinite(g0)) g0 = std::numeric_limits<double>::max();
                rects.g.push_back(g0);
                rects.Resize(1);
                rects.Classify(0);
                evals = 1;
            }
            size_t best = 0;
            // scratch of the iterations
            struct Job { size_t rect, first, dims_begin, dims_end; };
            std::vector<Job>                       jobs;
            std::vector<size_t>                    job_dims, class_best, hull, order;
            std::vector<std::pair<double, size_t>> pts;   // (size, rect), ascending size
            std::vector<double>                    up, gp, gm, w;
            const size_t                           none = std::numeric_limits<size_t>::max();
            while (evals < max_evals)
            {
                // --- potentially optimal rectangles: the best rectangle of every size class (first one on ties) ---
                class_best.assign(rects.cls_of_key.size(), none);
                for (size_t i = 0; i < rects.size(); ++i)
                {
                    size_t& b = class_best[rects.cls[i]];
                    if (b == none || rects.g[i] < rects.g[b]) b = i;
                }
                pts.clear();
                for (size_t b : class_best)
                    if (b != none) pts.emplace_back(rects.diag[b], b);
                std::sort(pts.begin(), pts.end());
                // start at the class holding the overall best value (largest such class), hull towards larger sizes
                size_t start = 0;
                for (size_t k = 0; k < pts.size(); ++k)
                    if (rects.g[pts[k].second] <= rects.g[pts[start].second]) start = k;
                hull.clear();   // indices into pts
                for (size_t k = start; k < pts.size(); ++k)
                {
                    while (hull.size() >= 2)
                    {
                        const auto& a = pts[hull[hull.size() - 2]];
                        const auto& b = pts[hull.back()];
                        const auto& c = pts[k];
                        // b is above the chord a-c  ->  not on the lower hull
                        const double cross = (b.first - a.first) * (rects.g[c.second] - rects.g[a.second]) -
                                             (rects.g[b.second] - rects.g[a.second]) * (c.first - a.first);
                        if (cross <= 0.0) hull.pop_back();
                        else break;
                    }
                    // the hull must go DOWN-RIGHT to UP-RIGHT only through points that beat their left neighbour's slope; a
                    // point with a larger value than the previous hull point is still potentially optimal (large K)
                    hull.push_back(k);
                }
                // --- samples of this iteration (one batch) ---
                jobs.clear();
                job_dims.clear();
                recycle_xs();
                for (size_t hk : hull)
                {
                    const size_t         ri = pts[hk].second;
                    const unsigned char* lv = rects.level.data() + ri * n;
                    unsigned char        lmin = 255;
                    for (size_t i = 0; i < n; ++i) lmin = std::min(lmin, lv[i]);
                    if (lmin >= 30) continue;   // side 3^-30: nothing left to resolve
                    Job job;
                    job.rect       = ri;
                    job.first      = xs.size();
                    job.dims_begin = job_dims.size();
                    for (size_t i = 0; i < n; ++i)
                        if (lv[i] == lmin) job_dims.push_back(i);
                    job.dims_end   = job_dims.size();
                    const int need = 2 * static_cast<int>(job.dims_end - job.dims_begin);
                    if (evals + static_cast<int>(xs.size()) + need > max_evals)   // the budget is a cap
                    {
                        job_dims.resize(job.dims_begin);
                        break;
                    }
                    const double  delta = tables.m1[lmin];
                    const double* cc    = rects.c.data() + ri * n;
                    for (size_t d = job.dims_begin; d < job.dims_end; ++d)
                    {
                        const size_t i = job_dims[d];
                        up.assign(cc, cc + n);
                        up[i] = cc[i] + delta;
                        to_box(up.data(), next_x());
                        up[i] = cc[i] - delta;
                        to_box(up.data(), next_x());
                    }
                    jobs.push_back(job);
                }
                if (jobs.empty()) break;
                f(xs, vals);
                evals += static_cast<int>(xs.size());
                // --- trisect ---
                for (const Job& job : jobs)
                {
                    const size_t m = job.dims_end - job.dims_begin;
                    gp.resize(m); gm.resize(m); w.resize(m);
                    for (size_t k = 0; k < m; ++k)
                    {
                        gp[k] = -vals[job.first + 2 * k];
                        gm[k] = -vals[job.first + 2 * k + 1];
                        if (!std::isfinite(gp[k])) gp[k] = std::numeric_limits<double>::max();
                        if (!std::isfinite(gm[k])) gm[k] = std::numeric_limits<double>::max();
                        w[k] = std::min(gp[k], gm[k]);
                    }
                    order.resize(m);
                    std::iota(order.begin(), order.end(), 0);
                    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return w[a] < w[b]; });
                    const unsigned char lmin  = rects.level[job.rect * n + job_dims[job.dims_begin]];
                    const double        delta = tables.m1[lmin];
                    for (size_t k : order)
                    {
                        const size_t i = job_dims[job.dims_begin + k];
                        rects.level[job.rect * n + i] += 1;
                        const size_t first = rects.Child(job.rect, i, delta, gp[k], none);
                        rects.Child(job.rect, i, -delta, gm[k], first);
                    }
                    rects.Classify(job.rect);   // the parent has shrunk
                }
                for (size_t i = 0; i < rects.size(); ++i)
                    if (rects.g[i] < rects.g[best]) best = i;
            }
            if (best_value) *best_value = -rects.g[best];
            if (evals_used) *evals_used = evals;
            std::vector<double> x;
            to_box(rects.c.data() + best * n, x);
            return x;
        }
    } // namespace optim
} // namespace sequential_line_search
