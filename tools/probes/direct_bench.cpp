// Host cost of the batched DIRECT (host/direct.cpp) without a device: a cheap synthetic objective, the time inside the objective
// subtracted.  Prints the bookkeeping time per run, the batches, and a hash over every evaluated point and the result (two
// implementations that print the same hash walked the same trajectory).
//   g++ -O2 -std=c++17 -I../../include -I../../sequential-line-search_amd/host direct_bench.cpp ../../sequential-line-search_amd/host/direct.cpp -o bin/direct_bench
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "device.hpp"

using namespace sequential_line_search;

int main(int argc, char** argv)
{
    const int    D     = argc > 1 ? atoi(argv[1]) : 32;
    const int    evals = argc > 2 ? atoi(argv[2]) : 50 * D;
    const int    reps  = argc > 3 ? atoi(argv[3]) : 200;
    double       inside = 0.0;
    uint64_t     hash = 1469598103934665603ull;
    long         batches = 0, points = 0;
    auto         mix = [&](double v) { uint64_t b; std::memcpy(&b, &v, 8); hash = (hash ^ b) * 1099511628211ull; };
    optim::BatchObjective f = [&](const std::vector<std::vector<double>>& xs, std::vector<double>& vals) {
        const auto t0 = std::chrono::steady_clock::now();
        vals.resize(xs.size());
        for (size_t k = 0; k < xs.size(); ++k)
        {
            double s = 0.0;
            for (int i = 0; i < D; ++i)
            {
                const double u = xs[k][i] - 0.3 - 0.01 * i;
                s += -u * u + 0.05 * std::cos(9.0 * xs[k][i] + i);
                mix(xs[k][i]);
            }
            vals[k] = s;
        }
        ++batches; points += (long)xs.size();
        inside += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    std::vector<double> lo(D, 0.0), hi(D, 1.0);
    const auto          t0 = std::chrono::steady_clock::now();
    double              bv = 0; int used = 0;
    std::vector<double> x;
    for (int r = 0; r < reps; ++r)
    {
        hash = 1469598103934665603ull;   // the hash printed is the LAST run's: every run must walk the same trajectory
        x    = optim::DirectMaximize(f, lo, hi, evals, &bv, &used);
    }
    const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (double v : x) mix(v);
    mix(bv);
    printf("D %d max_evals %d: used %d, %.1f batches per run, bookkeeping %.1f us per run (objective %.1f us), best %.12g, hash %016llx\n", D, evals, used,
           (double)batches / reps, (total - inside) / reps * 1e6, inside / reps * 1e6, bv, (unsigned long long)hash);
    return 0;
}
