// Sequential line search on the D-dimensional bump exp(-|x - 0.4|^2) with a simulated user who picks the best point
// of every slider by brute force -- the scenario of the reference's demos/sequential_line_search_nd (main.cpp:11-37,63-76)
// as a CLI:   sequential_line_search_nd [D=8] [n_iterations=10] [seed=1] [use_MAP_hyperparams=1]
// use_MAP_hyperparams defaults to the reference demo's setting (main.cpp:21: true, which is also the constructor's default,
// sequential-line-search.hpp:37): the kernel hyper-parameters are estimated jointly with the goodness values on every submit.
// 0 selects the fixed-hyper-parameter variant (K cached, src/preference-regressor.cpp:363-371).
// Prints per iteration: objective value at the maximiser, residual norm |x - 0.4|, wall time of SubmitFeedbackData.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <sequential-line-search/sequential-line-search.hpp>
#include <sequential-line-search/utils.hpp>

using namespace sequential_line_search;
using Eigen::VectorXd;

static double Objective(const VectorXd& x)
{
    double q = 0.0;
    for (long i = 0; i < x.size(); ++i) q += (x(i) - 0.4) * (x(i) - 0.4);
    return std::exp(-q);
}

int main(int argc, char** argv)
{
    const int D      = argc > 1 ? std::atoi(argv[1]) : 8;
    const int n_iter = argc > 2 ? std::atoi(argv[2]) : 10;
    utils::SetRandomSeed(argc > 3 ? std::atoi(argv[3]) : 1);
    const bool use_map_hyperparams = argc > 4 ? std::atoi(argv[4]) != 0 : true;   // main.cpp:21

    SequentialLineSearchOptimizer optimizer(D, true, use_map_hyperparams, KernelType::ArdMatern52Kernel,
                                            AcquisitionFuncType::ExpectedImprovement);
    optimizer.SetHyperparams(0.50, 0.50, 0.001, 0.10, 0.01);   // main.cpp:11-15

    for (int it = 0; it < n_iter; ++it)
    {
        // simulated user: brute-force line search over the slider (main.cpp:63-76)
        double best_t = 0.0, best_v = -1.0;
        for (int k = 0; k <= 1000; ++k)
        {
            const double t = k / 1000.0;
            const double v = Objective(optimizer.CalcPointFromSliderPosition(t));
            if (v > best_v) { best_v = v; best_t = t; }
        }
        const auto t0 = std::chrono::steady_clock::now();
        optimizer.SubmitFeedbackData(best_t);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

        const VectorXd x = optimizer.GetMaximizer();
        double         r = 0.0;
        for (long i = 0; i < x.size(); ++i) r += (x(i) - 0.4) * (x(i) - 0.4);
        std::cout << "iter " << it + 1 << "  objective " << Objective(x) << "  residual " << std::sqrt(r) << "  ms " << ms << std::endl;
    }
    return 0;
}
