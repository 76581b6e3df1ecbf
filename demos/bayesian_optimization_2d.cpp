// Bayesian optimisation of the two-bump objective of the reference's 2-D GUI demo,
//     f(x) = exp(-|x - (0.3, 0.3)|^2 / 0.3^2) + 1.5 exp(-|x - (0.7, 0.7)|^2 / 0.4^2)
// (demos/bayesian_optimization_2d_gui/core.cpp:80-88), with the loop of its Core::proceedOptimization (:26-55): first point
// uniform in [0.05, 0.95]^2, then x = FindNextPoint(regressor), GaussianProcessRegressor(X, y) refitted with MAP
// hyper-parameters after every observation, current estimate = the data point with the largest predicted mean.  No GUI:
//   bayesian_optimization_2d [n_iterations=25] [seed=1]
// The maximum of the sum is f = 1.532995 at x = (0.682333, 0.682333) (the low bump's tail pulls it off (0.7, 0.7); scipy from both
// bump centres, tests/test_gpu_host_cpp.py).
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <sequential-line-search/acquisition-function.hpp>
#include <sequential-line-search/gaussian-process-regressor.hpp>
#include <sequential-line-search/utils.hpp>

using namespace sequential_line_search;
using Eigen::MatrixXd;
using Eigen::VectorXd;

static double Bump(const VectorXd& x, double cx, double cy, double sigma)
{
    const double dx = x(0) - cx, dy = x(1) - cy;
    return std::exp(-(dx * dx + dy * dy) / (sigma * sigma));
}
static double Objective(const VectorXd& x) { return Bump(x, 0.3, 0.3, 0.3) + 1.5 * Bump(x, 0.7, 0.7, 0.4); }

int main(int argc, char** argv)
{
    const int n_iterations = argc > 1 ? std::atoi(argv[1]) : 25;
    utils::SetRandomSeed(argc > 2 ? std::atoi(argv[2]) : 1);
    MatrixXd X(2, 0);
    VectorXd y(0);
    std::shared_ptr<GaussianProcessRegressor> regressor;
    VectorXd x_max(2);
    double   y_max = NAN;
    for (int it = 0; it < n_iterations; ++it)
    {
        VectorXd x(2);
        if (X.cols() == 0)
        {
            const VectorXd u = utils::GenerateRandomVector(2);
            x(0) = 0.05 + 0.90 * u(0);
            x(1) = 0.05 + 0.90 * u(1);
        }
        else
        {
            x = acquisition_func::FindNextPoint(*regressor);
        }
        const double v = Objective(x);
        X              = eig::AppendCol(X, x);
        VectorXd y_new(y.size() + 1);
        for (long i = 0; i < y.size(); ++i) y_new(i) = y(i);
        y_new(y.size()) = v;
        y               = y_new;
        regressor       = std::make_shared<GaussianProcessRegressor>(X, y);
        x_max           = regressor->PredictMaximumPointFromData();
        y_max           = regressor->PredictMu(x_max);
        std::cout << "iter " << it + 1 << "  x " << x(0) << " " << x(1) << "  y " << v << "  x_max " << x_max(0) << " " << x_max(1)
                  << "  y_max " << y_max << std::endl;
    }
    std::cout << "maximizer " << x_max(0) << " " << x_max(1) << " maximum " << y_max << std::endl;
    return 0;
}
