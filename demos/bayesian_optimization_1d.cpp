// Bayesian optimisation of f(x) = 1 - 1.5 x sin(13 x) on [0,1] with GaussianProcessRegressor (MAP hyper-parameters) and
// EI -- the scenario of the reference's demos/bayesian_optimization_1d (objective: core.cpp:70-73), as a CLI:
//   bayesian_optimization_1d [n_trials=1] [n_iterations=20] [seed=1]
// Prints the found maximiser per trial (true optimum: x = 0.852733, f = 2.273928).
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

static double Objective(const VectorXd& x) { return 1.0 - 1.5 * x(0) * std::sin(x(0) * 13.0); }

int main(int argc, char** argv)
{
    const int n_trials     = argc > 1 ? std::atoi(argv[1]) : 1;
    const int n_iterations = argc > 2 ? std::atoi(argv[2]) : 20;
    utils::SetRandomSeed(argc > 3 ? std::atoi(argv[3]) : 1);

    for (int trial = 0; trial < n_trials; ++trial)
    {
        MatrixXd X(1, 0);
        VectorXd y(0);
        std::shared_ptr<GaussianProcessRegressor> regressor;
        VectorXd x_max(1);
        double   y_max = NAN;
        for (int it = 0; it < n_iterations; ++it)
        {
            const VectorXd x = (X.cols() == 0) ? utils::GenerateRandomVector(1) : acquisition_func::FindNextPoint(*regressor);
            const double   v = Objective(x);
            X                = eig::AppendCol(X, x);
            VectorXd y_new(y.size() + 1);
            for (long i = 0; i < y.size(); ++i) y_new(i) = y(i);
            y_new(y.size()) = v;
            y               = y_new;
            regressor       = std::make_shared<GaussianProcessRegressor>(X, y);
            // current estimate: the data point with the largest predicted mean (core.cpp:39-48)
            x_max = regressor->PredictMaximumPointFromData();
            y_max = regressor->PredictMu(x_max);
            std::cout << "iter " << it + 1 << "  x " << x(0) << "  y " << v << "  x_max " << x_max(0) << "  y_max " << y_max
                      << "  a " << regressor->GetKernelHyperparams()(0) << "  r " << regressor->GetKernelHyperparams()(1) << "  b "
                      << regressor->GetNoiseHyperparam() << std::endl;
        }
        std::cout << "trial " << trial + 1 << " maximizer " << x_max(0) << " maximum " << y_max << std::endl;
    }
    return 0;
}
