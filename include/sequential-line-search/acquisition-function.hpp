// Acquisition functions and their maximisation
// (reference surface: include/sequential-line-search/acquisition-function.hpp:11-79).
#ifndef SEQUENTIAL_LINE_SEARCH_ACQUISITION_FUNCTION_HPP
#define SEQUENTIAL_LINE_SEARCH_ACQUISITION_FUNCTION_HPP

#include <memory>
#include <sequential-line-search/eigen-lite.hpp>
#include <sequential-line-search/regressor.hpp>
#include <vector>

namespace sequential_line_search
{
    enum class AcquisitionFuncType
    {
        ExpectedImprovement,
        GaussianProcessUpperConfidenceBound,
    };

    namespace acquisition_func
    {
        /// Acquisition value at x (0 if the regressor holds no data).  `..._hyperparam` is the GP-UCB trade-off weight
        /// (ignored for EI).
        double CalcAcquisitionValue(const Regressor& regressor, const Eigen::VectorXd& x, const AcquisitionFuncType func_type,
                                    const double gaussian_process_upper_confidence_bound_hyperparam = 1.0);

        Eigen::VectorXd CalcAcquisitionValueDerivative(const Regressor& regressor, const Eigen::VectorXd& x,
                                                       const AcquisitionFuncType func_type,
                                                       const double gaussian_process_upper_confidence_bound_hyperparam = 1.0);

        /// Batched value (and gradient, D x M, if grad != nullptr) for the columns of Xs: one device pass.
        Eigen::VectorXd CalcAcquisitionValues(const Regressor& regressor, const Eigen::MatrixXd& Xs, const AcquisitionFuncType func_type,
                                              const double gaussian_process_upper_confidence_bound_hyperparam = 1.0,
                                              Eigen::MatrixXd* grad = nullptr);

        /// Maximiser of the acquisition function over [0,1]^D: `num_global_search_iters` random starts, each refined by a
        /// bounded L-BFGS limited to `num_local_search_iters` objective evaluations, all starts advanced in lock step on the
        /// GPU (the reference's SEQUENTIAL_LINE_SEARCH_USE_PARALLELIZED_MULTI_START_SEARCH branch).
        Eigen::VectorXd FindNextPoint(const Regressor& regressor, const unsigned num_global_search_iters = 100,
                                      const unsigned            num_local_search_iters = 50,
                                      const AcquisitionFuncType func_type              = AcquisitionFuncType::ExpectedImprovement,
                                      const double              gaussian_process_upper_confidence_bound_hyperparam = 1.0);

        /// Same from an explicit start set (D x S), for reproducible runs and multi-GPU sharding; returns also the value.
        Eigen::VectorXd FindNextPointFromStarts(const Regressor& regressor, const Eigen::MatrixXd& starts,
                                                const unsigned num_local_search_iters, const AcquisitionFuncType func_type,
                                                const double gaussian_process_upper_confidence_bound_hyperparam, double* value = nullptr);

        /// Sequential batch of `num_points` maximisers [Schonlau+ 1998]: after each point the predictive variance is updated
        /// with the new point, the mean is kept.
        std::vector<Eigen::VectorXd> FindNextPoints(const Regressor& regressor, const unsigned num_points,
                                                    const unsigned            num_global_search_iters = 100,
                                                    const unsigned            num_local_search_iters  = 50,
                                                    const AcquisitionFuncType func_type = AcquisitionFuncType::ExpectedImprovement,
                                                    const double gaussian_process_upper_confidence_bound_hyperparam = 1.0);
    } // namespace acquisition_func
} // namespace sequential_line_search

#endif
