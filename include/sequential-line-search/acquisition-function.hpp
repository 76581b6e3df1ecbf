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

    /// How FindNextPoint / FindNextPoints search [0,1]^D (reference: the two branches of FindGlobalSolution,
    /// src/acquisition-function.cpp:112-167, chosen there at COMPILE time by the CMake option
    /// SEQUENTIAL_LINE_SEARCH_USE_PARALLELIZED_MULTI_START_SEARCH, default OFF):
    ///   DirectThenLbfgs      DIRECT with num_global_search_iters evaluations, then one L-BFGS of num_local_search_iters
    ///                        evaluations from its result (:155-165, the reference's default build);
    ///   ParallelMultiStart   num_global_search_iters random starts, one L-BFGS each, best end point (:121-153).
    /// Here the choice is a run-time setting.  Its initial value is ParallelMultiStart if this library was compiled with
    /// -DSEQUENTIAL_LINE_SEARCH_USE_PARALLELIZED_MULTI_START_SEARCH, else DirectThenLbfgs; the environment variable
    /// SLS_GLOBAL_SEARCH=direct|multistart overrides that, SetGlobalSearchStrategy overrides both.
    enum class GlobalSearchStrategy
    {
        DirectThenLbfgs,
        ParallelMultiStart,
    };

    namespace acquisition_func
    {
        void                 SetGlobalSearchStrategy(GlobalSearchStrategy strategy);
        GlobalSearchStrategy GetGlobalSearchStrategy();

        /// Relative stopping tolerances of the local (L-BFGS) searches.  Every search of the reference goes through
        /// nloptutil::solve, whose defaults are relative_func_tolerance = relative_param_tolerance = 1e-6 (NLopt's ftol_rel /
        /// xtol_rel): a local search ends with the first accepted step that changes the value, or every
        /// coordinate, by less than that fraction -- long before the evaluation cap on most of C3's acquisition landscapes (the
        /// cap only polishes the 7th to 10th digit).  Initial values 1e-6 / 1e-6; SLS_LOCAL_SEARCH_TOL=<v> sets both (0 = off: run
        /// to the cap, the behaviour before round 5's last revision).  The MAP fits keep running to convergence or their caps
        /// (their optima are this layer's parity anchors; SLS_MAP_FIT_TOL opts in, see host/device.hpp).
        void SetLocalSearchTolerances(double relative_func_tolerance, double relative_param_tolerance);
        void GetLocalSearchTolerances(double* relative_func_tolerance, double* relative_param_tolerance);
        /// The same pair for the two MAP fits (GaussianProcessRegressor::PerformMapEstimation, PreferenceRegressor's).  Initial
        /// values 0 / 0 (off) -- a DEVIATION from the reference, whose fits run under nloptutil::solve's 1e-6 defaults like every
        /// other search: pass (1e-6, 1e-6) for the reference's evaluation counts (INTEGRATION.md 2 says why off is the default).
        void SetMapFitTolerances(double relative_func_tolerance, double relative_param_tolerance);
        void GetMapFitTolerances(double* relative_func_tolerance, double* relative_param_tolerance);

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

        /// Maximiser of the acquisition function over [0,1]^D by the current GlobalSearchStrategy.  DirectThenLbfgs: every
        /// DIRECT iteration's sample points are one batched device evaluation.  ParallelMultiStart: all starts advance in
        /// lock step on the GPU (sharded over device::Devices() when several are configured).
        Eigen::VectorXd FindNextPoint(const Regressor& regressor, const unsigned num_global_search_iters = 100,
                                      const unsigned            num_local_search_iters = 50,
                                      const AcquisitionFuncType func_type              = AcquisitionFuncType::ExpectedImprovement,
                                      const double              gaussian_process_upper_confidence_bound_hyperparam = 1.0);

        /// The DirectThenLbfgs branch, whatever the current strategy (deterministic: DIRECT draws no random numbers).
        Eigen::VectorXd FindNextPointDirect(const Regressor& regressor, const unsigned num_global_search_iters,
                                            const unsigned num_local_search_iters, const AcquisitionFuncType func_type,
                                            const double gaussian_process_upper_confidence_bound_hyperparam, double* value = nullptr);

        /// The ParallelMultiStart branch from an explicit start set (D x S), for reproducible runs and multi-GPU sharding;
        /// returns also the value.
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
