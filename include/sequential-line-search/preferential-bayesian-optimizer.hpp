// Preferential Bayesian optimisation with discrete-choice queries
// (reference surface: include/sequential-line-search/preferential-bayesian-optimizer.hpp:16-160).  Search space [0,1]^D.
#ifndef SEQUENTIAL_LINE_SEARCH_PREFERENTIAL_BAYESIAN_OPTIMIZER_HPP
#define SEQUENTIAL_LINE_SEARCH_PREFERENTIAL_BAYESIAN_OPTIMIZER_HPP

#include <functional>
#include <memory>
#include <sequential-line-search/acquisition-function.hpp>
#include <sequential-line-search/current-best-selection-strategy.hpp>
#include <sequential-line-search/eigen-lite.hpp>
#include <sequential-line-search/kernel-type.hpp>
#include <string>
#include <vector>

namespace sequential_line_search
{
    class PreferenceRegressor;
    class PreferenceDataManager;

    /// (number of dimensions, number of options per query) -> the first query.
    using InitialQueryGenerator = std::function<std::vector<Eigen::VectorXd>(const int, const int)>;

    std::vector<Eigen::VectorXd> GenerateRandomPoints(const int num_dims, const int num_options);

    class PreferentialBayesianOptimizer
    {
    public:
        PreferentialBayesianOptimizer(const int num_dims, const bool use_map_hyperparams = true,
                                      const KernelType             kernel_type             = KernelType::ArdMatern52Kernel,
                                      const AcquisitionFuncType    acquisition_func_type   = AcquisitionFuncType::ExpectedImprovement,
                                      const InitialQueryGenerator& initial_query_generator = GenerateRandomPoints,
                                      const CurrentBestSelectionStrategy current_best_selection_strategy =
                                          CurrentBestSelectionStrategy::LargestExpectValue,
                                      const int num_options = 2);

        void SetHyperparams(const double kernel_signal_var = 0.500, const double kernel_length_scale = 0.500,
                            const double noise_level = 0.005, const double kernel_hyperparams_prior_var = 0.250,
                            const double btl_scale = 0.010);

        /// option_index: zero-based index into GetCurrentOptions().  num_map_estimation_iters <= 0: heuristic
        /// 10 (D + number of data points).
        void SubmitFeedbackData(const int option_index, const int num_map_estimation_iters = 0);

        /// Same with arbitrary options instead of the ones proposed by the optimizer.
        void SubmitCustomFeedbackData(const Eigen::VectorXd& chosen_option, const std::vector<Eigen::VectorXd>& other_options,
                                      const int num_map_estimation_iters = 0);

        /// Builds the next query: option 0 = current best, the others from FindNextPoints.  Non-positive arguments select
        /// the reference's multi-start heuristic (500 D starts, 10 D local evaluations).
        void DetermineNextQuery(const int num_global_search_iters = 0, const int num_local_search_iters = 0);

        const std::vector<Eigen::VectorXd>& GetCurrentOptions() const { return m_current_options; }
        Eigen::VectorXd                     GetMaximizer() const;

        double GetPreferenceValueMean(const Eigen::VectorXd& point) const;
        double GetPreferenceValueStdev(const Eigen::VectorXd& point) const;
        double GetAcquisitionFuncValue(const Eigen::VectorXd& point) const;

        /// Batched forms of the three accessors above (additions of this build: SURVEY.md 8(f3)): one query point per column of
        /// `points` (D x M), ONE device pass each -- what the reference's GUI demo does pixel by pixel
        /// (demos/bayesian_optimization_2d_gui/mainwidget.cpp:41-72).  Zeros while there is no data, like the scalar forms.
        Eigen::VectorXd GetPreferenceValueMeans(const Eigen::MatrixXd& points) const;
        Eigen::VectorXd GetPreferenceValueStdevs(const Eigen::MatrixXd& points) const;
        Eigen::VectorXd GetAcquisitionFuncValues(const Eigen::MatrixXd& points) const;

        const Eigen::MatrixXd& GetRawDataPoints() const;
        void                   DampData(const std::string& directory_path) const;

        void SetGaussianProcessUpperConfidenceBoundHyperparam(const double hyperparam)
        {
            m_gaussian_process_upper_confidence_bound_hyperparam = hyperparam;
        }

    private:
        const bool m_use_map_hyperparams;
        const int  m_num_options;

        const CurrentBestSelectionStrategy m_current_best_selection_strategy;

        std::shared_ptr<PreferenceRegressor>   m_regressor;
        std::shared_ptr<PreferenceDataManager> m_data;
        std::vector<Eigen::VectorXd>           m_current_options;

        double m_kernel_signal_var;
        double m_kernel_length_scale;
        double m_noise_level;
        double m_kernel_hyperparams_prior_var;
        double m_btl_scale;

        const KernelType          m_kernel_type;
        const AcquisitionFuncType m_acquisition_func_type;

        double m_gaussian_process_upper_confidence_bound_hyperparam;

        void PerformMapEstimation(const int num_map_estimation_iters);
    };
} // namespace sequential_line_search

#endif
