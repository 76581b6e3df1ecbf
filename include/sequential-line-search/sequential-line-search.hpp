// Optimizer facade of sequential line search [Koyama+ 2017]
// (reference surface: include/sequential-line-search/sequential-line-search.hpp:18-131).  Search space: [0,1]^D.
#ifndef SEQUENTIAL_LINE_SEARCH_SEQUENTIAL_LINE_SEARCH_HPP
#define SEQUENTIAL_LINE_SEARCH_SEQUENTIAL_LINE_SEARCH_HPP

#include <functional>
#include <memory>
#include <sequential-line-search/acquisition-function.hpp>
#include <sequential-line-search/current-best-selection-strategy.hpp>
#include <sequential-line-search/eigen-lite.hpp>
#include <sequential-line-search/kernel-type.hpp>
#include <string>
#include <utility>

namespace sequential_line_search
{
    class PreferenceRegressor;
    class Slider;
    class PreferenceDataManager;

    std::pair<Eigen::VectorXd, Eigen::VectorXd> GenerateRandomSliderEnds(const int num_dims);
    std::pair<Eigen::VectorXd, Eigen::VectorXd> GenerateCenteredFixedLengthRandomSliderEnds(const int num_dims);

    class SequentialLineSearchOptimizer
    {
    public:
        SequentialLineSearchOptimizer(
            const int num_dims, const bool use_slider_enlargement = true, const bool use_map_hyperparams = true,
            const KernelType          kernel_type           = KernelType::ArdMatern52Kernel,
            const AcquisitionFuncType acquisition_func_type = AcquisitionFuncType::ExpectedImprovement,
            const std::function<std::pair<Eigen::VectorXd, Eigen::VectorXd>(const int)>& initial_query_generator = GenerateRandomSliderEnds,
            const CurrentBestSelectionStrategy current_best_selection_strategy = CurrentBestSelectionStrategy::LargestExpectValue);

        /// With MAP enabled the kernel values are prior medians and initial guesses; otherwise they are used directly.
        void SetHyperparams(const double kernel_signal_var = 0.500, const double kernel_length_scale = 0.500,
                            const double noise_level = 0.005, const double kernel_hyperparams_prior_var = 0.250,
                            const double btl_scale = 0.010);

        /// slider_position in [0,1]: 0 = first end-point, 1 = second.  Effort is set by the reference's heuristic
        /// (100 MAP evaluations, 10 parallel starts, 10 D local evaluations).
        void SubmitFeedbackData(const double slider_position);
        void SubmitFeedbackData(const double slider_position, const int num_map_estimation_iters, const int num_global_search_iters,
                                const int num_local_search_iters);

        std::pair<Eigen::VectorXd, Eigen::VectorXd> GetSliderEnds() const;
        Eigen::VectorXd                             CalcPointFromSliderPosition(const double slider_position) const;
        Eigen::VectorXd                             GetMaximizer() const;

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

        void DampData(const std::string& directory_path) const;

        void SetGaussianProcessUpperConfidenceBoundHyperparam(const double hyperparam)
        {
            m_gaussian_process_upper_confidence_bound_hyperparam = hyperparam;
        }

    private:
        const bool m_use_slider_enlargement;
        const bool m_use_map_hyperparams;

        const CurrentBestSelectionStrategy m_current_best_selection_strategy;

        std::shared_ptr<PreferenceRegressor>   m_regressor;
        std::shared_ptr<Slider>                m_slider;
        std::shared_ptr<PreferenceDataManager> m_data;

        double m_kernel_signal_var;
        double m_kernel_length_scale;
        double m_noise_level;
        double m_kernel_hyperparams_prior_var;
        double m_btl_scale;

        const KernelType          m_kernel_type;
        const AcquisitionFuncType m_acquisition_func_type;

        double m_gaussian_process_upper_confidence_bound_hyperparam;
    };
} // namespace sequential_line_search

#endif
