// SequentialLineSearchOptimizer facade (reference: src/sequential-line-search.cpp).
#include <sequential-line-search/acquisition-function.hpp>
#include <sequential-line-search/preference-data-manager.hpp>
#include <sequential-line-search/preference-regressor.hpp>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sequential-line-search/sequential-line-search.hpp>
#include <sequential-line-search/slider.hpp>
#include <sequential-line-search/utils.hpp>

using Eigen::VectorXd;

namespace sequential_line_search
{
    // reference: src/sequential-line-search.cpp:11-23
    std::pair<VectorXd, VectorXd> GenerateRandomSliderEnds(const int num_dims)
    {
        const VectorXd a = utils::GenerateRandomVector(num_dims);
        const VectorXd b = utils::GenerateRandomVector(num_dims);
        return {a, b};
    }
    std::pair<VectorXd, VectorXd> GenerateCenteredFixedLengthRandomSliderEnds(const int num_dims)
    {
        const VectorXd c   = VectorXd::Constant(num_dims, 0.50);
        const VectorXd dir = 0.5 * VectorXd::Random(num_dims);
        return {c + dir, c - dir};
    }

    // reference: src/sequential-line-search.cpp:25-50
    SequentialLineSearchOptimizer::SequentialLineSearchOptimizer(
        const int num_dims, const bool use_slider_enlargement, const bool use_map_hyperparams, const KernelType kernel_type,
        const AcquisitionFuncType acquisition_func_type,
        const std::function<std::pair<VectorXd, VectorXd>(const int)>& initial_query_generator,
        const CurrentBestSelectionStrategy                             current_best_selection_strategy)
        : m_use_slider_enlargement(use_slider_enlargement),
          m_use_map_hyperparams(use_map_hyperparams),
          m_current_best_selection_strategy(current_best_selection_strategy),
          m_kernel_signal_var(0.500),
          m_kernel_length_scale(0.500),
          m_noise_level(0.005),
          m_kernel_hyperparams_prior_var(0.250),
          m_btl_scale(0.010),
          m_kernel_type(kernel_type),
          m_acquisition_func_type(acquisition_func_type),
          m_gaussian_process_upper_confidence_bound_hyperparam(1.0)
    {
        const auto ends = initial_query_generator(num_dims);
        m_data          = std::make_shared<PreferenceDataManager>();
        m_regressor     = nullptr;
        m_slider        = std::make_shared<Slider>(ends.first, ends.second, false);
    }

    void SequentialLineSearchOptimizer::SetHyperparams(const double kernel_signal_var, const double kernel_length_scale,
                                                       const double noise_level, const double kernel_hyperparams_prior_var,
                                                       const double btl_scale)
    {
        m_kernel_signal_var            = kernel_signal_var;
        m_kernel_length_scale          = kernel_length_scale;
        m_noise_level                  = noise_level;
        m_kernel_hyperparams_prior_var = kernel_hyperparams_prior_var;
        m_btl_scale                    = btl_scale;
    }

    // reference: src/sequential-line-search.cpp:65-79 -- the effort heuristic follows the maximiser branch in use: 10 starts
    // for the parallel multi-start search, 50 D DIRECT evaluations otherwise; 10 D local evaluations either way
    void SequentialLineSearchOptimizer::SubmitFeedbackData(const double slider_position)
    {
        const int  num_dims                = static_cast<int>(GetMaximizer().size());
        const bool multi_start             = acquisition_func::GetGlobalSearchStrategy() == GlobalSearchStrategy::ParallelMultiStart;
        const int  num_global_search_iters = multi_start ? 10 : 50 * num_dims;
        SubmitFeedbackData(slider_position, 100, num_global_search_iters, 10 * num_dims);
    }

    // reference: src/sequential-line-search.cpp:81-123
    void SequentialLineSearchOptimizer::SubmitFeedbackData(const double slider_position, const int num_map_estimation_iters,
                                                           const int num_global_search_iters, const int num_local_search_iters)
    {
        const VectorXd x_chosen   = CalcPointFromSliderPosition(slider_position);
        const VectorXd x_prev_max = m_slider->original_end_0;
        const VectorXd x_prev_ei  = m_slider->original_end_1;

        m_data->AddNewPoints(x_chosen, {x_prev_max, x_prev_ei}, true);

        static const bool timing = std::getenv("SLS_HOST_TIMING") != nullptr;   // once per process; stderr: ms in the MAP fit / in the maximiser
        const auto t0     = std::chrono::steady_clock::now();
        m_regressor = std::make_shared<PreferenceRegressor>(m_data->GetX(), m_data->GetD(), m_use_map_hyperparams, m_kernel_signal_var,
                                                            m_kernel_length_scale, m_noise_level, m_kernel_hyperparams_prior_var,
                                                            m_btl_scale, num_map_estimation_iters, m_kernel_type);

        const auto     t1     = std::chrono::steady_clock::now();
        const VectorXd x_plus = (m_current_best_selection_strategy == CurrentBestSelectionStrategy::LargestExpectValue)
                                    ? m_regressor->FindArgMax()
                                    : x_chosen;
        const VectorXd x_acquisition =
            acquisition_func::FindNextPoint(*m_regressor, num_global_search_iters, num_local_search_iters, m_acquisition_func_type,
                                            m_gaussian_process_upper_confidence_bound_hyperparam);
        if (timing)
            std::fprintf(stderr, "SubmitFeedbackData: N %d  MAP fit %.2f ms  next point %.2f ms\n", (int)m_data->GetX().cols(),
                         std::chrono::duration<double, std::milli>(t1 - t0).count(),
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());

        m_slider = std::make_shared<Slider>(x_plus, x_acquisition, m_use_slider_enlargement);
    }

    std::pair<VectorXd, VectorXd> SequentialLineSearchOptimizer::GetSliderEnds() const { return {m_slider->end_0, m_slider->end_1}; }

    VectorXd SequentialLineSearchOptimizer::CalcPointFromSliderPosition(const double slider_position) const
    {
        return m_slider->GetValue(slider_position);
    }

    VectorXd SequentialLineSearchOptimizer::GetMaximizer() const { return m_slider->original_end_0; }

    double SequentialLineSearchOptimizer::GetPreferenceValueMean(const VectorXd& point) const
    {
        return m_regressor ? m_regressor->PredictMu(point) : 0.0;
    }
    double SequentialLineSearchOptimizer::GetPreferenceValueStdev(const VectorXd& point) const
    {
        return m_regressor ? m_regressor->PredictSigma(point) : 0.0;
    }
    double SequentialLineSearchOptimizer::GetAcquisitionFuncValue(const VectorXd& point) const
    {
        return m_regressor ? acquisition_func::CalcAcquisitionValue(*m_regressor, point, m_acquisition_func_type,
                                                                    m_gaussian_process_upper_confidence_bound_hyperparam)
                           : 0.0;
    }

    Eigen::VectorXd SequentialLineSearchOptimizer::GetPreferenceValueMeans(const Eigen::MatrixXd& points) const
    {
        Eigen::VectorXd mu = Eigen::VectorXd::Zero(points.cols()), sigma;
        if (m_regressor) m_regressor->PredictBatch(points, mu, sigma);
        return mu;
    }
    Eigen::VectorXd SequentialLineSearchOptimizer::GetPreferenceValueStdevs(const Eigen::MatrixXd& points) const
    {
        Eigen::VectorXd mu, sigma = Eigen::VectorXd::Zero(points.cols());
        if (m_regressor) m_regressor->PredictBatch(points, mu, sigma);
        return sigma;
    }
    Eigen::VectorXd SequentialLineSearchOptimizer::GetAcquisitionFuncValues(const Eigen::MatrixXd& points) const
    {
        if (!m_regressor) return Eigen::VectorXd::Zero(points.cols());
        return acquisition_func::CalcAcquisitionValues(*m_regressor, points, m_acquisition_func_type,
                                                       m_gaussian_process_upper_confidence_bound_hyperparam);
    }

    const Eigen::MatrixXd& SequentialLineSearchOptimizer::GetRawDataPoints() const { return m_data->GetX(); }

    void SequentialLineSearchOptimizer::DampData(const std::string& directory_path) const
    {
        if (m_regressor) m_regressor->DampData(directory_path);
    }
} // namespace sequential_line_search
