// PreferentialBayesianOptimizer facade (reference: src/preferential-bayesian-optimizer.cpp) -- the second caller of the
// device path (PreferenceRegressor MAP + FindNextPoints).
#include <sequential-line-search/preference-data-manager.hpp>
#include <sequential-line-search/preference-regressor.hpp>
#include <sequential-line-search/preferential-bayesian-optimizer.hpp>
#include <sequential-line-search/utils.hpp>
#include <stdexcept>

using Eigen::VectorXd;

namespace sequential_line_search
{
    // reference: src/preferential-bayesian-optimizer.cpp:11-20
    std::vector<VectorXd> GenerateRandomPoints(const int num_dims, const int num_options)
    {
        std::vector<VectorXd> options;
        for (int i = 0; i < num_options; ++i) options.push_back(utils::GenerateRandomVector(num_dims));
        return options;
    }

    // reference: :22-47
    PreferentialBayesianOptimizer::PreferentialBayesianOptimizer(const int num_dims, const bool use_map_hyperparams,
                                                                 const KernelType kernel_type, const AcquisitionFuncType acquisition_func_type,
                                                                 const InitialQueryGenerator&       initial_query_generator,
                                                                 const CurrentBestSelectionStrategy current_best_selection_strategy,
                                                                 const int                          num_options)
        : m_use_map_hyperparams(use_map_hyperparams),
          m_num_options(num_options),
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
        if (num_options < 2) throw std::invalid_argument("PreferentialBayesianOptimizer: num_options must be >= 2");
        m_data            = std::make_shared<PreferenceDataManager>();
        m_current_options = initial_query_generator(num_dims, num_options);
        if (static_cast<int>(m_current_options.size()) != num_options)
            throw std::invalid_argument("PreferentialBayesianOptimizer: the initial query generator returned a wrong number of options");
    }

    void PreferentialBayesianOptimizer::SetHyperparams(const double kernel_signal_var, const double kernel_length_scale,
                                                       const double noise_level, const double kernel_hyperparams_prior_var,
                                                       const double btl_scale)
    {
        m_kernel_signal_var            = kernel_signal_var;
        m_kernel_length_scale          = kernel_length_scale;
        m_noise_level                  = noise_level;
        m_kernel_hyperparams_prior_var = kernel_hyperparams_prior_var;
        m_btl_scale                    = btl_scale;
    }

    // reference: :64-81
    void PreferentialBayesianOptimizer::SubmitFeedbackData(const int option_index, const int num_map_estimation_iters)
    {
        if (option_index < 0 || option_index >= static_cast<int>(m_current_options.size()))
            throw std::out_of_range("PreferentialBayesianOptimizer::SubmitFeedbackData: option_index");
        std::vector<VectorXd> others;
        for (int i = 0; i < static_cast<int>(m_current_options.size()); ++i)
            if (i != option_index) others.push_back(m_current_options[i]);
        SubmitCustomFeedbackData(m_current_options[option_index], others, num_map_estimation_iters);
    }

    // reference: :83-93
    void PreferentialBayesianOptimizer::SubmitCustomFeedbackData(const VectorXd& chosen_option, const std::vector<VectorXd>& other_options,
                                                                 const int num_map_estimation_iters)
    {
        m_data->AddNewPoints(chosen_option, other_options, true);
        PerformMapEstimation(num_map_estimation_iters);
    }

    // reference: :95-141 (heuristic of the parallel multi-start branch: 500 D starts, 10 D local evaluations)
    void PreferentialBayesianOptimizer::DetermineNextQuery(const int num_global_search_iters, const int num_local_search_iters)
    {
        if (!m_regressor) throw std::logic_error("PreferentialBayesianOptimizer::DetermineNextQuery called before any feedback");
        const int num_dims = static_cast<int>(GetMaximizer().size());
        const int n_global = num_global_search_iters > 0 ? num_global_search_iters : 500 * num_dims;
        const int n_local  = num_local_search_iters > 0 ? num_local_search_iters : 10 * num_dims;

        const VectorXd x_plus = (m_current_best_selection_strategy == CurrentBestSelectionStrategy::LargestExpectValue)
                                    ? m_regressor->FindArgMax()
                                    : m_data->GetLastSelectedDataPoint();
        const std::vector<VectorXd> next = acquisition_func::FindNextPoints(*m_regressor, m_num_options - 1, n_global, n_local,
                                                                            m_acquisition_func_type,
                                                                            m_gaussian_process_upper_confidence_bound_hyperparam);
        m_current_options[0] = x_plus;
        for (int i = 1; i < m_num_options; ++i) m_current_options[i] = next[i - 1];
    }

    VectorXd PreferentialBayesianOptimizer::GetMaximizer() const { return m_current_options[0]; }

    double PreferentialBayesianOptimizer::GetPreferenceValueMean(const VectorXd& point) const
    {
        return m_regressor ? m_regressor->PredictMu(point) : 0.0;
    }
    double PreferentialBayesianOptimizer::GetPreferenceValueStdev(const VectorXd& point) const
    {
        return m_regressor ? m_regressor->PredictSigma(point) : 0.0;
    }
    double PreferentialBayesianOptimizer::GetAcquisitionFuncValue(const VectorXd& point) const
    {
        return m_regressor ? acquisition_func::CalcAcquisitionValue(*m_regressor, point, m_acquisition_func_type,
                                                                    m_gaussian_process_upper_confidence_bound_hyperparam)
                           : 0.0;
    }

    Eigen::VectorXd PreferentialBayesianOptimizer::GetPreferenceValueMeans(const Eigen::MatrixXd& points) const
    {
        Eigen::VectorXd mu = Eigen::VectorXd::Zero(points.cols()), sigma;
        if (m_regressor) m_regressor->PredictBatch(points, mu, sigma);
        return mu;
    }
    Eigen::VectorXd PreferentialBayesianOptimizer::GetPreferenceValueStdevs(const Eigen::MatrixXd& points) const
    {
        Eigen::VectorXd mu, sigma = Eigen::VectorXd::Zero(points.cols());
        if (m_regressor) m_regressor->PredictBatch(points, mu, sigma);
        return sigma;
    }
    Eigen::VectorXd PreferentialBayesianOptimizer::GetAcquisitionFuncValues(const Eigen::MatrixXd& points) const
    {
        if (!m_regressor) return Eigen::VectorXd::Zero(points.cols());
        return acquisition_func::CalcAcquisitionValues(*m_regressor, points, m_acquisition_func_type,
                                                       m_gaussian_process_upper_confidence_bound_hyperparam);
    }

    const Eigen::MatrixXd& PreferentialBayesianOptimizer::GetRawDataPoints() const { return m_data->GetX(); }

    void PreferentialBayesianOptimizer::DampData(const std::string& directory_path) const
    {
        if (m_regressor) m_regressor->DampData(directory_path);
    }

    // reference: :184-end (heuristic budget 10 (D + number of points))
    void PreferentialBayesianOptimizer::PerformMapEstimation(const int num_map_estimation_iters)
    {
        const int iters = num_map_estimation_iters > 0
                              ? num_map_estimation_iters
                              : 10 * (static_cast<int>(GetMaximizer().size()) + m_data->GetNumDataPoints());
        m_regressor = std::make_shared<PreferenceRegressor>(m_data->GetX(), m_data->GetD(), m_use_map_hyperparams, m_kernel_signal_var,
                                                            m_kernel_length_scale, m_noise_level, m_kernel_hyperparams_prior_var,
                                                            m_btl_scale, iters, m_kernel_type);
    }
} // namespace sequential_line_search
