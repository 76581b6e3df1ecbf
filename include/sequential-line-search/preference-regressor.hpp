// Preference-GP regression [Chu & Ghahramani 2005; Brochu+ 2007]
// (reference surface: include/sequential-line-search/preference-regressor.hpp:20-88).
#ifndef SEQUENTIAL_LINE_SEARCH_PREFERENCE_REGRESSOR_HPP
#define SEQUENTIAL_LINE_SEARCH_PREFERENCE_REGRESSOR_HPP

#include <memory>
#include <sequential-line-search/eigen-lite.hpp>
#include <sequential-line-search/preference.hpp>
#include <sequential-line-search/regressor.hpp>
#include <string>
#include <vector>

namespace sequential_line_search
{
    namespace device
    {
        struct GpHandle;
    }

#ifndef SLS_HAVE_REAL_EIGEN
    /// Stand-in for Eigen::LLT<MatrixXd>: holds the lower factor; solve() runs the device block substitution.
    class CholeskyFactor
    {
    public:
        CholeskyFactor() {}
        explicit CholeskyFactor(const Eigen::MatrixXd& lower) : m_L(lower) {}
        const Eigen::MatrixXd& matrixL() const { return m_L; }
        Eigen::VectorXd        solve(const Eigen::VectorXd& b) const;

    private:
        Eigen::MatrixXd m_L;
    };
    using LltType = CholeskyFactor;
#else
    using LltType = Eigen::LLT<Eigen::MatrixXd>;
#endif

    class PreferenceRegressor : public Regressor
    {
    public:
        PreferenceRegressor(const Eigen::MatrixXd& X, const std::vector<Preference>& D, const bool use_map_hyperparams = false,
                            const double default_kernel_signal_var = 0.500, const double default_kernel_length_scale = 0.500,
                            const double default_noise_level = 0.005, const double kernel_hyperparams_prior_var = 0.250,
                            const double btl_scale = 0.010, const unsigned num_map_estimation_iters = 100,
                            const KernelType kernel_type = KernelType::ArdMatern52Kernel);

        double PredictMu(const Eigen::VectorXd& x) const override;
        double PredictSigma(const Eigen::VectorXd& x) const override;

        Eigen::VectorXd PredictMuDerivative(const Eigen::VectorXd& x) const override;
        Eigen::VectorXd PredictSigmaDerivative(const Eigen::VectorXd& x) const override;

        const bool m_use_map_hyperparams;

        /// The observed data point with the largest estimated goodness value.
        Eigen::VectorXd FindArgMax() const;

        // Data
        Eigen::MatrixXd         m_X;
        std::vector<Preference> m_D;

        double          m_noise_hyperparam;
        Eigen::VectorXd m_kernel_hyperparams;

        /// K = K_f + b I at the final hyper-parameters, and its Cholesky factor.
        Eigen::MatrixXd m_K;
        LltType         m_K_llt;

        /// Writes <prefix>X.csv and <prefix>D.csv into dir_path.
        void DampData(const std::string& dir_path, const std::string& prefix = "") const;

        const Eigen::MatrixXd& GetLargeX() const override { return m_X; }
        const Eigen::VectorXd& GetSmallY() const override { return m_y; }

        const Eigen::VectorXd& GetKernelHyperparams() const override { return m_kernel_hyperparams; }
        double                 GetNoiseHyperparam() const override { return m_noise_hyperparam; }

        const double m_default_kernel_signal_var;
        const double m_default_kernel_length_scale;
        const double m_default_noise_level;
        const double m_kernel_hyperparams_prior_var;
        const double m_btl_scale;

        sls_gp* GetDeviceHandle() const override;

        /// Value of the MAP objective at the solution (diagnostics / tests).
        double GetMapObjectiveValue() const { return m_map_objective; }

    private:
        Eigen::VectorXd m_y;
        double          m_map_objective = 0.0;

        void PerformMapEstimation(const unsigned num_iters);

        std::shared_ptr<device::GpHandle> m_handle;
    };
} // namespace sequential_line_search

#endif
