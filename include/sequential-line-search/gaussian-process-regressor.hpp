// GP regression on (X, y) (reference surface: include/sequential-line-search/gaussian-process-regressor.hpp:13-58).
#ifndef SEQUENTIAL_LINE_SEARCH_GAUSSIAN_PROCESS_REGRESSOR_HPP
#define SEQUENTIAL_LINE_SEARCH_GAUSSIAN_PROCESS_REGRESSOR_HPP

#include <atomic>
#include <memory>
#include <sequential-line-search/eigen-lite.hpp>
#include <sequential-line-search/regressor.hpp>

namespace sequential_line_search
{
    namespace device
    {
        struct GpHandle;
    }

    class GaussianProcessRegressor : public Regressor
    {
    public:
        /// Hyper-parameters (a, b, r) come from MAP estimation of the log marginal likelihood with log-normal priors.
        GaussianProcessRegressor(const Eigen::MatrixXd& X, const Eigen::VectorXd& y,
                                 const KernelType kernel_type = KernelType::ArdMatern52Kernel);

        /// Specified hyper-parameters are used as they are.  `materialize_matrices = false` (extension) skips the
        /// 2 x N^2 device->host copies of m_K_y / m_K_y_inv for THIS object (large N, or internal helper regressors).
        GaussianProcessRegressor(const Eigen::MatrixXd& X, const Eigen::VectorXd& y, const Eigen::VectorXd& kernel_hyperparams,
                                 double noise_hyperparam, const KernelType kernel_type = KernelType::ArdMatern52Kernel,
                                 bool materialize_matrices = true);

        double PredictMu(const Eigen::VectorXd& x) const override;
        double PredictSigma(const Eigen::VectorXd& x) const override;

        Eigen::VectorXd PredictMuDerivative(const Eigen::VectorXd& x) const override;
        Eigen::VectorXd PredictSigmaDerivative(const Eigen::VectorXd& x) const override;

        // Available after construction (copied back from the device unless materialisation is switched off)
        Eigen::MatrixXd m_K_y;
        Eigen::MatrixXd m_K_y_inv;

        /// Process-wide default for objects constructed afterwards (atomic; read once per construction): set false to skip
        /// the 2 x N^2 device->host copies of m_K_y / m_K_y_inv for large N.  Library code never writes it.
        static std::atomic<bool> s_materialize_matrices;

        const Eigen::MatrixXd& GetLargeX() const override { return m_X; }
        const Eigen::VectorXd& GetSmallY() const override { return m_y; }

        const Eigen::VectorXd& GetKernelHyperparams() const override { return m_kernel_hyperparams; }
        double                 GetNoiseHyperparam() const override { return m_noise_hyperparam; }

        sls_gp* GetDeviceHandle() const override;

        /// Extension: what the MAP fit of the first constructor did (all zero for the second constructor).
        struct MapFitStats
        {
            double final_value  = 0.0;   ///< log posterior at the returned (a, b, r)
            double direct_value = 0.0;   ///< best value of the DIRECT phase
            double prior_value  = 0.0;   ///< value at the prior medians (the reference's x_ini)
            int    evals_direct = 0;
            int    evals_local  = 0;
            double seconds      = 0.0;
        };
        const MapFitStats& GetMapFitStats() const { return m_map_stats; }

        /// Extension: add one observation without refitting (O(N^2) update on the device; hyper-parameters unchanged).
        /// m_K_y / m_K_y_inv are refreshed only if this object materialises them.
        void AppendPoint(const Eigen::VectorXd& x, double y);

    private:
        void PerformMapEstimation();
        void BuildDeviceState();

        Eigen::MatrixXd m_X;
        Eigen::VectorXd m_y;
        Eigen::VectorXd m_kernel_hyperparams;
        double          m_noise_hyperparam;
        bool            m_materialize = true;
        MapFitStats     m_map_stats;

        std::shared_ptr<device::GpHandle> m_handle;
    };
} // namespace sequential_line_search

#endif
