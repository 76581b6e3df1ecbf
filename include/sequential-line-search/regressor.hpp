// Abstract regressor + Gram-matrix builders (reference surface: include/sequential-line-search/regressor.hpp:10-72).
// Every builder below runs on the MI355X through the C ABI (include/sls_hip.h).
#ifndef SEQUENTIAL_LINE_SEARCH_REGRESSOR_HPP
#define SEQUENTIAL_LINE_SEARCH_REGRESSOR_HPP

#include <sequential-line-search/eigen-lite.hpp>
#include <sequential-line-search/kernel-type.hpp>
#include <vector>

struct sls_gp;

namespace sequential_line_search
{
    class Regressor
    {
    public:
        Regressor(const KernelType kernel_type);
        virtual ~Regressor() {}

        unsigned GetNumDims() const { return GetLargeX().rows(); }

        virtual double PredictMu(const Eigen::VectorXd& x) const    = 0;
        virtual double PredictSigma(const Eigen::VectorXd& x) const = 0;

        virtual Eigen::VectorXd PredictMuDerivative(const Eigen::VectorXd& x) const    = 0;
        virtual Eigen::VectorXd PredictSigmaDerivative(const Eigen::VectorXd& x) const = 0;

        virtual const Eigen::VectorXd& GetKernelHyperparams() const = 0;
        virtual double                 GetNoiseHyperparam() const   = 0;

        virtual const Eigen::MatrixXd& GetLargeX() const = 0;
        virtual const Eigen::VectorXd& GetSmallY() const = 0;

        /// argmax_i PredictMu(x_i) over the data points.  For the built-in regressors this is hoisted to fit time
        /// (mu(x_i) = y_i - b alpha_i); a user subclass without a device handle gets the reference's N x PredictMu loop.
        Eigen::VectorXd PredictMaximumPointFromData() const;

        Kernel                   GetKernel() const { return m_kernel; }
        KernelThetaDerivative    GetKernelThetaDerivative() const { return m_kernel_theta_derivative; }
        KernelFirstArgDerivative GetKernelFirstArgDerivative() const { return m_kernel_first_arg_derivative; }
        KernelType               GetKernelType() const { return m_kernel_type; }

        /// Device-resident state (K_y^-1, alpha, ...) or nullptr; used by acquisition_func for batched evaluation.
        virtual sls_gp* GetDeviceHandle() const { return nullptr; }

        // Batched forms (not in the reference; D x M input, one query point per column).
        void PredictBatch(const Eigen::MatrixXd& Xs, Eigen::VectorXd& mu, Eigen::VectorXd& sigma) const;

    protected:
        KernelType               m_kernel_type;
        Kernel                   m_kernel;
        KernelThetaDerivative    m_kernel_theta_derivative;
        KernelFirstArgDerivative m_kernel_first_arg_derivative;
    };

    // k
    Eigen::VectorXd CalcSmallK(const Eigen::VectorXd& x, const Eigen::MatrixXd& X, const Eigen::VectorXd& kernel_hyperparameters,
                               const Kernel kernel);
    // K_y = K_f + sigma^2 I
    Eigen::MatrixXd CalcLargeKY(const Eigen::MatrixXd& X, const Eigen::VectorXd& kernel_hyperparameters, const double noise_level,
                                const Kernel kernel);
    // K_f
    Eigen::MatrixXd CalcLargeKF(const Eigen::MatrixXd& X, const Eigen::VectorXd& kernel_hyperparameters, const Kernel kernel);
    // partial k / partial x  (D x N).  Rarely needed: the predictors contract it on the device without forming it.
    Eigen::MatrixXd CalcSmallKSmallXDerivative(const Eigen::VectorXd& x, const Eigen::MatrixXd& X,
                                               const Eigen::VectorXd&         kernel_hyperparameters,
                                               const KernelFirstArgDerivative kernel_first_arg_derivative);
    // partial K_y / partial theta: (D+1) dense N x N matrices.  Kept for API compatibility only (O(D N^2) memory);
    // the MAP objectives use the fused contraction of sls_nll_eval instead.
    std::vector<Eigen::MatrixXd> CalcLargeKYThetaDerivative(const Eigen::MatrixXd& X, const Eigen::VectorXd& kernel_hyperparameters,
                                                            const KernelThetaDerivative kernel_theta_derivative);
    // partial K_y / partial sigma^2 = I
    Eigen::MatrixXd CalcLargeKYNoiseLevelDerivative(const Eigen::MatrixXd& X, const Eigen::VectorXd& kernel_hyperparameters,
                                                    const double noise_level);
} // namespace sequential_line_search

#endif
