// Regressor base + Gram builders over the C ABI (reference: src/regressor.cpp).
#include <cmath>
#include <sequential-line-search/regressor.hpp>
#include <stdexcept>

#include "device.hpp"

using Eigen::MatrixXd;
using Eigen::VectorXd;

namespace sequential_line_search
{
    namespace
    {
        double ScaledSquaredDistance(const VectorXd& xa, const VectorXd& xb, const VectorXd& theta)
        {
            double q = 0.0;
            for (long i = 0; i < xa.size(); ++i)
            {
                const double d = (xa(i) - xb(i)) / theta(1 + i);
                q += d * d;
            }
            return q;
        }
        int KernelId(KernelType t) { return t == KernelType::ArdSquaredExponentialKernel ? SLS_KERNEL_ARD_SQUARED_EXPONENTIAL : SLS_KERNEL_ARD_MATERN52; }
    } // namespace

    // mathtoolbox kernel functions; scalar host forms for API compatibility only
    namespace kernels
    {
        double ArdSquaredExp(const VectorXd& xa, const VectorXd& xb, const VectorXd& theta)
        {
            return theta(0) * std::exp(-0.5 * ScaledSquaredDistance(xa, xb, theta));
        }
        VectorXd ArdSquaredExpThetaDerivative(const VectorXd& xa, const VectorXd& xb, const VectorXd& theta)
        {
            const double e = std::exp(-0.5 * ScaledSquaredDistance(xa, xb, theta));
            VectorXd     g(theta.size());
            g(0) = e;
            for (long i = 0; i < xa.size(); ++i)
            {
                const double d = xa(i) - xb(i), l = theta(1 + i);
                g(1 + i)       = theta(0) * e * d * d / (l * l * l);
            }
            return g;
        }
        VectorXd ArdSquaredExpFirstArgDerivative(const VectorXd& xa, const VectorXd& xb, const VectorXd& theta)
        {
            const double k = ArdSquaredExp(xa, xb, theta);
            VectorXd     g(xa.size());
            for (long i = 0; i < xa.size(); ++i) g(i) = -k * (xa(i) - xb(i)) / (theta(1 + i) * theta(1 + i));
            return g;
        }
        double ArdMatern52(const VectorXd& xa, const VectorXd& xb, const VectorXd& theta)
        {
            const double q = ScaledSquaredDistance(xa, xb, theta), s = std::sqrt(5.0 * q);
            return theta(0) * (1.0 + s + (5.0 / 3.0) * q) * std::exp(-s);
        }
        VectorXd ArdMatern52ThetaDerivative(const VectorXd& xa, const VectorXd& xb, const VectorXd& theta)
        {
            const double q = ScaledSquaredDistance(xa, xb, theta), s = std::sqrt(5.0 * q), e = std::exp(-s);
            const double c = theta(0) * (5.0 / 3.0) * (1.0 + s) * e;
            VectorXd     g(theta.size());
            g(0) = (1.0 + s + (5.0 / 3.0) * q) * e;
            for (long i = 0; i < xa.size(); ++i)
            {
                const double d = xa(i) - xb(i), l = theta(1 + i);
                g(1 + i)       = c * d * d / (l * l * l);
            }
            return g;
        }
        VectorXd ArdMatern52FirstArgDerivative(const VectorXd& xa, const VectorXd& xb, const VectorXd& theta)
        {
            const double q = ScaledSquaredDistance(xa, xb, theta), s = std::sqrt(5.0 * q);
            const double c = theta(0) * (5.0 / 3.0) * (1.0 + s) * std::exp(-s);
            VectorXd     g(xa.size());
            for (long i = 0; i < xa.size(); ++i) g(i) = -c * (xa(i) - xb(i)) / (theta(1 + i) * theta(1 + i));
            return g;
        }
        KernelType TypeOf(Kernel k)
        {
            if (k == ArdSquaredExp) return KernelType::ArdSquaredExponentialKernel;
            if (k == ArdMatern52) return KernelType::ArdMatern52Kernel;
            throw std::invalid_argument("sequential_line_search: unknown kernel function pointer (only the built-in ARD "
                                        "squared-exponential / Matern-5/2 kernels run on the device; no host fallback)");
        }
        KernelType TypeOf(KernelThetaDerivative k, int)
        {
            if (k == ArdSquaredExpThetaDerivative || k == ArdSquaredExpFirstArgDerivative) return KernelType::ArdSquaredExponentialKernel;
            if (k == ArdMatern52ThetaDerivative || k == ArdMatern52FirstArgDerivative) return KernelType::ArdMatern52Kernel;
            throw std::invalid_argument("sequential_line_search: unknown kernel derivative function pointer");
        }
    } // namespace kernels

    // reference: src/regressor.cpp:8-27
    Regressor::Regressor(const KernelType kernel_type) : m_kernel_type(kernel_type)
    {
        if (kernel_type == KernelType::ArdSquaredExponentialKernel)
        {
            m_kernel                      = kernels::ArdSquaredExp;
            m_kernel_theta_derivative     = kernels::ArdSquaredExpThetaDerivative;
            m_kernel_first_arg_derivative = kernels::ArdSquaredExpFirstArgDerivative;
        }
        else
        {
            m_kernel                      = kernels::ArdMatern52;
            m_kernel_theta_derivative     = kernels::ArdMatern52ThetaDerivative;
            m_kernel_first_arg_derivative = kernels::ArdMatern52FirstArgDerivative;
        }
    }

    // reference: src/regressor.cpp:29-43
    VectorXd Regressor::PredictMaximumPointFromData() const
    {
        const MatrixXd& X = GetLargeX();
        if (sls_gp* h = GetDeviceHandle())
        {
            int best = 0;
            device::Check(sls_gp_get_summary(h, &best, nullptr, nullptr), "sls_gp_get_summary");
            return eig::Col(X, best);
        }
        int    best = 0;
        double bv   = -INFINITY;
        for (long i = 0; i < X.cols(); ++i)
        {
            const double f = PredictMu(eig::Col(X, i));
            if (f > bv) { bv = f; best = static_cast<int>(i); }
        }
        return eig::Col(X, best);
    }

    void Regressor::PredictBatch(const MatrixXd& Xs, VectorXd& mu, VectorXd& sigma) const
    {
        const long M = Xs.cols();
        mu           = VectorXd(M);
        sigma        = VectorXd(M);
        if (sls_gp* h = GetDeviceHandle())
        {
            device::Check(sls_gp_predict(h, Xs.data(), static_cast<int>(M), mu.data(), sigma.data()), "sls_gp_predict");
            return;
        }
        for (long m = 0; m < M; ++m)
        {
            mu(m)    = PredictMu(eig::Col(Xs, m));
            sigma(m) = PredictSigma(eig::Col(Xs, m));
        }
    }

    // reference: src/regressor.cpp:45-59
    VectorXd CalcSmallK(const VectorXd& x, const MatrixXd& X, const VectorXd& theta, const Kernel kernel)
    {
        VectorXd k(X.cols());
        device::Check(sls_gram_cross(device::Context(), X.data(), static_cast<int>(X.rows()), static_cast<int>(X.cols()), x.data(), 1,
                                     theta.data(), KernelId(kernels::TypeOf(kernel)), k.data()),
                      "sls_gram_cross");
        return k;
    }

    // reference: src/regressor.cpp:61-71
    MatrixXd CalcLargeKY(const MatrixXd& X, const VectorXd& theta, const double noise_level, const Kernel kernel)
    {
        MatrixXd K(X.cols(), X.cols());
        device::Check(sls_gram(device::Context(), X.data(), static_cast<int>(X.rows()), static_cast<int>(X.cols()), theta.data(),
                               noise_level, KernelId(kernels::TypeOf(kernel)), K.data()),
                      "sls_gram");
        return K;
    }

    // reference: src/regressor.cpp:73-89
    MatrixXd CalcLargeKF(const MatrixXd& X, const VectorXd& theta, const Kernel kernel) { return CalcLargeKY(X, theta, 0.0, kernel); }

    // reference: src/regressor.cpp:91-108.  dk_i/dx = -c_i (x - x_i) / l^2 with the weights c from the device cross-Gram
    // (c = k for the squared-exponential kernel; for Matern the scalar form is evaluated per column).
    MatrixXd CalcSmallKSmallXDerivative(const VectorXd& x, const MatrixXd& X, const VectorXd& theta,
                                        const KernelFirstArgDerivative kernel_first_arg_derivative)
    {
        const KernelType t = kernels::TypeOf(kernel_first_arg_derivative, 0);
        MatrixXd         J(X.rows(), X.cols());
        if (t == KernelType::ArdSquaredExponentialKernel)
        {
            const VectorXd k = CalcSmallK(x, X, theta, kernels::ArdSquaredExp);
            for (long i = 0; i < X.cols(); ++i)
                for (long d = 0; d < X.rows(); ++d) J(d, i) = -k(i) * (x(d) - X(d, i)) / (theta(1 + d) * theta(1 + d));
        }
        else
        {
            for (long i = 0; i < X.cols(); ++i) eig::SetCol(J, i, kernel_first_arg_derivative(x, eig::Col(X, i), theta));
        }
        return J;
    }

    // reference: src/regressor.cpp:110-134 (API compatibility; the library's own MAP code never builds this tensor)
    std::vector<MatrixXd> CalcLargeKYThetaDerivative(const MatrixXd& X, const VectorXd& theta, const KernelThetaDerivative dk)
    {
        kernels::TypeOf(dk, 0);
        const long            N = X.cols();
        std::vector<MatrixXd> tensor(theta.size(), MatrixXd(N, N));
        for (long i = 0; i < N; ++i)
            for (long j = i; j < N; ++j)
            {
                const VectorXd g = dk(eig::Col(X, i), eig::Col(X, j), theta);
                for (long p = 0; p < theta.size(); ++p) tensor[p](i, j) = tensor[p](j, i) = g(p);
            }
        return tensor;
    }

    // reference: src/regressor.cpp:136-141
    MatrixXd CalcLargeKYNoiseLevelDerivative(const MatrixXd& X, const VectorXd&, const double) { return MatrixXd::Identity(X.cols(), X.cols()); }
} // namespace sequential_line_search
