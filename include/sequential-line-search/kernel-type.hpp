// Kernel selection of the regressors (public surface of the reference: include/sequential-line-search/kernel-type.hpp:8-20).
#ifndef SEQUENTIAL_LINE_SEARCH_KERNEL_TYPE_HPP
#define SEQUENTIAL_LINE_SEARCH_KERNEL_TYPE_HPP

#include <sequential-line-search/eigen-lite.hpp>

namespace sequential_line_search
{
    enum class KernelType
    {
        ArdSquaredExponentialKernel,
        ArdMatern52Kernel,
    };

    // Host-callable scalar forms (theta = (a, l_1..l_D)); kept for source compatibility with code that takes the
    // function pointers.  The library itself never loops over them: the Gram / cross-covariance builders recognise
    // these pointers and run the gfx950 kernels instead.
    using Kernel                   = double (*)(const Eigen::VectorXd&, const Eigen::VectorXd&, const Eigen::VectorXd&);
    using KernelThetaDerivative    = Eigen::VectorXd (*)(const Eigen::VectorXd&, const Eigen::VectorXd&, const Eigen::VectorXd&);
    using KernelFirstArgDerivative = Eigen::VectorXd (*)(const Eigen::VectorXd&, const Eigen::VectorXd&, const Eigen::VectorXd&);

    namespace kernels
    {
        double          ArdSquaredExp(const Eigen::VectorXd& xa, const Eigen::VectorXd& xb, const Eigen::VectorXd& theta);
        Eigen::VectorXd ArdSquaredExpThetaDerivative(const Eigen::VectorXd& xa, const Eigen::VectorXd& xb, const Eigen::VectorXd& theta);
        Eigen::VectorXd ArdSquaredExpFirstArgDerivative(const Eigen::VectorXd& xa, const Eigen::VectorXd& xb, const Eigen::VectorXd& theta);
        double          ArdMatern52(const Eigen::VectorXd& xa, const Eigen::VectorXd& xb, const Eigen::VectorXd& theta);
        Eigen::VectorXd ArdMatern52ThetaDerivative(const Eigen::VectorXd& xa, const Eigen::VectorXd& xb, const Eigen::VectorXd& theta);
        Eigen::VectorXd ArdMatern52FirstArgDerivative(const Eigen::VectorXd& xa, const Eigen::VectorXd& xb, const Eigen::VectorXd& theta);

        /// Kernel type behind one of the pointers above; throws std::invalid_argument for a foreign function (the device
        /// path has no generic-callback mode and there is deliberately no host fallback).
        KernelType TypeOf(Kernel k);
        KernelType TypeOf(KernelThetaDerivative k, int);
    } // namespace kernels
} // namespace sequential_line_search

#endif
