// acquisition_func over the C ABI (reference: src/acquisition-function.cpp).
#include <cmath>
#include <sequential-line-search/acquisition-function.hpp>
#include <sequential-line-search/gaussian-process-regressor.hpp>
#include <sequential-line-search/utils.hpp>
#include <stdexcept>

#include "device.hpp"

using Eigen::MatrixXd;
using Eigen::VectorXd;

namespace sequential_line_search
{
    namespace
    {
        int AcqId(AcquisitionFuncType t)
        {
            return t == AcquisitionFuncType::ExpectedImprovement ? SLS_ACQ_EXPECTED_IMPROVEMENT : SLS_ACQ_GP_UCB;
        }

        // mathtoolbox::GetExpectedImprovement / ...Derivative for regressors WITHOUT a device
        // handle (user subclasses of Regressor): the reference's call structure on top of the virtual Predict* methods.
        double NormalCdf(double u) { return 0.5 * std::erfc(-u / std::sqrt(2.0)); }
        double NormalPdf(double u) { return std::exp(-0.5 * u * u) / std::sqrt(2.0 * M_PI); }

        double GenericValue(const Regressor& r, const VectorXd& x, AcquisitionFuncType type, double h)
        {
            const double mu = r.PredictMu(x), sigma = r.PredictSigma(x);
            if (type == AcquisitionFuncType::GaussianProcessUpperConfidenceBound) return mu + h * sigma;
            const double diff = mu - r.PredictMu(r.PredictMaximumPointFromData());
            const double u    = diff / sigma;
            const double ei   = diff * NormalCdf(u) + sigma * NormalPdf(u);
            return (sigma < 1e-10 || std::isnan(ei)) ? 0.0 : ei;
        }
        VectorXd GenericDerivative(const Regressor& r, const VectorXd& x, AcquisitionFuncType type, double h)
        {
            const VectorXd dm = r.PredictMuDerivative(x), ds = r.PredictSigmaDerivative(x);
            if (type == AcquisitionFuncType::GaussianProcessUpperConfidenceBound) return dm + h * ds;
            const double mu = r.PredictMu(x), sigma = r.PredictSigma(x);
            const double u  = (mu - r.PredictMu(r.PredictMaximumPointFromData())) / sigma;
            VectorXd     g  = NormalCdf(u) * dm + NormalPdf(u) * ds;
            bool         bad = sigma < 1e-10;
            for (long i = 0; i < g.size(); ++i) bad = bad || std::isnan(g(i));
            return bad ? VectorXd::Zero(x.size()) : g;
        }

        MatrixXd RandomStarts(unsigned num_dim, unsigned count)
        {
            // reference: x_ini = 0.5 (Random + 1) per start (src/acquisition-function.cpp:127), here drawn up front so that
            // the whole start set goes to the device in one transfer
            MatrixXd starts(num_dim, count);
            for (unsigned i = 0; i < count; ++i) eig::SetCol(starts, i, utils::GenerateRandomVector(num_dim));
            return starts;
        }

        sls_gp* RequireHandle(const Regressor& r)
        {
            sls_gp* h = r.GetDeviceHandle();
            if (!h)
                throw std::invalid_argument("acquisition_func: the multi-start maximiser needs a device-resident regressor "
                                            "(GaussianProcessRegressor / PreferenceRegressor); there is no host fallback");
            return h;
        }
    } // namespace

    // reference: src/acquisition-function.cpp:170-198
    double acquisition_func::CalcAcquisitionValue(const Regressor& regressor, const VectorXd& x, const AcquisitionFuncType func_type,
                                                  const double hyperparam)
    {
        if (regressor.GetSmallY().rows() == 0) return 0.0;
        if (sls_gp* h = regressor.GetDeviceHandle())
        {
            double v = 0.0;
            device::Check(sls_acq_eval(h, AcqId(func_type), hyperparam, x.data(), 1, &v, nullptr), "sls_acq_eval");
            return v;
        }
        return GenericValue(regressor, x, func_type, hyperparam);
    }

    // reference: src/acquisition-function.cpp:200-230
    VectorXd acquisition_func::CalcAcquisitionValueDerivative(const Regressor& regressor, const VectorXd& x,
                                                              const AcquisitionFuncType func_type, const double hyperparam)
    {
        if (regressor.GetSmallY().rows() == 0) return VectorXd::Zero(x.size());
        if (sls_gp* h = regressor.GetDeviceHandle())
        {
            double   v = 0.0;
            VectorXd g(x.size());
            device::Check(sls_acq_eval(h, AcqId(func_type), hyperparam, x.data(), 1, &v, g.data()), "sls_acq_eval");
            return g;
        }
        return GenericDerivative(regressor, x, func_type, hyperparam);
    }

    VectorXd acquisition_func::CalcAcquisitionValues(const Regressor& regressor, const MatrixXd& Xs, const AcquisitionFuncType func_type,
                                                     const double hyperparam, MatrixXd* grad)
    {
        const long M = Xs.cols();
        VectorXd   v = VectorXd::Zero(M);
        if (grad) *grad = MatrixXd::Zero(Xs.rows(), M);
        if (regressor.GetSmallY().rows() == 0) return v;
        device::Check(sls_acq_eval(RequireHandle(regressor), AcqId(func_type), hyperparam, Xs.data(), static_cast<int>(M), v.data(),
                                   grad ? grad->data() : nullptr),
                      "sls_acq_eval");
        return v;
    }

    VectorXd acquisition_func::FindNextPointFromStarts(const Regressor& regressor, const MatrixXd& starts,
                                                       const unsigned num_local_search_iters, const AcquisitionFuncType func_type,
                                                       const double hyperparam, double* value)
    {
        VectorXd x(starts.rows());
        double   v   = 0.0;
        long     idx = 0;
        if (device::Multi() != nullptr)
        {
            // several GPUs configured (device::SetDevices / $SLS_DEVICES): the iterations of the reference's parallel
            // multi-start loop (:125-141) share only the const regressor, so the start set is split over the devices, each
            // holding a replica of the fitted state, and the per-device winners meet in ONE ncclAllGather.
            RequireHandle(regressor);
            const int kernel = regressor.GetKernelType() == KernelType::ArdSquaredExponentialKernel ? SLS_KERNEL_ARD_SQUARED_EXPONENTIAL
                                                                                                    : SLS_KERNEL_ARD_MATERN52;
            device::MultiGpHandle replicas(regressor.GetLargeX(), regressor.GetSmallY(), regressor.GetKernelHyperparams(),
                                           regressor.GetNoiseHyperparam(), kernel);
            device::Check(sls_multi_acq_maximize(replicas.h, AcqId(func_type), hyperparam, starts.data(),
                                                 static_cast<int>(starts.cols()), static_cast<int>(num_local_search_iters), nullptr,
                                                 x.data(), &v, &idx, nullptr),
                          "sls_multi_acq_maximize");
            if (value) *value = v;
            return x;
        }
        device::Check(sls_acq_maximize(RequireHandle(regressor), AcqId(func_type), hyperparam, starts.data(),
                                       static_cast<int>(starts.cols()), static_cast<int>(num_local_search_iters), nullptr, 0, x.data(),
                                       &v, &idx, nullptr, nullptr),
                      "sls_acq_maximize");
        if (value) *value = v;
        return x;
    }

    // reference: src/acquisition-function.cpp:232-244 + FindGlobalSolution :112-153 (parallelised multi-start branch)
    VectorXd acquisition_func::FindNextPoint(const Regressor& regressor, const unsigned num_global_search_iters,
                                             const unsigned num_local_search_iters, const AcquisitionFuncType func_type,
                                             const double hyperparam)
    {
        const unsigned num_dim = regressor.GetNumDims();
        return FindNextPointFromStarts(regressor, RandomStarts(num_dim, num_global_search_iters), num_local_search_iters, func_type,
                                       hyperparam);
    }

    // reference: src/acquisition-function.cpp:246-298
    std::vector<VectorXd> acquisition_func::FindNextPoints(const Regressor& regressor, const unsigned num_points,
                                                           const unsigned num_global_search_iters,
                                                           const unsigned num_local_search_iters, const AcquisitionFuncType func_type,
                                                           const double hyperparam)
    {
        const unsigned        num_dim = regressor.GetNumDims();
        std::vector<VectorXd> points;
        sls_gp*               mean_handle = RequireHandle(regressor);
        const VectorXd        theta       = regressor.GetKernelHyperparams();

        // Dummy regressor that only tracks the variance.  Like the reference (:261-262, :293) it is built with the
        // DEFAULT kernel type (Matern-5/2), not the regressor's own -- reproduced on purpose.
        // It never exposes m_K_y / m_K_y_inv, so it is built without the host copies (per-object flag: no global state).
        std::shared_ptr<GaussianProcessRegressor> temp = std::make_shared<GaussianProcessRegressor>(
            regressor.GetLargeX(), regressor.GetSmallY(), theta, regressor.GetNoiseHyperparam(), KernelType::ArdMatern52Kernel,
            /*materialize_matrices=*/false);

        for (unsigned i = 0; i < num_points; ++i)
        {
            const MatrixXd starts = RandomStarts(num_dim, num_global_search_iters);
            VectorXd       x_star(num_dim);
            double         v   = 0.0;
            long           idx = 0;
            device::Check(sls_acq_maximize_pair(mean_handle, temp->GetDeviceHandle(), AcqId(func_type), hyperparam, starts.data(),
                                                static_cast<int>(starts.cols()), static_cast<int>(num_local_search_iters), nullptr,
                                                x_star.data(), &v, &idx),
                          "sls_acq_maximize_pair");
            points.push_back(x_star);
            if (points.size() != num_points)
            {
                // append the new point to the dummy regressor; its value is irrelevant for the variance (the reference
                // uses PredictMu, :286-289).  The reference rebuilds the regressor (:293); here the fitted state grows by a
                // rank-1 update on the device.
                temp->AppendPoint(x_star, temp->PredictMu(x_star));
            }
        }
        return points;
    }
} // namespace sequential_line_search
