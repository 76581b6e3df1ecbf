// GaussianProcessRegressor over the C ABI (reference: src/gaussian-process-regressor.cpp).
#include <chrono>
#include <memory>
#include <cmath>
#include <sequential-line-search/gaussian-process-regressor.hpp>

#include "device.hpp"

using Eigen::MatrixXd;
using Eigen::VectorXd;

namespace sequential_line_search
{
    std::atomic<bool> GaussianProcessRegressor::s_materialize_matrices{true};

    namespace
    {
        int KernelId(KernelType t) { return t == KernelType::ArdSquaredExponentialKernel ? SLS_KERNEL_ARD_SQUARED_EXPONENTIAL : SLS_KERNEL_ARD_MATERN52; }
    }

    // reference: src/gaussian-process-regressor.cpp:198-212
    GaussianProcessRegressor::GaussianProcessRegressor(const MatrixXd& X, const VectorXd& y, const KernelType kernel_type)
        : Regressor(kernel_type), m_X(X), m_y(y), m_noise_hyperparam(0.0), m_materialize(s_materialize_matrices.load())
    {
        if (X.rows() == 0) return;   // inert object, like the reference
        PerformMapEstimation();
        BuildDeviceState();
    }

    // reference: src/gaussian-process-regressor.cpp:214-232
    GaussianProcessRegressor::GaussianProcessRegressor(const MatrixXd& X, const VectorXd& y, const VectorXd& kernel_hyperparams,
                                                       double noise_hyperparam, const KernelType kernel_type,
                                                       bool materialize_matrices)
        : Regressor(kernel_type), m_X(X), m_y(y), m_kernel_hyperparams(kernel_hyperparams), m_noise_hyperparam(noise_hyperparam),
          m_materialize(materialize_matrices && s_materialize_matrices.load())
    {
        if (X.rows() == 0) return;
        BuildDeviceState();
    }

    void GaussianProcessRegressor::BuildDeviceState()
    {
        m_handle = std::make_shared<device::GpHandle>(m_X, m_y, m_kernel_hyperparams, m_noise_hyperparam, KernelId(m_kernel_type));
        if (m_materialize)
        {
            const long N = m_X.cols();
            m_K_y        = MatrixXd(N, N);
            m_K_y_inv    = MatrixXd(N, N);
            device::Check(sls_gp_get_matrix(m_handle->h, SLS_GP_K_Y, m_K_y.data()), "sls_gp_get_matrix(K_y)");
            device::Check(sls_gp_get_matrix(m_handle->h, SLS_GP_K_Y_INV, m_K_y_inv.data()), "sls_gp_get_matrix(K_y_inv)");
        }
    }

    void GaussianProcessRegressor::AppendPoint(const VectorXd& x, double y)
    {
        device::Check(sls_gp_append_point(m_handle->h, x.data(), y), "sls_gp_append_point");
        m_X = eig::AppendCol(m_X, x);
        VectorXd yn(m_y.size() + 1);
        for (long i = 0; i < m_y.size(); ++i) yn(i) = m_y(i);
        yn(m_y.size()) = y;
        m_y            = yn;
        if (m_materialize)
        {
            const long N = m_X.cols();
            m_K_y        = MatrixXd(N, N);
            m_K_y_inv    = MatrixXd(N, N);
            device::Check(sls_gp_get_matrix(m_handle->h, SLS_GP_K_Y, m_K_y.data()), "sls_gp_get_matrix(K_y)");
            device::Check(sls_gp_get_matrix(m_handle->h, SLS_GP_K_Y_INV, m_K_y_inv.data()), "sls_gp_get_matrix(K_y_inv)");
        }
    }

    sls_gp* GaussianProcessRegressor::GetDeviceHandle() const { return m_handle ? m_handle->h : nullptr; }

    // reference: src/gaussian-process-regressor.cpp:234-272 -- single-point forms of the batched device evaluation
    double GaussianProcessRegressor::PredictMu(const VectorXd& x) const
    {
        double mu = 0.0;
        device::Check(sls_gp_predict(m_handle->h, x.data(), 1, &mu, nullptr), "sls_gp_predict");
        return mu;
    }
    double GaussianProcessRegressor::PredictSigma(const VectorXd& x) const
    {
        double sigma = 0.0;
        device::Check(sls_gp_predict(m_handle->h, x.data(), 1, nullptr, &sigma), "sls_gp_predict");
        return sigma;
    }
    VectorXd GaussianProcessRegressor::PredictMuDerivative(const VectorXd& x) const
    {
        VectorXd g(x.size());
        device::Check(sls_gp_predict_grad(m_handle->h, x.data(), 1, g.data(), nullptr), "sls_gp_predict_grad");
        return g;
    }
    VectorXd GaussianProcessRegressor::PredictSigmaDerivative(const VectorXd& x) const
    {
        VectorXd g(x.size());
        device::Check(sls_gp_predict_grad(m_handle->h, x.data(), 1, nullptr, g.data()), "sls_gp_predict_grad");
        return g;
    }

    // reference: src/gaussian-process-regressor.cpp:274-299.  Objective + gradient on the device (sls_gp_nll_grad);
    // drivers: DIRECT(300) as the reference (host/direct.cpp), then a bounded L-BFGS in log-parameters in place of
    // TNEWTON(1000) (same bounds [1e-8, 50]; NLopt is unavailable, iterates are not comparable, the optimum is).
    void GaussianProcessRegressor::PerformMapEstimation()
    {
        const int         D = static_cast<int>(m_X.rows());
        const auto        t_start = std::chrono::steady_clock::now();
        device::NllHandle nll(m_X, KernelId(m_kernel_type));
        double            ftol_rel = 0.0, xtol_rel = 0.0;   // off unless SLS_MAP_FIT_TOL is set (device.hpp: optim::MapFitTolerances)
        optim::MapFitTolerances(&ftol_rel, &xtol_rel);
        device::Check(sls_nll_set_tolerances(nll.h, ftol_rel, xtol_rel), "sls_nll_set_tolerances");
        const double      lo = std::log(1e-8), hi = std::log(5e+01);

        auto objective = [&](const std::vector<double>& z, std::vector<double>* grad) -> double {
            std::vector<double> x(z.size()), g(z.size());
            for (size_t i = 0; i < z.size(); ++i) x[i] = std::exp(z[i]);
            double    v  = 0.0;
            const int rc = sls_gp_nll_grad(nll.h, m_y.data(), x.data(), &v, grad ? g.data() : nullptr);
            if (rc == SLS_ERR_NOT_SPD) return -HUGE_VAL;   // numerically singular K_y: reject the trial point
            device::Check(rc, "sls_gp_nll_grad");
            if (grad)
            {
                grad->resize(z.size());
                for (size_t i = 0; i < z.size(); ++i) (*grad)[i] = g[i] * x[i];   // d/d log x
            }
            return v;
        };

        // global phase: DIRECT, 300 evaluations on the reference's box [1e-8, 50]^(D+2) in the reference's (linear)
        // parameters (:291-294); DIRECT ignores x_ini.  The prior medians (the reference's x_ini) stay in the race.
        // the points of one DIRECT iteration are independent: ONE device call (one workgroup per point for N <= 128); with several
        // devices configured (device::SetDevices / SLS_DEVICES) and a problem large enough for the tiled pipeline, the points of
        // a batch are dealt over the devices -- the part of a MAP fit that shards (the N^3 factorisation of one evaluation does not)
        std::unique_ptr<device::MultiNllHandle> multi;
        if (device::Multi() && m_X.cols() > 128) multi.reset(new device::MultiNllHandle(m_X, KernelId(m_kernel_type)));
        const optim::BatchObjective batch = [&](const std::vector<std::vector<double>>& xs, std::vector<double>& values) {
            values.resize(xs.size());
            if (xs.empty()) return;
            std::vector<double> flat(xs.size() * (D + 2));
            for (size_t k = 0; k < xs.size(); ++k)
                for (int i = 0; i < D + 2; ++i) flat[k * (D + 2) + i] = xs[k][i];
            if (multi)
                device::Check(sls_multi_gp_nll_batch(multi->h, m_y.data(), flat.data(), static_cast<int>(xs.size()), values.data()),
                              "sls_multi_gp_nll_batch");
            else
                device::Check(sls_gp_nll_batch(nll.h, m_y.data(), flat.data(), static_cast<int>(xs.size()), values.data()), "sls_gp_nll_batch");
        };
        const std::vector<double> lin_lower(D + 2, 1e-8), lin_upper(D + 2, 5e+01);
        double                    direct_v = 0.0;
        const std::vector<double> xg       = optim::DirectMaximize(batch, lin_lower, lin_upper, 300, &direct_v, &m_map_stats.evals_direct);
        std::vector<double> best(D + 2);
        best[0] = std::log(0.5); best[1] = std::log(1e-4);
        for (int d = 0; d < D; ++d) best[2 + d] = std::log(0.5);
        m_map_stats.direct_value = direct_v;
        m_map_stats.prior_value  = objective(best, nullptr);
        if (direct_v > m_map_stats.prior_value)
            for (int i = 0; i < D + 2; ++i) best[i] = std::log(xg[i]);
        // local phase: bounded quasi-Newton in log-parameters from the global phase's point (reference: TNEWTON, 1000
        // evaluations, :295; same bounds)
        const std::vector<double> lower(D + 2, lo), upper(D + 2, hi);
        // N <= 128, D <= 128: the whole local phase is one launch (sls_gp_map_fit); otherwise one device objective call per evaluation
        std::vector<double> z(D + 2);
        const int rc_fit = sls_gp_map_fit(nll.h, m_y.data(), best.data(), lower.data(), upper.data(), 1000, 0, z.data(),
                                          &m_map_stats.final_value, &m_map_stats.evals_local);
        if (rc_fit == SLS_ERR_UNSUPPORTED)
            z = optim::MaximizeBounded(objective, best, lower, upper, 1000, &m_map_stats.final_value, &m_map_stats.evals_local, ftol_rel, xtol_rel);
        else
            device::Check(rc_fit, "sls_gp_map_fit");
        m_map_stats.seconds         = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();

        m_kernel_hyperparams    = VectorXd(D + 1);
        m_kernel_hyperparams(0) = std::exp(z[0]);
        m_noise_hyperparam      = std::exp(z[1]);
        for (int d = 0; d < D; ++d) m_kernel_hyperparams(1 + d) = std::exp(z[2 + d]);
    }
} // namespace sequential_line_search
