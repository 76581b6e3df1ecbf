// PreferenceRegressor over the C ABI (reference: src/preference-regressor.cpp).
#include <chrono>
#include <cmath>
#include <fstream>
#include <sequential-line-search/preference-regressor.hpp>
#include <sequential-line-search/utils.hpp>

#include "device.hpp"

using Eigen::MatrixXd;
using Eigen::VectorXd;

namespace sequential_line_search
{
    namespace
    {
        int KernelId(KernelType t) { return t == KernelType::ArdSquaredExponentialKernel ? SLS_KERNEL_ARD_SQUARED_EXPONENTIAL : SLS_KERNEL_ARD_MATERN52; }
    }

#ifndef SLS_HAVE_REAL_EIGEN
    VectorXd CholeskyFactor::solve(const VectorXd& b) const
    {
        VectorXd x = b;
        device::Check(sls_potrs(device::Context(), m_L.data(), static_cast<int>(m_L.rows()), x.data(), 1), "sls_potrs");
        return x;
    }
#endif

    // reference: src/preference-regressor.cpp:262-291
    PreferenceRegressor::PreferenceRegressor(const MatrixXd& X, const std::vector<Preference>& D, const bool use_map_hyperparams,
                                             const double default_kernel_signal_var, const double default_kernel_length_scale,
                                             const double default_noise_level, const double kernel_hyperparams_prior_var,
                                             const double btl_scale, const unsigned num_map_estimation_iters, const KernelType kernel_type)
        : Regressor(kernel_type),
          m_use_map_hyperparams(use_map_hyperparams),
          m_X(X),
          m_D(D),
          m_noise_hyperparam(default_noise_level),
          m_default_kernel_signal_var(default_kernel_signal_var),
          m_default_kernel_length_scale(default_kernel_length_scale),
          m_default_noise_level(default_noise_level),
          m_kernel_hyperparams_prior_var(kernel_hyperparams_prior_var),
          m_btl_scale(btl_scale)
    {
        if (X.cols() == 0 || D.size() == 0) return;

        static const bool timing = std::getenv("SLS_HOST_TIMING") != nullptr;   // once per process; stderr: where the constructor's time goes
        const auto        t0     = std::chrono::steady_clock::now();
        PerformMapEstimation(num_map_estimation_iters);
        const auto t1 = std::chrono::steady_clock::now();

        // final K, its Cholesky factor and the predictive state live on the device; the public members are copies
        m_handle = std::make_shared<device::GpHandle>(m_X, m_y, m_kernel_hyperparams, m_noise_hyperparam, KernelId(m_kernel_type));
        // PredictSigma / PredictSigmaDerivative of this class solve with the Cholesky factor (:299-313, :323-330); the explicit
        // inverse is GaussianProcessRegressor's formula
        device::Check(sls_gp_set_sigma_mode(m_handle->h, SLS_SIGMA_CHOLESKY_SOLVE), "sls_gp_set_sigma_mode");
        const auto t2 = std::chrono::steady_clock::now();
        const long M = m_X.cols();
        m_K          = MatrixXd(M, M);
        MatrixXd L(M, M);
        device::Check(sls_gp_get_matrix(m_handle->h, SLS_GP_K_Y, m_K.data()), "sls_gp_get_matrix(K)");
        device::Check(sls_gp_get_matrix(m_handle->h, SLS_GP_CHOL_L, L.data()), "sls_gp_get_matrix(L)");
#ifndef SLS_HAVE_REAL_EIGEN
        m_K_llt = CholeskyFactor(L);
#else
        m_K_llt = Eigen::LLT<MatrixXd>(m_K);
#endif
        if (timing)
        {
            const auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "  PreferenceRegressor: MAP estimation %.3f ms, predictor handle %.3f ms, K and L copies %.3f ms\n", ms(t0, t1), ms(t1, t2),
                         ms(t2, std::chrono::steady_clock::now()));
        }
    }

    sls_gp* PreferenceRegressor::GetDeviceHandle() const { return m_handle ? m_handle->h : nullptr; }

    // reference: src/preference-regressor.cpp:293-330
    double PreferenceRegressor::PredictMu(const VectorXd& x) const
    {
        double mu = 0.0;
        device::Check(sls_gp_predict(m_handle->h, x.data(), 1, &mu, nullptr), "sls_gp_predict");
        return mu;
    }
    double PreferenceRegressor::PredictSigma(const VectorXd& x) const
    {
        double sigma = 0.0;
        device::Check(sls_gp_predict(m_handle->h, x.data(), 1, nullptr, &sigma), "sls_gp_predict");
        return sigma;
    }
    VectorXd PreferenceRegressor::PredictMuDerivative(const VectorXd& x) const
    {
        VectorXd g(x.size());
        device::Check(sls_gp_predict_grad(m_handle->h, x.data(), 1, g.data(), nullptr), "sls_gp_predict_grad");
        return g;
    }
    VectorXd PreferenceRegressor::PredictSigmaDerivative(const VectorXd& x) const
    {
        VectorXd g(x.size());
        device::Check(sls_gp_predict_grad(m_handle->h, x.data(), 1, nullptr, g.data()), "sls_gp_predict_grad");
        return g;
    }

    // reference: src/preference-regressor.cpp:332-403.  The objective and its gradient (:129-259) run through
    // sls_pref_map_fit / sls_pref_objective; NLopt's LD_TNEWTON is replaced by a bounded L-BFGS with the same evaluation budget
    // (on the device for the reference's own problem sizes, optim::MaximizeBounded of device.hpp otherwise: same statements).
    // Hyper-parameters are optimised in log-space (their box [1e-8, 10] spans nine decades).
    void PreferenceRegressor::PerformMapEstimation(const unsigned num_iters)
    {
        const int M = static_cast<int>(m_X.cols());
        const int d = static_cast<int>(m_X.rows());

        std::vector<unsigned> flat;
        std::vector<int>      offs{0};
        for (const Preference& p : m_D)
        {
            flat.insert(flat.end(), p.begin(), p.end());
            offs.push_back(static_cast<int>(flat.size()));
        }
        sls_pref_cfg cfg;
        cfg.use_map_hyperparams = m_use_map_hyperparams ? 1 : 0;
        cfg.default_a           = m_default_kernel_signal_var;
        cfg.default_r           = m_default_kernel_length_scale;
        cfg.default_b           = m_default_noise_level;
        cfg.prior_var           = m_kernel_hyperparams_prior_var;
        cfg.btl_scale           = m_btl_scale;
#ifdef SEQUENTIAL_LINE_SEARCH_USE_NOISELESS_FORMULATION
        cfg.noiseless = 1;
#else
        cfg.noiseless = 0;
#endif
        device::NllHandle nll(m_X, KernelId(m_kernel_type));
        double            ftol_rel = 0.0, xtol_rel = 0.0;   // off unless SLS_MAP_FIT_TOL is set (device.hpp: optim::MapFitTolerances)
        optim::MapFitTolerances(&ftol_rel, &xtol_rel);
        device::Check(sls_nll_set_tolerances(nll.h, ftol_rel, xtol_rel), "sls_nll_set_tolerances");

        const int           opt_dim = m_use_map_hyperparams ? M + 2 + d : M;
        std::vector<double> lower(opt_dim, -1e+01), upper(opt_dim, +1e+01), z0(opt_dim, 0.0);
        if (m_use_map_hyperparams)
        {
            for (int i = M; i < opt_dim; ++i)
            {
                lower[i] = std::log(1e-08);
                upper[i] = std::log(1e+01);
            }
            z0[M + 0] = std::log(m_default_kernel_signal_var);
            z0[M + 1] = cfg.noiseless ? 0.5 * (lower[M + 1] + upper[M + 1]) : std::log(m_default_noise_level);
            for (int i = 0; i < d; ++i) z0[M + 2 + i] = std::log(m_default_kernel_length_scale);
        }

        auto objective = [&](const std::vector<double>& z, std::vector<double>* grad) -> double {
            std::vector<double> x(z), g(z.size());
            for (int i = M; i < opt_dim; ++i) x[i] = std::exp(z[i]);
            double    v  = 0.0;
            const int rc = sls_pref_objective(nll.h, flat.data(), offs.data(), static_cast<int>(m_D.size()), x.data(), &cfg, &v,
                                              grad ? g.data() : nullptr);
            if (rc == SLS_ERR_NOT_SPD) return -HUGE_VAL;
            device::Check(rc, "sls_pref_objective");
            if (grad)
            {
                *grad = g;
                for (int i = M; i < opt_dim; ++i) (*grad)[i] = g[i] * x[i];
            }
            return v;
        };

        // M <= 128, D <= 128 (with or without hyper-parameters): the whole fit is ONE launch -- objective, BTL terms and the optimiser run
        // on the device (sls_pref_map_fit); otherwise the same optimiser runs here with one sls_pref_objective call per evaluation
        std::vector<double> z(opt_dim);
        const int rc_fit = sls_pref_map_fit(nll.h, flat.data(), offs.data(), static_cast<int>(m_D.size()), &cfg, z0.data(), lower.data(),
                                            upper.data(), static_cast<int>(num_iters), 0, z.data(), &m_map_objective, nullptr);
        if (rc_fit == SLS_ERR_UNSUPPORTED)
            z = optim::MaximizeBounded(objective, z0, lower, upper, static_cast<int>(num_iters), &m_map_objective, nullptr, ftol_rel, xtol_rel);
        else
            device::Check(rc_fit, "sls_pref_map_fit");

        m_y = VectorXd(M);
        for (int i = 0; i < M; ++i) m_y(i) = z[i];
        m_kernel_hyperparams = VectorXd(d + 1);
        if (m_use_map_hyperparams)
        {
            m_kernel_hyperparams(0) = std::exp(z[M + 0]);
            m_noise_hyperparam      = cfg.noiseless ? 0.0 : std::exp(z[M + 1]);
            for (int i = 0; i < d; ++i) m_kernel_hyperparams(1 + i) = std::exp(z[M + 2 + i]);
        }
        else
        {
            m_kernel_hyperparams(0) = m_default_kernel_signal_var;
            for (int i = 0; i < d; ++i) m_kernel_hyperparams(1 + i) = m_default_kernel_length_scale;
            m_noise_hyperparam = cfg.noiseless ? 0.0 : m_default_noise_level;
        }
    }

    // reference: src/preference-regressor.cpp:405-410
    VectorXd PreferenceRegressor::FindArgMax() const
    {
        int i = 0;
        m_y.maxCoeff(&i);
        return eig::Col(m_X, i);
    }

    // reference: src/preference-regressor.cpp:412-432
    void PreferenceRegressor::DampData(const std::string& dir_path, const std::string& prefix) const
    {
        utils::ExportMatrixToCsv(dir_path + "/" + prefix + "X.csv", m_X);
        std::ofstream ofs(dir_path + "/" + prefix + "D.csv");
        for (const Preference& p : m_D)
        {
            for (size_t j = 0; j < p.size(); ++j) ofs << p[j] << (j + 1 != p.size() ? "," : "");
            ofs << std::endl;
        }
    }
} // namespace sequential_line_search
