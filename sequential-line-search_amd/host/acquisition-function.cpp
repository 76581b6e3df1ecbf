// acquisition_func over the C ABI (reference: src/acquisition-function.cpp).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <string>
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

        GlobalSearchStrategy InitialStrategy()
        {
            if (const char* env = std::getenv("SLS_GLOBAL_SEARCH"))
            {
                const std::string v(env);
                if (v == "direct") return GlobalSearchStrategy::DirectThenLbfgs;
                if (v == "multistart") return GlobalSearchStrategy::ParallelMultiStart;
            }
#ifdef SEQUENTIAL_LINE_SEARCH_USE_PARALLELIZED_MULTI_START_SEARCH
            return GlobalSearchStrategy::ParallelMultiStart;
#else
            return GlobalSearchStrategy::DirectThenLbfgs;   // the reference's default build (CMakeLists.txt:33)
#endif
        }
        std::atomic<int> g_strategy{-1};

        sls_gp* RequireHandle(const Regressor& r)
        {
            sls_gp* h = r.GetDeviceHandle();
            if (!h)
                throw std::invalid_argument("acquisition_func: the multi-start maximiser needs a device-resident regressor "
                                            "(GaussianProcessRegressor / PreferenceRegressor); there is no host fallback");
            return h;
        }
    } // namespace

    namespace
    {
        sls_lbfgs_opts LocalSearchOpts()
        {
            sls_lbfgs_opts o;
            sls_lbfgs_default_opts(&o);
            optim::SearchTolerances(&o.ftol_rel, &o.xtol_rel);   // nloptutil::solve's defaults unless changed
            return o;
        }
    } // namespace
    void acquisition_func::SetLocalSearchTolerances(double f, double x) { optim::SetSearchTolerances(f, x); }
    void acquisition_func::GetLocalSearchTolerances(double* f, double* x) { optim::SearchTolerances(f, x); }
    void acquisition_func::SetMapFitTolerances(double f, double x) { optim::SetMapFitTolerances(f, x); }
    void acquisition_func::GetMapFitTolerances(double* f, double* x) { optim::MapFitTolerances(f, x); }

    void acquisition_func::SetGlobalSearchStrategy(GlobalSearchStrategy strategy) { g_strategy.store(static_cast<int>(strategy)); }
    GlobalSearchStrategy acquisition_func::GetGlobalSearchStrategy()
    {
        int v = g_strategy.load();
        if (v < 0)
        {
            v = static_cast<int>(InitialStrategy());
            g_strategy.store(v);
        }
        return static_cast<GlobalSearchStrategy>(v);
    }

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
        // several GPUs configured (device::SetDevices / $SLS_DEVICES): the iterations of the reference's parallel multi-start
        // loop (:125-141) share only the const regressor, so the start set is split over the devices, each holding a replica of
        // the fitted state, and the per-device winners meet in ONE ncclAllGather.  The replicas belong to the regressor's device
        // handle: they are built on the first call (the shard on the primary's device IS the primary: no second fit there) and
        // reused by every later one.
        const sls_lbfgs_opts lopts = LocalSearchOpts();   // nloptutil::solve's relative tolerances
        if (std::shared_ptr<device::MultiGpHandle> replicas = device::ReplicasFor(RequireHandle(regressor), regressor.GetLargeX().cols()))
        {
            device::Check(sls_multi_acq_maximize(replicas->h, AcqId(func_type), hyperparam, starts.data(),
                                                 static_cast<int>(starts.cols()), static_cast<int>(num_local_search_iters), &lopts,
                                                 x.data(), &v, &idx, nullptr),
                          "sls_multi_acq_maximize");
            if (value) *value = v;
            return x;
        }
        device::Check(sls_acq_maximize(RequireHandle(regressor), AcqId(func_type), hyperparam, starts.data(),
                                       static_cast<int>(starts.cols()), static_cast<int>(num_local_search_iters), &lopts, 0, x.data(),
                                       &v, &idx, nullptr, nullptr),
                      "sls_acq_maximize");
        if (value) *value = v;
        return x;
    }

    // reference: src/acquisition-function.cpp:155-165 -- DIRECT (num_global_search_iters evaluations), then L-BFGS
    // (num_local_search_iters evaluations) from its result.  Every DIRECT iteration is ONE batched device evaluation.
    VectorXd acquisition_func::FindNextPointDirect(const Regressor& regressor, const unsigned num_global_search_iters,
                                                   const unsigned num_local_search_iters, const AcquisitionFuncType func_type,
                                                   const double hyperparam, double* value)
    {
        const unsigned num_dim = regressor.GetNumDims();
        sls_gp*        h       = RequireHandle(regressor);
        // SLS_HOST_TIMING: where the time of one call goes (stderr) -- DIRECT's own bookkeeping on the host, its batched device
        // evaluations, the local phase
        static const bool timing = std::getenv("SLS_HOST_TIMING") != nullptr;   // read once per process
        using clk         = std::chrono::steady_clock;
        double ms_dev = 0.0;
        int    batches = 0;
        const optim::BatchObjective objective = [&](const std::vector<std::vector<double>>& xs, std::vector<double>& values) {
            const auto t0 = clk::now();
            MatrixXd Xs(num_dim, static_cast<long>(xs.size()));
            for (size_t m = 0; m < xs.size(); ++m)
                for (unsigned d = 0; d < num_dim; ++d) Xs(d, static_cast<long>(m)) = xs[m][d];
            values.resize(xs.size());
            device::Check(sls_acq_eval(h, AcqId(func_type), hyperparam, Xs.data(), static_cast<int>(xs.size()), values.data(), nullptr),
                          "sls_acq_eval");
            ms_dev += std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            ++batches;
        };
        const std::vector<double> lower(num_dim, 0.0), upper(num_dim, 1.0);
        const auto                t_direct0 = clk::now();
        const std::vector<double> xg = optim::DirectMaximize(objective, lower, upper, static_cast<int>(num_global_search_iters));
        const double ms_direct = std::chrono::duration<double, std::milli>(clk::now() - t_direct0).count();
        const auto   t_local0  = clk::now();
        struct Report {
            bool on; double direct, dev; int batches; clk::time_point t0;
            ~Report() {
                if (on)
                    std::fprintf(stderr, "  FindNextPointDirect: DIRECT %.2f ms (%d batched evaluations: %.2f ms on the device side, %.2f ms host bookkeeping), local phase %.2f ms\n",
                                 direct, batches, dev, direct - dev, std::chrono::duration<double, std::milli>(clk::now() - t0).count());
            }
        } report{timing, ms_direct, ms_dev, batches, t_local0};
        MatrixXd start(num_dim, 1);
        for (unsigned d = 0; d < num_dim; ++d) start(d, 0) = xg[d];
        if (num_local_search_iters == 0)
        {
            VectorXd x(num_dim);
            for (unsigned d = 0; d < num_dim; ++d) x(d) = xg[d];
            if (value) *value = CalcAcquisitionValue(regressor, x, func_type, hyperparam);
            return x;
        }
        VectorXd x(num_dim);
        double   v   = 0.0;
        long     idx = 0;
        const sls_lbfgs_opts lopts = LocalSearchOpts();   // nloptutil::solve's relative tolerances
        device::Check(sls_acq_maximize(h, AcqId(func_type), hyperparam, start.data(), 1, static_cast<int>(num_local_search_iters), &lopts,
                                       0, x.data(), &v, &idx, nullptr, nullptr),
                      "sls_acq_maximize");
        if (value) *value = v;
        return x;
    }

    // reference: src/acquisition-function.cpp:232-244 + FindGlobalSolution :112-167 (both branches, chosen at run time)
    VectorXd acquisition_func::FindNextPoint(const Regressor& regressor, const unsigned num_global_search_iters,
                                             const unsigned num_local_search_iters, const AcquisitionFuncType func_type,
                                             const double hyperparam)
    {
        const unsigned num_dim = regressor.GetNumDims();
        if (GetGlobalSearchStrategy() == GlobalSearchStrategy::DirectThenLbfgs)
            return FindNextPointDirect(regressor, num_global_search_iters, num_local_search_iters, func_type, hyperparam);
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
            // FindGlobalSolution on objective_for_multiple_points (:265-278): both branches, as in FindNextPoint
            MatrixXd starts;
            if (GetGlobalSearchStrategy() == GlobalSearchStrategy::DirectThenLbfgs)
            {
                const optim::BatchObjective objective = [&](const std::vector<std::vector<double>>& xs, std::vector<double>& values) {
                    MatrixXd Xs(num_dim, static_cast<long>(xs.size()));
                    for (size_t m = 0; m < xs.size(); ++m)
                        for (unsigned d = 0; d < num_dim; ++d) Xs(d, static_cast<long>(m)) = xs[m][d];
                    values.resize(xs.size());
                    device::Check(sls_acq_eval_pair(mean_handle, temp->GetDeviceHandle(), AcqId(func_type), hyperparam, Xs.data(),
                                                    static_cast<int>(xs.size()), values.data(), nullptr),
                                  "sls_acq_eval_pair");
                };
                const std::vector<double> lower(num_dim, 0.0), upper(num_dim, 1.0);
                const std::vector<double> xg = optim::DirectMaximize(objective, lower, upper, static_cast<int>(num_global_search_iters));
                starts = MatrixXd(num_dim, 1);
                for (unsigned d = 0; d < num_dim; ++d) starts(d, 0) = xg[d];
            }
            else
            {
                starts = RandomStarts(num_dim, num_global_search_iters);
            }
            VectorXd x_star(num_dim);
            double   v   = 0.0;
            long     idx = 0;
            const sls_lbfgs_opts lopts = LocalSearchOpts();   // nloptutil::solve's relative tolerances
            device::Check(sls_acq_maximize_pair(mean_handle, temp->GetDeviceHandle(), AcqId(func_type), hyperparam, starts.data(),
                                                static_cast<int>(starts.cols()),
                                                std::max(1, static_cast<int>(num_local_search_iters)), &lopts, x_star.data(), &v, &idx),
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
