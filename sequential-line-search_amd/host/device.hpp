// Host-side RAII over the C ABI (include/sls_hip.h): process-wide context, GP and MAP-objective handles.
#pragma once
#include <sls_hip.h>

#include <sequential-line-search/device.hpp>
#include <sequential-line-search/eigen-lite.hpp>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace sequential_line_search
{
    namespace device
    {
        /// Throws std::runtime_error carrying sls_last_error() when rc != 0 (the reference never returns error codes; a
        /// missing GPU or a non-SPD matrix must not be silent).
        void Check(int rc, const char* what);

        /// Lazily created context on device $SLS_DEVICE (default 0).
        sls_ctx* Context();

        // SetDevices / Devices: public, include/sequential-line-search/device.hpp
        /// The process-wide multi-device handle for Devices(), shared: a reconfiguration (SetDevices) lets go of it, but it lives
        /// on until the last handle built on it has gone (another thread may be in the middle of a call).
        struct MultiRef
        {
            sls_multi* m = nullptr;
            explicit MultiRef(sls_multi* m_) : m(m_) {}
            ~MultiRef();
            MultiRef(const MultiRef&)            = delete;
            MultiRef& operator=(const MultiRef&) = delete;
        };
        /// nullptr when only one device is configured.
        std::shared_ptr<MultiRef> Multi();

        struct MultiGpHandle
        {
            std::shared_ptr<MultiRef> multi;   // keeps the communicators alive
            sls_multi_gp*             h = nullptr;
            /// replicas of a fitted handle: the shard on the primary's device is the primary itself (sls_multi_gp_create_from)
            MultiGpHandle(std::shared_ptr<MultiRef> multi, sls_gp* primary);
            ~MultiGpHandle();
            MultiGpHandle(const MultiGpHandle&)            = delete;
            MultiGpHandle& operator=(const MultiGpHandle&) = delete;
        };
        /// The replicas of `primary` on the configured devices, created on first use and kept until the primary handle dies
        /// (GpHandle's destructor), the device configuration changes or the primary has grown (n_points differs): a second
        /// FindNextPoint on the same regressor fits nothing.  nullptr when only one device is configured.
        std::shared_ptr<MultiGpHandle> ReplicasFor(sls_gp* primary, long n_points);
        void                           ForgetReplicas(sls_gp* primary);
        /// how many times replicas were built (tests: "zero sls_gp_create on the second call")
        long ReplicaBuilds();

        struct GpHandle
        {
            sls_gp* h = nullptr;
            GpHandle(const Eigen::MatrixXd& X, const Eigen::VectorXd& y, const Eigen::VectorXd& theta, double b, int kernel);
            ~GpHandle();
            GpHandle(const GpHandle&)            = delete;
            GpHandle& operator=(const GpHandle&) = delete;
        };

        /// MAP objective handles on every configured device (Devices().size() >= 2): batches of independent evaluations are
        /// dealt round-robin over them (sls_multi_gp_nll_batch).
        struct MultiNllHandle
        {
            std::shared_ptr<MultiRef> multi;
            sls_multi_nll*            h = nullptr;
            MultiNllHandle(const Eigen::MatrixXd& X, int kernel);
            ~MultiNllHandle();
            MultiNllHandle(const MultiNllHandle&)            = delete;
            MultiNllHandle& operator=(const MultiNllHandle&) = delete;
        };

        struct NllHandle
        {
            sls_nll* h = nullptr;
            NllHandle(const Eigen::MatrixXd& X, int kernel);
            ~NllHandle();
            NllHandle(const NllHandle&)            = delete;
            NllHandle& operator=(const NllHandle&) = delete;
        };
    } // namespace device

    namespace optim
    {
        /// value = f(x), and the gradient into `grad` when grad != nullptr.
        using Objective = std::function<double(const std::vector<double>& x, std::vector<double>* grad)>;

        /// Bounded L-BFGS MAXIMISER (projected gradient, Armijo backtracking, m = 8), at most max_evals objective
        /// evaluations.  Stand-in for nloptutil::solve(..., LD_LBFGS / LD_TNEWTON, ..., is_max = true, max_evals):
        /// NLopt is not available, so iterates differ from the reference while the optimum is the same.
        /// ftol_rel / xtol_rel: NLopt's relative stopping tests on every accepted step (0 = off), see SearchTolerances.
        std::vector<double> MaximizeBounded(const Objective& f, std::vector<double> x0, const std::vector<double>& lower,
                                            const std::vector<double>& upper, int max_evals, double* best_value = nullptr,
                                            int* evals_used = nullptr, double ftol_rel = 0.0, double xtol_rel = 0.0);

        /// The relative tolerances every search of this layer runs with: nloptutil::solve's defaults relative_func_tolerance =
        /// relative_param_tolerance = 1e-6, SLS_LOCAL_SEARCH_TOL=<v> sets both (0 = off).  The setting of the
        /// acquisition maximiser's local searches ONLY (acquisition_func::SetLocalSearchTolerances is its setter); the MAP fits have
        /// their own pair, below.
        void SetSearchTolerances(double ftol_rel, double xtol_rel);
        void SearchTolerances(double* ftol_rel, double* xtol_rel);
        /// The MAP fits do NOT take those by default: their optima are what pins this layer to independent implementations (scipy,
        /// tests/golden/map_optima*.npz), and NLopt's tests look at ONE step -- on the slow tail of the joint preference fit a step
        /// below 1e-6 still leaves 4e-4 of the objective on the table (tests/test_gpu_map_device.py).  SLS_MAP_FIT_TOL=<v> opts in
        /// (both tolerances; what nloptutil::solve's defaults would do to the reference's fits is v = 1e-6), and so does
        /// SetMapFitTolerances (acquisition_func::SetMapFitTolerances / set_map_fit_tolerances in the Python module) at run time.
        /// This is a deviation from the reference, where the same nloptutil::solve defaults apply to the TNEWTON MAP fits too:
        /// INTEGRATION.md 2.
        void SetMapFitTolerances(double ftol_rel, double xtol_rel);
        void MapFitTolerances(double* ftol_rel, double* xtol_rel);

        /// values[k] = f(xs[k]) for a whole batch of points (one device call per batch).
        using BatchObjective = std::function<void(const std::vector<std::vector<double>>& xs, std::vector<double>& values)>;

        /// DIRECT global MAXIMISER on the box [lower, upper] with at most max_evals objective evaluations (host/direct.cpp).
        /// Stand-in for nloptutil::solve(x0, upper, lower, f, nlopt::GN_DIRECT, data, is_max = true, max_evals) -- DIRECT
        /// ignores x0.  Returns the centre of the best rectangle.
        std::vector<double> DirectMaximize(const BatchObjective& f, const std::vector<double>& lower, const std::vector<double>& upper,
                                           int max_evals, double* best_value = nullptr, int* evals_used = nullptr);
    } // namespace optim
} // namespace sequential_line_search
