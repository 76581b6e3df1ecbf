// C entry points into the host C++ layer for ctypes-driven tests and tools: the two MAP fits of the reference
// (src/gaussian-process-regressor.cpp:274-299, src/preference-regressor.cpp:332-403) run through the restated classes
// exactly as a C++ caller would run them; only plain pointers cross the boundary.  Column-major inputs as Eigen.
#include <sequential-line-search/acquisition-function.hpp>
#include <sequential-line-search/gaussian-process-regressor.hpp>
#include <sequential-line-search/preference-regressor.hpp>

#include <cstring>
#include <exception>
#include <string>

using namespace sequential_line_search;

namespace
{
    thread_local std::string g_err;
    KernelType KernelOf(int k) { return k == 0 ? KernelType::ArdSquaredExponentialKernel : KernelType::ArdMatern52Kernel; }
} // namespace

extern "C" const char* slsh_last_error() { return g_err.c_str(); }

/// GaussianProcessRegressor(X, y, kernel): out_x = (a, b, r_1..r_D); stats = {final value, DIRECT value, value at the
/// prior medians, DIRECT evaluations, local evaluations, seconds, 0, 0}.
extern "C" int slsh_gp_map_fit(const double* X, int D, int N, const double* y, int kernel, double* out_x, double* stats)
{
    try
    {
        Eigen::MatrixXd Xm(D, N);
        std::memcpy(Xm.data(), X, sizeof(double) * D * N);
        Eigen::VectorXd ym(N);
        std::memcpy(ym.data(), y, sizeof(double) * N);
        const bool prev = GaussianProcessRegressor::s_materialize_matrices.exchange(false);
        GaussianProcessRegressor gp(Xm, ym, KernelOf(kernel));
        GaussianProcessRegressor::s_materialize_matrices.store(prev);
        out_x[0] = gp.GetKernelHyperparams()(0);
        out_x[1] = gp.GetNoiseHyperparam();
        for (int d = 0; d < D; ++d) out_x[2 + d] = gp.GetKernelHyperparams()(1 + d);
        const auto& s = gp.GetMapFitStats();
        stats[0] = s.final_value; stats[1] = s.direct_value; stats[2] = s.prior_value;
        stats[3] = s.evals_direct; stats[4] = s.evals_local; stats[5] = s.seconds; stats[6] = stats[7] = 0.0;
        return 0;
    }
    catch (const std::exception& e)
    {
        g_err = e.what();
        return -1;
    }
}

/// PreferenceRegressor(X, D, use_map, a, r, b, prior_var, btl, iters, kernel): out_y[M] goodness values, out_hyp =
/// (a, b, r_1..r_D) as used / estimated, *objective = MAP objective at the returned point.
extern "C" int slsh_pref_map_fit(const double* X, int D, int M, const unsigned* prefs_flat, const int* offsets, int n_prefs,
                                 int use_map, double a, double r, double b, double prior_var, double btl_scale, unsigned iters,
                                 int kernel, double* out_y, double* out_hyp, double* objective)
{
    try
    {
        Eigen::MatrixXd Xm(D, M);
        std::memcpy(Xm.data(), X, sizeof(double) * D * M);
        std::vector<Preference> prefs;
        for (int p = 0; p < n_prefs; ++p) prefs.emplace_back(std::vector<unsigned>(prefs_flat + offsets[p], prefs_flat + offsets[p + 1]));
        PreferenceRegressor reg(Xm, prefs, use_map != 0, a, r, b, prior_var, btl_scale, iters, KernelOf(kernel));
        for (int i = 0; i < M; ++i) out_y[i] = reg.GetSmallY()(i);
        out_hyp[0] = reg.GetKernelHyperparams()(0);
        out_hyp[1] = reg.GetNoiseHyperparam();
        for (int d = 0; d < D; ++d) out_hyp[2 + d] = reg.GetKernelHyperparams()(1 + d);
        if (objective) *objective = reg.GetMapObjectiveValue();
        return 0;
    }
    catch (const std::exception& e)
    {
        g_err = e.what();
        return -1;
    }
}

/// The reference's default maximiser branch (src/acquisition-function.cpp:155-165: DIRECT, then ONE L-BFGS from its result) on a
/// PreferenceRegressor with FIXED hyper-parameters (use_map = false; the goodness values are fitted), with the given relative
/// tolerances for the local search (0 = run to the evaluation cap; the previous setting is restored).  out_x[D], *out_value = the
/// acquisition value at out_x.  For the tests that hold the early stop to a bound where it picks the answer.
extern "C" int slsh_find_next_point_direct(const double* X, int D, int M, const unsigned* prefs_flat, const int* offsets, int n_prefs,
                                           double a, double r, double b, double prior_var, double btl_scale, int kernel, unsigned num_global,
                                           unsigned num_local, double ftol_rel, double xtol_rel, double* out_x, double* out_value)
{
    double f0 = 0.0, x0 = 0.0;
    acquisition_func::GetLocalSearchTolerances(&f0, &x0);
    try
    {
        Eigen::MatrixXd Xm(D, M);
        std::memcpy(Xm.data(), X, sizeof(double) * D * M);
        std::vector<Preference> prefs;
        for (int p = 0; p < n_prefs; ++p) prefs.emplace_back(std::vector<unsigned>(prefs_flat + offsets[p], prefs_flat + offsets[p + 1]));
        PreferenceRegressor reg(Xm, prefs, false, a, r, b, prior_var, btl_scale, 100, KernelOf(kernel));
        acquisition_func::SetLocalSearchTolerances(ftol_rel, xtol_rel);
        double                v = 0.0;
        const Eigen::VectorXd x = acquisition_func::FindNextPointDirect(reg, num_global, num_local, AcquisitionFuncType::ExpectedImprovement, 1.0, &v);
        acquisition_func::SetLocalSearchTolerances(f0, x0);
        for (int d = 0; d < D; ++d) out_x[d] = x(d);
        *out_value = acquisition_func::CalcAcquisitionValue(reg, x, AcquisitionFuncType::ExpectedImprovement, 1.0);
        (void)v;
        return 0;
    }
    catch (const std::exception& e)
    {
        acquisition_func::SetLocalSearchTolerances(f0, x0);
        g_err = e.what();
        return -1;
    }
}
