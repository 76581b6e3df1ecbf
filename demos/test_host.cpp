// Behavioural checks of the host C++ layer (run on the GPU box by tests/test_gpu_host_cpp.py).  Exit code 0 = pass.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <thread>
#include <sequential-line-search/acquisition-function.hpp>
#include <sequential-line-search/device.hpp>
#include <sequential-line-search/gaussian-process-regressor.hpp>
#include <sequential-line-search/preference-data-manager.hpp>
#include <sequential-line-search/preference-regressor.hpp>
#include <sequential-line-search/preferential-bayesian-optimizer.hpp>
#include <sequential-line-search/sequential-line-search.hpp>
#include <sequential-line-search/slider.hpp>
#include <sequential-line-search/utils.hpp>

#include "device.hpp"   // host/device.hpp: the replica cache and the shared multi-device handle (internal API)

using namespace sequential_line_search;
using Eigen::MatrixXd;
using Eigen::VectorXd;

static int g_fail = 0;
#define EXPECT(cond)                                                                  \
    do {                                                                              \
        if (!(cond)) { std::cout << "FAIL " << __LINE__ << ": " #cond << std::endl; ++g_fail; } \
    } while (0)

static MatrixXd RandomPoints(int D, int N)
{
    MatrixXd X(D, N);
    for (int i = 0; i < N; ++i) eig::SetCol(X, i, utils::GenerateRandomVector(D));
    return X;
}

int main()
{
    utils::SetRandomSeed(7);
    // ---- free functions and fixed-hyper-parameter GPR against the scalar kernel definitions ----
    {
        const int      D = 3, N = 40;
        const MatrixXd X = RandomPoints(D, N);
        VectorXd       y(N), theta(D + 1);
        for (int i = 0; i < N; ++i) y(i) = std::sin(3.0 * X(0, i)) + X(1, i) * X(2, i);
        theta(0) = 0.5; theta(1) = 0.4; theta(2) = 0.5; theta(3) = 0.6;
        for (KernelType kt : {KernelType::ArdSquaredExponentialKernel, KernelType::ArdMatern52Kernel})
        {
            GaussianProcessRegressor gp(X, y, theta, 0.01, kt);
            const MatrixXd           K = CalcLargeKY(X, theta, 0.01, gp.GetKernel());
            double                   err = 0.0;
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j)
                {
                    const double ref = gp.GetKernel()(eig::Col(X, i), eig::Col(X, j), theta) + (i == j ? 0.01 : 0.0);
                    err              = std::max(err, std::abs(K(i, j) - ref));
                    err              = std::max(err, std::abs(gp.m_K_y(i, j) - ref));
                }
            EXPECT(err < 1e-12);
            // K_y * K_y_inv = I
            double res = 0.0;
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j)
                {
                    double s = 0.0;
                    for (int k = 0; k < N; ++k) s += gp.m_K_y(i, k) * gp.m_K_y_inv(k, j);
                    res = std::max(res, std::abs(s - (i == j ? 1.0 : 0.0)));
                }
            EXPECT(res < 1e-8);
            // mu(x) = k^T K^-1 y from the public members, sigma^2 = a - k^T K^-1 k; gradients by finite differences
            const VectorXd x = utils::GenerateRandomVector(D);
            const VectorXd k = CalcSmallK(x, X, theta, gp.GetKernel());
            double         mu = 0.0, kk = 0.0;
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j)
                {
                    mu += k(i) * gp.m_K_y_inv(i, j) * y(j);
                    kk += k(i) * gp.m_K_y_inv(i, j) * k(j);
                }
            EXPECT(std::abs(gp.PredictMu(x) - mu) < 1e-8);
            EXPECT(std::abs(gp.PredictSigma(x) - std::sqrt(theta(0) - kk)) < 1e-8);
            const VectorXd dm = gp.PredictMuDerivative(x), ds = gp.PredictSigmaDerivative(x);
            const VectorXd de = acquisition_func::CalcAcquisitionValueDerivative(gp, x, AcquisitionFuncType::ExpectedImprovement);
            for (int d = 0; d < D; ++d)
            {
                VectorXd xp = x, xm = x;
                xp(d) += 1e-6; xm(d) -= 1e-6;
                EXPECT(std::abs((gp.PredictMu(xp) - gp.PredictMu(xm)) / 2e-6 - dm(d)) < 1e-5);
                EXPECT(std::abs((gp.PredictSigma(xp) - gp.PredictSigma(xm)) / 2e-6 - ds(d)) < 1e-5);
                const double fp = acquisition_func::CalcAcquisitionValue(gp, xp, AcquisitionFuncType::ExpectedImprovement);
                const double fm = acquisition_func::CalcAcquisitionValue(gp, xm, AcquisitionFuncType::ExpectedImprovement);
                EXPECT(std::abs((fp - fm) / 2e-6 - de(d)) < 1e-5);
            }
            // PredictMaximumPointFromData == argmax of PredictMu over the data (the reference's loop)
            int    best = 0;
            double bv   = -1e300;
            for (int i = 0; i < N; ++i)
            {
                const double f = gp.PredictMu(eig::Col(X, i));
                if (f > bv) { bv = f; best = i; }
            }
            EXPECT((gp.PredictMaximumPointFromData() - eig::Col(X, best)).norm() == 0.0);
            // batched == single
            MatrixXd Xs = RandomPoints(D, 9), G;
            const VectorXd v = acquisition_func::CalcAcquisitionValues(gp, Xs, AcquisitionFuncType::GaussianProcessUpperConfidenceBound, 2.0, &G);
            for (int m = 0; m < 9; ++m)
                EXPECT(std::abs(v(m) - acquisition_func::CalcAcquisitionValue(gp, eig::Col(Xs, m), AcquisitionFuncType::GaussianProcessUpperConfidenceBound, 2.0)) < 1e-10);
            // maximiser: never worse than the best start, inside the box
            const MatrixXd starts = RandomPoints(D, 64);
            double         vmax   = 0.0;
            const VectorXd xs = acquisition_func::FindNextPointFromStarts(gp, starts, 30, AcquisitionFuncType::ExpectedImprovement, 1.0, &vmax);
            const VectorXd v0 = acquisition_func::CalcAcquisitionValues(gp, starts, AcquisitionFuncType::ExpectedImprovement);
            EXPECT(vmax >= v0.maxCoeff() - 1e-15);
            for (int d = 0; d < D; ++d) EXPECT(xs(d) >= 0.0 && xs(d) <= 1.0);
            // multi-GPU path behind the same call: three logical shards on device 0 (replicated fit, starts split 22/21/21,
            // per-shard winners merged by first maximum) must return the single-device winner bit for bit
            device::SetDevices({0, 0, 0});
            const long     builds0 = device::ReplicaBuilds();
            double         vmulti  = 0.0;
            const VectorXd xm = acquisition_func::FindNextPointFromStarts(gp, starts, 30, AcquisitionFuncType::ExpectedImprovement, 1.0, &vmulti);
            EXPECT(vmulti == vmax && (xm - xs).norm() == 0.0);
            EXPECT(device::ReplicaBuilds() == builds0 + 1);
            // the replicas belong to the regressor's handle: a second (and third) call on the same regressor fits nothing
            double         vagain = 0.0;
            const VectorXd xa = acquisition_func::FindNextPointFromStarts(gp, starts, 30, AcquisitionFuncType::ExpectedImprovement, 1.0, &vagain);
            acquisition_func::FindNextPointFromStarts(gp, starts, 10, AcquisitionFuncType::GaussianProcessUpperConfidenceBound, 1.5);
            EXPECT(device::ReplicaBuilds() == builds0 + 1);
            EXPECT(vagain == vmax && (xa - xs).norm() == 0.0);
            // a reconfiguration while a handle built on the old configuration is still alive must not pull it from under that handle
            {
                std::shared_ptr<device::MultiRef> old_multi = device::Multi();
                device::SetDevices({0, 0});
                double         v2 = 0.0;
                const VectorXd x2 = acquisition_func::FindNextPointFromStarts(gp, starts, 30, AcquisitionFuncType::ExpectedImprovement, 1.0, &v2);
                EXPECT(v2 == vmax && (x2 - xs).norm() == 0.0);
                EXPECT(device::ReplicaBuilds() == builds0 + 2);             // new configuration: new replicas
                EXPECT(old_multi && old_multi->m != nullptr && sls_multi_size(old_multi->m) == 3);   // still alive and usable
            }
            device::SetDevices({0});
        }
        // empty regressor: acquisition value 0 (src/acquisition-function.cpp:176-179)
        GaussianProcessRegressor empty(MatrixXd(0, 0), VectorXd(0));
        EXPECT(acquisition_func::CalcAcquisitionValue(empty, VectorXd::Constant(1, 0.3), AcquisitionFuncType::ExpectedImprovement) == 0.0);
    }
    // ---- the reference calls the const predictors from many worker threads on ONE regressor
    //      (src/acquisition-function.cpp:125-144): concurrent calls must give the sequential answers ----
    {
        const int      D = 3, N = 50, T = 8, M = 40;
        const MatrixXd X = RandomPoints(D, N);
        VectorXd       y(N), theta(D + 1);
        for (int i = 0; i < N; ++i) y(i) = std::cos(2.0 * X(0, i)) - X(1, i);
        theta(0) = 0.5; theta(1) = 0.4; theta(2) = 0.5; theta(3) = 0.6;
        GaussianProcessRegressor gp(X, y, theta, 0.01);
        const MatrixXd           Q = RandomPoints(D, T * M);
        std::vector<double>      seq(T * M), par(T * M);
        for (int i = 0; i < T * M; ++i) seq[i] = acquisition_func::CalcAcquisitionValue(gp, eig::Col(Q, i), AcquisitionFuncType::ExpectedImprovement) + gp.PredictSigma(eig::Col(Q, i));
        std::vector<std::thread> workers;
        for (int t = 0; t < T; ++t)
            workers.emplace_back([&, t]() {
                for (int i = t * M; i < (t + 1) * M; ++i)
                    par[i] = acquisition_func::CalcAcquisitionValue(gp, eig::Col(Q, i), AcquisitionFuncType::ExpectedImprovement) + gp.PredictSigma(eig::Col(Q, i));
            });
        for (auto& w : workers) w.join();
        bool same = true;
        for (int i = 0; i < T * M; ++i) same = same && (seq[i] == par[i]);
        EXPECT(same);
    }
    // ---- the same pattern as a throughput check: 8 host threads x 1000 single-point PredictMu on ONE regressor against one thread
    // (src/acquisition-function.cpp:125-144: the reference's workers share a const regressor).  Each call borrows a stream + mapped
    // block of the handle (sls_gp::EvalSlot) instead of the context's lock: the calls overlap ----
    {
        const int D = 8, N = 60, T = 8, M = 1000;
        const MatrixXd X = RandomPoints(D, N);
        VectorXd       y(N), theta(D + 1);
        for (int i = 0; i < N; ++i) y(i) = std::cos(2.0 * X(0, i)) - X(1, i);
        theta(0) = 0.5;
        for (int d = 0; d < D; ++d) theta(1 + d) = 0.5;
        GaussianProcessRegressor gp(X, y, theta, 0.01);
        const MatrixXd           Q = RandomPoints(D, M);
        std::vector<double>      one(M), many(T * M);
        for (int i = 0; i < 50; ++i) gp.PredictMu(eig::Col(Q, i));   // warm-up
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < M; ++i) one[i] = gp.PredictMu(eig::Col(Q, i));
        const double s1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::vector<std::thread> workers;
        const auto t1 = std::chrono::steady_clock::now();
        for (int t = 0; t < T; ++t)
            workers.emplace_back([&, t]() {
                for (int i = 0; i < M; ++i) many[t * M + i] = gp.PredictMu(eig::Col(Q, i));
            });
        for (auto& w : workers) w.join();
        const double s8 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        bool same = true;
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < M; ++i) same = same && (many[t * M + i] == one[i]);
        EXPECT(same);
        const double ratio = (T * M / s8) / (M / s1);
        std::cout << "concurrent PredictMu: 1 thread " << 1e6 * s1 / M << " us per call, " << T << " threads " << 1e6 * s8 / (T * M)
                  << " us per call: throughput x" << ratio << std::endl;
        // measured on MI355X (round 5): one thread 19.6 us per call (round 4, through the context's lock and two copies: ~70 us),
        // eight threads 6.8 us per call in aggregate = x2.9 of one thread of THIS build, x10 of round 4's; what remains is the HIP
        // runtime's own serialisation of launches.  The bound leaves room for a loaded host.
        EXPECT(ratio >= 2.0);
    }
    // ---- GP MAP: 1-D BO reaches the known optimum of 1 - 1.5 x sin(13 x) ----
    // EI with a zero-mean GP can stall in the local optimum x = 0.378 (f = 1.555) for some start sets -- the model is the
    // reference's; with few points the MAP length scale is long and the model over-confident) -- so the known-answer check
    // runs 40 iterations over several seeds: most runs must reach the global optimum, none may end below f(0) = 1.
    {
        int reached = 0;
        for (int seed = 1; seed <= 6; ++seed)
        {
            utils::SetRandomSeed(seed);
            MatrixXd X(1, 0);
            VectorXd y(0);
            std::shared_ptr<GaussianProcessRegressor> gp;
            for (int it = 0; it < 40; ++it)
            {
                const VectorXd x = (it == 0) ? utils::GenerateRandomVector(1) : acquisition_func::FindNextPoint(*gp);
                X                = eig::AppendCol(X, x);
                VectorXd yn(y.size() + 1);
                for (long i = 0; i < y.size(); ++i) yn(i) = y(i);
                yn(y.size()) = 1.0 - 1.5 * x(0) * std::sin(13.0 * x(0));
                y            = yn;
                gp           = std::make_shared<GaussianProcessRegressor>(X, y);
            }
            const VectorXd xb = gp->PredictMaximumPointFromData();
            const double   mb = gp->PredictMu(xb);
            std::cout << "1-D BO seed " << seed << ": x_max " << xb(0) << "  mu " << mb << "  (true 0.852733 / 2.273928)" << std::endl;
            if (std::abs(xb(0) - 0.852733) < 2e-2 && std::abs(mb - 2.273928) < 2e-2) ++reached;
            EXPECT(mb > 0.99);   // never worse than the boundary value f(0) = 1
        }
        EXPECT(reached >= 4);   // 40 iterations: measured 6 of 6 (7 of 8 over seeds 1..8); 20 iterations reach it for ~40 % of the seeds
    }
    // ---- data manager: merge semantics of src/preference-data-manager.cpp ----
    {
        PreferenceDataManager dm;
        VectorXd a(2), b(2), c(2);
        a(0) = 0.1; a(1) = 0.1; b(0) = 0.5; b(1) = 0.5; c(0) = 0.9; c(1) = 0.9;
        dm.AddNewPoints(a, {b, c});
        EXPECT(dm.GetNumDataPoints() == 3 && dm.GetD().size() == 1 && dm.GetD()[0][0] == 0);
        VectorXd e(2), f(2);
        e(0) = 0.3; e(1) = 0.7; f(0) = 0.5 + 1e-6; f(1) = 0.5;   // f merges with b
        dm.AddNewPoints(e, {a, f});                               // a is an exact duplicate of point 0
        EXPECT(dm.GetNumDataPoints() == 4);
        for (const Preference& p : dm.GetD())
            for (unsigned idx : p) EXPECT(idx < 4u);
        EXPECT((dm.GetLastSelectedDataPoint() - e).norm() < 1e-12);
    }
    // ---- slider enlargement: closed form keeps the ends inside the box, on the same line, not shorter ----
    {
        VectorXd p(3), q(3);
        p(0) = 0.45; p(1) = 0.5; p(2) = 0.9; q(0) = 0.55; q(1) = 0.4; q(2) = 0.8;
        Slider s(p, q, true);
        const double l0 = (p - q).norm(), l1 = (s.end_0 - s.end_1).norm();
        EXPECT(l1 >= l0 - 1e-12);
        const VectorXd d0 = (1.0 / l0) * (p - q), d1 = (1.0 / l1) * (s.end_0 - s.end_1);
        EXPECT((d0 - d1).norm() < 1e-9);
        EXPECT((s.original_end_0 - p).norm() == 0.0);
        EXPECT(std::abs(l1 - std::max(1.25 * l0, 0.25)) < 1e-9);   // interior segment: scale 1.25, then the minimum length 0.25
    }
    // ---- preference regressor + full sequential line search loop: residual shrinks on the bump objective; with the
    // constructor's default (use_map_hyperparams = true, sequential-line-search.hpp:37: the reference demo's setting) and with
    // fixed hyper-parameters ----
    for (const bool use_map : {true, false})
    {
        const int D = 4;
        utils::SetRandomSeed(7);
        SequentialLineSearchOptimizer opt = use_map ? SequentialLineSearchOptimizer(D) : SequentialLineSearchOptimizer(D, true, false, KernelType::ArdMatern52Kernel);
        opt.SetHyperparams(0.5, 0.5, 0.001, 0.1, 0.01);
        auto objective = [](const VectorXd& x) {
            double qd = 0.0;
            for (long i = 0; i < x.size(); ++i) qd += (x(i) - 0.4) * (x(i) - 0.4);
            return std::exp(-qd);
        };
        double first = -1.0, last = -1.0;
        for (int it = 0; it < 8; ++it)
        {
            double bt = 0.0, bv = -1.0;
            for (int k = 0; k <= 200; ++k)
            {
                const double v = objective(opt.CalcPointFromSliderPosition(k / 200.0));
                if (v > bv) { bv = v; bt = k / 200.0; }
            }
            opt.SubmitFeedbackData(bt);
            const double v = objective(opt.GetMaximizer());
            if (it == 0) first = v;
            last = v;
            EXPECT(opt.GetPreferenceValueStdev(opt.GetMaximizer()) >= 0.0);
            EXPECT(opt.GetAcquisitionFuncValue(opt.GetSliderEnds().second) >= 0.0);
        }
        std::cout << "SLS D=4 (use_map_hyperparams " << use_map << "): objective at maximiser " << first << " -> " << last << std::endl;
        EXPECT(last >= first - 1e-9);
        EXPECT(last > 0.9);
    }
    // ---- preference regressor with MAP hyper-parameters + FindNextPoints (Schonlau batch) ----
    {
        const int      D = 2;
        const MatrixXd X = RandomPoints(D, 9);
        std::vector<Preference> prefs{Preference(0, 1, 2), Preference(3, 4, 5), Preference(6, 7, 8), Preference(0, 3), Preference(0, 6)};
        PreferenceRegressor fixed(X, prefs, false), map(X, prefs, true);
        EXPECT((fixed.FindArgMax() - eig::Col(X, 0)).norm() == 0.0);
        EXPECT(map.GetMapObjectiveValue() >= fixed.GetMapObjectiveValue() - 50.0);   // different objectives; both finite
        EXPECT(std::isfinite(map.GetMapObjectiveValue()) && map.GetNoiseHyperparam() > 0.0);
        const VectorXd b  = fixed.m_K_llt.solve(fixed.GetSmallY());
        double         mu = 0.0;
        const VectorXd x  = utils::GenerateRandomVector(D);
        const VectorXd k  = CalcSmallK(x, X, fixed.GetKernelHyperparams(), fixed.GetKernel());
        for (int i = 0; i < 9; ++i) mu += k(i) * b(i);
        EXPECT(std::abs(mu - fixed.PredictMu(x)) < 1e-9);   // PredictMu = k^T K_llt.solve(y) (src/preference-regressor.cpp:293-297)
        // DampData round trip (X.csv / D.csv) -> identical regressor
        fixed.DampData("/tmp", "slshost_");
        const MatrixXd X2 = utils::ImportMatrixFromCsv("/tmp/slshost_X.csv");
        const auto     D2 = utils::ImportPreferencesFromCsv("/tmp/slshost_D.csv");
        EXPECT(X2.rows() == X.rows() && X2.cols() == X.cols() && D2.size() == prefs.size());
        std::vector<Preference> prefs2;
        for (const auto& p : D2) prefs2.push_back(Preference(p));
        PreferenceRegressor reloaded(X2, prefs2, false);
        EXPECT(std::abs(reloaded.PredictMu(x) - fixed.PredictMu(x)) < 1e-12);
        const auto pts = acquisition_func::FindNextPoints(fixed, 3, 32, 20);
        EXPECT(pts.size() == 3);
        EXPECT((pts[0] - pts[1]).norm() > 1e-3 && (pts[1] - pts[2]).norm() > 1e-3);   // variance update pushes the batch apart
    }
    // ---- preferential Bayesian optimisation (pairwise comparison) on the bump objective ----
    {
        const int D = 3;
        utils::SetRandomSeed(11);
        PreferentialBayesianOptimizer pbo(D);   // the default: joint MAP estimation of the hyper-parameters
        pbo.SetHyperparams(0.5, 0.5, 0.001, 0.1, 0.01);
        auto objective = [](const VectorXd& x) {
            double qd = 0.0;
            for (long i = 0; i < x.size(); ++i) qd += (x(i) - 0.4) * (x(i) - 0.4);
            return std::exp(-qd);
        };
        double first = -1.0, last = -1.0;
        for (int it = 0; it < 10; ++it)
        {
            const auto& opts = pbo.GetCurrentOptions();
            EXPECT(opts.size() == 2);
            const int choice = objective(opts[0]) >= objective(opts[1]) ? 0 : 1;
            pbo.SubmitFeedbackData(choice);
            pbo.DetermineNextQuery(64, 20);
            const double v = objective(pbo.GetMaximizer());
            if (it == 0) first = v;
            last = v;
            for (const VectorXd& o : pbo.GetCurrentOptions())
                for (long d = 0; d < o.size(); ++d) EXPECT(o(d) >= 0.0 && o(d) <= 1.0);
        }
        std::cout << "PBO D=3: objective at maximiser " << first << " -> " << last << std::endl;
        EXPECT(last >= first - 1e-9);
        EXPECT(pbo.GetRawDataPoints().cols() >= 10);
    }
    std::cout << (g_fail ? "HOST TESTS FAILED" : "HOST TESTS PASSED") << std::endl;
    return g_fail ? 1 : 0;
}
