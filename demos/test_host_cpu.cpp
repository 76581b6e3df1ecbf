// Host-logic checks that need no GPU (run by tests/test_host_logic_cpu.py): data manager merge semantics, slider
// enlargement, BTL helpers, the bounded L-BFGS driver, CSV round trip, kernel scalar forms, error behaviour without a device.
#include <fstream>
#include <cmath>
#include <iostream>
#include <sequential-line-search/gaussian-process-regressor.hpp>
#include <sequential-line-search/preference-data-manager.hpp>
#include <sequential-line-search/slider.hpp>
#include <sequential-line-search/utils.hpp>
#include <stdexcept>

#include "device.hpp"

using namespace sequential_line_search;
using Eigen::MatrixXd;
using Eigen::VectorXd;

static int g_fail = 0;
#define EXPECT(cond)                                                                  \
    do {                                                                              \
        if (!(cond)) { std::cout << "FAIL " << __LINE__ << ": " #cond << std::endl; ++g_fail; } \
    } while (0)

static VectorXd V(std::initializer_list<double> l) { return VectorXd(l); }

int main(int argc, char** argv)
{
    const bool expect_no_gpu = argc > 1 && std::string(argv[1]) == "--no-gpu";
    // ---- PreferenceDataManager (reference: src/preference-data-manager.cpp) ----
    {
        PreferenceDataManager dm;
        dm.AddNewPoints(V({0.1, 0.1}), {V({0.5, 0.5}), V({0.9, 0.9})});
        EXPECT(dm.GetNumDataPoints() == 3);
        EXPECT(dm.GetD().size() == 1 && dm.GetD()[0].size() == 3 && dm.GetD()[0][0] == 0 && dm.GetD()[0][2] == 2);
        // second observation: the chosen point, the previous best (exact duplicate of point 0) and a near-duplicate of point 1
        dm.AddNewPoints(V({0.3, 0.7}), {V({0.1, 0.1}), V({0.5 + 5e-5, 0.5})});
        // 6 points, two merges -> 4; merged points move to the end as midpoints
        EXPECT(dm.GetNumDataPoints() == 4);
        const MatrixXd& X = dm.GetX();
        EXPECT(std::abs(X(0, 0) - 0.9) < 1e-15);                       // untouched points keep their order: c, e, then merged
        EXPECT(std::abs(X(0, 1) - 0.3) < 1e-15 && std::abs(X(1, 1) - 0.7) < 1e-15);
        EXPECT(std::abs(X(0, 2) - 0.1) < 1e-15);                       // a merged with its duplicate
        EXPECT(std::abs(X(0, 3) - (0.5 + 2.5e-5)) < 1e-12);            // b merged with b + 5e-5 -> midpoint
        for (const Preference& p : dm.GetD())
            for (unsigned idx : p) EXPECT(idx < 4u);
        EXPECT(dm.GetD()[1][0] == 1 && dm.GetD()[1][1] == 2 && dm.GetD()[1][2] == 3);
        EXPECT(dm.GetD()[0][0] == 2 && dm.GetD()[0][1] == 3 && dm.GetD()[0][2] == 0);
        EXPECT((dm.GetLastSelectedDataPoint() - V({0.3, 0.7})).norm() < 1e-15);
        // no merging requested
        PreferenceDataManager dm2;
        dm2.AddNewPoints(V({0.1}), {V({0.2})});
        dm2.AddNewPoints(V({0.1}), {V({0.2})}, false);
        EXPECT(dm2.GetNumDataPoints() == 4);
    }
    // ---- Slider (reference: src/slider.cpp) ----
    {
        Slider plain(V({0.2, 0.2}), V({0.4, 0.6}), false);
        EXPECT((plain.GetValue(0.0) - V({0.2, 0.2})).norm() == 0.0 && (plain.GetValue(1.0) - V({0.4, 0.6})).norm() == 0.0);
        EXPECT((plain.GetValue(0.5) - V({0.3, 0.4})).norm() < 1e-15);
        // interior: stretched by 1.25 about the centre
        Slider s(V({0.3, 0.3}), V({0.7, 0.7}), true);
        EXPECT((s.end_0 - V({0.25, 0.25})).norm() < 1e-12 && (s.end_1 - V({0.75, 0.75})).norm() < 1e-12);
        EXPECT((s.original_end_0 - V({0.3, 0.3})).norm() == 0.0);
        // against the wall: end_1 side is clipped by the box, end_0 side gets the full factor
        Slider w(V({0.5, 0.5}), V({0.98, 0.5}), true);
        EXPECT(w.end_1(0) <= 1.0 && w.end_1(0) > 0.98 && std::abs(w.end_1(1) - 0.5) < 1e-12);
        EXPECT(std::abs(w.end_0(0) - (0.74 - 1.25 * 0.24)) < 1e-12);
        // shorter than the minimum length 0.25 -> extended to it
        Slider tiny(V({0.50, 0.5}), V({0.52, 0.5}), true);
        EXPECT(std::abs((tiny.end_0 - tiny.end_1).norm() - 0.25) < 1e-9);
    }
    // ---- BTL (reference: include/sequential-line-search/utils.hpp:25-52) ----
    {
        const VectorXd f = V({0.02, -0.01, 0.005});
        const double   s = 0.01;
        const double   p = utils::CalcBtl(f, s);
        EXPECT(std::abs(p - std::exp(2.0) / (std::exp(2.0) + std::exp(-1.0) + std::exp(0.5))) < 1e-15);
        const VectorXd d = utils::CalcBtlDerivative(f, s);
        for (int i = 0; i < 3; ++i)
        {
            VectorXd fp = f, fm = f;
            fp(i) += 1e-7; fm(i) -= 1e-7;
            EXPECT(std::abs((utils::CalcBtl(fp, s) - utils::CalcBtl(fm, s)) / 2e-7 - d(i)) < 1e-5);
        }
        EXPECT(std::abs(d(0) + d(1) + d(2)) < 1e-10);   // the probabilities are shift invariant
    }
    // ---- bounded L-BFGS maximiser used by the MAP drivers ----
    {
        // concave quadratic with the optimum outside the box in one coordinate
        const std::vector<double> c{0.3, 1.7, -0.4};
        auto f = [&](const std::vector<double>& x, std::vector<double>* g) {
            double v = 0.0;
            if (g) g->resize(3);
            for (int i = 0; i < 3; ++i)
            {
                v -= (i + 1) * (x[i] - c[i]) * (x[i] - c[i]);
                if (g) (*g)[i] = -2.0 * (i + 1) * (x[i] - c[i]);
            }
            return v;
        };
        double                    best = 0.0;
        const std::vector<double> x = optim::MaximizeBounded(f, {0.9, 0.1, 0.9}, {0, 0, 0}, {1, 1, 1}, 60, &best);
        EXPECT(std::abs(x[0] - 0.3) < 1e-7 && std::abs(x[1] - 1.0) < 1e-12 && std::abs(x[2] - 0.0) < 1e-12);
        EXPECT(std::abs(best - (-2.0 * 0.49 - 3.0 * 0.16)) < 1e-9);
        // Rosenbrock (as a maximisation), budgeted
        auto rosen = [](const std::vector<double>& x, std::vector<double>* g) {
            const double a = 1.0 - x[0], b = x[1] - x[0] * x[0];
            if (g) { g->resize(2); (*g)[0] = -(-2.0 * a - 400.0 * x[0] * b); (*g)[1] = -(200.0 * b); }
            return -(a * a + 100.0 * b * b);
        };
        const std::vector<double> xr = optim::MaximizeBounded(rosen, {-1.2, 1.0}, {-2, -2}, {2, 2}, 300);
        EXPECT(std::abs(xr[0] - 1.0) < 1e-4 && std::abs(xr[1] - 1.0) < 1e-4);
    }
    // ---- DIRECT (host/direct.cpp): the global phase of the reference's default maximiser branch ----
    {
        int  calls = 0, points = 0, max_batch = 0;
        auto counted = [&](auto fn) {
            return [&, fn](const std::vector<std::vector<double>>& xs, std::vector<double>& v) {
                ++calls; points += static_cast<int>(xs.size()); max_batch = std::max(max_batch, static_cast<int>(xs.size()));
                v.resize(xs.size());
                for (size_t k = 0; k < xs.size(); ++k) v[k] = fn(xs[k]);
            };
        };
        // 1-D demo objective 1 - 1.5 x sin(13 x): global maximum 2.273928 at 0.852733, local one 1.555 at 0.378
        double bv = 0.0; int used = 0;
        auto x1 = optim::DirectMaximize(counted([](const std::vector<double>& x) { return 1.0 - 1.5 * x[0] * std::sin(13.0 * x[0]); }),
                                        {0.0}, {1.0}, 60, &bv, &used);
        EXPECT(std::abs(x1[0] - 0.852733) < 5e-3 && std::abs(bv - 2.273928) < 1e-3);
        EXPECT(used <= 60 && used == points && calls < used);   // the budget is a cap; iterations are batched
        // 2-D Branin (negated), rescaled box: three global minima with value 0.397887
        calls = points = max_batch = 0;
        auto branin = [](const std::vector<double>& x) {
            const double a = 1.0, b = 5.1 / (4 * M_PI * M_PI), c = 5.0 / M_PI, r = 6.0, s = 10.0, t = 1.0 / (8 * M_PI);
            const double u = x[1] - b * x[0] * x[0] + c * x[0] - r;
            return -(a * u * u + s * (1 - t) * std::cos(x[0]) + s);
        };
        auto x2 = optim::DirectMaximize(counted(branin), {-5.0, 0.0}, {10.0, 15.0}, 400, &bv, &used);
        EXPECT(std::abs(-bv - 0.397887) < 2e-3 && used <= 400);
        EXPECT(max_batch >= 4);   // several potentially optimal rectangles per iteration share one batch
        EXPECT(x2[0] >= -5.0 && x2[0] <= 10.0 && x2[1] >= 0.0 && x2[1] <= 15.0);
        // 6-D: optimum off-centre, anisotropic; 50 D evaluations (the facade's heuristic) get close, never leave the box
        auto quad = [](const std::vector<double>& x) {
            double q = 0.0;
            for (size_t i = 0; i < x.size(); ++i) q += (1.0 + i) * (x[i] - 0.3) * (x[i] - 0.3);
            return std::exp(-q);
        };
        auto x6 = optim::DirectMaximize(counted(quad), std::vector<double>(6, 0.0), std::vector<double>(6, 1.0), 300, &bv, &used);
        EXPECT(bv > 0.9 && used <= 300);
        for (double v : x6) EXPECT(v >= 0.0 && v <= 1.0);
        // a budget smaller than one full first iteration: the centre is returned, nothing is exceeded
        auto xc = optim::DirectMaximize(counted(quad), std::vector<double>(6, 0.0), std::vector<double>(6, 1.0), 5, &bv, &used);
        EXPECT(used == 1 && xc[0] == 0.5);
    }
    // ---- CSV round trip ----
    {
        utils::SetRandomSeed(5);
        MatrixXd X(3, 4);
        for (int j = 0; j < 4; ++j) eig::SetCol(X, j, utils::GenerateRandomVector(3));
        utils::ExportMatrixToCsv("/tmp/sls_cpu_X.csv", X);
        const MatrixXd Y = utils::ImportMatrixFromCsv("/tmp/sls_cpu_X.csv");
        EXPECT(Y.rows() == 3 && Y.cols() == 4);
        double err = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) err = std::max(err, std::abs(X(i, j) - Y(i, j)));
        EXPECT(err == 0.0);
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 3; ++i) EXPECT(X(i, j) >= 0.0 && X(i, j) <= 1.0);
        utils::SetRandomSeed(5);
        EXPECT((utils::GenerateRandomVector(3) - eig::Col(X, 0)).norm() == 0.0);   // seeded stream is reproducible
        {   // a ragged file is rejected, not read out of bounds
            std::ofstream f("/tmp/sls_cpu_ragged.csv");
            f << "1,2\n3,4,5\n";
        }
        bool threw = false;
        try { utils::ImportMatrixFromCsv("/tmp/sls_cpu_ragged.csv"); } catch (const std::runtime_error&) { threw = true; }
        EXPECT(threw);
    }
    // ---- kernel scalar forms: derivative consistency, Matern finite at coincident points ----
    {
        const VectorXd xa = V({0.2, 0.7}), xb = V({0.5, 0.4}), th = V({0.6, 0.3, 0.8});
        for (auto k : {kernels::ArdSquaredExp, kernels::ArdMatern52})
        {
            auto dk = (k == kernels::ArdSquaredExp) ? kernels::ArdSquaredExpFirstArgDerivative : kernels::ArdMatern52FirstArgDerivative;
            auto dt = (k == kernels::ArdSquaredExp) ? kernels::ArdSquaredExpThetaDerivative : kernels::ArdMatern52ThetaDerivative;
            const VectorXd g = dk(xa, xb, th), gt = dt(xa, xb, th);
            for (int i = 0; i < 2; ++i)
            {
                VectorXd p = xa, m = xa;
                p(i) += 1e-6; m(i) -= 1e-6;
                EXPECT(std::abs((k(p, xb, th) - k(m, xb, th)) / 2e-6 - g(i)) < 1e-7);
            }
            for (int i = 0; i < 3; ++i)
            {
                VectorXd p = th, m = th;
                p(i) += 1e-6; m(i) -= 1e-6;
                EXPECT(std::abs((k(xa, xb, p) - k(xa, xb, m)) / 2e-6 - gt(i)) < 1e-7);
            }
            EXPECT(std::abs(k(xa, xa, th) - 0.6) < 1e-15 && dk(xa, xa, th).norm() == 0.0);
            EXPECT(kernels::TypeOf(k) == (k == kernels::ArdSquaredExp ? KernelType::ArdSquaredExponentialKernel : KernelType::ArdMatern52Kernel));
        }
        bool threw = false;
        try { kernels::TypeOf(static_cast<Kernel>([](const VectorXd&, const VectorXd&, const VectorXd&) { return 0.0; })); }
        catch (const std::invalid_argument&) { threw = true; }
        EXPECT(threw);   // foreign kernel callbacks are rejected, not evaluated on the host
    }
    // ---- without a GPU every device-backed call must fail loudly (no silent CPU path) ----
    if (expect_no_gpu)
    {
        bool threw = false;
        try
        {
            MatrixXd X(1, 2);
            X(0, 0) = 0.1; X(0, 1) = 0.9;
            GaussianProcessRegressor gp(X, V({1.0, 2.0}), V({0.5, 0.3}), 0.01);
        }
        catch (const std::runtime_error& e)
        {
            threw = std::string(e.what()).find("no CPU fallback") != std::string::npos || std::string(e.what()).find("no HIP device") != std::string::npos;
        }
        EXPECT(threw);
        GaussianProcessRegressor empty(MatrixXd(0, 0), VectorXd(0));   // the inert object needs no device
        EXPECT(empty.GetSmallY().rows() == 0);
    }
    std::cout << (g_fail ? "HOST CPU TESTS FAILED" : "HOST CPU TESTS PASSED") << std::endl;
    return g_fail ? 1 : 0;
}
