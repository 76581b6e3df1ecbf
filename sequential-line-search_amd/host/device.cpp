#include "device.hpp"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>

namespace sequential_line_search
{
    namespace device
    {
        void Check(int rc, const char* what)
        {
            if (rc != 0) throw std::runtime_error(std::string(what) + ": " + sls_last_error());
        }

        sls_ctx* Context()
        {
            static sls_ctx*   ctx = nullptr;
            static std::mutex mtx;
            std::lock_guard<std::mutex> lock(mtx);
            if (!ctx)
            {
                const char* env = std::getenv("SLS_DEVICE");
                Check(sls_ctx_create(env ? std::atoi(env) : 0, &ctx), "sls_ctx_create");
            }
            return ctx;
        }

        namespace
        {
            std::mutex        g_multi_mtx;
            std::vector<int>  g_devices;
            bool              g_devices_set = false;
            std::shared_ptr<MultiRef> g_multi;
            std::vector<int>          g_multi_devices;
            struct ReplicaEntry
            {
                std::shared_ptr<MultiGpHandle> replicas;
                std::shared_ptr<MultiRef>      multi;
                long                           generation = -1;   // sls_gp_generation of the primary the replicas were built from
            };
            std::mutex                       g_replica_mtx;
            // never destroyed: static destructors run after the HIP runtime may have shut down, and an entry of a leaked or static
            // regressor would then destroy device handles on a dead runtime (entries of destroyed regressors are erased by ~GpHandle)
            std::map<sls_gp*, ReplicaEntry>& g_replicas = *new std::map<sls_gp*, ReplicaEntry>();
            long                             g_replica_builds = 0;

            void LoadDevicesFromEnv()
            {
                if (g_devices_set) return;
                g_devices_set = true;
                g_devices.clear();
                if (const char* env = std::getenv("SLS_DEVICES"))
                {
                    std::string tok;
                    for (const char* p = env;; ++p)
                    {
                        if (*p == ',' || *p == '\0')
                        {
                            if (!tok.empty()) g_devices.push_back(std::atoi(tok.c_str()));
                            tok.clear();
                            if (*p == '\0') break;
                        }
                        else tok.push_back(*p);
                    }
                }
                if (g_devices.empty())
                {
                    const char* env = std::getenv("SLS_DEVICE");
                    g_devices.push_back(env ? std::atoi(env) : 0);
                }
            }
        } // namespace

        void SetDevices(const std::vector<int>& devices)
        {
            std::lock_guard<std::mutex> lock(g_multi_mtx);
            if (devices.empty()) throw std::invalid_argument("device::SetDevices: empty device list");
            g_devices     = devices;
            g_devices_set = true;
        }

        std::vector<int> Devices()
        {
            std::lock_guard<std::mutex> lock(g_multi_mtx);
            LoadDevicesFromEnv();
            return g_devices;
        }

        MultiRef::~MultiRef() { sls_multi_destroy(m); }

        std::shared_ptr<MultiRef> Multi()
        {
            std::lock_guard<std::mutex> lock(g_multi_mtx);
            LoadDevicesFromEnv();
            if (g_devices.size() < 2) return nullptr;
            if (g_multi && g_multi_devices != g_devices) g_multi.reset();   // the old one lives on while handles still use it
            if (!g_multi)
            {
                sls_multi* m = nullptr;
                Check(sls_multi_create(g_devices.data(), static_cast<int>(g_devices.size()), &m), "sls_multi_create");
                g_multi         = std::make_shared<MultiRef>(m);
                g_multi_devices = g_devices;
            }
            return g_multi;
        }

        MultiGpHandle::MultiGpHandle(std::shared_ptr<MultiRef> multi_, sls_gp* primary) : multi(std::move(multi_))
        {
            Check(sls_multi_gp_create_from(multi->m, primary, &h), "sls_multi_gp_create_from");
        }
        MultiGpHandle::~MultiGpHandle() { sls_multi_gp_destroy(h); }

        std::shared_ptr<MultiGpHandle> ReplicasFor(sls_gp* primary, long n_points)
        {
            std::shared_ptr<MultiRef> multi = Multi();
            if (!multi || !primary) return nullptr;
            std::lock_guard<std::mutex> lock(g_replica_mtx);
            (void)n_points;
            // the primary's predictor generation changes with every fit, in-place refit (sls_gp_refit_dev through GetDeviceHandle()),
            // appended point and sigma-mode switch: replicas built from an earlier state are never reused
            long generation = 0;
            Check(sls_gp_generation(primary, &generation), "sls_gp_generation");
            ReplicaEntry& e = g_replicas[primary];
            if (!e.replicas || e.multi != multi || e.generation != generation)
            {
                e.replicas.reset();
                e.replicas   = std::make_shared<MultiGpHandle>(multi, primary);
                e.multi      = multi;
                e.generation = generation;
                ++g_replica_builds;
            }
            return e.replicas;
        }
        void ForgetReplicas(sls_gp* primary)
        {
            std::lock_guard<std::mutex> lock(g_replica_mtx);
            g_replicas.erase(primary);
        }
        long ReplicaBuilds()
        {
            std::lock_guard<std::mutex> lock(g_replica_mtx);
            return g_replica_builds;
        }

        GpHandle::GpHandle(const Eigen::MatrixXd& X, const Eigen::VectorXd& y, const Eigen::VectorXd& theta, double b, int kernel)
        {
            Check(sls_gp_create(Context(), X.data(), static_cast<int>(X.rows()), static_cast<int>(X.cols()), y.data(), theta.data(), b,
                                kernel, &h),
                  "sls_gp_create");
        }
        GpHandle::~GpHandle()
        {
            ForgetReplicas(h);   // the replicas borrow this handle as one of their shards: they go first
            sls_gp_destroy(h);
        }

        NllHandle::NllHandle(const Eigen::MatrixXd& X, int kernel)
        {
            Check(sls_nll_create(Context(), X.data(), static_cast<int>(X.rows()), static_cast<int>(X.cols()), kernel, &h),
                  "sls_nll_create");
        }
        NllHandle::~NllHandle() { sls_nll_destroy(h); }

        MultiNllHandle::MultiNllHandle(const Eigen::MatrixXd& X, int kernel) : multi(Multi())
        {
            Check(sls_multi_nll_create(multi->m, X.data(), static_cast<int>(X.rows()), static_cast<int>(X.cols()), kernel, &h),
                  "sls_multi_nll_create");
        }
        MultiNllHandle::~MultiNllHandle() { sls_multi_nll_destroy(h); }
    } // namespace device

    namespace optim
    {
        namespace
        {
            std::atomic<double> g_ftol_rel{-1.0}, g_xtol_rel{-1.0};
            void                InitTolerances()
            {
                if (g_ftol_rel.load() >= 0.0) return;
                const char*  e = std::getenv("SLS_LOCAL_SEARCH_TOL");   // read once per process
                const double v = e ? std::max(0.0, std::atof(e)) : 1e-6;
                g_xtol_rel.store(v);
                g_ftol_rel.store(v);
            }
        } // namespace
        void SetSearchTolerances(double f, double x)
        {
            g_xtol_rel.store(std::max(0.0, x));
            g_ftol_rel.store(std::max(0.0, f));
        }
        void SearchTolerances(double* f, double* x)
        {
            InitTolerances();
            if (f) *f = g_ftol_rel.load();
            if (x) *x = g_xtol_rel.load();
        }

        namespace
        {
            std::atomic<double> g_map_ftol_rel{-1.0}, g_map_xtol_rel{-1.0};
            void                InitMapTolerances()
            {
                if (g_map_ftol_rel.load() >= 0.0) return;
                const char*  e = std::getenv("SLS_MAP_FIT_TOL");   // read once per process; off unless set
                const double v = e ? std::max(0.0, std::atof(e)) : 0.0;
                g_map_xtol_rel.store(v);
                g_map_ftol_rel.store(v);
            }
        } // namespace
        void SetMapFitTolerances(double f, double x)
        {
            g_map_xtol_rel.store(std::max(0.0, x));
            g_map_ftol_rel.store(std::max(0.0, f));
        }
        void MapFitTolerances(double* f, double* x)
        {
            InitMapTolerances();
            if (f) *f = g_map_ftol_rel.load();
            if (x) *x = g_map_xtol_rel.load();
        }

        std::vector<double> MaximizeBounded(const Objective& f, std::vector<double> x, const std::vector<double>& lo,
                                            const std::vector<double>& hi, int max_evals, double* best_value, int* evals_used,
                                            double ftol_rel, double xtol_rel)
        {
            bool stalled = false;   // NLopt's relative tests fired on an accepted step
            const size_t n = x.size();
            const int    m = 8;
            auto clampv = [&](std::vector<double>& v) {
                for (size_t i = 0; i < n; ++i) v[i] = std::min(hi[i], std::max(lo[i], v[i]));
            };
            clampv(x);
            std::vector<double> g(n), gt(n), xt(n), d(n), pg(n);
            // minimise phi = -f
            int    evals = 0;
            double fx    = -f(x, &g);
            ++evals;
            for (auto& v : g) v = -v;
            std::vector<std::vector<double>> S, Y;
            std::vector<double>              rho;
            while (evals < max_evals && !stalled)
            {
                double pgmax = 0.0, pgn2 = 0.0;
                for (size_t i = 0; i < n; ++i)
                {
                    double v = g[i];
                    if ((x[i] <= lo[i] && v > 0.0) || (x[i] >= hi[i] && v < 0.0)) v = 0.0;
                    pg[i] = v;
                    pgmax = std::max(pgmax, std::fabs(v));
                    pgn2 += v * v;
                }
                if (!(pgmax > 0.0)) break;
                d = pg;
                std::vector<double> al(S.size());
                for (int h = static_cast<int>(S.size()) - 1; h >= 0; --h)
                {
                    double dot = 0.0;
                    for (size_t i = 0; i < n; ++i) dot += S[h][i] * d[i];
                    al[h] = rho[h] * dot;
                    for (size_t i = 0; i < n; ++i) d[i] -= al[h] * Y[h][i];
                }
                double gamma = 1.0 / std::max(1.0, std::sqrt(pgn2));
                if (!S.empty())
                {
                    double sy = 0.0, yy = 0.0;
                    for (size_t i = 0; i < n; ++i)
                    {
                        sy += S.back()[i] * Y.back()[i];
                        yy += Y.back()[i] * Y.back()[i];
                    }
                    gamma = sy / yy;
                }
                for (auto& v : d) v *= gamma;
                for (size_t h = 0; h < S.size(); ++h)
                {
                    double dot = 0.0;
                    for (size_t i = 0; i < n; ++i) dot += Y[h][i] * d[i];
                    const double beta = rho[h] * dot;
                    for (size_t i = 0; i < n; ++i) d[i] += S[h][i] * (al[h] - beta);
                }
                double gd = 0.0;
                for (size_t i = 0; i < n; ++i)
                {
                    d[i] = (pg[i] == 0.0) ? 0.0 : -d[i];
                    gd += pg[i] * d[i];
                }
                if (!(gd < 0.0))
                {
                    S.clear(); Y.clear(); rho.clear();
                    gamma = 1.0 / std::max(1.0, std::sqrt(pgn2));
                    gd    = 0.0;
                    for (size_t i = 0; i < n; ++i)
                    {
                        d[i] = -gamma * pg[i];
                        gd += pg[i] * d[i];
                    }
                    if (!(gd < 0.0)) break;
                }
                double t        = 1.0;
                bool   accepted = false;
                for (int bt = 0; bt <= 30 && evals < max_evals; ++bt)
                {
                    for (size_t i = 0; i < n; ++i) xt[i] = x[i] + t * d[i];
                    clampv(xt);
                    double ss = 0.0, gs = 0.0;
                    for (size_t i = 0; i < n; ++i)
                    {
                        ss += (xt[i] - x[i]) * (xt[i] - x[i]);
                        gs += g[i] * (xt[i] - x[i]);
                    }
                    if (ss == 0.0) break;
                    const double ft = -f(xt, &gt);
                    ++evals;
                    for (auto& v : gt) v = -v;
                    if (std::isfinite(ft) && ft <= fx + 1e-4 * gs)
                    {
                        std::vector<double> s(n), y(n);
                        double              sy = 0.0, yy = 0.0;
                        for (size_t i = 0; i < n; ++i)
                        {
                            s[i] = xt[i] - x[i];
                            y[i] = gt[i] - g[i];
                            sy += s[i] * y[i];
                            yy += y[i] * y[i];
                        }
                        if (sy > 1e-10 * yy && sy > 0.0)
                        {
                            if (static_cast<int>(S.size()) == m)
                            {
                                S.erase(S.begin()); Y.erase(Y.begin()); rho.erase(rho.begin());
                            }
                            S.push_back(s); Y.push_back(y); rho.push_back(1.0 / sy);
                        }
                        // NLopt's relative stopping tests (nlopt/src/util/stop.c: relstop) on the accepted step
                        if (ftol_rel > 0.0 && (std::fabs(ft - fx) < ftol_rel * 0.5 * (std::fabs(ft) + std::fabs(fx)) || ft == fx)) stalled = true;
                        if (xtol_rel > 0.0)
                        {
                            bool moved = false;
                            for (size_t i = 0; i < n; ++i)
                                if (!(std::fabs(xt[i] - x[i]) < xtol_rel * 0.5 * (std::fabs(xt[i]) + std::fabs(x[i])) || xt[i] == x[i])) moved = true;
                            if (!moved) stalled = true;
                        }
                        x = xt; g = gt; fx = ft;
                        accepted = true;
                        break;
                    }
                    t *= 0.5;
                }
                if (!accepted) break;
            }
            if (best_value) *best_value = -fx;
            if (evals_used) *evals_used = evals;
            return x;
        }
    } // namespace optim
} // namespace sequential_line_search
