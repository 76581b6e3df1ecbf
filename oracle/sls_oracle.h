/*
 * sls_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the Gaussian-process regression + acquisition
 * maximisation hot path of yuki-koyama/sequential-line-search.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call
 * this; the shipped library (sequential-line-search_amd/csrc) never does.
 *
 * PARITY UNPINNED: the reference cannot be built here (Eigen, mathtoolbox,
 * NLopt, parallel-util are absent; every external/ submodule is empty) and it
 * holds no golden vectors.  The scalar kernels / EI / priors follow the
 * published mathtoolbox definitions; the oracle is
 * instead pinned against independent implementations (mpmath, scipy, sklearn)
 * through the fixtures in tests/golden/ (tests/golden/make_fixtures.py).
 *
 * Conventions (same as the reference): all matrices column-major double;
 * X is D x N, one data point per column (reference: Eigen::MatrixXd);
 * theta = (a, l_1..l_D) (kernel_hyperparams), b = noise level.
 *
 * Reference line citations are relative to /root/reference.
 */
#ifndef SLS_ORACLE_H
#define SLS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { SLSO_KERNEL_ARD_SE = 0, SLSO_KERNEL_ARD_MATERN52 = 1 };        /* kernel-type.hpp:8-12 */
enum { SLSO_ACQ_EI = 0, SLSO_ACQ_UCB = 1 };                           /* acquisition-function.hpp:11-15 */
enum { SLSO_REG_GPR = 0, SLSO_REG_PREF = 1 };

/* ---- mathtoolbox kernel functions (regressor.cpp:14-16,21-23; Appendix A) ---- */
double slso_kernel(int kernel, const double* xa, const double* xb, const double* theta, int D);
void   slso_kernel_theta_derivative(int kernel, const double* xa, const double* xb, const double* theta, int D, double* out /*D+1*/);
void   slso_kernel_first_arg_derivative(int kernel, const double* xa, const double* xb, const double* theta, int D, double* out /*D*/);

/* ---- regressor.cpp free functions ---- */
void slso_calc_small_k(int kernel, const double* x, const double* X, int D, int N, const double* theta, double* k /*N*/);               /* :45-59 */
void slso_calc_large_kf(int kernel, const double* X, int D, int N, const double* theta, double* K /*NxN*/);                              /* :73-89 */
void slso_calc_large_ky(int kernel, const double* X, int D, int N, const double* theta, double b, double* K /*NxN*/);                    /* :61-71 */
void slso_calc_small_k_small_x_derivative(int kernel, const double* x, const double* X, int D, int N, const double* theta, double* J /*DxN*/); /* :91-108 */
void slso_calc_large_ky_theta_derivative(int kernel, const double* X, int D, int N, const double* theta, double* T /*(D+1) x N x N*/);  /* :110-134 */

/* ---- dense linear algebra (Eigen stand-ins) ---- */
int    slso_cholesky(double* A, int n);                               /* Eigen::LLT: lower factor in place, upper zeroed; 0 ok */
void   slso_chol_solve(const double* L, int n, double* B, int nrhs);  /* LLT::solve in place */
void   slso_trsm_lower(const double* L, int n, double* B, int nrhs);  /* B <- L^-1 B */
void   slso_trsm_lower_t(const double* L, int n, double* B, int nrhs);/* B <- L^-T B */
int    slso_lu_inverse(const double* A, int n, double* Ainv);         /* MatrixXd::inverse() = PartialPivLU */
void   slso_spd_inverse_from_chol(const double* L, int n, double* Ainv);
double slso_logdet_from_chol(const double* L, int n);                 /* mathtoolbox CalcLogDetOfSymmetricPositiveDefiniteMatrix */

/* ---- mathtoolbox scalar functions ---- */
double slso_log_lognormal(double x, double mu, double sigma2);             /* GetLogOfLogNormalDist */
double slso_log_lognormal_derivative(double x, double mu, double sigma2);  /* GetLogOfLogNormalDistDerivative */
double slso_norm_pdf(double u);
double slso_norm_cdf(double u);
double slso_btl(const double* f, int n, double scale);                         /* utils.hpp:25-29 */
void   slso_btl_derivative(const double* f, int n, double scale, double* d);   /* utils.hpp:31-52 */

/* ---- regressor object: GaussianProcessRegressor (fixed hyper-parameters ctor,
 *      gaussian-process-regressor.cpp:214-232) or the predictive state of
 *      PreferenceRegressor (preference-regressor.cpp:289-330) ---- */
typedef struct slso_regressor {
    int     reg_type, kernel, D, N;
    double* X;      /* D x N */
    double* y;      /* N     */
    double* theta;  /* 1 + D */
    double  b;
    double* K;      /* K_y (m_K_y / m_K)        */
    double* Kinv;   /* GPR: m_K_y_inv (LU inverse, as the reference); PREF: unused (NULL) */
    double* L;      /* Cholesky factor (PREF: m_K_llt; GPR: used by the hoisted mode only) */
    /* hoisted caches (not in the reference; identical results up to rounding) */
    double* alpha;  /* K^-1 y */
    double* KinvC;  /* Cholesky-based K^-1, used by the hoisted batched evaluation */
    int     best_index;
    double  mu_best;
} slso_regressor;

slso_regressor* slso_regressor_create(int reg_type, int kernel, const double* X, int D, int N, const double* y,
                                      const double* theta, double b);
void slso_regressor_free(slso_regressor* r);
int    slso_regressor_best_index(const slso_regressor* r);   /* hoisted arg max_i (y_i - b alpha_i): what regressor.cpp:29-43 returns */
double slso_regressor_mu_best(const slso_regressor* r);

/* as-written predictive quantities (O(N^2) .. O(D N^2) per call, like the reference) */
double slso_predict_mu(const slso_regressor* r, const double* x);                        /* GPR :234-239 / PREF :293-297 */
double slso_predict_sigma(const slso_regressor* r, const double* x);                     /* :241-255 / :299-313 */
void   slso_predict_mu_derivative(const slso_regressor* r, const double* x, double* g);  /* :257-263 / :315-321 */
void   slso_predict_sigma_derivative(const slso_regressor* r, const double* x, double* g); /* :265-272 / :323-330 */
int    slso_predict_maximum_point_from_data(const slso_regressor* r, double* x_best);    /* regressor.cpp:29-43; returns index */

/* as-written acquisition (acquisition-function.cpp:170-230): recomputes x_best on every call => O(N^3) */
double slso_acq_value_as_written(const slso_regressor* r, const double* x, int acq, double ucb_h);
void   slso_acq_derivative_as_written(const slso_regressor* r, const double* x, int acq, double ucb_h, double* g);

/* hoisted evaluation (alpha, mu+ cached; w = K^-1 k once): value and gradient at M points, Xs is D x M */
void slso_predict_batch(const slso_regressor* r, const double* Xs, int M, double* mu, double* sigma);
void slso_predict_grad_batch(const slso_regressor* r, const double* Xs, int M, double* dmu /*DxM*/, double* dsigma /*DxM*/);
void slso_acq_eval_batch(const slso_regressor* r, const double* Xs, int M, int acq, double ucb_h, double* val,
                         double* grad /*DxM or NULL*/);

/* ---- multi-start maximiser (acquisition-function.cpp:121-153 skeleton) ----
 * S independent bounded L-BFGS runs from explicit starts (D x S), each limited to
 * n_local objective evaluations; returns argmax over the S end points (first max,
 * like Eigen::maxCoeff).  The L-BFGS itself is the build's own algorithm (NLopt is
 * absent; SURVEY.md 7 "maximiser parity"), specified in DESIGN.md 5 and implemented
 * identically by the HIP path.  x_stars (D x S) / y_stars (S) may be NULL. */
typedef struct slso_lbfgs_opts {
    int    history;      /* m, default 6 */
    double c1;           /* Armijo constant, default 1e-4 */
    double shrink;       /* backtracking factor, default 0.5 */
    double gtol;         /* projected-gradient inf-norm stop, default 0 (never) */
    int    max_backtracks; /* default 20 */
    /* NLopt's relative stopping tests on every accepted step (nlopt/src/util/stop.c: relstop); 0 = off.  nloptutil::solve's
     * defaults, which every search of the reference runs with, are 1e-6 for both. */
    double ftol_rel;     /* |f' - f| < ftol_rel (|f'| + |f|) / 2 or f' == f */
    double xtol_rel;     /* every d: |x'_d - x_d| < xtol_rel (|x'_d| + |x_d|) / 2 or x'_d == x_d */
} slso_lbfgs_opts;
void slso_lbfgs_default_opts(slso_lbfgs_opts* o);
int  slso_acq_maximize(const slso_regressor* r, int acq, double ucb_h, const double* starts, int S, int n_local,
                       const slso_lbfgs_opts* opts, double* x_out /*D*/, double* val_out, double* x_stars, double* y_stars,
                       int n_threads);
/* Same run, plus per-start diagnostics (S each, may be NULL): the smallest relative distance of any Armijo test from its
 * threshold and the evaluation at which it occurred -- starts whose margin is at rounding level may legitimately take the
 * other branch on an implementation with a different summation order. */
int  slso_acq_maximize_diag(const slso_regressor* r, int acq, double ucb_h, const double* starts, int S, int n_local,
                            const slso_lbfgs_opts* opts, double* x_out, double* val_out, double* x_stars, double* y_stars,
                            int n_threads, double* armijo_margin, int* armijo_eval);

/* ---- GP marginal likelihood MAP objective (gaussian-process-regressor.cpp:141-193, 66-127) ----
 * x = (a, b, r_1..r_D); returns log p; grad (D+2) may be NULL.
 * as_written=1: explicit LU inverse + dense (D+1) x N x N derivative tensor + per-theta traces.
 * as_written=0: Cholesky + fused W = alpha alpha^T - K^-1 contraction. */
double slso_gp_map_objective(int kernel, const double* X, int D, int N, const double* y, const double* x, double* grad,
                             int as_written);

/* ---- preference-GP MAP objective (preference-regressor.cpp:129-259, 53-115) ----
 * prefs_flat / pref_offsets: CSR of the std::vector<Preference> D (n_prefs tuples).
 * x = (y_1..y_M [, a, b, r_1..r_D] if use_map); returns objective; grad may be NULL. */
typedef struct slso_pref_cfg {
    int    use_map_hyperparams;
    double default_a, default_r, default_b, prior_var, btl_scale;
    int    noiseless;  /* SEQUENTIAL_LINE_SEARCH_USE_NOISELESS_FORMULATION */
} slso_pref_cfg;
double slso_pref_objective(int kernel, const double* X, int D, int M, const unsigned* prefs_flat, const int* pref_offsets,
                           int n_prefs, const double* x, const slso_pref_cfg* cfg, double* grad);

/* ---- synthetic inputs: SplitMix64 -> uniform / Box-Muller normal ---- */
void slso_fill_uniform(double* out, long n, unsigned long long seed);
void slso_fill_normal(double* out, long n, unsigned long long seed);

#ifdef __cplusplus
}
#endif
#endif
