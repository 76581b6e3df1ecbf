/*
 * sls_hip.h -- C ABI of the MI355X-native GP regression + acquisition-maximisation
 * hot path (drop-in for the compute behind sequential-line-search's Regressor /
 * GaussianProcessRegressor / PreferenceRegressor / acquisition_func surface).
 *
 * The reference has no FFI layer: its "operator API" is the C++
 * public surface itself.  Each entry point below names the reference function(s)
 * whose arithmetic it replaces (paths relative to the reference tree).  The C++
 * classes in include/sequential-line-search/ and the tests bind exactly these.
 *
 * Conventions
 *   - plain C types only; every matrix is column-major double (Eigen::MatrixXd
 *     layout).  X is D x N: one data point per column, D contiguous doubles.
 *   - theta = (a, l_1..l_D)  (kernel_hyperparams), b = noise level.
 *   - pointers are HOST pointers unless the parameter name ends in _dev.
 *   - return 0 on success, <0 on error; sls_last_error() describes the last
 *     failure of the calling thread.  There is NO CPU fallback: every entry point
 *     runs hand-written gfx950 kernels and fails if no GPU is present.
 *   - a context owns one HIP stream and one (recursive) lock; a context and its
 *     handles may be shared by any number of host threads -- the reference shares one
 *     const regressor across hardware_concurrency worker threads
 *     (src/acquisition-function.cpp:125-144).  Two kinds of calls:
 *       * const evaluations of at most 64 points on a small handle (N <= 512, D <= 128:
 *         sls_gp_predict, sls_gp_predict_grad, sls_acq_eval) do NOT take the context's
 *         lock: each borrows one of the context's evaluation slots (own stream, own
 *         mapped result block) under a SHARED lock on the handle's fitted state and
 *         runs concurrently with the others (SLS_EVAL_SLOTS=0 restores the lock);
 *       * everything else -- fits, refits, sls_gp_append_point, sls_gp_set_sigma_mode,
 *         maximisers, MAP fits, larger evaluations -- holds the context's lock for the
 *         whole call (mutators also take the handle's state lock exclusively) and is
 *         SERIALISED in arrival order.
 *     A handle must outlive every call in flight on it: sls_gp_destroy waits for
 *     the evaluations that hold its state lock, but a call that STARTS after the
 *     destroy is a use after free, as with any C handle.  For independent streams of
 *     large work use one context per thread (or sls_multi, one per GPU).
 */
#ifndef SLS_HIP_H
#define SLS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* include/sequential-line-search/kernel-type.hpp:8-12 */
#define SLS_KERNEL_ARD_SQUARED_EXPONENTIAL 0
#define SLS_KERNEL_ARD_MATERN52 1
/* include/sequential-line-search/acquisition-function.hpp:11-15 */
#define SLS_ACQ_EXPECTED_IMPROVEMENT 0
#define SLS_ACQ_GP_UCB 1

#define SLS_OK 0
#define SLS_ERR_INVALID (-1)
#define SLS_ERR_HIP (-2)
#define SLS_ERR_NOT_SPD (-3)
#define SLS_ERR_NO_DEVICE (-4)
#define SLS_ERR_UNSUPPORTED (-5) /* the problem is outside what this entry point runs on the device; use the general path */

typedef struct sls_ctx sls_ctx;
typedef struct sls_gp sls_gp;

const char* sls_last_error(void);
int sls_version(void);

/* ---- context ------------------------------------------------------------ */
int sls_ctx_create(int device, sls_ctx** out);
/* Handles created from a context (sls_gp, sls_nll, sls_comm) may be destroyed after it: with live handles sls_ctx_destroy only
 * marks the context, the handles keep working, and the last one to go frees it (garbage-collected bindings finalise
 * objects in no particular order). */
int sls_ctx_destroy(sls_ctx* ctx);
/* Run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL restores the own stream. */
int sls_ctx_set_stream(sls_ctx* ctx, void* hip_stream);
int sls_ctx_synchronize(sls_ctx* ctx);
/* Upper bound on candidates evaluated per device pass (workspace = 3 * chunk * N_pad doubles). Default 16384. */
int sls_ctx_set_candidate_chunk(sls_ctx* ctx, int chunk);

/* Device blocks released by handles and calls are cached per device (exact-size reuse; at most SLS_POOL_MB, default 16384,
 * MB): re-creating a regressor of the same shape costs no hipMalloc.  This returns every cached block of `device` to the
 * driver. */
int sls_device_trim_cache(int device);

/* ---- free functions of src/regressor.cpp --------------------------------- */
/* CalcLargeKY (src/regressor.cpp:61-71); b = 0 gives CalcLargeKF (:73-89).  K_out is N x N. */
int sls_gram(sls_ctx* ctx, const double* X, int D, int N, const double* theta, double b, int kernel, double* K_out);
/* CalcSmallK (src/regressor.cpp:45-59) for M query points at once: Ks_out is N x M, column m = k(Xs[:,m], X). */
int sls_gram_cross(sls_ctx* ctx, const double* X, int D, int N, const double* Xs, int M, const double* theta, int kernel,
                   double* Ks_out);

/* ---- dense factorisation (Eigen::LLT / MatrixXd::inverse stand-ins) ------ */
/* Eigen::LLT<MatrixXd>(K) (src/preference-regressor.cpp:162,290): A (N x N, symmetric, lower read) -> lower factor L
 * (upper zeroed).  SLS_ERR_NOT_SPD if a pivot is not positive. */
int sls_potrf(sls_ctx* ctx, double* A, int N);
/* LLT::solve (src/preference-regressor.cpp:165,296,309,320,329): B (N x nrhs) <- (L L^T)^-1 B. */
int sls_potrs(sls_ctx* ctx, const double* L, int N, double* B, int nrhs);
/* K^-1 from its Cholesky factor (replaces m_K_y.inverse(), src/gaussian-process-regressor.cpp:159,211,231). */
int sls_potri(sls_ctx* ctx, const double* L, int N, double* Ainv);

/* ---- GP handle ----------------------------------------------------------- */
/* GaussianProcessRegressor(X, y, kernel_hyperparams, noise, kernel_type) (src/gaussian-process-regressor.cpp:214-232) and the
 * post-MAP state of PreferenceRegressor (src/preference-regressor.cpp:289-290): builds K_y, its Cholesky factor, K_y^-1,
 * alpha = K_y^-1 y and the hoisted PredictMaximumPointFromData (src/regressor.cpp:29-43) on the device. */
int sls_gp_create(sls_ctx* ctx, const double* X, int D, int N, const double* y, const double* theta, double b, int kernel,
                  sls_gp** out);
int sls_gp_destroy(sls_gp* gp);
/* Append one observation (x, y) to a fitted handle in O(N^2) (Schur-complement update of K_y^-1, one new row of the
 * Cholesky factor): what FindNextPoints needs after every accepted point instead of re-building the dummy regressor
 * (src/acquisition-function.cpp:280-293). */
int sls_gp_append_point(sls_gp* gp, const double* x, double y);

#define SLS_GP_K_Y 0        /* N x N  m_K_y / m_K          (gaussian-process-regressor.hpp:31, preference-regressor.hpp:57) */
#define SLS_GP_K_Y_INV 1    /* N x N  m_K_y_inv            (gaussian-process-regressor.hpp:32) */
#define SLS_GP_CHOL_L 2     /* N x N  m_K_llt.matrixL()    (preference-regressor.hpp:60) */
#define SLS_GP_ALPHA 3      /* N      K_y^-1 y */
#define SLS_GP_MU_DATA 4    /* N      PredictMu at every data point (regressor.cpp:34-37) */
int sls_gp_get_matrix(sls_gp* gp, int what, double* out);
/* How the predictive deviation is formed.  The two reference classes differ (and so do the two modes, by rounding only):
 *   SLS_SIGMA_EXPLICIT_INVERSE (default)  sigma^2 = a - k^T K_y^-1 k with the explicit inverse, GaussianProcessRegressor::PredictSigma
 *                                         (src/gaussian-process-regressor.cpp:241-255, m_K_y_inv);
 *   SLS_SIGMA_CHOLESKY_SOLVE              sigma^2 = a - k . LLT.solve(k) = a - |L^-1 k|^2, PreferenceRegressor::PredictSigma
 *                                         (src/preference-regressor.cpp:299-313); the sigma gradient uses L^-T (L^-1 k) (:323-330).
 * For ill-conditioned K_y (cond >= 1e6, sigma <= 3e-3) the first form is only accurate to cond(K_y) eps a / (2 sigma), the second to
 * ~1e-12; a handle that stands for a PreferenceRegressor should therefore use the second, as the reference does. */
#define SLS_SIGMA_EXPLICIT_INVERSE 0
#define SLS_SIGMA_CHOLESKY_SOLVE 1
int sls_gp_set_sigma_mode(sls_gp* gp, int mode);
/* best_index / x_best: PredictMaximumPointFromData (src/regressor.cpp:29-43); mu_best = PredictMu(x_best);
 * logdet = CalcLogDetOfSymmetricPositiveDefiniteMatrix(K_y).  Any out pointer may be NULL. */
int sls_gp_get_summary(sls_gp* gp, int* best_index, double* mu_best, double* logdet);
/* A number that changes whenever the predictor behind the handle changes (fit, sls_gp_refit_dev, sls_gp_append_point,
 * sls_gp_set_sigma_mode) and is never repeated within the process, not even by a new handle at a recycled address: callers that
 * cache something derived from the handle (the host layer's replicas on other GPUs) compare it instead of the pointer. */
int sls_gp_generation(sls_gp* gp, long* generation);

/* PredictMu / PredictSigma for M points (gaussian-process-regressor.cpp:234-255, preference-regressor.cpp:293-313). */
int sls_gp_predict(sls_gp* gp, const double* Xs, int M, double* mu, double* sigma);
/* PredictMuDerivative / PredictSigmaDerivative (:257-272 / :315-330) for M points: dmu, dsigma are D x M. */
int sls_gp_predict_grad(sls_gp* gp, const double* Xs, int M, double* dmu, double* dsigma);
/* acquisition_func::CalcAcquisitionValue / CalcAcquisitionValueDerivative (src/acquisition-function.cpp:170-230)
 * for M points; grad (D x M) may be NULL. */
int sls_acq_eval(sls_gp* gp, int acq_type, double ucb_h, const double* Xs, int M, double* val, double* grad);

/* ---- multi-start maximiser ------------------------------------------------ */
/* FindGlobalSolution, parallelised multi-start branch (src/acquisition-function.cpp:121-153): S bounded L-BFGS runs
 * on [0,1]^D from the explicit starts (D x S), n_local objective evaluations each, all S advanced in lock step on the
 * device; returns the first maximum over the end points (Eigen maxCoeff semantics).
 * idx_out is the winning start's index plus start_index_offset (so that ranks sharding one global start set report
 * global indices).  x_stars (D x S) / y_stars (S) may be NULL. */
typedef struct sls_lbfgs_opts {
    /* sizeof(sls_lbfgs_opts) as the CALLER's header declares it (sls_lbfgs_default_opts fills it in).  The struct is caller-
     * allocated and has grown once already (ftol_rel / xtol_rel, round 5): the library reads struct_size bytes and takes its
     * defaults for members beyond them, and refuses a size it does not know (0, or larger than its own). */
    int struct_size;
    int history;        /* L-BFGS memory m (1..8), default 6 */
    double c1;          /* Armijo constant, default 1e-4 */
    double shrink;      /* backtracking factor, default 0.5 */
    double gtol;        /* projected-gradient inf-norm stop, default 0 */
    int max_backtracks; /* default 20 */
    /* NLopt's relative stopping tests (nlopt/src/util/stop.c, relstop), applied to every ACCEPTED step x -> x', f -> f'; 0 = off
     * (the default of this ABI: a start then runs until it cannot move or reaches its cap).  The reference's searches go through
     * nloptutil::solve, whose defaults are ftol_rel = xtol_rel = 1e-6: the host layer passes those.
     *   ftol_rel: stop when |f' - f| < ftol_rel (|f'| + |f|) / 2 or f' == f
     *   xtol_rel: stop when for every d  |x'_d - x_d| < xtol_rel (|x'_d| + |x_d|) / 2 or x'_d == x_d
     * The accepted point is kept; the start leaves the batch. */
    double ftol_rel;
    double xtol_rel;
} sls_lbfgs_opts;
void sls_lbfgs_default_opts(sls_lbfgs_opts* o);
int sls_acq_maximize(sls_gp* gp, int acq_type, double ucb_h, const double* starts, int S, int n_local,
                     const sls_lbfgs_opts* opts, long start_index_offset, double* x_out, double* val_out, long* idx_out,
                     double* x_stars, double* y_stars);
/* Statistics of the last sls_acq_maximize* call on this handle.  The starts advance in lock step over an ACTIVE SET: a start
 * that can no longer move (stationary projected gradient, null step, exhausted backtracking) is finished and leaves the
 * batch -- NLopt's max_evals is a cap per start, not a quota (src/acquisition-function.cpp:128-129).
 *   evals_issued  objective evaluations actually performed (sum over rounds of the live starts)
 *   evals_cap     S * n_local
 *   rounds        lock-step rounds executed (<= n_local)
 *   live_at_end   starts still moving when the cap was reached
 * Any out pointer may be NULL.  (The single-launch wavefront path for small problems counts, inside the kernel, the evaluations
 * of starts that were still moving, and reports rounds = n_local.) */
int sls_acq_last_stats(sls_gp* gp, long* evals_issued, long* evals_cap, int* rounds, int* live_at_end);
/* Same, with the starts already resident in HBM (D x S column-major device buffer) -- the timed path of bench.py. */
int sls_acq_maximize_dev(sls_gp* gp, int acq_type, double ucb_h, const double* starts_dev, int S, int n_local,
                         const sls_lbfgs_opts* opts, long start_index_offset, double* x_out, double* val_out,
                         long* idx_out);
/* objective_for_multiple_points (src/acquisition-function.cpp:63-110), the inner objective of FindNextPoints (:246-298):
 * predictive mean (and mu+) from gp_mean, predictive deviation from gp_sigma (the variance-updated dummy regressor). */
int sls_acq_eval_pair(sls_gp* gp_mean, sls_gp* gp_sigma, int acq_type, double ucb_h, const double* Xs, int M, double* val,
                      double* grad);
int sls_acq_maximize_pair(sls_gp* gp_mean, sls_gp* gp_sigma, int acq_type, double ucb_h, const double* starts, int S, int n_local,
                          const sls_lbfgs_opts* opts, double* x_out, double* val_out, long* idx_out);
/* Re-fit an existing handle in place from device-resident X (D x N), y (N): the timed "GP fit" of bench.py. */
int sls_gp_refit_dev(sls_gp* gp, const double* X_dev, const double* y_dev);

/* ---- multi-GPU maximisation ------------------------------------------------------------------------------
 * FindGlobalSolution's multi-start loop (src/acquisition-function.cpp:121-153) shards over its starts: the iterations share
 * only the const regressor (:125-141).  Every GPU holds a replica of the fitted state, runs a contiguous slice of the starts
 * with the global index offset, and contributes (value, global index, x[D]) to ONE ncclAllGather (RCCL over xGMI); the
 * first maximum (highest value, ties -> lowest global index: Eigen maxCoeff, :146-153) is taken on the gathered records.
 * RCCL is dlopen'ed on first use; single-GPU callers never load it. */

/* (1) one process, n devices, one host thread per device.  `devices` may name a GPU more than once (logical shards on one
 * device, used by tests on a 1-GPU box): RCCL allows one rank per GPU, so the records are then merged on the host;
 * sls_multi_exchange() says which exchange is in use ("ncclAllGather" or "host merge: <why>"). */
typedef struct sls_multi sls_multi;
typedef struct sls_multi_gp sls_multi_gp;
int sls_multi_create(const int* devices, int n, sls_multi** out);
int sls_multi_destroy(sls_multi* m);
int sls_multi_size(const sls_multi* m);
const char* sls_multi_exchange(const sls_multi* m);
sls_ctx* sls_multi_ctx(sls_multi* m, int shard);
/* sls_gp_create on every device (concurrently). */
int sls_multi_gp_create(sls_multi* m, const double* X, int D, int N, const double* y, const double* theta, double b, int kernel,
                        sls_multi_gp** out);
/* Replicas of an EXISTING fitted handle (what FindNextPoint* needs when several devices are configured): the shard on the
 * primary's own device is the primary itself -- no second fit there --, the other devices fit from its data.  `primary` must outlive
 * the result and must not be refitted or grown (sls_gp_append_point) while the replicas are in use. */
int sls_multi_gp_create_from(sls_multi* m, sls_gp* primary, sls_multi_gp** out);
int sls_multi_gp_destroy(sls_multi_gp* g);
sls_gp* sls_multi_gp_shard(sls_multi_gp* g, int shard);
/* sls_acq_maximize over all devices: starts (D x S, host) are split into contiguous slices; idx_out is the global start
 * index of the winner; evals_issued (may be NULL) sums the shards' sls_acq_last_stats. */
int sls_multi_acq_maximize(sls_multi_gp* g, int acq_type, double ucb_h, const double* starts, int S, int n_local,
                           const sls_lbfgs_opts* opts, double* x_out, double* val_out, long* idx_out, long* evals_issued);
/* sls_gp_predict over all devices: the M points (D x M, host) are split into contiguous column slices, shard r predicts its
 * slice on its device and writes mu / sigma at the slice's offset (gaussian-process-regressor.cpp:234-255 for M points; no
 * collective -- the outputs are disjoint).  A point's arithmetic does not depend on its slice: bit-identical to sls_gp_predict. */
int sls_multi_gp_predict(sls_multi_gp* g, const double* Xs, int M, double* mu, double* sigma);
/* GP MAP objective over the devices of `m`: the B independent points of one DIRECT iteration of PerformMapEstimation
   (src/gaussian-process-regressor.cpp:294) dealt round-robin, point k on shard k mod n, values gathered through host memory (no
   collective; the N^3 factorisation of ONE evaluation does not shard).  Bit-identical to sls_gp_nll_batch on one device. */
typedef struct sls_multi_nll sls_multi_nll;
int sls_multi_nll_create(sls_multi* m, const double* X, int D, int N, int kernel, sls_multi_nll** out);
int sls_multi_nll_destroy(sls_multi_nll* g);
int sls_multi_gp_nll_batch(sls_multi_nll* g, const double* y, const double* xs, int B, double* values);

/* (2) one process per GPU (torch.distributed.run / mpirun).  Rank 0 calls sls_comm_unique_id and distributes the 128 bytes
 * (any side channel); every rank then calls sls_comm_create on its context.  sls_comm_allgather_best is the single exchange
 * of a step: this rank's (value, global index, x[D]) in, the global first maximum out -- identical on every rank. */
typedef struct sls_comm sls_comm;
int sls_comm_unique_id(char* out128);
int sls_comm_create(sls_ctx* ctx, const char* id128, int rank, int world, sls_comm** out);
int sls_comm_destroy(sls_comm* c);
int sls_comm_allgather_best(sls_comm* c, double value, long index, const double* x, int D, double* value_out, long* index_out,
                            double* x_out);

/* ---- MAP objectives (hyper-parameter / goodness-value estimation) -------------------------------- */
/* Device state for repeated evaluations of the GP log-likelihood terms on a fixed design matrix X (D x N). */
typedef struct sls_nll sls_nll;
int sls_nll_create(sls_ctx* ctx, const double* X, int D, int N, int kernel, sls_nll** out);
int sls_nll_destroy(sls_nll* h);
/* NLopt's relative stopping tests for the MAP fits of this handle (sls_gp_map_fit, sls_pref_map_fit): applied to every accepted step
 * exactly as sls_lbfgs_opts.ftol_rel / xtol_rel are in the acquisition maximiser (on the optimiser's own variables: the logarithms of
 * the hyper-parameters).  0 = off, the default of a new handle; the reference's fits run through nloptutil::solve with 1e-6 / 1e-6
 * (SURVEY.md Appendix A), which the host layer sets.  A GP fit reaches its optimum to machine precision within 20-40 evaluations
 * and then backtracks for another 40-300 before it can no longer move; with the tests it ends there. */
int sls_nll_set_tolerances(sls_nll* h, double ftol_rel, double xtol_rel);
/* Core of both MAP objectives, for K_y = K_f(theta) + b I and a vector y (N):
 *   quad = y^T K_y^-1 y, logdet = log|K_y| (CalcLogDetOfSymmetricPositiveDefiniteMatrix), alpha = K_y^-1 y (may be NULL),
 *   grad_theta[p] = 1/2 alpha^T (dK/dtheta_p) alpha - 1/2 tr(K_y^-1 dK/dtheta_p), p = 0..D  (may be NULL)
 *   grad_b        = the same with dK/db = I                                              (may be NULL)
 * i.e. calc_grad_theta / calc_grad_b (src/gaussian-process-regressor.cpp:66-106) and CalcObjectiveThetaDerivative /
 * CalcObjectiveNoiseLevelDerivative (src/preference-regressor.cpp:53-115) without their prior terms, computed without
 * the (D+1) x N x N tensor of CalcLargeKYThetaDerivative (src/regressor.cpp:110-134).  A repeated (theta, b) re-uses
 * the cached factorisation (the reference's cached m_K / m_K_llt, src/preference-regressor.cpp:161-162,363-371). */
int sls_nll_eval(sls_nll* h, const double* y, const double* theta, double b, double* quad, double* logdet, double* alpha,
                 double* grad_theta, double* grad_b);
/* objective(x, grad) of src/gaussian-process-regressor.cpp:141-193 with x = (a, b, r_1..r_D) and the file's fixed
 * log-normal priors (:18-24).  grad (D + 2) may be NULL. */
int sls_gp_nll_grad(sls_nll* h, const double* y, const double* x, double* value, double* grad);
/* values[k] = the same objective at xs[k] = (a, b, r_1..r_D), k < B, no gradients: the B independent evaluations of one DIRECT
   iteration of PerformMapEstimation (src/gaussian-process-regressor.cpp:294 evaluates them one by one).
     N <= 128: ONE launch, one workgroup per parameter set (bit-identical to B single evaluations).
     N >  128: bordered factorisations -- row N of K_y carries y, so y^T K_y^-1 y and log|K_y| come from the factor alone (no
               inverse) --, several parameter sets per persistent launch, each on its own share of the chip (a factorisation of this
               size is bound by its serial chain and leaves most CUs idle): 3.7x the rate of one full evaluation after the other
               at N = 4096, B = 8.  Values are bit-identical to the same call with one point at a time and agree with
               sls_gp_nll_grad's value to rounding.
   Every parameter set is validated up front (a > 0, b >= 0, r > 0: SLS_ERR_INVALID otherwise); a parameter set whose K_y is not
   positive definite yields -HUGE_VAL instead of an error. */
int sls_gp_nll_batch(sls_nll* h, const double* y, const double* xs, int B, double* values);
/* objective(x, grad) of src/preference-regressor.cpp:129-259.  x = (y_1..y_M [, a, b, r_1..r_D] if use_map_hyperparams);
 * prefs_flat / pref_offsets: CSR image of std::vector<Preference> (n_prefs tuples, first index = preferred point).
 * grad (same length as x) may be NULL.  For M <= 128 the whole objective -- BTL terms (include/sequential-line-search/utils.hpp:
 * 25-52, no max-subtraction: same overflow behaviour) included -- is one single-workgroup launch; larger problems run the
 * tiled pipeline with the O(#preferences) BTL terms on the host. */
typedef struct sls_pref_cfg {
    int use_map_hyperparams;
    double default_a, default_r, default_b, prior_var, btl_scale;
    int noiseless; /* SEQUENTIAL_LINE_SEARCH_USE_NOISELESS_FORMULATION (src/preference-regressor.cpp:48-50,139-143) */
} sls_pref_cfg;
int sls_pref_objective(sls_nll* h, const unsigned* prefs_flat, const int* pref_offsets, int n_prefs, const double* x,
                       const sls_pref_cfg* cfg, double* value, double* grad);
/* PreferenceRegressor::PerformMapEstimation (src/preference-regressor.cpp:332-403) as ONE device launch for M <= 128 data points
 * and D <= 128 dimensions, with or without use_map_hyperparams: the objective above -- Bradley-Terry-Luce terms included -- is maximised on the device by
 * the bounded L-BFGS that stands in for nloptutil::solve(..., LD_TNEWTON, ..., num_iters) (:377).  Variables z = (y_1..y_M
 * [, log a, log b, log r_1..log r_D]); z0 / lower / upper / z_out have that length; max_evals = NLopt's max_evals.
 * evals_per_launch: 0 = the whole fit in one launch; k > 0 = launches of k evaluations each, continued from device-resident
 * state (same machine code, same bits: the test hook behind "one launch == one launch per evaluation").
 * Returns SLS_ERR_UNSUPPORTED (nothing computed) outside those limits: drive sls_pref_objective from the host instead. */
int sls_pref_map_fit(sls_nll* h, const unsigned* prefs_flat, const int* pref_offsets, int n_prefs, const sls_pref_cfg* cfg,
                     const double* z0, const double* lower, const double* upper, int max_evals, int evals_per_launch,
                     double* z_out, double* value, int* evals_used);
/* Local phase of GaussianProcessRegressor::PerformMapEstimation (src/gaussian-process-regressor.cpp:295: TNEWTON from the DIRECT
 * point) in one launch for N <= 128, D <= 128: maximises sls_gp_nll_grad's objective over z = (log a, log b, log r_1..log r_D). */
int sls_gp_map_fit(sls_nll* h, const double* y, const double* z0, const double* lower, const double* upper, int max_evals,
                   int evals_per_launch, double* z_out, double* value, int* evals_used);

/* ---- instrumentation ------------------------------------------------------ */
/* Per-kernel accumulated device time (ms, HIP events on the context's stream) and launch counts since the last reset.
 * Names: "gram", "potri" (N <= 4096: factor + L^-1 + K^-1 in one launch) or "potrf", "trtri", "lauum" (larger N, or
 * SLS_POTRI_FUSED=0), "fit_small" (N <= 128: the whole fit in one launch), "cross_gram", "acq_gemm", "grad_gemm", "finalize",
 * "lbfgs"; "potrf_fallbacks": launches = how often a single-launch factorisation gave up and was recomputed. */
int sls_prof_enable(sls_ctx* ctx, int on);
int sls_prof_reset(sls_ctx* ctx);
int sls_prof_get(sls_ctx* ctx, const char* name, double* total_ms, long* launches);

/* The library's run-time switches (SLS_* environment variables: A/B runs and test hooks, DESIGN.md "Run-time switches") are
 * parsed ONCE per process, at first use.  This re-reads the environment: for tests that switch paths inside one process. */
int sls_tuning_reload(void);

#ifdef __cplusplus
}
#endif
#endif
