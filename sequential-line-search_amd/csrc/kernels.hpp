// Launch wrappers of the hand-written gfx950 kernels (definitions in kernels_*.hip).
//
// Device-side data layout: every matrix keeps its LONG dimension contiguous.
//   training points   XtT [i + d*Np]      scaled + centred  x~ = (x - 0.5) / l_d, zero padded (Np = ceil128(N))
//   candidates        XsT [n + d*Sp]      same transform, n = candidate index         (Sp = ceil128(S))
//   K, L, Linv, Kinv  [i + j*Np]          column-major Np x Np, identity-padded
//   K*, C*, P         [n + i*Sp]          candidate-major: row i (training point) is contiguous over candidates
//   G_mu, G_sigma     [n + d*Sp]
// Padding to the 128-wide GEMM tile keeps every kernel free of edge code.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include "tuning.hpp"

#include <mutex>
#include <map>
#include <set>
#include <utility>

namespace slsk {

// hipFuncAttributeMaxDynamicSharedMemorySize is per (device, function): one process may drive several GPUs (sls_multi),
// so the opt-in is recorded per current device, under a lock (launch wrappers run on one host thread per device).
inline void ensure_dyn_lds(const void* fn, int bytes) {
    static std::mutex mtx;
    static std::map<std::pair<int, const void*>, int> done;   // (device, kernel) -> largest size set so far
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mtx);
    int& have = done[{dev, fn}];
    if (bytes > have) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        have = bytes;
    }
}

// Geometry of the CURRENT device, read once per device: compute units, and the XCDs (each with its own L2) they sit in.  HIP has no
// attribute for the latter; gfx950 builds an XCD from 32 active CUs and hands out workgroups round-robin over a partition's XCDs
// (SPX: 256 CUs = 8 XCDs; a CPX partition: 32 CUs = 1).  A CU count that is not a multiple of 32 is treated as one XCD.
struct ChipGeometry {
    int n_cu, n_xcd;
};
inline ChipGeometry chip_geometry() {
    static std::mutex mtx;
    static std::map<int, ChipGeometry> table;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mtx);
    auto it = table.find(dev);
    if (it != table.end()) return it->second;
    int v = 0;
    (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    ChipGeometry g;
    g.n_cu = v > 0 ? v : 64;
    g.n_xcd = (g.n_cu % 32 == 0) ? g.n_cu / 32 : 1;
    table[dev] = g;
    return g;
}

struct KernelSpec {
    int kernel;   // SLS_KERNEL_*
    double a;     // signal variance theta[0]
};

// ---- kernels_gram.hip -------------------------------------------------------
// X (D x M column-major, device) -> XT[i + d*ld] = (X[d,i]-0.5)*inv_ell[d] for i<M,d<D, 0 elsewhere (i<Mp, d<Dcols); norms[i].
void launch_prep_points(hipStream_t s, const double* X, int D, int M, const double* inv_ell, double* XT, long ld, int Mp,
                        int Dcols, double* norms);
// same from candidate-major raw coordinates xr[n + d*ldr]
void launch_prep_cands(hipStream_t s, const double* xr, long ldr, int D, int M, const double* inv_ell, double* XT, long ld,
                       int Mp, int Dcols, double* norms);
// XaT[i + d*ld] = alpha[i] * XT[i + d*ld]
void launch_scale_rows(hipStream_t s, const double* XT, const double* alpha, double* XaT, long ld, int Np, int Dcols);
// K[i + j*Np] = k(x_i, x_j) (+ b on the diagonal); identity in the padding.  lower_only: skip tiles above the diagonal.
void launch_gram_sym(hipStream_t s, const double* XT, long ld, int Dp, const double* nx, int Np, int N, KernelSpec ks, double b,
                     double* K, bool lower_only);
// Ks[n + i*ldk] = k(xs_n, x_i), Cs likewise with the derivative weight c (Matern; Cs == Ks for SE);
// mu_part[t*ldk + n] = sum_{i in tile t} alpha_i k, ca_part likewise with c.  alpha may be NULL (no partials).
void launch_cross_gram(hipStream_t s, const double* XsT, long lds_, const double* ns, int Sp, const double* XT, long ld,
                       const double* nx, int Np, int N, int Dp, KernelSpec ks, const double* alpha, double* Ks, double* Cs,
                       long ldk, double* mu_part, double* ca_part);

// ---- kernels_small.hip ---------------------------------------------------------
// Whole MAP-objective evaluation for N <= 128, D <= 128 in one single-workgroup launch.  Inputs a, b, l_1..l_D, y_1..y_N travel
// in the kernel argument block (D <= 32) or in in_dev = [a, b, l.., y..] (device).  out (device, NLL_SMALL_OUT_DOUBLES):
// [0] sum W.*K_f, [1] d/db, [2] y^T alpha, [3] logdet, [4] potrf info, [8..8+D) d/dl (only if want_grad),
// [136..136+N) alpha.  X: raw D x N column-major design matrix (device); XTr: its transpose [i + d * 128] (device);
// kc (want_grad only): 2 x 128 x 128 doubles of scratch per launch (kernel values and derivative weights of the pairs, written by
// the Gram pass and read back by the gradient pass of the same workgroup).
constexpr int NLL_SMALL_MAX_N = 128;
constexpr int NLL_SMALL_MAX_D = 128;
constexpr int NLL_SMALL_MAX_ARG_D = 32;    // up to here the length scales travel in the kernel argument block (no upload)
constexpr int NLL_SMALL_OUT_GL = 8, NLL_SMALL_OUT_ALPHA = 8 + NLL_SMALL_MAX_D, NLL_SMALL_OUT_DOUBLES = NLL_SMALL_OUT_ALPHA + NLL_SMALL_MAX_N;
constexpr size_t NLL_SMALL_KC_DOUBLES = 2 * 128 * 128;
struct NllSmallArgs {
    const double* X;
    const double* XTr;
    double* kc;
    int x_lds = 1;            // 0: never stage the design matrix in LDS (A/B and tests: both paths agree to rounding)
    int D, N, want_grad;
    int* info;
    double* out;
    const double* in_dev;
    // batch > 1: workgroup w evaluates the parameter set in_dev + w * in_stride (in_dev must be set) into out + w * out_stride,
    // with its own pivot-failure word info[w]
    int batch;
    long in_stride, out_stride;
    double a, b;
    double ell[NLL_SMALL_MAX_ARG_D];
    double y[NLL_SMALL_MAX_N];
};
void launch_nll_small(hipStream_t s, int kernel, const NllSmallArgs& args);

// Whole fit of a GP handle for N <= 128 (Np = 128) in one single-workgroup launch (gp_fit_small_kernel): every output of
// gp_fit_device (capi.hip).  All pointers device; matrices 128 x 128 column-major, XT / XaT [i + d * 128] with Dcols columns.
struct GpFitSmallArgs {
    const double *X, *y, *inv_ell;   // X: raw D x N column-major (its transpose passes through XaT before alpha o X~ lands there)
    int D, N, Dcols;
    double a, b;
    double *XT, *nx, *XaT, *L, *Linv, *U, *Kinv, *alpha, *mu_data, *scal;   // scal[0] = max_i mu(x_i), scal[1] = log|K_y|
    long* d_idx;                     // first arg max of mu over the data points
    int* info;                       // 1 + index of the first non-positive pivot, or 0
    double* summary = nullptr;       // mapped host memory (or nullptr): [0] scal[0], [1] scal[1], [2] d_idx[0], [3] *info, [4] 0
    int x_lds = 1;                   // as NllSmallArgs::x_lds
};
void launch_gp_fit_small(hipStream_t s, int kernel, const GpFitSmallArgs& args);

// Whole MAP FIT for N <= 128 in one single-workgroup launch (map_opt_kernel): the bounded L-BFGS of host/device.cpp
// (optim::MaximizeBounded) with the objective evaluated in place -- the preference objective of
// src/preference-regressor.cpp:129-259 (Bradley-Terry-Luce terms from a CSR / CSC image of m_D, GP term, log-normal priors)
// or the GP marginal-likelihood objective of src/gaussian-process-regressor.cpp:141-193.
// Variables z (n = ny + nh <= 320: N <= 128 goodness values + D + 2 <= 130 hyper-parameters):  z[0..ny) = goodness values y
// (ny = N or 0: y fixed = y_fixed);  z[ny..ny+nh) = (a, b, l_1..l_D) (nh = D + 2 or 0: fixed a0, b0, r0), logarithms if log_hyper.
// The optimiser state lives in registers, 64 variables per register: 3 per lane up to 192 variables (C3: 91 + 34), 5 beyond.
constexpr int MAP_OPT_MAX_VARS = 320;
constexpr int MAP_OPT_HIST = 8;
constexpr int MAP_OPT_TRACE_SLOTS = 24;
constexpr int MAP_OPT_STATE_DOUBLES = (4 + 2 * MAP_OPT_HIST) * MAP_OPT_MAX_VARS + 32;
constexpr int MAP_OPT_OUT_X = 8, MAP_OPT_OUT_G = 8 + MAP_OPT_MAX_VARS, MAP_OPT_OUT_DOUBLES = 8 + 2 * MAP_OPT_MAX_VARS;
struct MapOptArgs {
    const double* X;          // D x N column-major design matrix (device)
    const double* XTr;        // its transpose [i + d * 128] (device)
    double* kc;               // NLL_SMALL_KC_DOUBLES of scratch (nh > 0)
    int x_lds = 1;            // as NllSmallArgs::x_lds
    int D, N, ny, nh, log_hyper, noiseless;
    const double* y_fixed;    // device, N (ny == 0)
    double a0, b0, r0;        // fixed hyper-parameters (nh == 0)
    double mu_a, mu_b, mu_r, s2_a, s2_b, s2_r;   // log-normal priors of a, b, l_d (nh > 0)
    double btl_scale;
    int n_prefs, flat_len;
    const int *pref_off, *pref_flat;   // CSR of the preference tuples: members of tuple p = pref_flat[pref_off[p] .. pref_off[p+1])
    const int *csc_off, *csc_ent;      // transposed: flat positions that name data point j, in increasing order
    double* btl_scratch;               // flat_len doubles (device); used when flat_len > 640 (else the LDS scratch)
    const double *z0, *lower, *upper;  // device, n each
    int max_evals;            // cap on objective evaluations of the whole fit
    double ftol_rel, xtol_rel;   // NLopt's relative stopping tests on accepted steps (sls_nll_set_tolerances), 0 = off
    int budget;               // evaluations performed by THIS launch (>= max_evals: the whole fit in one launch)
    int eval_only;            // 1: evaluate at z0, write value and gradient, stop
    int fresh;                // 1: start from z0; 0: continue from `state`
    double* state;            // MAP_OPT_STATE_DOUBLES (device): optimiser state, stored at the end of every launch
    double* out;              // MAP_OPT_OUT_DOUBLES: [0] objective at x, [1] evaluations used, [2] finished, [3] last pivot info,
                              // [8 .. 8+n) x, [8+320 .. 8+320+n) gradient (eval_only)
    int* info;
    // optional (SLS_MAP_TRACE=1, probes): MAP_OPT_TRACE_SLOTS ticks of the 100 MHz clock thread 0 spent in [0] publishing the trial
    // point, [8] Gram, [9] Cholesky, [10] log-det, [11] inverse, [1] alpha, [12] gradient weights, [13] length-scale contraction,
    // [3] BTL terms, [4] value + gradient slots, [5] optimiser, [6] evaluations, [7] kernel
    long long* trace;
};
void launch_map_opt(hipStream_t s, int kernel, const MapOptArgs& args);

// ---- kernels_acq.hip ---------------------------------------------------------
// P[n + i*ldk] = Cs * (Kinv Ks)  and the partial column sums kw_part (Ks.*W), cw_part (Cs.*W) per 128-row tile of i.
// kw_part / cw_part: [2 tn + half][candidate] (whole tiles use the even slot).  Returns the index (in the kernel's grouped tile
// order) of the first tile that ran as two half tiles (= the tile count if none): FinalizeArgs::split_first.
int launch_acq_gemm(hipStream_t s, const double* Ks, const double* Cs, long ldk, int Sp, const double* Kinv, int Np, double* P,
                    double* kw_part, double* cw_part, int* sync = nullptr);
// kw_part = per-tile partial sums of (Linv Ks)^2 (triangular contraction), cw_part = 0: the no-gradient form of acq_gemm
void launch_var_gemm(hipStream_t s, const double* Ks, long ldk, int Sp, const double* Linv, int Np, double* kw_part,
                     double* cw_part);
// Gs[n + d*ldk] = sum_i P[n,i] XT[i,d] ;  Gm[n + d*ldk] = sum_i Cs[n,i] XaT[i,d]   (d < Dcols)
// part (D <= 64 only, may be NULL): scratch of 8 * Sp * 64 doubles; used when grad_gemm_wants_split(Sp) to spread the
// contraction of a small launch over four times as many workgroups (same bits as the unsplit form).
// Always, since round 3: four times as many workgroups of a quarter of the work each fill the 768 slots evenly at every active-set
// size (C4: 85.9 -> 79.2 ms per step including grad_reduce4_kernel).  SLS_GRAD_SPLIT_TILES=t: split only launches of fewer than
// t tiles (0: never) -- the tests run both forms against each other.
inline bool grad_gemm_wants_split(int Sp) { return !tune_set(TUNE_GRAD_SPLIT_TILES) || 2 * (Sp / 128) < tune(TUNE_GRAD_SPLIT_TILES, 0); }
void launch_grad_gemm(hipStream_t s, const double* P, const double* Cs, long ldk, int Sp, const double* XT, const double* XaT,
                      long ld, int Np, int Dcols, double* Gs, double* Gm, double* part = nullptr);
struct FinalizeArgs {
    int S, D, nbt;            // candidates in this chunk, dims, number of 128-row tiles of i
    int ntm, split_first;     // candidate tiles of the chunk; first half-split tile of launch_acq_gemm (INT_MAX: none)
    long ldk;                 // candidate leading dimension of this chunk
    const double *mu_part, *ca_part, *kw_part, *cw_part, *Gs, *Gm, *XsT, *inv_ell;
    const double* kw_solve_part = nullptr;   // solve-based sigma (see WaveArgs::solve_sigma): partial sums of (L^-1 k)^2 from var_gemm
    double a, mu_best, ucb_h;
    int acq;                  // SLS_ACQ_* (used when val/grad requested)
    // outputs (any may be NULL); all candidate-major with leading dimension ldo, offset already applied
    long ldo;
    double *mu, *sigma, *dmu, *dsigma, *val, *grad;
};
void launch_finalize(hipStream_t s, const FinalizeArgs& a);

// acquisition value / gradient from separately predicted mean and deviation (objective_for_multiple_points,
// src/acquisition-function.cpp:63-110): all arrays candidate-major with leading dimension ld
void launch_combine(hipStream_t s, int S, int D, long ld, const double* mu, const double* sigma, const double* dmu,
                    const double* dsigma, int acq, double mu_best, double ucb_h, double* val, double* grad);

struct LbfgsState {
    int S, D, m;
    long ld;                  // Sp
    double *x, *g, *dir, *xt, *scr;   // [n + d*ld]
    double *Sh, *Yh;          // [h][d][n]
    double *rho;              // [h][n]
    double *f, *t;            // [n]
    int *hlen, *hpos, *nbt, *done;    // [n]
    double c1, shrink, gtol;
    int max_backtracks;
    double ftol_rel = 0.0, xtol_rel = 0.0;   // NLopt's relative stopping tests on accepted steps (sls_lbfgs_opts), 0 = off
    // Active set: this round evaluated `nlive` trial points; column j of (val, grad) belongs to start live[j]
    // (live == nullptr: identity, nlive == S).  ldv = leading dimension of grad.
    const int* live;
    int nlive;
    long ldv;
};
// first: (val, grad) are the objective at the starts; otherwise at the trial points xt.  Writes the next trial points into
// xt (start-indexed) and marks starts that can no longer move (null step, exhausted backtracking, stationary) as done.
void launch_lbfgs_step(hipStream_t s, const LbfgsState& st, const double* val, const double* grad, bool first);
// Stable compaction of the starts that are still moving: live_out[0..count) = { n in live_in (or 0..n_in) : !done[n] } in
// increasing order, count -> *count_out (device); then xc[j + d*ldc] = xt[live_out[j] + d*ld] (padding columns up to the
// next multiple of 128 are filled with 0.5 so that the tile kernels read finite numbers).
void launch_compact_live(hipStream_t s, const int* live_in, int n_in, const int* done, int* live_out, int* count_out,
                         int* block_counts /* (n_in + 1023) / 1024 ints of scratch */);
void launch_gather_trials(hipStream_t s, const double* xt, long ld, int D, const int* live, const int* count_dev, int n_max,
                          double* xc, long ldc);
void launch_clamp_starts(hipStream_t s, const double* starts /*D x S col-major*/, int D, int S, double* xt, long ld, int Sp);
// first maximum of y[n] = -f[n]: out[0] = value, idx_out[0] = index
// best (mapped host memory): [0] max of -f, [1] its first index, [2] *counter (or 0), [8 .. 8 + D) = x[index + d ldx]
void launch_argmax_neg_gather(hipStream_t s, const double* f, int S, const double* x, long ldx, int D, double* best,
                              const unsigned long long* counter);

// NLopt's relative stopping tests (nlopt/src/util/stop.c: relstop) as the maximisers apply them to an accepted step; tolerance 0 = off.
//   lbfgs_f_stalled: |f' - f| < tol (|f'| + |f|) / 2 or f' == f
//   lbfgs_x_moved:   1 when coordinate d still moved by more than its tolerance (summed over d: 0 = every coordinate stalled)
__host__ __device__ inline bool lbfgs_f_stalled(double f_old, double f_new, double tol) {
    return tol > 0.0 && (fabs(f_new - f_old) < tol * 0.5 * (fabs(f_new) + fabs(f_old)) || f_new == f_old);
}
__host__ __device__ inline double lbfgs_x_moved(double x_old, double x_new, double tol) {
    return (fabs(x_new - x_old) < tol * 0.5 * (fabs(x_new) + fabs(x_old)) || x_new == x_old) ? 0.0 : 1.0;
}

// ---- kernels_wave.hip: one wavefront per start, whole L-BFGS in one launch (small problems) ----
struct WaveArgs {
    int S, D, N, Np, m, n_local, acq, matern;
    int lds_per_wave, Dr;                 // filled by the launcher
    int stage_kinv, stage_xt;             // filled by the launcher: K^-1 / XT copied into LDS behind the four waves' areas
    double a, mu_best, ucb_h, c1, shrink, gtol;
    int max_backtracks;
    double ftol_rel = 0.0, xtol_rel = 0.0;   // as in LbfgsState
    const double *XT, *inv_ell, *Kinv, *alpha, *starts;   // XT [i + d*Np] scaled; starts D x S column-major (raw)
    // solve_sigma (handles of a PreferenceRegressor, sls_gp_set_sigma_mode): sigma^2 = a - |L^-1 k|^2 and w = L^-T (L^-1 k) as the
    // reference's k . LLT.solve(k) (src/preference-regressor.cpp:299-313,323-330) instead of the explicit K^-1 of
    // GaussianProcessRegressor (src/gaussian-process-regressor.cpp:241-255).  Linv = L^-1 (zero above the diagonal), U = Linv^T.
    int solve_sigma = 0;
    int stage_ld = 0, stage_cols = 0;     // filled by the launcher: leading dimension / columns of the staged first-pass matrix
    int xch = 0;                          // filled by the launcher: doubles of the cooperative form's exchange area (0: one wave per start)
    const double *Linv = nullptr, *U = nullptr;
    double *x_out, *f_out;                // x_out[n + d*ld], f_out[n] = -acq at the end point
    unsigned long long* useful;           // optional: += evaluations of starts that were still moving (the others run idle)
    long ld;
    // evaluation-only mode (n_local == 0): `starts` holds the M query points; any of these may be NULL
    double *ev_mu, *ev_sigma, *ev_dmu, *ev_dsigma, *ev_val, *ev_grad;
    // optional (SLS_WAVE_TRACE=1, probes): ticks of the 100 MHz clock wave 0 of workgroup 0 spent in [0] the kernel-vector loop,
    // [1] w = K^-1 k, [2] the four reductions, [3] the gradient loop + acquisition, [4] L-BFGS direction, [5] step bookkeeping,
    // [6] evaluations, [7] whole kernel
    long long* trace = nullptr;
};
constexpr int WAVE_PATH_MAX_NP = 512;     // 8 rows per lane
constexpr int WAVE_PATH_MAX_D = 128;      // 2 dimensions per lane
void launch_maximize_wave(hipStream_t s, WaveArgs a);

// ---- kernels_chol.hip ---------------------------------------------------------
// In-place lower Cholesky of the Np x Np matrix A (ld = Np); Linv receives the 128x128 diagonal-block inverses
// (rest of Linv untouched).  info (device int) gets 1 + index of the first non-positive pivot, or stays 0.
// nbo: 128-columns per outer block of the multi-launch schedule (1 = one-level algorithm; default from potrf_default_nbo(Np)).
constexpr int NB = 128;            // block size of the factorisation = the GEMM tile
int potrf_default_nbo(int Np);
// one 128 x 128 diagonal block: Cholesky in place + its inverse into Tout (kernels_chol.hip; the multi-launch schedule's step)
void launch_chol_diag(hipStream_t s, double* A, long lda, double* Tout, long ldt, int* info, int global_off);
void launch_potrf(hipStream_t s, double* A, int Np, double* Linv, int* info, int nbo = 0, int* dataflow_sync = nullptr);
void launch_gemm_splitk_nt(hipStream_t s, const double* A, long lda, const double* B, long ldb, double* Cpart, long ldc,
                           long part_stride, int mt, int nt, int K, int chunks);
// Single-launch factorisation (SLS_POTRF_MODE=3, default): per-tile ownership and ready flags, no grid barriers.  sync =
// potrf_dataflow_sync_ints(Np) ints of device scratch; returns false when not applicable.  info[1] != 0 afterwards: a bounded
// wait expired (the kernel aborted; results undefined).  launch_potrf takes this path when dataflow_sync != nullptr, Np >= 384
// and SLS_POTRF_MODE is not 0 (0 = multi-launch schedule).
size_t potrf_dataflow_sync_ints(int Np);
bool launch_potrf_dataflow(hipStream_t s, double* A, int Np, double* Linv, int* info, int* sync, long long* trace = nullptr);
// nprob independent Np x Np factorisations in ONE launch, each on its own share of the chip's workgroups: problem q lives at
// A + q strideA / Linv + q strideA (doubles), uses sync + q stride_sync (ints) and reports into info[2 q], info[2 q + 1].
// block_inverses = false: the 128 x 128 inverses T_jj are not formed (callers that only need L).  Same bits per problem as a
// launch of its own (a tile's arithmetic does not depend on how many workgroups take part).
int potrf_dataflow_max_problems(int Np);
bool launch_potrf_dataflow_batch(hipStream_t s, double* A, int Np, double* Linv, int* info, int* sync, int nprob, long strideA,
                                 long stride_sync, bool block_inverses, long long* trace = nullptr);
// The same launch with a second team of workgroups that builds Linv = L^-1, U = Linv^T and Kinv = A^-1 behind the factorisation
// (N <= 4096; kernels_chol.hip: potri_team).  launch_potri picks it when it applies and falls back to launch_potrf + launch_trtri
// + launch_lauum otherwise (SLS_POTRI_FUSED=0: always); returns true when the fused launch ran.
struct PotriFused {
    double* U;
    double* Kinv;
};
bool launch_potri_dataflow(hipStream_t s, double* A, int Np, double* Linv, double* U, double* Kinv, int* info, int* sync,
                           long long* trace = nullptr);
bool potri_fused_applies(int Np, bool have_sync);   // sizes / switches only: the launch itself may still decline (too few CUs)
bool launch_potri(hipStream_t s, double* A, int Np, double* Linv, double* U, double* Kinv, int* info, int* dataflow_sync,
                  bool linv_zeroed = true);   // false: Linv was not cleared by the caller (only the separate launches need it: done inside)
int potrf_default_mode(int Np);
// Linv <- L^-1 (lower) given L and the diagonal-block inverses already in Linv; tmp is an Np x Np scratch.
void launch_trtri(hipStream_t s, const double* L, int Np, double* Linv, double* tmp, double* U);   // U <- (L^-1)^T
void launch_transpose_full(hipStream_t s, const double* src, double* dst, int Np);   // dst = src^T (Np x Np)
// diagonal-block inverses only (for potrs / potri on a caller-supplied factor)
void launch_diag_inverse(hipStream_t s, const double* L, int Np, double* Linv);
// Kinv <- Linv^T Linv (full symmetric)
void launch_lauum(hipStream_t s, const double* U, int Np, double* Kinv);                           // K^-1 = U U^T
// B (Np x Rp, ld = Np, Rp multiple of 128) <- (L L^T)^-1 B using the block inverses in Linv's diagonal
void launch_potrs(hipStream_t s, const double* L, const double* Linv, int Np, double* B, int Rp);
// y = A x; `part` is a caller-owned scratch of (Np/128) * Np doubles (per-chunk partial sums, reduced in fixed order)
void launch_gemv_n(hipStream_t s, const double* A, int Np, const double* x, double* y, double* part, bool lower = false);   // lower: A is lower triangular (zeros above the diagonal are not read)
void launch_gemv_t(hipStream_t s, const double* A, int Np, const double* x, double* y, bool lower = false);   // y = A^T x
void launch_zero_upper(hipStream_t s, double* A, int Np);
void launch_fill(hipStream_t s, double* p, long n, double v);
// mu_data[i] = y[i] - b*alpha[i] (i<N); out: first argmax + value; logdet = 2 sum log L_ii (i<N)
// mu_data = y - b alpha, scal[0] = its first maximum, d_idx[0] = where, scal[1] = 2 sum log L_ii; summary (mapped host memory or
// nullptr): [0] max, [1] log-det, [2] index, [3..4] info[0..1]
void launch_fit_summary(hipStream_t s, const double* y, const double* alpha, double b, int N, double* mu_data, const double* L, int Np,
                        const int* info, double* scal, long* d_idx, double* summary);
// bordered factorisation (value-only MAP objective): row N of each of the nprob matrices A + q strideA <- (y^T, c); after the
// factorisation out[2 q] = y^T K^-1 y (= |row N of L|^2) and out[2 q + 1] = log|K|
void launch_border_row(hipStream_t s, double* A, long strideA, int nprob, int Np, int N, const double* y, double c);
void launch_border_reduce(hipStream_t s, const double* L, long strideA, int nprob, int Np, int N, double* out);

// ---- rank-1 growth of the fitted state by one data point (kernels_chol.hip) ----
// scal_out[0] = k.u, scal_out[1] = l.l, scal_out[2] = u.y  (fixed-order block reductions)
void launch_append_dots(hipStream_t s, const double* k, const double* u, const double* l, const double* y, int N, double* scal_out);
// Kinv(0:N,0:N) += u u^T / s ; Kinv(N,0:N) = Kinv(0:N,N) = -u / s ; Kinv(N,N) = 1/s ;
// L(N,0:N) = l ; L(N,N) = lam ; Linv(N,0:N) = -u / lam ; Linv(N,N) = 1/lam ; alpha(0:N) += u (uy - eta)/s ; alpha(N) = (eta - uy)/s
void launch_append_update(hipStream_t s, double* Kinv, double* L, double* Linv, double* alpha, int Np, int N, const double* u,
                          const double* l, const double* scal /* k.u, l.l, u.y on device */, double kappa, double eta);

// generic C = alpha * opA opB^T + beta * C on full 128-tiles (mt x nt tiles, K multiple of 16)
void launch_gemm_plain(hipStream_t s, const double* A, long lda, bool a_kc, const double* B, long ldb, bool b_kc, double* C,
                       long ldc, int mt, int nt, int K, double alpha, double beta);

// ---- kernels_map.hip -----------------------------------------------------------
// MAP-gradient weights (replaces the (D+1) x N x N tensor of CalcLargeKYThetaDerivative, src/regressor.cpp:110-134):
// G[j + k*Np] = 1/2 (alpha_j alpha_k - Kinv_jk) c_jk  (0 in the padding);  wk_part[tile] = sum over the tile of
// 1/2 (alpha_j alpha_k - Kinv_jk) kf_jk  (kf = kernel value without noise).  ntiles = (Np/128)^2 partials.
void launch_nll_weight(hipStream_t s, const double* XT, long ld, int Dp, const double* nx, int Np, int N, KernelSpec ks,
                       const double* alpha, const double* Kinv, double* G, double* wk_part, double* row_part);
// row_part (nullptr: skip): [Np / 128][Np] partial row sums of G, one slice per tile column; launch_sum_chunks adds them in slice order
// out[0] = sum_t part[t] (fixed order);  out[1] = 1/2 (alpha.alpha - tr Kinv);  out[2] = y.alpha
void launch_nll_scalars(hipStream_t s, const double* part, int nparts, const double* alpha, const double* y,
                        const double* Kinv, int Np, int N, double* out, const double* Lfac = nullptr,
                        const int* info = nullptr);   // Lfac: out[4] = 2 sum log L_ii too;  info: out[5..6] = info[0..1]
// gl[p] = 2 * inv_ell[p] * sum_j XT[j,p] * (XT[j,p] * s_j - Y[j,p]),  p < D;  one more workgroup of the same launch does what
// launch_nll_scalars does (arguments from `part` on)
void launch_lengthscale_grad(hipStream_t s, const double* XT, const double* Y, const double* svec, const double* inv_ell,
                             long ld, int N, int D, double* gl, const double* part, int nparts, const double* alpha, const double* y,
                             const double* Kinv, int Np, double* out, const double* Lfac, const int* info);
// Y = sum of `chunks` partial products `stride` apart (in place in the first), and svec = sum of `row_chunks` slices of row_part
void launch_sum_chunks(hipStream_t s, double* Y, long n, int chunks, long stride, const double* row_part, int Np, int row_chunks,
                       double* svec);

}  // namespace slsk
