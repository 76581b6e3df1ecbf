/*
 * sls_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See sls_oracle.h.
 *
 * PARITY UNPINNED against the real reference binary (it cannot be built here);
 * pinned against mpmath / scipy / sklearn fixtures (tests/golden/).
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 */
#include "sls_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI 3.14159265358979323846264338327950288

static double* dalloc(long n) {
    double* p = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    return p;
}

/* ------------------------------------------------------------------------- */
/* mathtoolbox kernel functions                        */
/* ------------------------------------------------------------------------- */

static double scaled_sqdist(const double* xa, const double* xb, const double* theta, int D) {
    double q = 0.0;
    for (int i = 0; i < D; ++i) {
        const double d = xa[i] - xb[i];
        const double l = theta[1 + i];
        q += (d * d) / (l * l);
    }
    return q;
}

/* mathtoolbox::GetArdSquaredExpKernel / GetArdMatern52Kernel (src/regressor.cpp:14,21) */
double slso_kernel(int kernel, const double* xa, const double* xb, const double* theta, int D) {
    const double a = theta[0];
    const double q = scaled_sqdist(xa, xb, theta, D);
    if (kernel == SLSO_KERNEL_ARD_SE) return a * exp(-0.5 * q);
    const double s = sqrt(5.0 * q);
    return a * (1.0 + s + (5.0 / 3.0) * q) * exp(-s);
}

/* ...ThetaDerivative (src/regressor.cpp:15,22) */
void slso_kernel_theta_derivative(int kernel, const double* xa, const double* xb, const double* theta, int D, double* out) {
    const double a = theta[0];
    const double q = scaled_sqdist(xa, xb, theta, D);
    if (kernel == SLSO_KERNEL_ARD_SE) {
        const double e = exp(-0.5 * q);
        out[0] = e;
        for (int i = 0; i < D; ++i) {
            const double d = xa[i] - xb[i], l = theta[1 + i];
            out[1 + i] = a * e * d * d / (l * l * l);
        }
    } else {
        const double s = sqrt(5.0 * q), e = exp(-s);
        out[0] = (1.0 + s + (5.0 / 3.0) * q) * e;
        const double c = a * (5.0 / 3.0) * (1.0 + s) * e;
        for (int i = 0; i < D; ++i) {
            const double d = xa[i] - xb[i], l = theta[1 + i];
            out[1 + i] = c * d * d / (l * l * l);
        }
    }
}

/* ...FirstArgDerivative (src/regressor.cpp:16,23) */
void slso_kernel_first_arg_derivative(int kernel, const double* xa, const double* xb, const double* theta, int D, double* out) {
    const double a = theta[0];
    const double q = scaled_sqdist(xa, xb, theta, D);
    double c;
    if (kernel == SLSO_KERNEL_ARD_SE) {
        c = a * exp(-0.5 * q);
    } else {
        const double s = sqrt(5.0 * q);
        c = a * (5.0 / 3.0) * (1.0 + s) * exp(-s);
    }
    for (int i = 0; i < D; ++i) {
        const double d = xa[i] - xb[i], l = theta[1 + i];
        out[i] = -c * d / (l * l);
    }
}

/* ------------------------------------------------------------------------- */
/* src/regressor.cpp free functions                                           */
/* ------------------------------------------------------------------------- */

/* CalcSmallK, src/regressor.cpp:45-59 */
void slso_calc_small_k(int kernel, const double* x, const double* X, int D, int N, const double* theta, double* k) {
    for (int i = 0; i < N; ++i) k[i] = slso_kernel(kernel, x, X + (long)i * D, theta, D);
}

/* CalcLargeKF, src/regressor.cpp:73-89 */
void slso_calc_large_kf(int kernel, const double* X, int D, int N, const double* theta, double* K) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < N; ++i)
        for (int j = i; j < N; ++j) {
            const double v = slso_kernel(kernel, X + (long)i * D, X + (long)j * D, theta, D);
            K[i + (long)j * N] = v;
            K[j + (long)i * N] = v;
        }
}

/* CalcLargeKY, src/regressor.cpp:61-71 */
void slso_calc_large_ky(int kernel, const double* X, int D, int N, const double* theta, double b, double* K) {
    slso_calc_large_kf(kernel, X, D, N, theta, K);
    for (int i = 0; i < N; ++i) K[i + (long)i * N] += b;
}

/* CalcSmallKSmallXDerivative, src/regressor.cpp:91-108 */
void slso_calc_small_k_small_x_derivative(int kernel, const double* x, const double* X, int D, int N, const double* theta, double* J) {
    for (int i = 0; i < N; ++i) slso_kernel_first_arg_derivative(kernel, x, X + (long)i * D, theta, D, J + (long)i * D);
}

/* CalcLargeKYThetaDerivative, src/regressor.cpp:110-134: tensor[p] = dK/dtheta_p, p = 0..D */
void slso_calc_large_ky_theta_derivative(int kernel, const double* X, int D, int N, const double* theta, double* T) {
    double* g = dalloc(D + 1);
    for (int i = 0; i < N; ++i)
        for (int j = i; j < N; ++j) {
            slso_kernel_theta_derivative(kernel, X + (long)i * D, X + (long)j * D, theta, D, g);
            for (int p = 0; p <= D; ++p) {
                T[(long)p * N * N + i + (long)j * N] = g[p];
                T[(long)p * N * N + j + (long)i * N] = g[p];
            }
        }
    free(g);
}

/* ------------------------------------------------------------------------- */
/* dense linear algebra (Eigen stand-ins)                                     */
/* ------------------------------------------------------------------------- */

static int chol_unblocked(double* A, int n, long lda) {
    for (int j = 0; j < n; ++j) {
        double d = A[j + j * lda];
        for (int k = 0; k < j; ++k) d -= A[j + k * lda] * A[j + k * lda];
        if (!(d > 0.0)) return j + 1;
        d = sqrt(d);
        A[j + j * lda] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i + j * lda];
            for (int k = 0; k < j; ++k) s -= A[i + k * lda] * A[j + k * lda];
            A[i + j * lda] = s / d;
        }
    }
    return 0;
}

/* Eigen::LLT<MatrixXd> (src/preference-regressor.cpp:162,290): right-looking blocked, lower. */
int slso_cholesky(double* A, int n) {
    const int NB = 64;
    const long lda = n;
    for (int j = 0; j < n; j += NB) {
        const int jb = (n - j < NB) ? n - j : NB;
        int info = chol_unblocked(A + j + j * lda, jb, lda);
        if (info) return j + info;
        const int m = n - j - jb;
        if (m <= 0) continue;
        /* panel: L21 = A21 L11^-T */
#pragma omp parallel for schedule(static)
        for (int i = j + jb; i < n; ++i) {
            for (int c = 0; c < jb; ++c) {
                double s = A[i + (j + c) * lda];
                for (int k = 0; k < c; ++k) s -= A[i + (j + k) * lda] * A[(j + c) + (j + k) * lda];
                A[i + (j + c) * lda] = s / A[(j + c) + (j + c) * lda];
            }
        }
        /* trailing: A22 -= L21 L21^T (lower) */
#pragma omp parallel for schedule(dynamic, 4)
        for (int c = j + jb; c < n; ++c) {
            double* col = A + c * lda;
            for (int k = 0; k < jb; ++k) {
                const double l = A[c + (j + k) * lda];
                const double* pk = A + (j + k) * lda;
                for (int i = c; i < n; ++i) col[i] -= pk[i] * l;
            }
        }
    }
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < j; ++i) A[i + j * lda] = 0.0;
    return 0;
}

void slso_trsm_lower(const double* L, int n, double* B, int nrhs) {
#pragma omp parallel for schedule(static) if (nrhs > 1)
    for (int c = 0; c < nrhs; ++c) {
        double* b = B + (long)c * n;
        for (int j = 0; j < n; ++j) {
            const double v = b[j] / L[j + (long)j * n];
            b[j] = v;
            const double* col = L + (long)j * n;
            for (int i = j + 1; i < n; ++i) b[i] -= col[i] * v;
        }
    }
}

void slso_trsm_lower_t(const double* L, int n, double* B, int nrhs) {
#pragma omp parallel for schedule(static) if (nrhs > 1)
    for (int c = 0; c < nrhs; ++c) {
        double* b = B + (long)c * n;
        for (int j = n - 1; j >= 0; --j) {
            const double* col = L + (long)j * n;
            double s = b[j];
            for (int i = j + 1; i < n; ++i) s -= col[i] * b[i];
            b[j] = s / col[j];
        }
    }
}

/* Eigen::LLT::solve (src/preference-regressor.cpp:165,296,309,320,329) */
void slso_chol_solve(const double* L, int n, double* B, int nrhs) {
    slso_trsm_lower(L, n, B, nrhs);
    slso_trsm_lower_t(L, n, B, nrhs);
}

/* MatrixXd::inverse() -> PartialPivLU (src/gaussian-process-regressor.cpp:159,211,231) */
int slso_lu_inverse(const double* A, int n, double* Ainv) {
    const long ld = n;
    double* LU = dalloc((long)n * n);
    int* piv = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    memcpy(LU, A, sizeof(double) * (size_t)n * n);
    for (int j = 0; j < n; ++j) {
        int p = j;
        double best = fabs(LU[j + j * ld]);
        for (int i = j + 1; i < n; ++i)
            if (fabs(LU[i + j * ld]) > best) { best = fabs(LU[i + j * ld]); p = i; }
        piv[j] = p;
        if (best == 0.0) { free(LU); free(piv); return j + 1; }
        if (p != j)
            for (int c = 0; c < n; ++c) { double t = LU[j + c * ld]; LU[j + c * ld] = LU[p + c * ld]; LU[p + c * ld] = t; }
        const double d = LU[j + j * ld];
        for (int i = j + 1; i < n; ++i) LU[i + j * ld] /= d;
        for (int c = j + 1; c < n; ++c) {
            const double u = LU[j + c * ld];
            if (u != 0.0)
                for (int i = j + 1; i < n; ++i) LU[i + c * ld] -= LU[i + j * ld] * u;
        }
    }
    /* solve A X = I column by column */
#pragma omp parallel for schedule(static)
    for (int c = 0; c < n; ++c) {
        double* x = Ainv + c * ld;
        for (int i = 0; i < n; ++i) x[i] = 0.0;
        x[c] = 1.0;
        for (int j = 0; j < n; ++j) { const int p = piv[j]; if (p != j) { double t = x[j]; x[j] = x[p]; x[p] = t; } }
        for (int j = 0; j < n; ++j) { const double v = x[j]; if (v != 0.0) for (int i = j + 1; i < n; ++i) x[i] -= LU[i + j * ld] * v; }
        for (int j = n - 1; j >= 0; --j) { const double v = x[j] / LU[j + j * ld]; x[j] = v; for (int i = 0; i < j; ++i) x[i] -= LU[i + j * ld] * v; }
    }
    free(LU);
    free(piv);
    return 0;
}

void slso_spd_inverse_from_chol(const double* L, int n, double* Ainv) {
    const long ld = n;
#pragma omp parallel for schedule(dynamic, 8)
    for (int c = 0; c < n; ++c) {
        double* x = Ainv + c * ld;
        for (int i = 0; i < n; ++i) x[i] = 0.0;
        x[c] = 1.0;
        /* forward solve touches rows >= c only */
        for (int j = c; j < n; ++j) {
            const double v = x[j] / L[j + j * ld];
            x[j] = v;
            const double* col = L + j * ld;
            for (int i = j + 1; i < n; ++i) x[i] -= col[i] * v;
        }
        for (int j = n - 1; j >= 0; --j) {
            const double* col = L + j * ld;
            double s = x[j];
            for (int i = j + 1; i < n; ++i) s -= col[i] * x[i];
            x[j] = s / col[j];
        }
    }
}

/* mathtoolbox::CalcLogDetOfSymmetricPositiveDefiniteMatrix
 * (src/gaussian-process-regressor.cpp:175, src/preference-regressor.cpp:166) */
double slso_logdet_from_chol(const double* L, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += log(L[i + (long)i * n]);
    return 2.0 * s;
}

/* ------------------------------------------------------------------------- */
/* mathtoolbox scalar functions                                               */
/* ------------------------------------------------------------------------- */

/* GetLogOfLogNormalDist (src/gaussian-process-regressor.cpp:53-63, src/preference-regressor.cpp:184-190) */
double slso_log_lognormal(double x, double mu, double sigma2) {
    const double lx = log(x);
    return -lx - 0.5 * log(2.0 * PI * sigma2) - (lx - mu) * (lx - mu) / (2.0 * sigma2);
}
/* GetLogOfLogNormalDistDerivative (src/gaussian-process-regressor.cpp:38-48, src/preference-regressor.cpp:71,108) */
double slso_log_lognormal_derivative(double x, double mu, double sigma2) { return (mu - sigma2 - log(x)) / (sigma2 * x); }

double slso_norm_pdf(double u) { return exp(-0.5 * u * u) / sqrt(2.0 * PI); }
double slso_norm_cdf(double u) { return 0.5 * erfc(-u / sqrt(2.0)); }

/* utils::CalcBtl, include/sequential-line-search/utils.hpp:25-29 (no max-subtraction, as the reference) */
double slso_btl(const double* f, int n, double scale) {
    double sum = 0.0;
    for (int i = 0; i < n; ++i) sum += exp(f[i] / scale);
    return exp(f[0] / scale) / sum;
}
/* utils::CalcBtlDerivative, utils.hpp:31-52 */
void slso_btl_derivative(const double* f, int n, double scale, double* d) {
    const double btl = slso_btl(f, n, scale);
    const double tmp = -btl * btl / scale;
    double sum = 0.0;
    for (int i = 1; i < n; ++i) sum += exp((f[i] - f[0]) / scale);
    d[0] = tmp * (-sum);
    for (int i = 1; i < n; ++i) d[i] = tmp * exp((f[i] - f[0]) / scale);
}

/* ------------------------------------------------------------------------- */
/* regressor object                                                           */
/* ------------------------------------------------------------------------- */

slso_regressor* slso_regressor_create(int reg_type, int kernel, const double* X, int D, int N, const double* y,
                                      const double* theta, double b) {
    slso_regressor* r = (slso_regressor*)calloc(1, sizeof(slso_regressor));
    r->reg_type = reg_type; r->kernel = kernel; r->D = D; r->N = N; r->b = b;
    r->X = dalloc((long)D * N); memcpy(r->X, X, sizeof(double) * (size_t)D * N);
    r->y = dalloc(N); memcpy(r->y, y, sizeof(double) * (size_t)N);
    r->theta = dalloc(D + 1); memcpy(r->theta, theta, sizeof(double) * (size_t)(D + 1));
    if (N == 0) return r;
    r->K = dalloc((long)N * N);
    slso_calc_large_ky(kernel, X, D, N, theta, b, r->K);   /* gaussian-process-regressor.cpp:230 / preference-regressor.cpp:289 */
    r->L = dalloc((long)N * N);
    memcpy(r->L, r->K, sizeof(double) * (size_t)N * N);
    if (slso_cholesky(r->L, N) != 0) { /* not SPD: leave L as NaNs so that tests notice */
        for (long i = 0; i < (long)N * N; ++i) r->L[i] = NAN;
    }
    if (reg_type == SLSO_REG_GPR && N <= 1024) {          /* m_K_y_inv = m_K_y.inverse(), :231 (as-written mode only) */
        r->Kinv = dalloc((long)N * N);
        slso_lu_inverse(r->K, N, r->Kinv);
    }
    r->KinvC = dalloc((long)N * N);
    slso_spd_inverse_from_chol(r->L, N, r->KinvC);
    r->alpha = dalloc(N);
    memcpy(r->alpha, y, sizeof(double) * (size_t)N);
    slso_chol_solve(r->L, N, r->alpha, 1);
    /* hoisted PredictMaximumPointFromData (regressor.cpp:29-43): mu(x_i) = k_i^T K^-1 y = y_i - b alpha_i */
    int best = 0; double bv = -INFINITY;
    for (int i = 0; i < N; ++i) {
        const double m = r->y[i] - b * r->alpha[i];
        if (m > bv) { bv = m; best = i; }
    }
    r->best_index = best; r->mu_best = bv;
    return r;
}

/* hoisted PredictMaximumPointFromData: index and mu(x_best) cached at creation (no O(N^3) loop) */
int slso_regressor_best_index(const slso_regressor* r) { return r->best_index; }
double slso_regressor_mu_best(const slso_regressor* r) { return r->mu_best; }

void slso_regressor_free(slso_regressor* r) {
    if (!r) return;
    free(r->X); free(r->y); free(r->theta); free(r->K); free(r->Kinv); free(r->L); free(r->alpha); free(r->KinvC);
    free(r);
}

/* K^-1 v exactly as each reference class does it */
static void apply_Kinv_as_written(const slso_regressor* r, const double* v, double* out) {
    const int N = r->N;
    if (r->reg_type == SLSO_REG_GPR && r->Kinv) {
        /* m_K_y_inv * v as a column-major GEMV (column axpy sweeps, the traversal Eigen's product kernel uses; a
           row-wise dot form would stride by N and mis-state the reference's CPU cost) */
        for (int i = 0; i < N; ++i) out[i] = 0.0;
        for (int k = 0; k < N; ++k) {
            const double vk = v[k];
            const double* col = r->Kinv + (long)k * N;
            for (int i = 0; i < N; ++i) out[i] += col[i] * vk;
        }
    } else {
        memcpy(out, v, sizeof(double) * (size_t)N);
        slso_chol_solve(r->L, N, out, 1);          /* m_K_llt.solve(v) */
    }
}

/* PredictMu: gaussian-process-regressor.cpp:234-239 / preference-regressor.cpp:293-297 */
double slso_predict_mu(const slso_regressor* r, const double* x) {
    const int N = r->N;
    double* k = dalloc(N); double* t = dalloc(N);
    slso_calc_small_k(r->kernel, x, r->X, r->D, N, r->theta, k);
    apply_Kinv_as_written(r, r->y, t);
    double s = 0.0;
    for (int i = 0; i < N; ++i) s += k[i] * t[i];
    free(k); free(t);
    return s;
}

/* PredictSigma: :241-255 / :299-313 */
double slso_predict_sigma(const slso_regressor* r, const double* x) {
    const int N = r->N;
    double* k = dalloc(N); double* t = dalloc(N);
    slso_calc_small_k(r->kernel, x, r->X, r->D, N, r->theta, k);
    apply_Kinv_as_written(r, k, t);
    double s = 0.0;
    for (int i = 0; i < N; ++i) s += k[i] * t[i];
    free(k); free(t);
    const double sigma_2 = r->theta[0] - s;
    return sigma_2 < 0 ? 0.0 : sqrt(sigma_2);
}

/* PredictMuDerivative: :257-263 / :315-321 */
void slso_predict_mu_derivative(const slso_regressor* r, const double* x, double* g) {
    const int N = r->N, D = r->D;
    double* J = dalloc((long)D * N); double* t = dalloc(N);
    slso_calc_small_k_small_x_derivative(r->kernel, x, r->X, D, N, r->theta, J);
    apply_Kinv_as_written(r, r->y, t);
    for (int d = 0; d < D; ++d) { double s = 0.0; for (int i = 0; i < N; ++i) s += J[d + (long)i * D] * t[i]; g[d] = s; }
    free(J); free(t);
}

/* PredictSigmaDerivative: :265-272 / :323-330 (no sigma guard, like the reference) */
void slso_predict_sigma_derivative(const slso_regressor* r, const double* x, double* g) {
    const int N = r->N, D = r->D;
    double* J = dalloc((long)D * N); double* k = dalloc(N); double* t = dalloc(N);
    slso_calc_small_k_small_x_derivative(r->kernel, x, r->X, D, N, r->theta, J);
    slso_calc_small_k(r->kernel, x, r->X, D, N, r->theta, k);
    const double sigma = slso_predict_sigma(r, x);
    apply_Kinv_as_written(r, k, t);
    for (int d = 0; d < D; ++d) { double s = 0.0; for (int i = 0; i < N; ++i) s += J[d + (long)i * D] * t[i]; g[d] = -(1.0 / sigma) * s; }
    free(J); free(k); free(t);
}

/* Regressor::PredictMaximumPointFromData, src/regressor.cpp:29-43 (N x PredictMu) */
int slso_predict_maximum_point_from_data(const slso_regressor* r, double* x_best) {
    int best = 0; double bv = -INFINITY;
    for (int i = 0; i < r->N; ++i) {
        const double f = slso_predict_mu(r, r->X + (long)i * r->D);
        if (f > bv) { bv = f; best = i; }   /* Eigen maxCoeff: first maximum */
    }
    if (x_best) memcpy(x_best, r->X + (long)best * r->D, sizeof(double) * (size_t)r->D);
    return best;
}

/* mathtoolbox::GetExpectedImprovement / GetGaussianProcessUpperConfidenceBound (Appendix A) */
static double ei_value(double mu, double sigma, double mu_best) {
    const double diff = mu - mu_best;
    const double u = diff / sigma;
    const double ei = diff * slso_norm_cdf(u) + sigma * slso_norm_pdf(u);
    return (sigma < 1e-10 || isnan(ei)) ? 0.0 : ei;
}
static void ei_grad(double mu, double sigma, double mu_best, const double* dmu, const double* dsigma, int D, double* g) {
    const double diff = mu - mu_best;
    const double u = diff / sigma;
    const double Phi = slso_norm_cdf(u), phi = slso_norm_pdf(u);
    int bad = (sigma < 1e-10);
    for (int d = 0; d < D; ++d) { g[d] = Phi * dmu[d] + phi * dsigma[d]; if (isnan(g[d])) bad = 1; }
    if (bad) for (int d = 0; d < D; ++d) g[d] = 0.0;
}

/* acquisition_func::CalcAcquisitionValue, src/acquisition-function.cpp:170-198 */
double slso_acq_value_as_written(const slso_regressor* r, const double* x, int acq, double ucb_h) {
    if (r->N == 0) return 0.0;
    if (acq == SLSO_ACQ_EI) {
        double* xb = dalloc(r->D);
        slso_predict_maximum_point_from_data(r, xb);
        const double v = ei_value(slso_predict_mu(r, x), slso_predict_sigma(r, x), slso_predict_mu(r, xb));
        free(xb);
        return v;
    }
    return slso_predict_mu(r, x) + ucb_h * slso_predict_sigma(r, x);
}

/* acquisition_func::CalcAcquisitionValueDerivative, src/acquisition-function.cpp:200-230 */
void slso_acq_derivative_as_written(const slso_regressor* r, const double* x, int acq, double ucb_h, double* g) {
    const int D = r->D;
    if (r->N == 0) { for (int d = 0; d < D; ++d) g[d] = 0.0; return; }
    double* dm = dalloc(D); double* ds = dalloc(D);
    slso_predict_mu_derivative(r, x, dm);
    slso_predict_sigma_derivative(r, x, ds);
    if (acq == SLSO_ACQ_EI) {
        double* xb = dalloc(D);
        slso_predict_maximum_point_from_data(r, xb);
        ei_grad(slso_predict_mu(r, x), slso_predict_sigma(r, x), slso_predict_mu(r, xb), dm, ds, D, g);
        free(xb);
    } else {
        for (int d = 0; d < D; ++d) g[d] = dm[d] + ucb_h * ds[d];
    }
    free(dm); free(ds);
}

/* ------------------------------------------------------------------------- */
/* hoisted batched evaluation                                                 */
/* ------------------------------------------------------------------------- */

/* per-candidate quantities with w = K^-1 k from the Cholesky-based inverse */
static void eval_block(const slso_regressor* r, const double* Xs, int m0, int mb, double* mu, double* sigma, double* dmu,
                       double* dsigma, double* kbuf /*N*mb*/, double* wbuf /*N*mb*/, double* cbuf /*N*mb*/) {
    const int N = r->N, D = r->D;
    const double a = r->theta[0];
    for (int c = 0; c < mb; ++c) {
        const double* x = Xs + (long)(m0 + c) * D;
        for (int i = 0; i < N; ++i) {
            const double q = scaled_sqdist(x, r->X + (long)i * D, r->theta, D);
            if (r->kernel == SLSO_KERNEL_ARD_SE) {
                const double k = a * exp(-0.5 * q);
                kbuf[c * (long)N + i] = k; cbuf[c * (long)N + i] = k;
            } else {
                const double s = sqrt(5.0 * q), e = exp(-s);
                kbuf[c * (long)N + i] = a * (1.0 + s + (5.0 / 3.0) * q) * e;
                cbuf[c * (long)N + i] = a * (5.0 / 3.0) * (1.0 + s) * e;
            }
        }
    }
    /* W = KinvC * Kblock : row i of KinvC == column i (symmetric) */
    for (int i = 0; i < N; ++i) {
        const double* col = r->KinvC + (long)i * N;
        for (int c = 0; c < mb; ++c) {
            const double* k = kbuf + c * (long)N;
            double s = 0.0;
            for (int j = 0; j < N; ++j) s += col[j] * k[j];
            wbuf[c * (long)N + i] = s;
        }
    }
    for (int c = 0; c < mb; ++c) {
        const double* x = Xs + (long)(m0 + c) * D;
        const double* k = kbuf + c * (long)N; const double* w = wbuf + c * (long)N; const double* cc = cbuf + c * (long)N;
        double smu = 0.0, skw = 0.0;
        for (int i = 0; i < N; ++i) { smu += k[i] * r->alpha[i]; skw += k[i] * w[i]; }
        const double s2 = a - skw;
        const double sg = s2 < 0 ? 0.0 : sqrt(s2);
        if (mu) mu[m0 + c] = smu;
        if (sigma) sigma[m0 + c] = sg;
        if (dmu || dsigma) {
            for (int d = 0; d < D; ++d) {
                const double l2 = r->theta[1 + d] * r->theta[1 + d];
                double gm = 0.0, gs = 0.0;
                for (int i = 0; i < N; ++i) {
                    const double j = -cc[i] * (x[d] - r->X[d + (long)i * D]) / l2;   /* dk_i/dx_d */
                    gm += j * r->alpha[i];
                    gs += j * w[i];
                }
                if (dmu) dmu[d + (long)(m0 + c) * D] = gm;
                if (dsigma) dsigma[d + (long)(m0 + c) * D] = -(1.0 / sg) * gs;
            }
        }
    }
}

static void eval_all(const slso_regressor* r, const double* Xs, int M, double* mu, double* sigma, double* dmu, double* dsigma) {
    const int MB = 16;
    const int N = r->N;
#pragma omp parallel
    {
        double* kbuf = dalloc((long)N * MB); double* wbuf = dalloc((long)N * MB); double* cbuf = dalloc((long)N * MB);
#pragma omp for schedule(dynamic, 1)
        for (int m0 = 0; m0 < M; m0 += MB) {
            const int mb = (M - m0 < MB) ? M - m0 : MB;
            eval_block(r, Xs, m0, mb, mu, sigma, dmu, dsigma, kbuf, wbuf, cbuf);
        }
        free(kbuf); free(wbuf); free(cbuf);
    }
}

void slso_predict_batch(const slso_regressor* r, const double* Xs, int M, double* mu, double* sigma) {
    eval_all(r, Xs, M, mu, sigma, NULL, NULL);
}
void slso_predict_grad_batch(const slso_regressor* r, const double* Xs, int M, double* dmu, double* dsigma) {
    eval_all(r, Xs, M, NULL, NULL, dmu, dsigma);
}

void slso_acq_eval_batch(const slso_regressor* r, const double* Xs, int M, int acq, double ucb_h, double* val, double* grad) {
    const int D = r->D;
    if (r->N == 0) {   /* acquisition-function.cpp:176-179,206-209 */
        for (int m = 0; m < M; ++m) val[m] = 0.0;
        if (grad) for (long i = 0; i < (long)D * M; ++i) grad[i] = 0.0;
        return;
    }
    double* mu = dalloc(M); double* sg = dalloc(M);
    double* dm = grad ? dalloc((long)D * M) : NULL; double* ds = grad ? dalloc((long)D * M) : NULL;
    eval_all(r, Xs, M, mu, sg, dm, ds);
    for (int m = 0; m < M; ++m) {
        if (acq == SLSO_ACQ_EI) {
            val[m] = ei_value(mu[m], sg[m], r->mu_best);
            if (grad) ei_grad(mu[m], sg[m], r->mu_best, dm + (long)m * D, ds + (long)m * D, D, grad + (long)m * D);
        } else {
            val[m] = mu[m] + ucb_h * sg[m];
            if (grad) for (int d = 0; d < D; ++d) grad[d + (long)m * D] = dm[d + (long)m * D] + ucb_h * ds[d + (long)m * D];
        }
    }
    free(mu); free(sg); free(dm); free(ds);
}

/* ------------------------------------------------------------------------- */
/* multi-start bounded L-BFGS, lock-step over the S starts      */
/* ------------------------------------------------------------------------- */

void slso_lbfgs_default_opts(slso_lbfgs_opts* o) {
    o->history = 6; o->c1 = 1e-4; o->shrink = 0.5; o->gtol = 0.0; o->max_backtracks = 20; o->ftol_rel = 0.0; o->xtol_rel = 0.0;
}

typedef struct {
    double *x, *g, *d, *xt;      /* D each */
    double *S, *Y, *rho;         /* m x D, m x D, m */
    double f, t;
    int hlen, hpos, nbt, done, need_dir;
} lb_state;

/* direction for minimising phi = -acq from state (x, g); returns 0 if stationary */
static int lb_direction(lb_state* s, int D, int m, double gtol) {
    double* pg = s->xt; /* scratch */
    double pgmax = 0.0, pgn2 = 0.0;
    for (int d = 0; d < D; ++d) {
        double v = s->g[d];
        if ((s->x[d] <= 0.0 && v > 0.0) || (s->x[d] >= 1.0 && v < 0.0)) v = 0.0;
        pg[d] = v;
        if (fabs(v) > pgmax) pgmax = fabs(v);
        pgn2 += v * v;
    }
    if (!(pgmax > gtol)) return 0;
    double al[64];
    for (int d = 0; d < D; ++d) s->d[d] = pg[d];
    for (int h = 0; h < s->hlen; ++h) {
        const int idx = (s->hpos - 1 - h + 2 * m) % m;
        double dot = 0.0;
        for (int d = 0; d < D; ++d) dot += s->S[idx * D + d] * s->d[d];
        al[h] = s->rho[idx] * dot;
        for (int d = 0; d < D; ++d) s->d[d] -= al[h] * s->Y[idx * D + d];
    }
    double gamma;
    if (s->hlen > 0) {
        const int idx = (s->hpos - 1 + m) % m;
        double sy = 0.0, yy = 0.0;
        for (int d = 0; d < D; ++d) { sy += s->S[idx * D + d] * s->Y[idx * D + d]; yy += s->Y[idx * D + d] * s->Y[idx * D + d]; }
        gamma = sy / yy;
    } else {
        const double n = sqrt(pgn2);
        gamma = 1.0 / (n > 1.0 ? n : 1.0);
    }
    for (int d = 0; d < D; ++d) s->d[d] *= gamma;
    for (int h = s->hlen - 1; h >= 0; --h) {
        const int idx = (s->hpos - 1 - h + 2 * m) % m;
        double dot = 0.0;
        for (int d = 0; d < D; ++d) dot += s->Y[idx * D + d] * s->d[d];
        const double beta = s->rho[idx] * dot;
        for (int d = 0; d < D; ++d) s->d[d] += s->S[idx * D + d] * (al[h] - beta);
    }
    double gd = 0.0;
    for (int d = 0; d < D; ++d) { s->d[d] = (pg[d] == 0.0) ? 0.0 : -s->d[d]; gd += pg[d] * s->d[d]; }
    if (!(gd < 0.0)) {
        s->hlen = 0;
        const double n = sqrt(pgn2);
        gamma = 1.0 / (n > 1.0 ? n : 1.0);
        gd = 0.0;
        for (int d = 0; d < D; ++d) { s->d[d] = -gamma * pg[d]; gd += pg[d] * s->d[d]; }
        if (!(gd < 0.0)) return 0;
    }
    s->t = 1.0; s->nbt = 0;
    return 1;
}

/* diag_margin / diag_eval (S each, may be NULL): per start, the smallest relative distance of an Armijo test from its
   threshold, |ft - (f + c1 g.s)| / max(|f|, |ft|), over the run, and the evaluation at which it occurred.  A start whose
   margin is at rounding level may take the other branch on an implementation that sums in a different order (the HIP
   path): tests use this to tell such rounding-induced flips from real disagreements. */
static int acq_maximize_impl(const slso_regressor* r, int acq, double ucb_h, const double* starts, int S, int n_local,
                             const slso_lbfgs_opts* opts_in, double* x_out, double* val_out, double* x_stars, double* y_stars,
                             int n_threads, double* diag_margin, int* diag_eval) {
    const int D = r->D;
    if (diag_margin) for (int i = 0; i < S; ++i) { diag_margin[i] = INFINITY; if (diag_eval) diag_eval[i] = -1; }
    slso_lbfgs_opts o;
    if (opts_in) o = *opts_in; else slso_lbfgs_default_opts(&o);
    const int m = o.history > 64 ? 64 : o.history;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
    lb_state* st = (lb_state*)calloc((size_t)S, sizeof(lb_state));
    double* XT = dalloc((long)D * S); double* val = dalloc(S); double* grad = dalloc((long)D * S);
    for (int i = 0; i < S; ++i) {
        lb_state* s = &st[i];
        s->x = dalloc(D); s->g = dalloc(D); s->d = dalloc(D); s->xt = dalloc(D);
        s->S = dalloc((long)m * D); s->Y = dalloc((long)m * D); s->rho = dalloc(m);
        for (int d = 0; d < D; ++d) {
            double v = starts[d + (long)i * D];
            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            s->x[d] = v; XT[d + (long)i * D] = v;
        }
    }
    /* evaluation 1: the starts themselves */
    slso_acq_eval_batch(r, XT, S, acq, ucb_h, val, grad);
    for (int i = 0; i < S; ++i) {
        lb_state* s = &st[i];
        s->f = -val[i];
        for (int d = 0; d < D; ++d) s->g[d] = -grad[d + (long)i * D];
        s->need_dir = 1;
    }
    for (int ev = 1; ev < n_local; ++ev) {
        /* propose trial points */
        for (int i = 0; i < S; ++i) {
            lb_state* s = &st[i];
            if (!s->done && s->need_dir) {
                if (!lb_direction(s, D, m, o.gtol)) s->done = 1;
                s->need_dir = 0;
            }
            for (int d = 0; d < D; ++d) {
                double v = s->x[d];
                if (!s->done) { v = s->x[d] + s->t * s->d[d]; v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
                s->xt[d] = v; XT[d + (long)i * D] = v;
            }
        }
        slso_acq_eval_batch(r, XT, S, acq, ucb_h, val, grad);
        for (int i = 0; i < S; ++i) {
            lb_state* s = &st[i];
            if (s->done) continue;
            const double ft = -val[i];
            double gs = 0.0, ss = 0.0;
            for (int d = 0; d < D; ++d) { const double sd = s->xt[d] - s->x[d]; gs += s->g[d] * sd; ss += sd * sd; }
            if (ss == 0.0) { s->done = 1; continue; }
            if (diag_margin) {
                const double sc = fmax(fabs(s->f), fabs(ft));
                const double mg = fabs(ft - (s->f + o.c1 * gs)) / (sc > 0.0 ? sc : 1.0);
                if (mg < diag_margin[i]) { diag_margin[i] = mg; if (diag_eval) diag_eval[i] = ev + 1; }
            }
            if (ft <= s->f + o.c1 * gs) {
                double sy = 0.0, yy = 0.0;
                const int idx = s->hpos;
                for (int d = 0; d < D; ++d) {
                    const double sd = s->xt[d] - s->x[d];
                    const double yd = -grad[d + (long)i * D] - s->g[d];
                    s->S[idx * D + d] = sd; s->Y[idx * D + d] = yd;
                    sy += sd * yd; yy += yd * yd;
                }
                if (sy > 1e-10 * yy && sy > 0.0) {
                    s->rho[idx] = 1.0 / sy;
                    s->hpos = (s->hpos + 1) % m;
                    if (s->hlen < m) s->hlen++;
                }
                /* NLopt's relative stopping tests on the accepted step */
                if (o.ftol_rel > 0.0 && (fabs(ft - s->f) < o.ftol_rel * 0.5 * (fabs(ft) + fabs(s->f)) || ft == s->f)) s->done = 1;
                if (o.xtol_rel > 0.0) {
                    int moved = 0;
                    for (int d = 0; d < D; ++d) {
                        const double xo = s->x[d], xn = s->xt[d];
                        if (!(fabs(xn - xo) < o.xtol_rel * 0.5 * (fabs(xn) + fabs(xo)) || xn == xo)) moved = 1;
                    }
                    if (!moved) s->done = 1;
                }
                for (int d = 0; d < D; ++d) { s->x[d] = s->xt[d]; s->g[d] = -grad[d + (long)i * D]; }
                s->f = ft;
                s->need_dir = 1;
            } else {
                s->t *= o.shrink;
                s->nbt++;
                if (s->nbt > o.max_backtracks) s->done = 1;
            }
        }
    }
    /* y_stars.maxCoeff(&index), src/acquisition-function.cpp:146-153 (first maximum) */
    int best = 0; double bv = -INFINITY;
    for (int i = 0; i < S; ++i) {
        const double v = -st[i].f;
        if (y_stars) y_stars[i] = v;
        if (x_stars) memcpy(x_stars + (long)i * D, st[i].x, sizeof(double) * (size_t)D);
        if (v > bv) { bv = v; best = i; }
    }
    if (x_out) memcpy(x_out, st[best].x, sizeof(double) * (size_t)D);
    if (val_out) *val_out = bv;
    for (int i = 0; i < S; ++i) { lb_state* s = &st[i]; free(s->x); free(s->g); free(s->d); free(s->xt); free(s->S); free(s->Y); free(s->rho); }
    free(st); free(XT); free(val); free(grad);
    return best;
}

int slso_acq_maximize(const slso_regressor* r, int acq, double ucb_h, const double* starts, int S, int n_local,
                      const slso_lbfgs_opts* opts_in, double* x_out, double* val_out, double* x_stars, double* y_stars,
                      int n_threads) {
    return acq_maximize_impl(r, acq, ucb_h, starts, S, n_local, opts_in, x_out, val_out, x_stars, y_stars, n_threads, NULL, NULL);
}
int slso_acq_maximize_diag(const slso_regressor* r, int acq, double ucb_h, const double* starts, int S, int n_local,
                           const slso_lbfgs_opts* opts_in, double* x_out, double* val_out, double* x_stars, double* y_stars,
                           int n_threads, double* armijo_margin, int* armijo_eval) {
    return acq_maximize_impl(r, acq, ucb_h, starts, S, n_local, opts_in, x_out, val_out, x_stars, y_stars, n_threads,
                             armijo_margin, armijo_eval);
}

/* ------------------------------------------------------------------------- */
/* GP MAP objective (src/gaussian-process-regressor.cpp:18-24, 36-193)         */
/* ------------------------------------------------------------------------- */

double slso_gp_map_objective(int kernel, const double* X, int D, int N, const double* y, const double* x, double* grad,
                             int as_written) {
    const double a_mu = log(0.5), a_s2 = 0.5, b_mu = log(1e-4), b_s2 = 0.5, r_mu = log(0.5), r_s2 = 0.5;   /* :18-24 */
    const double a = x[0], b = x[1];
    double* theta = dalloc(D + 1);
    theta[0] = a;
    for (int d = 0; d < D; ++d) theta[1 + d] = x[2 + d];
    double* K = dalloc((long)N * N);
    slso_calc_large_ky(kernel, X, D, N, theta, b, K);                                  /* :158 */
    double* Kinv = dalloc((long)N * N);
    double* L = dalloc((long)N * N);
    memcpy(L, K, sizeof(double) * (size_t)N * N);
    slso_cholesky(L, N);
    if (as_written) slso_lu_inverse(K, N, Kinv);                                        /* :159 */
    else slso_spd_inverse_from_chol(L, N, Kinv);
    double* al = dalloc(N);
    if (as_written) { for (int i = 0; i < N; ++i) { double s = 0.0; for (int k = 0; k < N; ++k) s += Kinv[i + (long)k * N] * y[k]; al[i] = s; } }
    else { memcpy(al, y, sizeof(double) * (size_t)N); slso_chol_solve(L, N, al, 1); }
    double yKy = 0.0;
    for (int i = 0; i < N; ++i) yKy += y[i] * al[i];
    const double term1 = -0.5 * yKy;                                                    /* :174 */
    const double term2 = -0.5 * slso_logdet_from_chol(L, N);                            /* :175 */
    const double term3 = -0.5 * N * log(2.0 * PI);                                      /* :176 */
    double reg = slso_log_lognormal(a, a_mu, a_s2) + slso_log_lognormal(b, b_mu, b_s2); /* :179-190 */
    for (int d = 0; d < D; ++d) reg += slso_log_lognormal(x[2 + d], r_mu, r_s2);
    if (grad) {
        if (as_written) {
            double* T = dalloc((long)(D + 1) * N * N);
            slso_calc_large_ky_theta_derivative(kernel, X, D, N, theta, T);             /* calc_grad_theta :84 */
            for (int p = 0; p <= D; ++p) {
                const double* Tp = T + (long)p * N * N;
                double t1 = 0.0, tr = 0.0;
                for (int j = 0; j < N; ++j) {
                    double s = 0.0;
                    for (int i = 0; i < N; ++i) { s += al[i] * Tp[i + (long)j * N]; tr += Kinv[j + (long)i * N] * Tp[i + (long)j * N]; }
                    t1 += s * al[j];
                }
                const double prior = (p == 0) ? slso_log_lognormal_derivative(a, a_mu, a_s2)
                                              : slso_log_lognormal_derivative(x[1 + p], r_mu, r_s2);
                const double gval = 0.5 * t1 - 0.5 * tr + prior;                        /* :92-102 */
                if (p == 0) grad[0] = gval; else grad[1 + p] = gval;                    /* calc_grad :120-124 */
            }
            free(T);
            double aa = 0.0, tr = 0.0;                                                  /* calc_grad_b :66-77, dK/db = I */
            for (int i = 0; i < N; ++i) { aa += al[i] * al[i]; tr += Kinv[i + (long)i * N]; }
            grad[1] = 0.5 * aa - 0.5 * tr + slso_log_lognormal_derivative(b, b_mu, b_s2);
        } else {
            /* fused: W = alpha alpha^T - K^-1;  g_p = 1/2 sum_jk W_jk dK_jk/dtheta_p */
            double* g = dalloc(D + 1);
            double* acc = dalloc(D + 1);
            for (int p = 0; p <= D; ++p) acc[p] = 0.0;
            for (int j = 0; j < N; ++j)
                for (int i = 0; i < N; ++i) {
                    const double w = al[i] * al[j] - Kinv[i + (long)j * N];
                    slso_kernel_theta_derivative(kernel, X + (long)i * D, X + (long)j * D, theta, D, g);
                    for (int p = 0; p <= D; ++p) acc[p] += w * g[p];
                }
            grad[0] = 0.5 * acc[0] + slso_log_lognormal_derivative(a, a_mu, a_s2);
            for (int d = 0; d < D; ++d) grad[2 + d] = 0.5 * acc[1 + d] + slso_log_lognormal_derivative(x[2 + d], r_mu, r_s2);
            double aa = 0.0, tr = 0.0;
            for (int i = 0; i < N; ++i) { aa += al[i] * al[i]; tr += Kinv[i + (long)i * N]; }
            grad[1] = 0.5 * aa - 0.5 * tr + slso_log_lognormal_derivative(b, b_mu, b_s2);
            free(g); free(acc);
        }
    }
    free(theta); free(K); free(Kinv); free(L); free(al);
    return term1 + term2 + term3 + reg;
}

/* ------------------------------------------------------------------------- */
/* preference-GP MAP objective (src/preference-regressor.cpp:129-259)          */
/* ------------------------------------------------------------------------- */

double slso_pref_objective(int kernel, const double* X, int D, int M, const unsigned* prefs_flat, const int* pref_offsets,
                           int n_prefs, const double* x, const slso_pref_cfg* cfg, double* grad) {
    const int use_map = cfg->use_map_hyperparams;
    const double* yv = x;
    const double a = use_map ? x[M + 0] : cfg->default_a;                               /* :139 */
    const double b = cfg->noiseless ? 0.0 : (use_map ? x[M + 1] : cfg->default_b);      /* :140-144 */
    double* theta = dalloc(D + 1);
    theta[0] = a;
    for (int d = 0; d < D; ++d) theta[1 + d] = use_map ? x[M + 2 + d] : cfg->default_r; /* :145-147 */

    double obj = 0.0;
    double ftmp[64], dtmp[64];
    for (int p = 0; p < n_prefs; ++p) {                                                 /* :151-154, calc_log_likelihood :118-126 */
        const int o = pref_offsets[p], n = pref_offsets[p + 1] - o;
        for (int i = 0; i < n; ++i) ftmp[i] = yv[prefs_flat[o + i]];
        obj += log(slso_btl(ftmp, n, cfg->btl_scale));
    }
    double* K = dalloc((long)M * M);
    slso_calc_large_ky(kernel, X, D, M, theta, b, K);                                   /* :160-161 (same matrix as the cached m_K when !use_map) */
    double* L = dalloc((long)M * M);
    memcpy(L, K, sizeof(double) * (size_t)M * M);
    slso_cholesky(L, M);                                                                /* :162 */
    double* Kinv_y = dalloc(M);
    memcpy(Kinv_y, yv, sizeof(double) * (size_t)M);
    slso_chol_solve(L, M, Kinv_y, 1);                                                   /* :165 */
    double yKy = 0.0;
    for (int i = 0; i < M; ++i) yKy += yv[i] * Kinv_y[i];
    obj += -0.5 * yKy - 0.5 * slso_logdet_from_chol(L, M) - 0.5 * M * log(2.0 * PI);    /* :166-170 */
    if (use_map) {                                                                      /* :175-192 */
        obj += slso_log_lognormal(a, log(cfg->default_a), cfg->prior_var);
        if (!cfg->noiseless) obj += slso_log_lognormal(b, log(cfg->default_b), cfg->prior_var);
        for (int d = 0; d < D; ++d) obj += slso_log_lognormal(theta[1 + d], log(cfg->default_r), cfg->prior_var);
    }
    if (grad) {
        for (int i = 0; i < M; ++i) grad[i] = 0.0;
        for (int p = 0; p < n_prefs; ++p) {                                             /* :202-216 */
            const int o = pref_offsets[p], n = pref_offsets[p + 1] - o;
            for (int i = 0; i < n; ++i) ftmp[i] = yv[prefs_flat[o + i]];
            const double btl = slso_btl(ftmp, n, cfg->btl_scale);
            slso_btl_derivative(ftmp, n, cfg->btl_scale, dtmp);
            for (int i = 0; i < n; ++i) grad[prefs_flat[o + i]] += dtmp[i] / btl;
        }
        for (int i = 0; i < M; ++i) grad[i] += -Kinv_y[i];                              /* :219 */
        if (use_map) {
            /* CalcObjectiveThetaDerivative :77-115 and ...NoiseLevelDerivative :53-74, as written:
             * term_1 = 1/2 a^T dK a ; term_2 = -1/2 tr(K^-1 dK) via LLT.solve(dK).trace() */
            double* T = dalloc((long)(D + 1) * M * M);
            slso_calc_large_ky_theta_derivative(kernel, X, D, M, theta, T);
            double* Kinv = dalloc((long)M * M);
            slso_spd_inverse_from_chol(L, M, Kinv);
            for (int p = 0; p <= D; ++p) {
                const double* Tp = T + (long)p * M * M;
                double t1 = 0.0, tr = 0.0;
                for (int j = 0; j < M; ++j) {
                    double s = 0.0;
                    for (int i = 0; i < M; ++i) { s += Kinv_y[i] * Tp[i + (long)j * M]; tr += Kinv[j + (long)i * M] * Tp[i + (long)j * M]; }
                    t1 += s * Kinv_y[j];
                }
                const double pm = (p == 0) ? cfg->default_a : cfg->default_r;
                const double prior = slso_log_lognormal_derivative(theta[p], log(pm), cfg->prior_var);
                const double gval = 0.5 * t1 - 0.5 * tr + prior;
                if (p == 0) grad[M + 0] = gval; else grad[M + 1 + p] = gval;
            }
            if (cfg->noiseless) {
                grad[M + 1] = 0.0;                                                      /* :237-238 */
            } else {
                double aa = 0.0, tr = 0.0;
                for (int i = 0; i < M; ++i) { aa += Kinv_y[i] * Kinv_y[i]; tr += Kinv[i + (long)i * M]; }
                grad[M + 1] = 0.5 * aa - 0.5 * tr + slso_log_lognormal_derivative(b, log(cfg->default_b), cfg->prior_var);
            }
            free(T); free(Kinv);
        }
    }
    free(theta); free(K); free(L); free(Kinv_y);
    return obj;
}

/* ------------------------------------------------------------------------- */
/* synthetic inputs: SplitMix64                                               */
/* ------------------------------------------------------------------------- */

static unsigned long long splitmix64(unsigned long long* s) {
    unsigned long long z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void slso_fill_uniform(double* out, long n, unsigned long long seed) {
    unsigned long long s = seed;
    for (long i = 0; i < n; ++i) out[i] = (double)(splitmix64(&s) >> 11) * (1.0 / 9007199254740992.0);
}
void slso_fill_normal(double* out, long n, unsigned long long seed) {
    unsigned long long s = seed;
    for (long i = 0; i < n; i += 2) {
        double u1 = ((double)(splitmix64(&s) >> 11) + 1.0) * (1.0 / 9007199254740992.0);
        double u2 = (double)(splitmix64(&s) >> 11) * (1.0 / 9007199254740992.0);
        const double rr = sqrt(-2.0 * log(u1));
        out[i] = rr * cos(2.0 * PI * u2);
        if (i + 1 < n) out[i + 1] = rr * sin(2.0 * PI * u2);
    }
}
