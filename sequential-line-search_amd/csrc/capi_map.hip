// C-ABI: MAP objectives (GP marginal likelihood, preference objective).  Device: Gram + Cholesky + K^-1 + fused
// gradient contraction; host: the O(D) prior terms and the O(#preferences) Bradley-Terry-Luce terms.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>

#include "common.hpp"
#include "kernels.hpp"

using namespace slsk;

#define SLS_TRY slsk::note_entry(); try {
#define SLS_CATCH                                   \
    }                                               \
    catch (const slsk::HipFail& f) { return f.code; } \
    catch (const std::exception& e) {               \
        slsk::set_error("exception: %s", e.what()); \
        return SLS_ERR_INVALID;                     \
    }                                               \
    return SLS_OK;

struct sls_nll {
    sls_ctx* ctx = nullptr;
    int D = 0, N = 0, Np = 0, Dp = 0, Dcols = 0, kernel = 0;
    DBuf X, y, XT, nx, L, Linv, Kinv, alpha, G, Y, svec, parts, gemv_part, small_in, small_out, small_info;
    DBuf XTr, small_kc;   // one-workgroup path (N <= 128): the raw design matrix transposed [i + d * 128]; pair scratch of the gradient
    std::vector<double> cached_theta;
    double cached_b = -1.0;
    bool have_factor = false;
    double logdet = 0.0;
    double ftol_rel = 0.0, xtol_rel = 0.0;   // sls_nll_set_tolerances
    // results of the one-workgroup evaluation (kernels_small.hip): page-locked host memory the kernel writes DIRECTLY (mapped),
    // read after the stream synchronisation -- no device-to-host copy call per evaluation
    double* small_host = nullptr;      // host address
    double* small_host_dev = nullptr;  // the same memory as the device sees it
    size_t small_host_bytes = 0, mo_out_bytes = 0;
    // device-resident MAP fit (map_opt_kernel): index image of the preference tuples (uploaded when it changes), the vectors of
    // a call, the optimiser state, the result block in mapped host memory, a page-locked staging block
    DBuf mo_idx, mo_vec, mo_state, mo_btl;
    // value-only objective for several parameter sets at once (sls_gp_nll_batch, N > 128): P bordered matrices, their scaled
    // design matrices, flag tables and results
    DBuf bt_L, bt_T, bt_XT, bt_nx, bt_il, bt_sync, bt_out;
    int bt_P = 0;
    std::vector<int> mo_idx_host;      // what mo_idx holds
    double* mo_out = nullptr;          // mapped: host address
    double* mo_out_dev = nullptr;
    char* mo_stage = nullptr;          // page-locked
    size_t mo_stage_bytes = 0;
    // Page-locked, device-MAPPED block of the tiled evaluation in flight: [Dcols] inverse length scales, [Np] targets, [8 + Dcols]
    // results.  The kernels read the length scales from it and write the results into it directly, and the targets are uploaded
    // only when they change (a MAP fit evaluates the same y hundreds of times): in steady state an evaluation makes no copy call
    // at all.  (With pageable buffers every hipMemcpyAsync was a blocking staged copy: two up, four or five back per evaluation.)
    double* il_stage = nullptr;        // host address
    double* il_stage_dev = nullptr;    // the same memory as the device sees it
    size_t il_stage_bytes = 0;
    bool y_on_device = false;          // y.p holds y_stage()'s contents
    double* y_stage() const { return il_stage + Dcols; }
    double* res_stage() const { return il_stage + Dcols + Np; }
    double* res_dev() const { return il_stage_dev + Dcols + Np; }
    ~sls_nll() {   // page-locked blocks go back to the context (sls_nll_destroy holds its lock)
        ctx->host_give(il_stage, il_stage_bytes, true);
        ctx->host_give(small_host, small_host_bytes, true);
        ctx->host_give(mo_out, mo_out_bytes, true);
        ctx->host_give(mo_stage, mo_stage_bytes, false);
    }
};

extern "C" int sls_nll_create(sls_ctx* ctx, const double* X, int D, int N, int kernel, sls_nll** out) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (ctx) lock_ = std::unique_lock<std::recursive_mutex>(ctx->mtx);
    SLS_REQUIRE(ctx && X && out && D >= 1 && N >= 1, "sls_nll_create: bad argument");
    SLS_REQUIRE(kernel == SLS_KERNEL_ARD_SQUARED_EXPONENTIAL || kernel == SLS_KERNEL_ARD_MATERN52, "unknown kernel %d", kernel);
    SLS_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<sls_nll> h(new sls_nll());
    h->ctx = ctx; h->D = D; h->N = N; h->kernel = kernel;
    h->Np = round_up(N, 128); h->Dp = round_up(D, 16); h->Dcols = round_up(D, 128);
    const size_t Np = h->Np;
    h->X.ensure((size_t)D * N); h->y.ensure(Np);
    h->XT.ensure(Np * h->Dcols); h->nx.ensure(Np);
    h->L.ensure(Np * Np); h->Linv.ensure(Np * Np); h->Kinv.ensure(Np * Np);
    h->alpha.ensure(Np); h->svec.ensure(Np);
    // results (res_stage): [0..2] sums, [4] log|K_y|, [5..6] the factorisation's two info words, [8..] length-scale gradient
    h->gemv_part.ensure((Np / 128) * Np);
    SLS_HIP(hipMemcpyAsync(h->X.p, X, (size_t)D * N * 8, hipMemcpyHostToDevice, ctx->stream));
    std::vector<double> xtr;   // must outlive the synchronisation below
    if (N <= NLL_SMALL_MAX_N && D <= NLL_SMALL_MAX_D) {
        xtr.assign((size_t)128 * D, 0.0);
        for (int i = 0; i < N; ++i)
            for (int d = 0; d < D; ++d) xtr[i + (size_t)d * 128] = X[d + (size_t)i * D];
        h->XTr.ensure(xtr.size());
        SLS_HIP(hipMemcpyAsync(h->XTr.p, xtr.data(), xtr.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    launch_fill(ctx->stream, h->y.p, Np, 0.0);
    SLS_HIP(hipStreamSynchronize(ctx->stream));
    slsk::ctx_retain(ctx);
    *out = h.release();
    SLS_CATCH
}

extern "C" int sls_nll_set_tolerances(sls_nll* h, double ftol_rel, double xtol_rel) {
    if (!h) return SLS_ERR_INVALID;
    std::unique_lock<std::recursive_mutex> lock_(h->ctx->mtx);
    h->ftol_rel = ftol_rel > 0.0 ? ftol_rel : 0.0;
    h->xtol_rel = xtol_rel > 0.0 ? xtol_rel : 0.0;
    return SLS_OK;
}

extern "C" int sls_nll_destroy(sls_nll* h) {
    if (!h) return SLS_OK;
    slsk::note_entry();
    sls_ctx* c = h->ctx;
    {
        std::unique_lock<std::recursive_mutex> lock_(c->mtx);
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        delete h;
    }
    slsk::ctx_release(c);
    return SLS_OK;
}

// The factorisation of one evaluation, ENQUEUED only: K_y, L, L^-1 (+ its transpose in G), K_y^-1 and log|K_y| on the stream.
// Nothing is read back here -- the caller appends the rest of the evaluation, copies (d_info, log-det) back together with its
// own results and hands them to nll_factor_accept: one host synchronisation per evaluation instead of two (the one in the
// middle left the GPU idle for 40-100 us of a 3.1 ms evaluation at N = 4096).  false: (theta, b) is the cached factor.
static void nll_stage_ensure(sls_nll* h) {
    if (h->il_stage) return;
    h->il_stage = static_cast<double*>(h->ctx->host_take((size_t)(2 * h->Dcols + h->Np + 8) * 8, true, &h->il_stage_bytes));
    SLS_HIP(hipHostGetDevicePointer((void**)&h->il_stage_dev, h->il_stage, 0));
    h->y_on_device = false;
}
static bool nll_factor_enqueue(sls_nll* h, const double* theta, double b) {
    sls_ctx* c = h->ctx;
    const int D = h->D, N = h->N, Np = h->Np;
    if (h->have_factor && h->cached_b == b && (int)h->cached_theta.size() == D + 1 &&
        std::memcmp(h->cached_theta.data(), theta, sizeof(double) * (D + 1)) == 0)
        return false;
    h->have_factor = false;
    SLS_REQUIRE(theta[0] > 0.0, "signal variance must be positive");
    // Staged in a page-locked block of the handle: the upload is enqueued like everything else (a local buffer needed a
    // synchronisation here, in the middle of the evaluation).  The previous evaluation on this handle ended with one, so the block is free.
    nll_stage_ensure(h);
    double* il = h->il_stage;
    for (int d = 0; d < h->Dcols; ++d) il[d] = 0.0;
    for (int d = 0; d < D; ++d) {
        SLS_REQUIRE(theta[1 + d] > 0.0, "length scale %d must be positive", d);
        il[d] = 1.0 / theta[1 + d];
    }
    // no upload: the kernels read the mapped block (1 KB; the previous evaluation on this handle ended with a synchronisation)
    KernelSpec ks{h->kernel, theta[0]};
    launch_prep_points(c->stream, h->X.p, D, N, h->il_stage_dev, h->XT.p, Np, Np, h->Dcols, h->nx.p);
    launch_gram_sym(c->stream, h->XT.p, Np, h->Dp, h->nx.p, Np, N, ks, b, h->L.p, true);
    SLS_HIP(hipMemsetAsync(c->d_info, 0, 64, c->stream));
    c->potrf_tick_rearm();
    h->G.ensure((size_t)Np * Np);   // the gradient's weight matrix, written after the factorisation: holds (L^-1)^T until then
    // N <= 4096: one launch for the factorisation and the inverse (launch_potri); otherwise potrf + trtri + lauum.
    // L^-1's buffer is not cleared here: only the separate launches need that (and do it themselves); nothing in this file reads the
    // tiles above its diagonal (134 MB, 18 us at N = 4096).
    launch_potri(c->stream, h->L.p, Np, h->Linv.p, h->G.p, h->Kinv.p, c->d_info, c->potrf_df_sync(Np), false);
    return true;   // log|K_y| (result word 4) comes out of nll_scalars, which every evaluation launches anyway
}
// info2 = the two words of d_info behind the enqueued factorisation (pivot failure, single-launch Cholesky gave up).
// true: accepted (cached from now on); false: run the evaluation once more (the context has switched to the multi-launch
// Cholesky); throws when K_y is not positive definite.
static bool nll_factor_accept(sls_nll* h, const double* theta, double b, const int* info2, int attempt) {
    if (potrf_gave_up(h->ctx, info2[1], attempt)) return false;
    if (info2[0] != 0) {
        set_error("sls_nll_eval: K_y is not positive definite (pivot %d)", info2[0] - 1);
        throw HipFail{SLS_ERR_NOT_SPD};
    }
    h->cached_theta.assign(theta, theta + h->D + 1);
    h->cached_b = b;
    h->have_factor = true;
    return true;
}

// split-K factor of the gradient's Y = G X~ product: a power of two dividing nt, about 512 workgroups in the launch
static int nll_y_chunks(int nt, int ncols) {
    int c = 1;
    while (c * 2 <= nt && nt % (c * 2) == 0 && nt * ncols * c * 2 <= 512) c *= 2;
    return c;
}

// N <= 128: the whole evaluation is one single-workgroup launch (kernels_small.hip); SLS_NLL_SMALL=0 forces the tiled path
static bool nll_small_ok(const sls_nll* h) {
    if (h->N > NLL_SMALL_MAX_N || h->D > NLL_SMALL_MAX_D) return false;
    return tune_on(TUNE_NLL_SMALL);
}

static void nll_small_eval(sls_nll* h, const double* y, const double* theta, double b, double* quad, double* logdet,
                           double* alpha, double* grad_theta, double* grad_b) {
    sls_ctx* c = h->ctx;
    const int D = h->D, N = h->N;
    SLS_REQUIRE(theta[0] > 0.0, "signal variance must be positive");
    for (int d = 0; d < D; ++d) SLS_REQUIRE(theta[1 + d] > 0.0, "length scale %d must be positive", d);
    const bool want_grad = grad_theta || grad_b;
    NllSmallArgs args;
    args.X = h->X.p; args.XTr = h->XTr.p; args.D = D; args.N = N; args.want_grad = want_grad ? 1 : 0;
    if (want_grad) h->small_kc.ensure(NLL_SMALL_KC_DOUBLES);
    args.kc = h->small_kc.p;
    args.x_lds = tune_on(TUNE_SMALL_XLDS) ? 1 : 0;
    args.info = c->d_info;
    const bool zero_copy = tune_on(TUNE_SMALL_ZEROCOPY);
    if (zero_copy && !h->small_host) {
        h->small_host = static_cast<double*>(c->host_take(NLL_SMALL_OUT_DOUBLES * sizeof(double), true, &h->small_host_bytes));
        SLS_HIP(hipHostGetDevicePointer((void**)&h->small_host_dev, h->small_host, 0));
    }
    h->small_out.ensure(NLL_SMALL_OUT_DOUBLES);
    args.out = zero_copy ? h->small_host_dev : h->small_out.p;
    args.in_dev = nullptr;
    args.batch = 1; args.in_stride = 0; args.out_stride = 0;
    std::vector<double> in;   // staging for the D > 32 upload: must outlive the stream synchronisation below
    if (D <= NLL_SMALL_MAX_ARG_D) {
        args.a = theta[0]; args.b = b;
        for (int d = 0; d < D; ++d) args.ell[d] = theta[1 + d];
        std::memcpy(args.y, y, sizeof(double) * N);
    } else {
        in.resize(2 + D + N);
        in[0] = theta[0]; in[1] = b;
        for (int d = 0; d < D; ++d) in[2 + d] = theta[1 + d];
        for (int i = 0; i < N; ++i) in[2 + D + i] = y[i];
        h->small_in.ensure(in.size());
        SLS_HIP(hipMemcpyAsync(h->small_in.p, in.data(), in.size() * 8, hipMemcpyHostToDevice, c->stream));
        args.in_dev = h->small_in.p;
    }
    launch_nll_small(c->stream, h->kernel, args);
    double out_stack[NLL_SMALL_OUT_DOUBLES];
    const double* out = out_stack;
    const int nout = alpha ? NLL_SMALL_OUT_ALPHA + N : NLL_SMALL_OUT_ALPHA;
    if (zero_copy) {
        SLS_HIP(hipStreamSynchronize(c->stream));
        out = h->small_host;
    } else {
        SLS_HIP(hipMemcpyAsync(out_stack, h->small_out.p, nout * 8, hipMemcpyDeviceToHost, c->stream));
        SLS_HIP(hipStreamSynchronize(c->stream));
    }
    if (out[4] != 0.0) {
        set_error("sls_nll_eval: K_y is not positive definite (pivot %d)", (int)out[4] - 1);
        throw HipFail{SLS_ERR_NOT_SPD};
    }
    h->have_factor = false;   // the tiled path's cached factor (L, Linv, Kinv buffers) was not refreshed
    h->logdet = out[3];
    if (quad) *quad = out[2];
    if (logdet) *logdet = out[3];
    if (grad_b) *grad_b = out[1];
    if (grad_theta) {
        grad_theta[0] = out[0] / theta[0];
        for (int d = 0; d < D; ++d) grad_theta[1 + d] = out[NLL_SMALL_OUT_GL + d];
    }
    if (alpha) std::memcpy(alpha, out + NLL_SMALL_OUT_ALPHA, sizeof(double) * N);
}

static void nll_eval_impl(sls_nll* h, const double* y, const double* theta, double b, double* quad, double* logdet,
                          double* alpha, double* grad_theta, double* grad_b) {
    sls_ctx* c = h->ctx;
    const int D = h->D, N = h->N, Np = h->Np;
    SLS_REQUIRE(y && theta, "sls_nll_eval: NULL argument");
    SLS_REQUIRE(b >= 0.0, "noise level must be >= 0");
    if (nll_small_ok(h)) {
        nll_small_eval(h, y, theta, b, quad, logdet, alpha, grad_theta, grad_b);
        return;
    }
    const bool want_grad = grad_theta || grad_b;
    const int nt = Np / 128;
    nll_stage_ensure(h);
    const double* res = h->res_stage();
    for (int attempt = 0;; ++attempt) {
        // the targets go up first (nothing before them on the stream), and only when they differ from what is there
        if (!h->y_on_device || std::memcmp(h->y_stage(), y, sizeof(double) * N) != 0) {
            std::memcpy(h->y_stage(), y, sizeof(double) * N);
            SLS_HIP(hipMemcpyAsync(h->y.p, h->y_stage(), (size_t)N * 8, hipMemcpyHostToDevice, c->stream));
            h->y_on_device = true;
        }
        const bool fresh = nll_factor_enqueue(h, theta, b);
        // alpha = K_y^-1 y from the explicit (symmetric, full) inverse the gradient's weights use anyway: one pass over one matrix
        // (it was L^-1 twice: 69 -> 25 us at N = 4096)
        launch_gemv_t(c->stream, h->Kinv.p, Np, h->y.p, h->alpha.p);
        if (want_grad) {
            h->G.ensure((size_t)Np * Np);
            h->Y.ensure((size_t)Np * h->Dcols * nll_y_chunks(nt, h->Dcols / 128));
            h->parts.ensure((size_t)nt * nt);
            launch_nll_weight(c->stream, h->XT.p, Np, h->Dp, h->nx.p, Np, N, KernelSpec{h->kernel, theta[0]}, h->alpha.p, h->Kinv.p,
                              h->G.p, h->parts.p, grad_theta ? h->gemv_part.p : nullptr);
        } else {
            h->parts.ensure(1);
        }
        // the sums, and behind a fresh factorisation log|K_y| and its two info words (pivot failure, single launch gave up): a
        // workgroup of the last launch when there is a length-scale gradient, a launch of their own otherwise
        const double* Lf = fresh ? h->L.p : nullptr;
        const int* inf = fresh ? c->d_info : nullptr;
        const int nparts = want_grad ? nt * nt : 0;
        if (grad_theta) {
            // Y = G X~ has only nt x Dcols/128 output tiles: split the contraction so that the launch fills the chip
            const int yc = nll_y_chunks(nt, h->Dcols / 128);
            launch_gemm_splitk_nt(c->stream, h->G.p, Np, h->XT.p, Np, h->Y.p, Np, (long)Np * h->Dcols, nt, h->Dcols / 128, Np, yc);
            // the partial products of Y and the partial row sums (G 1) left by nll_weight, both added in chunk order
            launch_sum_chunks(c->stream, h->Y.p, (long)Np * h->Dcols, yc, (long)Np * h->Dcols, h->gemv_part.p, Np, nt, h->svec.p);
            launch_lengthscale_grad(c->stream, h->XT.p, h->Y.p, h->svec.p, h->il_stage_dev, Np, N, D, h->res_dev() + 8, h->parts.p, nparts,
                                    h->alpha.p, h->y.p, h->Kinv.p, Np, h->res_dev(), Lf, inf);
        } else {
            launch_nll_scalars(c->stream, h->parts.p, nparts, h->alpha.p, h->y.p, h->Kinv.p, Np, N, h->res_dev(), Lf, inf);
        }
        // the results are already in (mapped) host memory when the stream has drained: no copy back
        if (alpha) SLS_HIP(hipMemcpyAsync(alpha, h->alpha.p, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
        SLS_HIP(hipStreamSynchronize(c->stream));
        if (!fresh) break;
        const int info2[2] = {(int)res[5], (int)res[6]};
        if (nll_factor_accept(h, theta, b, info2, attempt)) {
            h->logdet = res[4];
            break;
        }   // else: once more on the multi-launch Cholesky
    }
    const double* sc = res;
    const double* gl = res + 8;
    if (quad) *quad = sc[2];
    if (logdet) *logdet = h->logdet;
    if (grad_b) *grad_b = sc[1];
    if (grad_theta) {
        grad_theta[0] = sc[0] / theta[0];
        for (int d = 0; d < D; ++d) grad_theta[1 + d] = gl[d];
    }
}

extern "C" int sls_nll_eval(sls_nll* h, const double* y, const double* theta, double b, double* quad, double* logdet,
                            double* alpha, double* grad_theta, double* grad_b) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (h) {
        lock_ = std::unique_lock<std::recursive_mutex>(h->ctx->mtx);
        (void)hipSetDevice(h->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    SLS_REQUIRE(h, "sls_nll_eval: NULL handle");
    nll_eval_impl(h, y, theta, b, quad, logdet, alpha, grad_theta, grad_b);
    SLS_CATCH
}

// mathtoolbox::GetLogOfLogNormalDist / ...Derivative
static double log_lognormal(double x, double mu, double s2) {
    const double lx = std::log(x);
    return -lx - 0.5 * std::log(2.0 * M_PI * s2) - (lx - mu) * (lx - mu) / (2.0 * s2);
}
static double log_lognormal_d(double x, double mu, double s2) { return (mu - s2 - std::log(x)) / (s2 * x); }

extern "C" int sls_gp_nll_grad(sls_nll* h, const double* y, const double* x, double* value, double* grad) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (h) {
        lock_ = std::unique_lock<std::recursive_mutex>(h->ctx->mtx);
        (void)hipSetDevice(h->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    SLS_REQUIRE(h && y && x, "sls_gp_nll_grad: NULL argument");
    const int D = h->D, N = h->N;
    // priors: src/gaussian-process-regressor.cpp:18-24
    const double a_mu = std::log(0.5), a_s2 = 0.5, b_mu = std::log(1e-4), b_s2 = 0.5, r_mu = std::log(0.5), r_s2 = 0.5;
    const double a = x[0], b = x[1];
    std::vector<double> theta(D + 1), gth(D + 1);
    theta[0] = a;
    for (int d = 0; d < D; ++d) theta[1 + d] = x[2 + d];
    double quad = 0, logdet = 0, gb = 0;
    nll_eval_impl(h, y, theta.data(), b, &quad, &logdet, nullptr, grad ? gth.data() : nullptr, grad ? &gb : nullptr);
    double reg = log_lognormal(a, a_mu, a_s2) + log_lognormal(b, b_mu, b_s2);
    for (int d = 0; d < D; ++d) reg += log_lognormal(x[2 + d], r_mu, r_s2);
    if (value) *value = -0.5 * quad - 0.5 * logdet - 0.5 * N * std::log(2.0 * M_PI) + reg;   // :174-192
    if (grad) {
        grad[0] = gth[0] + log_lognormal_d(a, a_mu, a_s2);                                   // calc_grad :120-124
        grad[1] = gb + log_lognormal_d(b, b_mu, b_s2);
        for (int d = 0; d < D; ++d) grad[2 + d] = gth[1 + d] + log_lognormal_d(x[2 + d], r_mu, r_s2);
    }
    SLS_CATCH
}

// values[k] = GP MAP objective at xs[k] = (a, b, r_1..r_D), k < B, without gradients: what DIRECT asks for per iteration
// (src/gaussian-process-regressor.cpp:294 runs them one by one).  N <= 128: ONE launch, one workgroup per parameter set;
// larger problems are evaluated in turn.  A parameter set whose K_y is not positive definite gets -HUGE_VAL.
extern "C" int sls_gp_nll_batch(sls_nll* h, const double* y, const double* xs, int B, double* values) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (h) {
        lock_ = std::unique_lock<std::recursive_mutex>(h->ctx->mtx);
        (void)hipSetDevice(h->ctx->device);
    }
    SLS_REQUIRE(h && y && xs && values && B >= 0, "sls_gp_nll_batch: bad argument");
    const int D = h->D, N = h->N;
    sls_ctx* c = h->ctx;
    const double a_mu = std::log(0.5), a_s2 = 0.5, b_mu = std::log(1e-4), b_s2 = 0.5, r_mu = std::log(0.5), r_s2 = 0.5;
    auto prior = [&](const double* x) {
        double reg = log_lognormal(x[0], a_mu, a_s2) + log_lognormal(x[1], b_mu, b_s2);
        for (int d = 0; d < D; ++d) reg += log_lognormal(x[2 + d], r_mu, r_s2);
        return reg;
    };
    auto sequential = [&](int k0, int k1) {
        std::vector<double> theta(D + 1);
        for (int k = k0; k < k1; ++k) {
            const double* x = xs + (size_t)k * (D + 2);
            theta[0] = x[0];
            for (int d = 0; d < D; ++d) theta[1 + d] = x[2 + d];
            double quad = 0, logdet = 0;
            try {
                nll_eval_impl(h, y, theta.data(), x[1], &quad, &logdet, nullptr, nullptr, nullptr);
                values[k] = -0.5 * quad - 0.5 * logdet - 0.5 * N * std::log(2.0 * M_PI) + prior(x);
            } catch (const HipFail& f) {
                if (f.code != SLS_ERR_NOT_SPD) throw;
                values[k] = -HUGE_VAL;
            }
        }
    };
    // every parameter set is validated up front, the same way on every path: a <= 0, b < 0 or a length scale <= 0 is an error
    for (int k = 0; k < B; ++k) {
        const double* x = xs + (size_t)k * (D + 2);
        SLS_REQUIRE(x[0] > 0.0 && x[1] >= 0.0, "sls_gp_nll_batch: signal variance must be positive, noise level >= 0 (point %d)", k);
        for (int d = 0; d < D; ++d) SLS_REQUIRE(x[2 + d] > 0.0, "length scale %d must be positive (point %d)", d, k);
    }
    if (!nll_small_ok(h)) {
        // N > 128: bordered factorisations (quad and log-det from the factor alone: no inverse), several parameter sets per
        // persistent launch, each on its own share of the chip -- one factorisation of this size is bound by its serial chain and
        // leaves most CUs idle.  SLS_NLL_BATCH=0: one full evaluation after the other (the round-3 path).
        const int Np2 = round_up(N + 1, 128);
        const int Pmax = !tune_on(TUNE_NLL_BATCH) ? 0 : potrf_dataflow_max_problems(Np2);
        if (Pmax < 1 || !c->potrf_df_available(Np2) || Np2 / 128 < 3) {   // the single-launch form is switched off, or too small for it
            sequential(0, B);
            return SLS_OK;
        }
        // info words: two per problem, read back from the 64 bytes of d_info cleared below (d_info + 32 belongs to acq_gemm)
        SLS_REQUIRE(Pmax <= 8, "sls_gp_nll_batch: %d problems per launch exceed the info block (8)", Pmax);
        const size_t mat = (size_t)Np2 * Np2, sync_ints = (potrf_dataflow_sync_ints(Np2) + 63) / 64 * 64;
        const int Dp = h->Dp, Dcols = h->Dcols;
        if (h->bt_P < Pmax) {
            h->bt_L.ensure(mat * Pmax); h->bt_T.ensure(mat * Pmax);
            h->bt_XT.ensure((size_t)Np2 * Dcols * Pmax); h->bt_nx.ensure((size_t)Np2 * Pmax); h->bt_il.ensure((size_t)Dcols * Pmax);
            h->bt_sync.ensure((sync_ints * Pmax + 1) / 2); h->bt_out.ensure(2 * Pmax);
            h->bt_P = Pmax;
        }
        SLS_HIP(hipMemcpyAsync(h->y.p, y, (size_t)N * 8, hipMemcpyHostToDevice, c->stream));
        h->y_on_device = false;   // not through the staging block: the next gradient evaluation uploads its own
        std::vector<double> il((size_t)Dcols * Pmax), out(2 * Pmax);
        for (int k0 = 0; k0 < B;) {
            const int P = std::min(Pmax, B - k0);
            std::fill(il.begin(), il.end(), 0.0);
            for (int q = 0; q < P; ++q)
                for (int d = 0; d < D; ++d) il[(size_t)q * Dcols + d] = 1.0 / xs[(size_t)(k0 + q) * (D + 2) + 2 + d];
            SLS_HIP(hipMemcpyAsync(h->bt_il.p, il.data(), (size_t)Dcols * P * 8, hipMemcpyHostToDevice, c->stream));
            SLS_HIP(hipStreamSynchronize(c->stream));   // `il` is rewritten for the next group
            for (int q = 0; q < P; ++q) {
                const double* x = xs + (size_t)(k0 + q) * (D + 2);
                double* XTq = h->bt_XT.p + (size_t)q * Np2 * Dcols;
                double* nxq = h->bt_nx.p + (size_t)q * Np2;
                launch_prep_points(c->stream, h->X.p, D, N, h->bt_il.p + (size_t)q * Dcols, XTq, Np2, Np2, Dcols, nxq);
                launch_gram_sym(c->stream, XTq, Np2, Dp, nxq, Np2, N, KernelSpec{h->kernel, x[0]}, x[1], h->bt_L.p + q * mat, true);
            }
            launch_border_row(c->stream, h->bt_L.p, (long)mat, P, Np2, N, h->y.p, 1e200);
            SLS_HIP(hipMemsetAsync(c->d_info, 0, 64, c->stream));
            c->potrf_tick_rearm();
            int info[16] = {0};
            bool ok = launch_potrf_dataflow_batch(c->stream, h->bt_L.p, Np2, h->bt_T.p, c->d_info, reinterpret_cast<int*>(h->bt_sync.p), P,
                                                  (long)mat, (long)sync_ints, false);
            if (ok) {
                launch_border_reduce(c->stream, h->bt_L.p, (long)mat, P, Np2, N, h->bt_out.p);
                SLS_HIP(hipMemcpyAsync(info, c->d_info, 16 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
                SLS_HIP(hipMemcpyAsync(out.data(), h->bt_out.p, 2 * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                SLS_HIP(hipStreamSynchronize(c->stream));
                int aborted = 0;
                for (int q = 0; q < P; ++q) aborted |= info[2 * q + 1];
                if (aborted) ok = !potrf_gave_up(c, aborted, 0);   // the context switches to the multi-launch schedule (or throws)
            }
            if (!ok) {
                sequential(k0, B);   // this group and the rest on the general path
                return SLS_OK;
            }
            for (int q = 0; q < P; ++q) {
                const double* x = xs + (size_t)(k0 + q) * (D + 2);
                values[k0 + q] = info[2 * q] != 0 ? -HUGE_VAL : -0.5 * out[2 * q] - 0.5 * out[2 * q + 1] - 0.5 * N * std::log(2.0 * M_PI) + prior(x);
            }
            k0 += P;
        }
        h->have_factor = false;
        return SLS_OK;
    }
    if (B <= 1) {
        sequential(0, B);
        return SLS_OK;
    }
    const size_t in_stride = 2 + D + N, out_stride = 8;
    std::vector<double> in((size_t)B * in_stride);
    for (int k = 0; k < B; ++k) {
        const double* x = xs + (size_t)k * (D + 2);
        double* o = in.data() + (size_t)k * in_stride;
        o[0] = x[0]; o[1] = x[1];
        for (int d = 0; d < D; ++d) o[2 + d] = x[2 + d];
        std::memcpy(o + 2 + D, y, sizeof(double) * N);
    }
    h->small_in.ensure(in.size());
    h->small_out.ensure(std::max<size_t>(NLL_SMALL_OUT_DOUBLES, (size_t)B * out_stride));
    h->small_info.ensure(((size_t)B + 1) / 2 + 1);
    SLS_HIP(hipMemcpyAsync(h->small_in.p, in.data(), in.size() * 8, hipMemcpyHostToDevice, c->stream));
    NllSmallArgs args;
    args.X = h->X.p; args.XTr = h->XTr.p; args.kc = nullptr; args.D = D; args.N = N; args.want_grad = 0;
    args.x_lds = tune_on(TUNE_SMALL_XLDS) ? 1 : 0;
    args.info = reinterpret_cast<int*>(h->small_info.p);
    args.out = h->small_out.p; args.in_dev = h->small_in.p;
    args.batch = B; args.in_stride = (long)in_stride; args.out_stride = (long)out_stride;
    args.a = args.b = 0.0;
    launch_nll_small(c->stream, h->kernel, args);
    std::vector<double> out((size_t)B * out_stride);
    SLS_HIP(hipMemcpyAsync(out.data(), h->small_out.p, out.size() * 8, hipMemcpyDeviceToHost, c->stream));
    SLS_HIP(hipStreamSynchronize(c->stream));
    h->have_factor = false;
    for (int k = 0; k < B; ++k) {
        const double* o = out.data() + (size_t)k * out_stride;
        values[k] = o[4] != 0.0 ? -HUGE_VAL : -0.5 * o[2] - 0.5 * o[3] - 0.5 * N * std::log(2.0 * M_PI) + prior(xs + (size_t)k * (D + 2));
    }
    SLS_CATCH
}

// ---- device-resident MAP fit / objective (map_opt_kernel, kernels_small.hip) --------------------------------------------
struct MapOptProblem {
    int ny = 0, nh = 0, log_hyper = 0, noiseless = 0;
    const double* y_fixed = nullptr;   // host, N (ny == 0)
    double a0 = 0, b0 = 0, r0 = 0, mu_a = 0, mu_b = 0, mu_r = 0, s2_a = 1, s2_b = 1, s2_r = 1, btl_scale = 1;
    const unsigned* prefs_flat = nullptr;
    const int* pref_offsets = nullptr;
    int n_prefs = 0;
};

// N <= 128 (one LDS image), D <= 128 (1/l, l and the length-scale gradient in the LDS scratch), with or without hyper-parameters;
// SLS_MAP_DEVICE=0 forces the host-driven optimiser / the host's BTL terms (A/B and tests)
static bool map_opt_supported(const sls_nll* h) {
    if (h->N > NLL_SMALL_MAX_N || h->D > NLL_SMALL_MAX_D) return false;
    return tune_on(TUNE_MAP_DEVICE);
}

// Runs the kernel: the whole fit in one launch (evals_per_launch <= 0 or >= max_evals), in launches of evals_per_launch
// evaluations continuing from the stored state, or one evaluation (eval_only: value + gradient at z0).
static void map_opt_run(sls_nll* h, const MapOptProblem& pb, const double* z0, const double* lower, const double* upper, int max_evals,
                        int evals_per_launch, bool eval_only, double* z_out, double* value, double* grad_out, int* evals_used,
                        bool* not_spd) {
    sls_ctx* c = h->ctx;
    const int D = h->D, N = h->N, n = pb.ny + pb.nh, P = pb.n_prefs;
    SLS_REQUIRE(n >= 1 && n <= MAP_OPT_MAX_VARS, "map fit: %d variables (limit %d)", n, MAP_OPT_MAX_VARS);
    SLS_REQUIRE(pb.ny == 0 || pb.ny == N, "map fit: ny must be 0 or N");
    SLS_REQUIRE(pb.nh == 0 || pb.nh == D + 2, "map fit: nh must be 0 or D + 2");
    const int F = P > 0 ? pb.pref_offsets[P] : 0;
    // index image: [pref_off (P+1)] [pref_flat (F)] [csc_off (N+1)] [csc_ent (F)]
    std::vector<int> idx((size_t)P + 1 + F + N + 1 + F, 0);
    int* poff = idx.data();
    int* pflat = poff + P + 1;
    int* coff = pflat + F;
    int* cent = coff + N + 1;
    for (int p = 0; p <= P; ++p) poff[p] = P > 0 ? pb.pref_offsets[p] : 0;
    for (int p = 0; p < P; ++p) SLS_REQUIRE(poff[p + 1] > poff[p], "preference tuple %d is empty", p);
    for (int q = 0; q < F; ++q) {
        SLS_REQUIRE(pb.prefs_flat[q] < (unsigned)N, "preference index %u out of range", pb.prefs_flat[q]);
        pflat[q] = (int)pb.prefs_flat[q];
        ++coff[pflat[q] + 1];
    }
    for (int j = 0; j < N; ++j) coff[j + 1] += coff[j];
    {
        std::vector<int> fill(coff, coff + N);
        for (int q = 0; q < F; ++q) cent[fill[pflat[q]]++] = q;   // increasing flat position = tuple order (:202-216)
    }
    const size_t vec_doubles = (size_t)3 * n + N;
    const size_t need = idx.size() * sizeof(int) + 8 + vec_doubles * 8;
    if (need > h->mo_stage_bytes) {
        c->host_give(h->mo_stage, h->mo_stage_bytes, false);
        h->mo_stage = nullptr;
        h->mo_stage_bytes = 0;
        h->mo_stage = static_cast<char*>(c->host_take(need * 2, false, &h->mo_stage_bytes));
    }
    if (!h->mo_out) {
        h->mo_out = static_cast<double*>(c->host_take(MAP_OPT_OUT_DOUBLES * sizeof(double), true, &h->mo_out_bytes));
        SLS_HIP(hipHostGetDevicePointer((void**)&h->mo_out_dev, h->mo_out, 0));
    }
    {
        const double* before = h->mo_idx.p;
        h->mo_idx.ensure((idx.size() + 1) / 2 + 1);
        if (h->mo_idx.p != before) h->mo_idx_host.clear();   // a new block: what the old one held is gone
    }
    h->mo_vec.ensure(vec_doubles);
    h->mo_state.ensure(MAP_OPT_STATE_DOUBLES + MAP_OPT_TRACE_SLOTS);   // + the optional section trace
    h->mo_btl.ensure(std::max(F, 1));
    // the previous call's copies have completed (every call ends with a stream synchronisation): the staging block is free
    double* vst = reinterpret_cast<double*>(h->mo_stage);
    std::memcpy(vst, z0, sizeof(double) * n);
    std::memcpy(vst + n, lower, sizeof(double) * n);
    std::memcpy(vst + 2 * n, upper, sizeof(double) * n);
    if (pb.ny == 0) std::memcpy(vst + 3 * n, pb.y_fixed, sizeof(double) * N);
    else std::memset(vst + 3 * n, 0, sizeof(double) * N);
    SLS_HIP(hipMemcpyAsync(h->mo_vec.p, vst, vec_doubles * 8, hipMemcpyHostToDevice, c->stream));
    if (idx != h->mo_idx_host) {
        int* ist = reinterpret_cast<int*>(h->mo_stage + vec_doubles * 8);
        std::memcpy(ist, idx.data(), idx.size() * sizeof(int));
        SLS_HIP(hipMemcpyAsync(h->mo_idx.p, ist, idx.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        h->mo_idx_host = idx;
    }
    MapOptArgs a;
    if (pb.nh > 0) h->small_kc.ensure(NLL_SMALL_KC_DOUBLES);
    a.x_lds = tune_on(TUNE_SMALL_XLDS) ? 1 : 0;
    a.X = h->X.p; a.XTr = h->XTr.p; a.kc = h->small_kc.p; a.D = D; a.N = N; a.ny = pb.ny; a.nh = pb.nh; a.log_hyper = pb.log_hyper; a.noiseless = pb.noiseless;
    a.y_fixed = h->mo_vec.p + 3 * n;
    a.a0 = pb.a0; a.b0 = pb.b0; a.r0 = pb.r0;
    a.mu_a = pb.mu_a; a.mu_b = pb.mu_b; a.mu_r = pb.mu_r; a.s2_a = pb.s2_a; a.s2_b = pb.s2_b; a.s2_r = pb.s2_r;
    a.btl_scale = pb.btl_scale;
    a.n_prefs = P; a.flat_len = F;
    const int* di = reinterpret_cast<const int*>(h->mo_idx.p);
    a.pref_off = di; a.pref_flat = di + P + 1; a.csc_off = di + P + 1 + F; a.csc_ent = di + P + 1 + F + N + 1;
    a.btl_scratch = h->mo_btl.p;
    a.z0 = h->mo_vec.p; a.lower = h->mo_vec.p + n; a.upper = h->mo_vec.p + 2 * n;
    a.max_evals = eval_only ? 1 : max_evals;
    a.ftol_rel = h->ftol_rel;
    a.xtol_rel = h->xtol_rel;
    a.eval_only = eval_only ? 1 : 0;
    a.state = h->mo_state.p;
    a.out = h->mo_out_dev;
    a.info = c->d_info;
    a.trace = nullptr;
    if (tune_set(TUNE_MAP_TRACE) && !eval_only) {
        a.trace = reinterpret_cast<long long*>(h->mo_state.p + MAP_OPT_STATE_DOUBLES);
        SLS_HIP(hipMemsetAsync(a.trace, 0, MAP_OPT_TRACE_SLOTS * sizeof(long long), c->stream));
    }
    const bool stepwise = !eval_only && evals_per_launch > 0 && evals_per_launch < max_evals;
    a.budget = stepwise ? evals_per_launch : a.max_evals;
    a.fresh = 1;
    const double* out = h->mo_out;
    for (;;) {
        launch_map_opt(c->stream, h->kernel, a);
        SLS_HIP(hipStreamSynchronize(c->stream));
        if (!stepwise || out[2] != 0.0 || (int)out[1] >= max_evals) break;
        a.fresh = 0;
    }
    if (a.trace) {
        long long tr[MAP_OPT_TRACE_SLOTS];
        SLS_HIP(hipMemcpy(tr, a.trace, sizeof(tr), hipMemcpyDeviceToHost));
        fprintf(stderr, "map_opt trace (us): N %d n %d prefs %d evals %lld | publish %.1f  gram %.1f  chol %.1f  logdet %.1f  inverse %.1f  "
                "alpha %.1f  grad-weights %.1f  grad-contraction %.1f  btl %.1f  value+gradient %.1f  optimiser %.1f  total %.1f\n", N, n, P, tr[6],
                tr[0] * 0.01, tr[8] * 0.01, tr[9] * 0.01, tr[10] * 0.01, tr[11] * 0.01, tr[1] * 0.01, tr[12] * 0.01, tr[13] * 0.01, tr[3] * 0.01,
                tr[4] * 0.01, tr[5] * 0.01, tr[7] * 0.01);
        fprintf(stderr, "   fine: inverse products %.1f barrier %.1f (mirror+barrier = rest) | grad-weights loop %.1f (block sum = rest) | contraction Y pass %.1f "
                "barrier %.1f (final = rest; row sums %.1f of the pass) | Gram dot products %.1f (elements = rest) | rejected trials %lld\n",
                tr[16] * 0.01, tr[17] * 0.01, tr[18] * 0.01, tr[19] * 0.01, tr[20] * 0.01, tr[22] * 0.01, tr[23] * 0.01, tr[21]);
    }
    h->have_factor = false;   // the tiled path's cached factor (L, Linv, Kinv buffers) was not refreshed
    if (value) *value = out[0];
    if (evals_used) *evals_used = (int)out[1];
    if (not_spd) *not_spd = out[3] != 0.0;
    if (z_out) std::memcpy(z_out, out + MAP_OPT_OUT_X, sizeof(double) * n);
    if (grad_out) std::memcpy(grad_out, out + MAP_OPT_OUT_G, sizeof(double) * n);
}

static MapOptProblem pref_problem(const sls_nll* h, const unsigned* prefs_flat, const int* pref_offsets, int n_prefs,
                                  const sls_pref_cfg* cfg, bool log_hyper) {
    MapOptProblem pb;
    pb.ny = h->N;
    pb.nh = cfg->use_map_hyperparams ? h->D + 2 : 0;
    pb.log_hyper = log_hyper ? 1 : 0;
    pb.noiseless = cfg->noiseless ? 1 : 0;
    pb.a0 = cfg->default_a; pb.b0 = cfg->default_b; pb.r0 = cfg->default_r;
    pb.mu_a = std::log(cfg->default_a); pb.mu_b = std::log(cfg->default_b); pb.mu_r = std::log(cfg->default_r);
    pb.s2_a = pb.s2_b = pb.s2_r = cfg->prior_var;
    pb.btl_scale = cfg->btl_scale;
    pb.prefs_flat = prefs_flat; pb.pref_offsets = pref_offsets; pb.n_prefs = n_prefs;
    return pb;
}

extern "C" int sls_pref_map_fit(sls_nll* h, const unsigned* prefs_flat, const int* pref_offsets, int n_prefs, const sls_pref_cfg* cfg,
                                const double* z0, const double* lower, const double* upper, int max_evals, int evals_per_launch,
                                double* z_out, double* value, int* evals_used) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (h) {
        lock_ = std::unique_lock<std::recursive_mutex>(h->ctx->mtx);
        (void)hipSetDevice(h->ctx->device);
    }
    SLS_REQUIRE(h && cfg && z0 && lower && upper && z_out && max_evals >= 1 && (n_prefs == 0 || (prefs_flat && pref_offsets)),
                "sls_pref_map_fit: bad argument");
    if (!map_opt_supported(h)) {
        set_error("sls_pref_map_fit: N = %d, D = %d outside the device-resident optimiser (N <= 128, D <= 128)", h->N, h->D);
        return SLS_ERR_UNSUPPORTED;
    }
    const MapOptProblem pb = pref_problem(h, prefs_flat, pref_offsets, n_prefs, cfg, true);
    map_opt_run(h, pb, z0, lower, upper, max_evals, evals_per_launch, false, z_out, value, nullptr, evals_used, nullptr);
    SLS_CATCH
}

extern "C" int sls_gp_map_fit(sls_nll* h, const double* y, const double* z0, const double* lower, const double* upper, int max_evals,
                              int evals_per_launch, double* z_out, double* value, int* evals_used) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (h) {
        lock_ = std::unique_lock<std::recursive_mutex>(h->ctx->mtx);
        (void)hipSetDevice(h->ctx->device);
    }
    SLS_REQUIRE(h && y && z0 && lower && upper && z_out && max_evals >= 1, "sls_gp_map_fit: bad argument");
    if (!map_opt_supported(h)) {
        set_error("sls_gp_map_fit: N = %d, D = %d outside the device-resident optimiser (N <= 128, D <= 128)", h->N, h->D);
        return SLS_ERR_UNSUPPORTED;
    }
    MapOptProblem pb;
    pb.ny = 0; pb.nh = h->D + 2; pb.log_hyper = 1; pb.y_fixed = y;
    // priors: src/gaussian-process-regressor.cpp:18-24
    pb.mu_a = std::log(0.5); pb.mu_b = std::log(1e-4); pb.mu_r = std::log(0.5);
    pb.s2_a = pb.s2_b = pb.s2_r = 0.5;
    map_opt_run(h, pb, z0, lower, upper, max_evals, evals_per_launch, false, z_out, value, nullptr, evals_used, nullptr);
    SLS_CATCH
}

// utils::CalcBtl / CalcBtlDerivative (include/sequential-line-search/utils.hpp:25-52), no max-subtraction like the reference
static double btl(const double* f, int n, double s) {
    double sum = 0.0;
    for (int i = 0; i < n; ++i) sum += std::exp(f[i] / s);
    return std::exp(f[0] / s) / sum;
}
static void btl_derivative(const double* f, int n, double s, double* d) {
    const double v = btl(f, n, s);
    const double tmp = -v * v / s;
    double sum = 0.0;
    for (int i = 1; i < n; ++i) sum += std::exp((f[i] - f[0]) / s);
    d[0] = tmp * (-sum);
    for (int i = 1; i < n; ++i) d[i] = tmp * std::exp((f[i] - f[0]) / s);
}

extern "C" int sls_pref_objective(sls_nll* h, const unsigned* prefs_flat, const int* pref_offsets, int n_prefs, const double* x,
                                  const sls_pref_cfg* cfg, double* value, double* grad) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (h) {
        lock_ = std::unique_lock<std::recursive_mutex>(h->ctx->mtx);
        (void)hipSetDevice(h->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    SLS_REQUIRE(h && x && cfg && (n_prefs == 0 || (prefs_flat && pref_offsets)), "sls_pref_objective: NULL argument");
    const int D = h->D, M = h->N;
    const bool use_map = cfg->use_map_hyperparams != 0;
    if (map_opt_supported(h)) {
        // one single-workgroup launch: BTL terms, GP term, priors and the whole gradient on the device (eval_only mode)
        if (use_map) {
            SLS_REQUIRE(x[M] > 0.0, "signal variance must be positive");
            SLS_REQUIRE(cfg->noiseless || x[M + 1] >= 0.0, "noise level must be >= 0");
            for (int d = 0; d < D; ++d) SLS_REQUIRE(x[M + 2 + d] > 0.0, "length scale %d must be positive", d);
        }
        const MapOptProblem pb = pref_problem(h, prefs_flat, pref_offsets, n_prefs, cfg, false);
        const int n = M + (use_map ? D + 2 : 0);
        std::vector<double> lo(n, -HUGE_VAL), hi(n, HUGE_VAL);
        bool bad = false;
        map_opt_run(h, pb, x, lo.data(), hi.data(), 1, 0, true, nullptr, value, grad, nullptr, &bad);
        if (bad) {
            set_error("sls_pref_objective: K_y is not positive definite");
            return SLS_ERR_NOT_SPD;
        }
        return SLS_OK;
    }
    const double a = use_map ? x[M + 0] : cfg->default_a;
    const double b = cfg->noiseless ? 0.0 : (use_map ? x[M + 1] : cfg->default_b);
    std::vector<double> theta(D + 1), gth(D + 1), alpha(M);
    theta[0] = a;
    for (int d = 0; d < D; ++d) theta[1 + d] = use_map ? x[M + 2 + d] : cfg->default_r;
    double obj = 0.0;
    std::vector<double> ftmp, dtmp;
    for (int p = 0; p < n_prefs; ++p) {   // :151-154
        const int o = pref_offsets[p], n = pref_offsets[p + 1] - o;
        ftmp.resize(n);
        for (int i = 0; i < n; ++i) {
            SLS_REQUIRE(prefs_flat[o + i] < (unsigned)M, "preference index %u out of range", prefs_flat[o + i]);
            ftmp[i] = x[prefs_flat[o + i]];
        }
        obj += std::log(btl(ftmp.data(), n, cfg->btl_scale));
    }
    double quad = 0, logdet = 0, gb = 0;
    const bool gtheta = grad && use_map;
    nll_eval_impl(h, x, theta.data(), b, &quad, &logdet, alpha.data(), gtheta ? gth.data() : nullptr, gtheta ? &gb : nullptr);
    obj += -0.5 * quad - 0.5 * logdet - 0.5 * M * std::log(2.0 * M_PI);   // :165-170
    if (use_map) {   // :175-192
        obj += log_lognormal(a, std::log(cfg->default_a), cfg->prior_var);
        if (!cfg->noiseless) obj += log_lognormal(b, std::log(cfg->default_b), cfg->prior_var);
        for (int d = 0; d < D; ++d) obj += log_lognormal(theta[1 + d], std::log(cfg->default_r), cfg->prior_var);
    }
    if (value) *value = obj;
    if (grad) {
        for (int i = 0; i < M; ++i) grad[i] = 0.0;
        for (int p = 0; p < n_prefs; ++p) {   // :202-216
            const int o = pref_offsets[p], n = pref_offsets[p + 1] - o;
            ftmp.resize(n);
            dtmp.resize(n);
            for (int i = 0; i < n; ++i) ftmp[i] = x[prefs_flat[o + i]];
            const double v = btl(ftmp.data(), n, cfg->btl_scale);
            btl_derivative(ftmp.data(), n, cfg->btl_scale, dtmp.data());
            for (int i = 0; i < n; ++i) grad[prefs_flat[o + i]] += dtmp[i] / v;
        }
        for (int i = 0; i < M; ++i) grad[i] += -alpha[i];   // :219
        if (use_map) {
            grad[M + 0] = gth[0] + log_lognormal_d(a, std::log(cfg->default_a), cfg->prior_var);
            grad[M + 1] = cfg->noiseless ? 0.0 : gb + log_lognormal_d(b, std::log(cfg->default_b), cfg->prior_var);   // :237-238
            for (int d = 0; d < D; ++d)
                grad[M + 2 + d] = gth[1 + d] + log_lognormal_d(theta[1 + d], std::log(cfg->default_r), cfg->prior_var);
        }
    }
    SLS_CATCH
}
