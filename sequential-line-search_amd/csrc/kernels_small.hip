// Fused MAP-objective evaluation for the reference's own operating sizes (N <= 128 data points; the demos run N <= 100):
// ONE workgroup, ONE launch replaces the ~20 launches of the tiled pipeline (prep, Gram, potrf, trtri, lauum, GEMVs,
// weight/gradient contractions, reductions), which at these sizes is pure launch latency (340 us wall per evaluation of
// the GP marginal likelihood at N = 20, of which < 80 us are kernels).
//
// Reference arithmetic: src/gaussian-process-regressor.cpp:66-127,141-193 (log marginal likelihood and its gradient
// wrt a, b, l_1..l_D) and the GP term of src/preference-regressor.cpp:53-115; kernel scalars src/regressor.cpp:14-23.
//   K_y = K_f(a, l) + b I  ->  L, L^-1 (chol_diag_steps, LDS resident)  ->  logdet = 2 sum log L_ii,
//   K^-1 = L^-T L^-1 (MFMA tiles, overwriting L),  alpha = K^-1 y,  quad = y^T alpha,
//   W = 1/2 (alpha alpha^T - K^-1):  d/da = sum W.*K_f / a,  d/db = tr W,  d/dl_p = (1/l_p) sum_ij W_ij c_ij (x~_ip - x~_jp)^2
// Everything lives in the 160 KB LDS image of the diagonal-block kernel; the 16 padding doubles of every LDS column
// (rows 128..143 of the [128 x 144] matrix) serve as scratch for alpha, y, 1/l and the reduction slots.
#include "chol_diag.hpp"
#include "kernels.hpp"
#include "wave_reduce.hpp"
#include "../../include/sls_hip.h"

namespace slsk {

constexpr int SMALL_OUT_GL = 8;        // out[8 .. 8+D): length-scale gradient (D <= NLL_SMALL_MAX_GRAD_D)
constexpr int SMALL_OUT_ALPHA = 32;    // out[32 .. 32+N): alpha

// scratch map (2048 doubles in the padding rows of the LDS matrix):
//   [0,128) alpha   [128,256) y   [256,384) 1/l   [384,1024) BTL contributions (map_opt)   [1024,1028) reduction slots
//   [1028,1032) a, b (map_opt)    [1040,1056) l (map_opt)   [1296,1552) gradient wrt the optimiser's variables (map_opt)
constexpr int SC_ALPHA = 0, SC_Y = 128, SC_INVL = 256, SC_BTL = 384, SC_BTL_MAX = 640, SC_RED = 1024, SC_AB = 1028,
              SC_ELL = 1040, SC_GZ = 1296;

__device__ __forceinline__ double& small_scratch(double* As, int k) { return As[(k >> 4) * DL + 128 + (k & 15)]; }

__device__ __forceinline__ double small_block_sum(double v, double* As) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) small_scratch(As, SC_RED + (threadIdx.x >> 6)) = v;
    __syncthreads();
    return (small_scratch(As, SC_RED) + small_scratch(As, SC_RED + 1)) + (small_scratch(As, SC_RED + 2) + small_scratch(As, SC_RED + 3));
}

template <bool MATERN>
__device__ __forceinline__ void small_kern(double a, double q, double& k, double& c) {
    if (!MATERN) {
        k = a * exp(-0.5 * q);
        c = k;
    } else {
        const double s = sqrt(5.0 * q), e = exp(-s);
        k = a * (1.0 + s + (5.0 / 3.0) * q) * e;
        c = a * (5.0 / 3.0) * (1.0 + s) * e;
    }
}

__device__ __forceinline__ double small_pair_q(const double* __restrict__ X, int D, int i, int j, double* As) {
    double q = 0.0;
    for (int d = 0; d < D; ++d) {
        const double t = (X[d + (long)i * D] - X[d + (long)j * D]) * small_scratch(As, SC_INVL + d);
        q = fma(t, t, q);
    }
    return q;
}

// K_y = K_f(a, l) + b I into the LDS image (1/l in the scratch) and its Cholesky factorisation on the leading ceil(N/16) blocks: L in the
// lower triangle of As, L^-T in its strictly-upper tiles, the inverses of the diagonal tiles in Ts.  Returns sum_i log L_ii.
template <bool MATERN>
__device__ __forceinline__ double small_build_factor(double* As, double* Ts, const double* __restrict__ X, int D, int N,
                                                     double a, double b, int* __restrict__ info) {
    const int tid = threadIdx.x;
    const int nb16 = (N + 15) >> 4, Nb = 16 * nb16;
    // ---- K_y (lower triangle + full diagonal tiles), identity padding up to the next multiple of 16 ----
    for (int idx = tid; idx < Nb * Nb; idx += 256) {
        const int i = idx % Nb, j = idx / Nb;
        if (i < j) continue;
        double v;
        if (i >= N) v = (i == j) ? 1.0 : 0.0;
        else if (i == j) v = a + b;
        else {
            double k, c;
            small_kern<MATERN>(a, small_pair_q(X, D, i, j, As), k, c);
            v = k;
        }
        As[i + j * DL] = v;
        if ((i >> 4) == (j >> 4)) As[j + i * DL] = v;
    }
    __syncthreads();

    chol_diag_steps<true>(As, Ts, info, 0, nb16);
    __syncthreads();

    // ---- log det ----
    const double ld = small_block_sum(tid < N ? log(As[tid + tid * DL]) : 0.0, As);
    return ld;
}

// K^-1 = L^-T L^-1 as a full symmetric image over the dead factor (after small_build_factor: L in the lower triangle of As, L^-T in its
// strictly-upper tiles, the inverses of the diagonal tiles in Ts)
__device__ __forceinline__ void small_inverse_in_place(double* As, double* Ts, int N) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, fk = lane >> 4;
    const int nb16 = (N + 15) >> 4;
    // ---- K^-1 = L^-T L^-1, lower tiles (i >= j) into the lower triangle (L is no longer needed) ----
    {
        int t = 0;
        for (int i = 0; i < nb16; ++i)
            for (int j = 0; j <= i; ++j, ++t) {
                if ((t & 3) != wave) continue;
                d4_t c = {0.0, 0.0, 0.0, 0.0};
                for (int k = i; k < nb16; ++k) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int kq = 4 * kk + fk;
                        const double af = (k == i) ? Ts[256 * i + kq + 16 * fl] : As[(16 * k + kq) * DL + 16 * i + fl];   // T[k][i] (kq, m)
                        const double bf = (k == j) ? Ts[256 * j + kq + 16 * fl] : As[(16 * k + kq) * DL + 16 * j + fl];   // T[k][j] (kq, n)
                        c = mfma16(bf, af, c);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) As[(16 * j + fk + 4 * q) * DL + 16 * i + fl] = c[q];
            }
    }
    __syncthreads();
    // mirror the strictly-lower tiles into the upper triangle (L^-T is no longer needed): K^-1 becomes a full symmetric image
    {
        int t = 0;
        for (int i = 1; i < nb16; ++i)
            for (int j = 0; j < i; ++j, ++t) {
                if ((t & 3) != wave) continue;
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = As[(16 * j + fl) * DL + 16 * i + ((fl + fk + 4 * q) & 15)];   // (row 16i + r, col 16j + fl)
#pragma unroll
                for (int q = 0; q < 4; ++q) As[(16 * i + ((fl + fk + 4 * q) & 15)) * DL + 16 * j + fl] = v[q];   // (row 16j + fl, col 16i + r)
            }
    }
    __syncthreads();
}


// K_y = K_f(a, l) + b I into the LDS image (1/l in the scratch), Cholesky + inverse on the leading ceil(N/16) blocks,
// K_y^-1 as a full symmetric image over the dead factor.  Returns sum_i log L_ii (= logdet / 2) in every thread.
template <bool MATERN>
__device__ __forceinline__ double small_factor_inverse(double* As, double* Ts, const double* __restrict__ X, int D, int N,
                                                       double a, double b, int* __restrict__ info) {
    const double ld = small_build_factor<MATERN>(As, Ts, X, D, N, a, b, info);
    small_inverse_in_place(As, Ts, N);
    return ld;
}

// alpha = K^-1 y (y in the scratch) into the scratch; gb = 1/2 (alpha.alpha - tr K^-1), quad = y.alpha in every thread
__device__ __forceinline__ void small_alpha(double* As, int N, double& gb, double& quad) {
    const int tid = threadIdx.x;
    if (tid < 128) {
        // eight LDS operand pairs in flight, then their fused multiply-adds in column order: the same sum as the plain loop (a
        // loop of dependent load -> fma trips ran at ~150 ns per column on the otherwise idle CU).  Columns N .. 8 ceil(N / 8) - 1
        // exist in the image (identity padding, inside the 16-aligned block) and meet y = 0 there.
        double s = 0.0;
        if (tid < N) {
            for (int j0 = 0; j0 < N; j0 += 8) {
                double av[8], yv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    av[u] = As[tid + (j0 + u) * DL];
                    yv[u] = small_scratch(As, SC_Y + j0 + u);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) s = fma(av[u], yv[u], s);
            }
        }
        small_scratch(As, SC_ALPHA + tid) = s;
    }
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    if (tid < N) {
        const double al = small_scratch(As, SC_ALPHA + tid);
        s1 = al * al - As[tid + tid * DL];
        s2 = small_scratch(As, SC_Y + tid) * al;
    }
    gb = 0.5 * small_block_sum(s1, As);
    quad = small_block_sum(s2, As);
}

// gradient contractions over the pairs i >= j:  sa = sum W.*K_f,  gl[d] = (1/l_d) sum W_ij c_ij (x~_id - x~_jd)^2 (d < D <= 16)
template <bool MATERN>
__device__ __forceinline__ void small_grad(double* As, const double* __restrict__ X, int D, int N, double a, bool want_grad,
                                           double& sa_t, double (&gl_t)[NLL_SMALL_MAX_GRAD_D]) {
    const int tid = threadIdx.x;
    double sa = 0.0;
    double gl[NLL_SMALL_MAX_GRAD_D];
#pragma unroll
    for (int d = 0; d < NLL_SMALL_MAX_GRAD_D; ++d) gl[d] = 0.0;
    if (want_grad) {
        for (int idx = tid; idx < N * N; idx += 256) {
            const int i = idx % N, j = idx / N;
            if (i < j) continue;
            const double w = (i == j ? 0.5 : 1.0) * (small_scratch(As, SC_ALPHA + i) * small_scratch(As, SC_ALPHA + j) - As[i + j * DL]);
            double q = 0.0, dd[NLL_SMALL_MAX_GRAD_D];
#pragma unroll
            for (int d = 0; d < NLL_SMALL_MAX_GRAD_D; ++d) {
                dd[d] = 0.0;
                if (d < D) {
                    const double t = (X[d + (long)i * D] - X[d + (long)j * D]) * small_scratch(As, SC_INVL + d);
                    dd[d] = t * t;
                    q += dd[d];
                }
            }
            if (D > NLL_SMALL_MAX_GRAD_D) q = small_pair_q(X, D, i, j, As);   // length-scale gradient not requested for such D (host check)
            if (i == j) q = 0.0;
            double k, c;
            small_kern<MATERN>(a, q, k, c);
            sa = fma(w, k, sa);
            const double g = w * c;
#pragma unroll
            for (int d = 0; d < NLL_SMALL_MAX_GRAD_D; ++d) gl[d] = fma(g, dd[d], gl[d]);
        }
    }
    sa_t = small_block_sum(sa, As);
#pragma unroll
    for (int d = 0; d < NLL_SMALL_MAX_GRAD_D; ++d) {
        gl_t[d] = 0.0;
        if (want_grad && d < D) gl_t[d] = small_block_sum(gl[d], As) * small_scratch(As, SC_INVL + d);   // D uniform: all threads take part
    }
}

template <bool MATERN>
__global__ __launch_bounds__(256) void nll_small_kernel(const NllSmallArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* As = reinterpret_cast<double*>(smem);
    double* Ts = As + 128 * DL;
    const int tid = threadIdx.x;
    const double* __restrict__ X = args.X;
    double* __restrict__ out = args.out + (long)blockIdx.x * args.out_stride;
    int* __restrict__ info = args.info + blockIdx.x;
    const double* __restrict__ in_dev = args.in_dev ? args.in_dev + (long)blockIdx.x * args.in_stride : nullptr;
    const int D = args.D, N = args.N, want_grad = args.want_grad;
    // hyper-parameters and targets travel in the kernel argument block (no upload); D > 32 (NLL_SMALL_MAX_ARG_D) falls back to a device buffer
    const double a = in_dev ? in_dev[0] : args.a, b = in_dev ? in_dev[1] : args.b;
    for (int d = tid; d < D; d += 256) small_scratch(As, SC_INVL + d) = 1.0 / (in_dev ? in_dev[2 + d] : args.ell[d]);
    for (int i = tid; i < 128; i += 256)
        small_scratch(As, SC_Y + i) = i < N ? (in_dev ? in_dev[2 + D + i] : args.y[i]) : 0.0;
    if (tid == 0) *info = 0;
    __syncthreads();

    const double ld = small_factor_inverse<MATERN>(As, Ts, X, D, N, a, b, info);
    double gb, quad;
    small_alpha(As, N, gb, quad);
    if (tid < N && args.batch <= 1) out[SMALL_OUT_ALPHA + tid] = small_scratch(As, SC_ALPHA + tid);   // batch mode: 8 output words per parameter set
    double sa_t, gl_t[NLL_SMALL_MAX_GRAD_D];
    small_grad<MATERN>(As, X, D, N, a, want_grad != 0, sa_t, gl_t);
    if (tid == 0) {
        if (want_grad) {
#pragma unroll
            for (int d = 0; d < NLL_SMALL_MAX_GRAD_D; ++d)
                if (d < D) out[SMALL_OUT_GL + d] = gl_t[d];
        }
        out[0] = sa_t;
        out[1] = gb;
        out[2] = quad;
        out[3] = 2.0 * ld;
        out[4] = (double)__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------------
// map_opt_kernel: the whole MAP fit (or one objective evaluation) in ONE single-workgroup launch.
//
// Objective (maximised), with y = z[0..ny) or the fixed targets and (a, b, l) = the hyper variables or the fixed defaults:
//   f = sum_p log BTL_p(y)                                   src/preference-regressor.cpp:151-154, utils.hpp:25-29 (no max-subtraction)
//     - 1/2 y^T K^-1 y - 1/2 log|K| - N/2 log 2 pi           :165-170  /  src/gaussian-process-regressor.cpp:174-180
//     + log-normal priors of a, b, l_d (nh > 0)              :175-192  /  :181-192
//   df/dy = sum_p dBTL_p / BTL_p - K^-1 y                    :199-221, utils.hpp:31-52
//   df/d(a, b, l)                                            :53-115   /  :66-127   (small_grad, D <= 16)
// Optimiser: optim::MaximizeBounded of host/device.cpp statement by statement (projected gradient, two-loop recursion over
// the last 8 pairs, Armijo backtracking by halving, at most 31 trials per direction) as a flat state machine with one
// objective evaluation per turn.  The optimiser state (x, g, direction, history) is REPLICATED in the registers of each of
// the four waves -- lane l owns variables l, l + 64, l + 128 -- so its ~25 reductions per iteration are wave
// shuffles without a workgroup barrier; all waves execute the same instructions on the same values and therefore take the
// same branches.  The evaluation itself is the one-workgroup pipeline of nll_small_kernel; for fixed hyper-parameters
// (the reference's use_map_hyperparams = false, src/preference-regressor.cpp:161-162) K^-1 is built once and stays in LDS.
// `budget` < max_evals: the launch stops after `budget` evaluations and leaves the state in `state`; the next launch (fresh = 0)
// continues from it with the same machine code -- the one-launch-per-evaluation form the tests compare the single launch with.
// ---------------------------------------------------------------------------------------------------------
constexpr int KV = MAP_OPT_MAX_VARS / 64;
constexpr int MH = MAP_OPT_HIST;

__device__ __forceinline__ double wave_dot(const double (&u)[KV], const double (&v)[KV]) {
    double p = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k) p += u[k] * v[k];
    return wave_sum(p);
}
// mathtoolbox::GetLogOfLogNormalDist / ...Derivative
__device__ __forceinline__ double dev_log_lognormal(double x, double mu, double s2) {
    const double lx = log(x);
    return -lx - 0.5 * log(2.0 * M_PI * s2) - (lx - mu) * (lx - mu) / (2.0 * s2);
}
__device__ __forceinline__ double dev_log_lognormal_d(double x, double mu, double s2) { return (mu - s2 - log(x)) / (s2 * x); }

template <bool MATERN>
__global__ __launch_bounds__(256) void map_opt_kernel(const MapOptArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* As = reinterpret_cast<double*>(smem);
    double* Ts = As + 128 * DL;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* __restrict__ X = args.X;
    const int D = args.D, N = args.N, ny = args.ny, nh = args.nh, n = ny + nh;
    const int P = args.n_prefs, F = args.flat_len;
    const bool btl_lds = F <= SC_BTL_MAX;
    int* __restrict__ info = args.info;
    double* __restrict__ state = args.state;
    double* __restrict__ out = args.out;
    long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool tracing = args.trace != nullptr;
    const long long tr_begin = tracing ? wall_clock64() : 0;
    long long t_prev = tr_begin;
#define MAP_T(slot) do { if (tracing) { const long long t_now = wall_clock64(); tr[slot] += t_now - t_prev; t_prev = t_now; } } while (0)

    // ---- optimiser state: one replica per wave ----
    double x[KV], g[KV], xt[KV], d[KV], lo[KV], hi[KV], S[MH][KV], Y[MH][KV], rho[MH];
    double fx = 0.0, t = 1.0, sy_last = 0.0, yy_last = 1.0;
    int cnt = 0, bt = 0, evals = 0, phase = 0, done = 0;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int e = lane + 64 * k;
        lo[k] = e < n ? args.lower[e] : 0.0;
        hi[k] = e < n ? args.upper[e] : 0.0;
        x[k] = g[k] = d[k] = 0.0;
        xt[k] = e < n ? fmin(hi[k], fmax(lo[k], args.z0[e])) : 0.0;
#pragma unroll
        for (int h = 0; h < MH; ++h) S[h][k] = Y[h][k] = 0.0;
    }
#pragma unroll
    for (int h = 0; h < MH; ++h) rho[h] = 0.0;
    if (!args.fresh) {
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int e = lane + 64 * k;
            x[k] = state[0 * MAP_OPT_MAX_VARS + e];
            g[k] = state[1 * MAP_OPT_MAX_VARS + e];
            xt[k] = state[2 * MAP_OPT_MAX_VARS + e];
            d[k] = state[3 * MAP_OPT_MAX_VARS + e];
#pragma unroll
            for (int h = 0; h < MH; ++h) {
                S[h][k] = state[(4 + h) * MAP_OPT_MAX_VARS + e];
                Y[h][k] = state[(4 + MH + h) * MAP_OPT_MAX_VARS + e];
            }
        }
        const double* sc = state + (4 + 2 * MH) * MAP_OPT_MAX_VARS;
        fx = sc[0]; t = sc[1]; sy_last = sc[2]; yy_last = sc[3];
#pragma unroll
        for (int h = 0; h < MH; ++h) rho[h] = sc[4 + h];
        cnt = (int)sc[12]; bt = (int)sc[13]; evals = (int)sc[14]; phase = (int)sc[15]; done = (int)sc[16];
    }

    // ---- fixed inputs ----
    for (int i = tid; i < 128; i += 256) small_scratch(As, SC_Y + i) = (ny == 0 && i < N) ? args.y_fixed[i] : 0.0;
    if (nh == 0)
        for (int dd = tid; dd < D; dd += 256) small_scratch(As, SC_INVL + dd) = 1.0 / args.r0;
    if (tid == 0) *info = 0;
    // the first preference tuple of this thread and the first tuple memberships of data point `tid`: indices in registers
    constexpr int RC = 4;
    int po = 0, pm = 0, pidx0 = 0, pidx1 = 0, pidx2 = 0, pidx3 = 0, co = 0, cm = 0, cidx0 = 0, cidx1 = 0, cidx2 = 0, cidx3 = 0;
    if (tid < P) {
        po = args.pref_off[tid];
        pm = args.pref_off[tid + 1] - po;
        pidx0 = args.pref_flat[po];
        if (pm > 1) pidx1 = args.pref_flat[po + 1];
        if (pm > 2) pidx2 = args.pref_flat[po + 2];
        if (pm > 3) pidx3 = args.pref_flat[po + 3];
    }
    if (tid < ny && P > 0) {
        co = args.csc_off[tid];
        cm = args.csc_off[tid + 1] - co;
        if (cm > 0) cidx0 = args.csc_ent[co];
        if (cm > 1) cidx1 = args.csc_ent[co + 1];
        if (cm > 2) cidx2 = args.csc_ent[co + 2];
        if (cm > 3) cidx3 = args.csc_ent[co + 3];
    }
    __syncthreads();

    bool have_factor = false, bad = false;
    double ld = 0.0, a = args.a0, b = args.noiseless ? 0.0 : args.b0;
    double f_last = 0.0;
    int budget = args.budget;
    while (budget > 0 && !done) {
        --budget;
        // ---- publish the trial point: y into the scratch, hyper-parameters in linear space ----
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                const int e = lane + 64 * k;
                if (e < ny) small_scratch(As, SC_Y + e) = xt[k];
                else if (e < n) {
                    const int hq = e - ny;
                    const double v = args.log_hyper ? exp(xt[k]) : xt[k];
                    if (hq == 0) small_scratch(As, SC_AB) = v;
                    else if (hq == 1) small_scratch(As, SC_AB + 1) = args.noiseless ? 0.0 : v;
                    else {
                        small_scratch(As, SC_ELL + hq - 2) = v;
                        small_scratch(As, SC_INVL + hq - 2) = 1.0 / v;
                    }
                }
            }
        }
        __syncthreads();
        if (nh) {
            a = small_scratch(As, SC_AB);
            b = small_scratch(As, SC_AB + 1);
        }
        if (nh || !have_factor) {
            ld = small_factor_inverse<MATERN>(As, Ts, X, D, N, a, b, info);
            have_factor = true;
            bad = __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        }
        MAP_T(0);
        double gb, quad;
        small_alpha(As, N, gb, quad);
        MAP_T(1);
        if (bad && nh && tid == 0) *info = 0;   // every thread has read it (barriers of small_alpha); the next factorisation starts clean
        double sa_t = 0.0, gl_t[NLL_SMALL_MAX_GRAD_D];
        if (nh) small_grad<MATERN>(As, X, D, N, a, true, sa_t, gl_t);

        MAP_T(2);
        // ---- Bradley-Terry-Luce terms: tuple p on thread p (, p + 256, ...) ----
        // contrib: the per-member terms d BTL_p / BTL_p, in the LDS scratch (flat_len <= 640) or in global memory -- two
        // instantiations of the same code, not a run-time pointer choice (a pointer that may be LDS or global is a generic
        // pointer: flat loads / stores)
        double btl_sum = 0.0, gy = 0.0;
        auto btl_terms = [&](auto&& contrib) {
            double lsum = 0.0;
            const double bs = args.btl_scale;
            // one tuple: o = its offset in the flat list, m = its size, (m0 .. m3) = its first RC members
            auto tuple_terms = [&](int o, int m, int m0, int m1, int m2, int m3) {
                // the first RC members from registers (a select chain: a dynamically indexed private array lives in scratch memory,
                // one flat load per member on the dependent path member -> y -> exp), the rest from global memory
                auto member = [&](int i) {
                    if (i >= RC) return args.pref_flat[o + i];
                    int r = m0;
                    r = (i == 1) ? m1 : r;
                    r = (i == 2) ? m2 : r;
                    r = (i == 3) ? m3 : r;
                    return r;
                };
                const double f0 = small_scratch(As, SC_Y + m0);
                double sum = 0.0;
                for (int i = 0; i < m; ++i) sum += exp(small_scratch(As, SC_Y + member(i)) / bs);
                const double v = exp(f0 / bs) / sum;                       // CalcBtl
                lsum += log(v);                                            // calc_log_likelihood
                const double tmp = -v * v / bs;                            // CalcBtlDerivative
                double sum2 = 0.0;
                for (int i = 1; i < m; ++i) {
                    const double r = exp((small_scratch(As, SC_Y + member(i)) - f0) / bs);   // used twice by the reference: once here
                    sum2 += r;
                    contrib(o + i) = r;
                }
                contrib(o) = (tmp * (-sum2)) / v;
                for (int i = 1; i < m; ++i) contrib(o + i) = (tmp * contrib(o + i)) / v;
            };
            static_assert(RC == 4, "tuple_terms takes four register members");
            if (tid < P) tuple_terms(po, pm, pidx0, pidx1, pidx2, pidx3);
            for (int p = tid + 256; p < P; p += 256) {       // more than 256 tuples: indices from global memory
                const int o = args.pref_off[p], m = args.pref_off[p + 1] - o;
                tuple_terms(o, m, args.pref_flat[o], m > 1 ? args.pref_flat[o + 1] : 0, m > 2 ? args.pref_flat[o + 2] : 0,
                            m > 3 ? args.pref_flat[o + 3] : 0);
            }
            btl_sum = small_block_sum(lsum, As);   // its barriers also publish the contributions (LDS or global, one CU)
            if (tid < ny) {
                for (int i = 0; i < cm; ++i) {                                                                      // :202-216, in tuple order
                    int e;
                    if (i >= RC) e = args.csc_ent[co + i];
                    else {
                        e = cidx0;
                        e = (i == 1) ? cidx1 : e;
                        e = (i == 2) ? cidx2 : e;
                        e = (i == 3) ? cidx3 : e;
                    }
                    gy += contrib(e);
                }
                gy -= small_scratch(As, SC_ALPHA + tid);                                                            // :219
            }
        };
        if (btl_lds) btl_terms([&](int q) -> double& { return small_scratch(As, SC_BTL + q); });
        else btl_terms([&](int q) -> double& { return args.btl_scratch[q]; });
        MAP_T(3);

        // ---- value ----
        double f = btl_sum + (-0.5 * quad - 0.5 * (2.0 * ld) - 0.5 * N * log(2.0 * M_PI));
        if (nh) {
            double reg = dev_log_lognormal(a, args.mu_a, args.s2_a);
            if (!args.noiseless) reg += dev_log_lognormal(b, args.mu_b, args.s2_b);
            for (int dd = 0; dd < D; ++dd) reg += dev_log_lognormal(small_scratch(As, SC_ELL + dd), args.mu_r, args.s2_r);
            f += reg;
        }
        f_last = f;
        // ---- gradient wrt the optimiser's variables (minimisation: phi = -f) ----
        if (tid < n) {
            double gz;
            if (tid < ny) gz = gy;
            else {
                const int hq = tid - ny;
                double gx, xv;
                if (hq == 0) {
                    xv = a;
                    gx = sa_t / a + dev_log_lognormal_d(a, args.mu_a, args.s2_a);
                } else if (hq == 1) {
                    xv = b;
                    gx = args.noiseless ? 0.0 : gb + dev_log_lognormal_d(b, args.mu_b, args.s2_b);
                } else {
                    xv = small_scratch(As, SC_ELL + hq - 2);
                    double gl = 0.0;
#pragma unroll
                    for (int dd = 0; dd < NLL_SMALL_MAX_GRAD_D; ++dd) gl = (hq - 2 == dd) ? gl_t[dd] : gl;
                    gx = gl + dev_log_lognormal_d(xv, args.mu_r, args.s2_r);
                }
                gz = args.log_hyper ? gx * xv : gx;
            }
            small_scratch(As, SC_GZ + tid) = -gz;
            if (args.eval_only) out[MAP_OPT_OUT_G + tid] = gz;
        }
        __syncthreads();
        MAP_T(4);
        ++evals;
        tr[6] += 1;
        if (args.eval_only) {
            done = 1;
#pragma unroll
            for (int k = 0; k < KV; ++k) x[k] = xt[k];
            fx = -f;
            break;
        }
        double gt[KV];
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int e = lane + 64 * k;
            gt[k] = e < n ? small_scratch(As, SC_GZ + e) : 0.0;
        }
        const double ft = bad ? HUGE_VAL : -f;

        // ---- advance the optimiser by one evaluation (per wave, no barriers) ----
        bool need_dir = false;
        if (phase == 0) {
#pragma unroll
            for (int k = 0; k < KV; ++k) { x[k] = xt[k]; g[k] = gt[k]; }
            fx = ft;
            phase = 1;
            need_dir = true;
        } else {
            double sv[KV], yv[KV];
#pragma unroll
            for (int k = 0; k < KV; ++k) { sv[k] = xt[k] - x[k]; yv[k] = gt[k] - g[k]; }
            const double gs = wave_dot(g, sv);
            if (isfinite(ft) && ft <= fx + 1e-4 * gs) {
                const double sy = wave_dot(sv, yv), yy = wave_dot(yv, yv);
                if (sy > 1e-10 * yy && sy > 0.0) {
                    if (cnt == MH) {
#pragma unroll
                        for (int h = 0; h + 1 < MH; ++h) {
                            rho[h] = rho[h + 1];
#pragma unroll
                            for (int k = 0; k < KV; ++k) { S[h][k] = S[h + 1][k]; Y[h][k] = Y[h + 1][k]; }
                        }
                        cnt = MH - 1;
                    }
#pragma unroll
                    for (int h = 0; h < MH; ++h)
                        if (h == cnt) {
                            rho[h] = 1.0 / sy;
#pragma unroll
                            for (int k = 0; k < KV; ++k) { S[h][k] = sv[k]; Y[h][k] = yv[k]; }
                        }
                    ++cnt;
                    sy_last = sy;
                    yy_last = yy;
                }
#pragma unroll
                for (int k = 0; k < KV; ++k) { x[k] = xt[k]; g[k] = gt[k]; }
                fx = ft;
                need_dir = true;
            } else {
                t *= 0.5;
                if (++bt > 30) done = 1;
            }
        }
        if (!done && need_dir) {
            if (evals >= args.max_evals) done = 1;
            else {
                double pg[KV], pm_ = 0.0, pn = 0.0;
#pragma unroll
                for (int k = 0; k < KV; ++k) {
                    double v = g[k];
                    if ((x[k] <= lo[k] && v > 0.0) || (x[k] >= hi[k] && v < 0.0)) v = 0.0;
                    pg[k] = v;
                    pm_ = fmax(pm_, fabs(v));
                    pn += v * v;
                }
                const double pgmax = wave_max(pm_), pgn2 = wave_sum(pn);
                if (!(pgmax > 0.0)) done = 1;
                else {
                    double al[MH];
#pragma unroll
                    for (int k = 0; k < KV; ++k) d[k] = pg[k];
#pragma unroll
                    for (int h = MH - 1; h >= 0; --h) {
                        al[h] = 0.0;
                        if (h < cnt) {
                            al[h] = rho[h] * wave_dot(S[h], d);
#pragma unroll
                            for (int k = 0; k < KV; ++k) d[k] -= al[h] * Y[h][k];
                        }
                    }
                    double gamma = cnt > 0 ? sy_last / yy_last : 1.0 / fmax(1.0, sqrt(pgn2));
#pragma unroll
                    for (int k = 0; k < KV; ++k) d[k] *= gamma;
#pragma unroll
                    for (int h = 0; h < MH; ++h)
                        if (h < cnt) {
                            const double beta = rho[h] * wave_dot(Y[h], d);
#pragma unroll
                            for (int k = 0; k < KV; ++k) d[k] += S[h][k] * (al[h] - beta);
                        }
#pragma unroll
                    for (int k = 0; k < KV; ++k) d[k] = (pg[k] == 0.0) ? 0.0 : -d[k];
                    double gd = wave_dot(pg, d);
                    if (!(gd < 0.0)) {
                        cnt = 0;
                        gamma = 1.0 / fmax(1.0, sqrt(pgn2));
#pragma unroll
                        for (int k = 0; k < KV; ++k) d[k] = -gamma * pg[k];
                        gd = wave_dot(pg, d);
                        if (!(gd < 0.0)) done = 1;
                    }
                    t = 1.0;
                    bt = 0;
                }
            }
        }
        if (!done) {
            if (evals >= args.max_evals) done = 1;
            else {
                double dv[KV];
#pragma unroll
                for (int k = 0; k < KV; ++k) {
                    xt[k] = fmin(hi[k], fmax(lo[k], x[k] + t * d[k]));
                    dv[k] = xt[k] - x[k];
                }
                if (wave_dot(dv, dv) == 0.0) done = 1;
            }
        }
        MAP_T(5);
    }
    if (tracing && tid == 0) {
        tr[7] = wall_clock64() - tr_begin;
        for (int q = 0; q < 8; ++q) args.trace[q] += tr[q];
    }

    // ---- results and the state for a continuation ----
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int e = lane + 64 * k;
            state[0 * MAP_OPT_MAX_VARS + e] = x[k];
            state[1 * MAP_OPT_MAX_VARS + e] = g[k];
            state[2 * MAP_OPT_MAX_VARS + e] = xt[k];
            state[3 * MAP_OPT_MAX_VARS + e] = d[k];
#pragma unroll
            for (int h = 0; h < MH; ++h) {
                state[(4 + h) * MAP_OPT_MAX_VARS + e] = S[h][k];
                state[(4 + MH + h) * MAP_OPT_MAX_VARS + e] = Y[h][k];
            }
            if (e < n) out[MAP_OPT_OUT_X + e] = x[k];
        }
        if (lane == 0) {
            double* sc = state + (4 + 2 * MH) * MAP_OPT_MAX_VARS;
            sc[0] = fx; sc[1] = t; sc[2] = sy_last; sc[3] = yy_last;
#pragma unroll
            for (int h = 0; h < MH; ++h) sc[4 + h] = rho[h];
            sc[12] = cnt; sc[13] = bt; sc[14] = evals; sc[15] = phase; sc[16] = done;
            out[0] = args.eval_only ? f_last : -fx;
            out[1] = evals;
            out[2] = done;
            out[3] = bad ? 1.0 : 0.0;
        }
    }
}

void launch_map_opt(hipStream_t s, int kernel, const MapOptArgs& args) {
    ensure_dyn_lds((const void*)map_opt_kernel<false>, DIAG_LDS_BYTES);
    ensure_dyn_lds((const void*)map_opt_kernel<true>, DIAG_LDS_BYTES);
    if (kernel == SLS_KERNEL_ARD_MATERN52)
        hipLaunchKernelGGL(map_opt_kernel<true>, dim3(1), dim3(256), DIAG_LDS_BYTES, s, args);
    else
        hipLaunchKernelGGL(map_opt_kernel<false>, dim3(1), dim3(256), DIAG_LDS_BYTES, s, args);
}

// ---------------------------------------------------------------------------------------------------------
// gp_fit_small_kernel: the whole fit of a GP handle for N <= 128 in ONE single-workgroup launch (GaussianProcessRegressor /
// PreferenceRegressor constructors, src/gaussian-process-regressor.cpp:198-232, src/preference-regressor.cpp:289-290, with the
// hoisted quantities of DESIGN.md 2): scaled design matrix + norms, K_y, L, L^-1, (L^-1)^T, K_y^-1, alpha, alpha o X~, the
// posterior mean at the data points with its first maximum, log|K_y|.  The tiled pipeline needs ~15 launches for the same at these
// sizes (175 us of device time + their launch overheads per fit; round 4, C3: one fit per SubmitFeedbackData).
// ---------------------------------------------------------------------------------------------------------
template <bool MATERN>
__global__ __launch_bounds__(256) void gp_fit_small_kernel(const GpFitSmallArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* As = reinterpret_cast<double*>(smem);
    double* Ts = As + 128 * DL;
    const int tid = threadIdx.x;
    const int D = p.D, N = p.N, Dcols = p.Dcols;
    constexpr int Np = 128;
    const int nb16 = (N + 15) >> 4, Nb = 16 * nb16;
    // ---- scaled, centred design matrix and its squared norms (prep_kernel's arithmetic) ----
    if (tid < Np) {
        double sq = 0.0;
        if (tid < N) {
            for (int d = 0; d < D; ++d) {
                const double v = (p.X[d + (long)tid * D] - 0.5) * p.inv_ell[d];
                p.XT[tid + (long)d * Np] = v;
                sq += v * v;
            }
            for (int d = D; d < Dcols; ++d) p.XT[tid + (long)d * Np] = 0.0;
        } else {
            for (int d = 0; d < Dcols; ++d) p.XT[tid + (long)d * Np] = 0.0;
        }
        p.nx[tid] = sq;
    }
    for (int d = tid; d < D; d += 256) small_scratch(As, SC_INVL + d) = p.inv_ell[d];
    for (int i = tid; i < 128; i += 256) small_scratch(As, SC_Y + i) = i < N ? p.y[i] : 0.0;
    if (tid == 0) *p.info = 0;
    __syncthreads();

    const double ld = small_build_factor<MATERN>(As, Ts, p.X, D, N, p.a, p.b, p.info);
    // ---- L, L^-1 and (L^-1)^T to global memory (identity padding outside the leading Nb x Nb block, zeros above / below) ----
    for (int idx = tid; idx < Np * Np; idx += 256) {
        const int i = idx & (Np - 1), j = idx >> 7;
        double l = 0.0, t = 0.0;
        if (i >= Nb || j >= Nb) {
            l = t = (i == j) ? 1.0 : 0.0;
        } else if (i >= j) {
            l = As[i + j * DL];
            // L^-1: same 16 x 16 tile -> the tile's inverse in Ts; other tiles -> the transposed slot in the upper triangle of As
            t = (i >> 4) == (j >> 4) ? Ts[256 * (i >> 4) + (i & 15) + 16 * (j & 15)] : As[j + i * DL];
        }
        p.L[idx] = l;
        p.Linv[idx] = t;
        p.U[j + (long)i * Np] = t;
    }
    __syncthreads();
    small_inverse_in_place(As, Ts, N);
    for (int idx = tid; idx < Np * Np; idx += 256) {
        const int i = idx & (Np - 1), j = idx >> 7;
        p.Kinv[idx] = (i < Nb && j < Nb) ? As[i + j * DL] : (i == j ? 1.0 : 0.0);
    }
    double gb, quad;
    small_alpha(As, N, gb, quad);
    // ---- alpha, alpha o X~, the posterior mean at the data points (mu(x_i) = y_i - b alpha_i) and its first maximum ----
    double mv = -INFINITY;
    int mi = 0x7fffffff;
    if (tid < Np) {
        const double al = tid < N ? small_scratch(As, SC_ALPHA + tid) : 0.0;
        p.alpha[tid] = al;
        for (int d = 0; d < Dcols; ++d) p.XaT[tid + (long)d * Np] = al * p.XT[tid + (long)d * Np];
        if (tid < N) {
            const double m = small_scratch(As, SC_Y + tid) - p.b * al;
            p.mu_data[tid] = m;
            mv = m;
            mi = tid;
        }
    }
    // first maximum (Eigen maxCoeff): highest value, ties -> lowest index; all -inf / NaN -> index 0 (argmax_kernel's rule)
    __syncthreads();
    double* rv = &small_scratch(As, SC_BTL);
    if (tid < 128) {
        small_scratch(As, SC_BTL + tid) = mv;
        small_scratch(As, SC_BTL + 128 + tid) = (double)mi;
    }
    __syncthreads();
    if (tid == 0) {
        double bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = 0; i < N; ++i) {
            const double v = small_scratch(As, SC_BTL + i);
            if (v > bv) { bv = v; bi = i; }
        }
        p.scal[0] = bi == 0x7fffffff ? small_scratch(As, SC_BTL) : bv;
        p.d_idx[0] = bi == 0x7fffffff ? 0 : bi;
        p.scal[1] = 2.0 * ld;
        if (p.summary) {   // what the host reads after the fit, in one mapped block: no copies back
            p.summary[0] = bi == 0x7fffffff ? small_scratch(As, SC_BTL) : bv;
            p.summary[1] = 2.0 * ld;
            p.summary[2] = bi == 0x7fffffff ? 0.0 : (double)bi;
            p.summary[3] = (double)__hip_atomic_load(p.info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.summary[4] = 0.0;
        }
        (void)rv;
    }
}

void launch_gp_fit_small(hipStream_t s, int kernel, const GpFitSmallArgs& args) {
    ensure_dyn_lds((const void*)gp_fit_small_kernel<false>, DIAG_LDS_BYTES);
    ensure_dyn_lds((const void*)gp_fit_small_kernel<true>, DIAG_LDS_BYTES);
    if (kernel == SLS_KERNEL_ARD_MATERN52)
        hipLaunchKernelGGL(gp_fit_small_kernel<true>, dim3(1), dim3(256), DIAG_LDS_BYTES, s, args);
    else
        hipLaunchKernelGGL(gp_fit_small_kernel<false>, dim3(1), dim3(256), DIAG_LDS_BYTES, s, args);
}

void launch_nll_small(hipStream_t s, int kernel, const NllSmallArgs& args) {
    ensure_dyn_lds((const void*)nll_small_kernel<false>, DIAG_LDS_BYTES);
    ensure_dyn_lds((const void*)nll_small_kernel<true>, DIAG_LDS_BYTES);
    if (kernel == SLS_KERNEL_ARD_MATERN52)
        hipLaunchKernelGGL(nll_small_kernel<true>, dim3(args.batch > 1 ? args.batch : 1), dim3(256), DIAG_LDS_BYTES, s, args);
    else
        hipLaunchKernelGGL(nll_small_kernel<false>, dim3(args.batch > 1 ? args.batch : 1), dim3(256), DIAG_LDS_BYTES, s, args);
}

}  // namespace slsk
