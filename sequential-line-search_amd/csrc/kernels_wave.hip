// Small-problem path of the multi-start maximiser: ONE WAVEFRONT PER START runs the complete bounded L-BFGS
//  inside a single kernel launch.
//
// The MFMA path (kernels_acq.hip) advances all starts in lock step with ~7 launches per evaluation round, each padded
// to 128-wide tiles; for the reference's real operating sizes (N <= a few hundred data points, 10..1000 starts,
// hundreds of rounds: demos/sequential_line_search_nd, demos/bayesian_optimization_1d) those rounds are pure device
// latency (~150 us each).  Here the per-candidate quantities are computed the way the reference does per point
// (PredictMu/Sigma/...Derivative, src/gaussian-process-regressor.cpp:234-272), but by a wavefront:
//   k_i, c_i        lanes over the data rows i (coalesced reads of the scaled design matrix XT[i + d*Np])
//   w = K^-1 k      lanes over rows, k_j broadcast from LDS, K^-1 columns read coalesced
//   mu, k.w, ...    wave butterfly reductions (__shfl_xor)
//   grad_d          lanes over the dimensions d, c_i alpha_i / c_i w_i broadcast from LDS
// and the L-BFGS state (x, g, direction, history) lives in registers / LDS with lanes over d.
// Per evaluation that is 2 N^2 + 6 N D flops on the VALU of one SIMD: faster than the tiled path while
// N <= 512 and the starts fit the chip a few times over.
#include <cstdlib>

#include "kernels.hpp"
#include "wave_reduce.hpp"
#include "../../include/sls_hip.h"

namespace slsk {

// STAGE: 0 K^-1 and the design matrix are read from global memory, 1 K^-1 from its LDS copy, 2 both from LDS.  A template
// parameter, not a run-time pointer choice: a pointer that may be global OR LDS is a generic pointer, and every read through it
// a flat_load -- measured in round 4 (SLS_WAVE_TRACE=1, N = 61): 7.7 us per evaluation in the K^-1 k loop alone.
// R = Np / 64 rows of the training set per lane, also a template parameter: with a run-time bound the `row r exists` tests of the
// unrolled per-row code became ~130 scalar branches per eight columns of the K^-1 k loop -- the other 7 of its 7.7 us.
// SOLVE: sigma from the Cholesky solve (two triangular passes with L^-1) instead of the explicit K^-1 -- WaveArgs::solve_sigma.
// With STAGE >= 1 the staged matrix is then the SYMMETRIC image S = L^-1 (lower triangle) + L^-T (upper triangle): the first pass
// v = L^-1 k reads S(i, j) for j <= i, the second w = L^-T v reads S(i, j) for j >= i -- both with lanes over i, i.e. the
// conflict-free access of the K^-1 loop, and one matrix in LDS instead of two.
// R = 1 serves N <= 64 (Np = 128, but no lane's second row exists): half the loads of every pass.
// COOP: the launch has ONE start and the four waves of its workgroup share every long sum (see evaluate()).
template <int STAGE, int R, bool SOLVE, bool COOP = false>
__global__ __launch_bounds__(256) void maximize_wave_kernel(WaveArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* smem = reinterpret_cast<double*>(smem_raw);
    // the wave index as a SCALAR: everything derived from it (the wave's LDS arrays, the chain a wave takes in the cooperative form, its
    // exchange slots) is then addressed with scalar arithmetic -- left in a vector register, every term of the cooperative chains
    // carried ~10 vector instructions of index arithmetic (v_min, v_mul_lo_u32, v_cmp, v_cndmask) for one multiply-add
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool tracing = p.trace != nullptr;
    const long long tr_begin = tracing ? wall_clock64() : 0;
    const long long tr_cyc0 = tracing ? clock64() : 0;
#define WAVE_T0 const long long t_0 = tracing ? wall_clock64() : 0; long long t_prev = t_0
#define WAVE_T(slot) do { if (tracing) { const long long t_now = wall_clock64(); tr[slot] += t_now - t_prev; t_prev = t_now; } } while (0)
    const int nraw = COOP ? 0 : blockIdx.x * 4 + wave;     // COOP: all four waves carry start 0 (identical state, shared sums)
    const bool live = COOP ? wave == 0 : nraw < p.S;
    const int n = live ? nraw : p.S - 1;          // surplus waves shadow the last start (uniform barrier counts)
    const int D = p.D, N = p.N, Np = p.Np, m = p.m;
    double* xs = smem + (long)wave * p.lds_per_wave;   // scaled trial point [D]
    double* kb = xs + p.Dr;                             // k_i            [Np]
    double* cab = kb + Np;                              // c_i alpha_i    [Np]
    double* cwb = cab + Np;                             // c_i w_i        [Np]
    double* Sh = cwb + Np;                              // history s      [m][Dr]
    double* Yh = Sh + m * p.Dr;                         // history y      [m][Dr]
    double* rho = Yh + m * p.Dr;                        // [m]
    const int d0 = lane, d1 = lane + 64;
    const bool has0 = d0 < D, has1 = d1 < D;
    // Few starts (the single start of the DIRECT -> L-BFGS branch, the 10..100 of a small multi-start): K^-1 and the design
    // matrix are copied into LDS once per workgroup and every evaluation reads them from there.  An evaluation is three short
    // loops whose every trip needs a global word (L2 at best: ~1 us from a chip that is otherwise idle); with one wavefront per
    // start nothing hides that, and 320 evaluations in sequence took 7 ms (22 us each) at N = 60, D = 32.  Values only move:
    // same bits.
    double* const xch = smem + 4L * p.lds_per_wave;         // COOP: exchange area for the four waves' partial sums (p.xch doubles)
    double* const sharedK = xch + p.xch;
    const int ldS = Np, colsS = p.stage_cols;
    double* const sharedX = sharedK + (STAGE >= 1 ? (long)ldS * colsS : 0L);
    if (STAGE >= 1) {
        if (SOLVE) {
            for (int idx = threadIdx.x; idx < ldS * colsS; idx += 256) {
                const int i = idx % ldS, j = idx / ldS;
                sharedK[idx] = i >= j ? p.Linv[i + (long)j * Np] : p.U[i + (long)j * Np];   // U = (L^-1)^T, every block kept
            }
        } else {
            for (int idx = threadIdx.x; idx < ldS * colsS; idx += 256) sharedK[idx] = p.Kinv[idx];
        }
    }
    if (STAGE >= 2)
        // leading dimension Np + 1 (odd): the kernel-vector loop reads XT with lanes over the ROWS i (stride 1), the gradient loop
        // with lanes over the DIMENSIONS d -- with the global layout's stride of Np = 128 doubles every lane of that second loop
        // hit the same LDS bank (a 32-way conflict per read: ~2 of the 2.7 us its N-long loop took at D = 32)
        for (int idx = threadIdx.x; idx < Np * D; idx += 256) sharedX[(idx % Np) + (long)(idx / Np) * (Np + 1)] = p.XT[idx];
    if (STAGE >= 1) __syncthreads();
    const double* __restrict__ KinvG = SOLVE ? p.Linv : p.Kinv;
    const double* __restrict__ UG = p.U;
    const double* __restrict__ XTG = p.XT;
    // element (i, j) of the first-pass matrix (K^-1, or L^-1 when SOLVE)
    auto kinv_at = [&](int i, int j) -> double {
        if constexpr (STAGE >= 1 && SOLVE) {
            const double x = sharedK[i + (long)j * ldS];   // unconditional LDS read, then a select (a predicated read costs an exec-mask round trip each)
            return j <= i ? x : 0.0;
        }
        else if constexpr (STAGE >= 1) return sharedK[i + (long)j * ldS];
        else return KinvG[i + (long)j * Np];
    };
    // element (i, j) of L^-T (second pass, SOLVE only): the transposed read of the staged L^-1, or U = (L^-1)^T from global memory
    auto linvT_at = [&](int i, int j) -> double {
        if constexpr (STAGE >= 1) {
            const double x = sharedK[i + (long)j * ldS];
            return j >= i ? x : 0.0;
        } else return UG[i + (long)j * Np];
    };
    auto xt_at = [&](int i, int d) -> double {
        if constexpr (STAGE >= 2) return sharedX[i + (long)d * (Np + 1)];
        else return XTG[i + (long)d * Np];
    };

    // per-lane constants of every evaluation, read once: 1 / l_d of this lane's dimensions, alpha_i of its rows
    const double il0 = has0 ? p.inv_ell[d0] : 0.0, il1 = has1 ? p.inv_ell[d1] : 0.0;
    double alr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) alr[r] = (lane + 64 * r < N) ? p.alpha[lane + 64 * r] : 0.0;
    // ---- objective: value and gradient of the acquisition function at xq (lanes over d) ----
    double last_mu = 0.0, last_sigma = 0.0, last_dm[2] = {0.0, 0.0}, last_ds[2] = {0.0, 0.0};   // predictive parts of the last call
    auto evaluate = [&](const double xq0, const double xq1, double& val, double& gr0, double& gr1) {
        WAVE_T0;
        tr[6] += 1;
        if (has0) xs[d0] = (xq0 - 0.5) * il0;
        if (has1) xs[d1] = (xq1 - 0.5) * il1;
        __syncthreads();
        // Every long sum of an evaluation runs as FOUR chains -- terms 0, 4, 8, .. / 1, 5, 9, .. / ..., each in increasing order --
        // added as (c0 + c1) + (c2 + c3): a fixed order, so a start's bits depend on nothing but its own data.  One wavefront per
        // start computes all four chains itself; when the launch has ONE start (COOP: the local phase of the DIRECT -> L-BFGS
        // branch), the workgroup's four waves -- which would otherwise shadow each other -- take one chain each and exchange the
        // partial sums through LDS: the same additions in the same order, a quarter of each loop per wave.
        // The reads of up to 16 loop trips are issued together, their arithmetic follows in order.
        auto combine4 = [](double c0, double c1, double c2, double c3) { return (c0 + c1) + (c2 + c3); };
        // out[k] = combine4 of the four waves' part[k] (COOP only): two workgroup barriers
        auto exchange = [&](auto& part, auto& out, auto K_) {
            constexpr int K = decltype(K_)::value;
#pragma unroll
            for (int k = 0; k < K; ++k) xch[(wave * K + k) * 64 + lane] = part[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < K; ++k)
                out[k] = combine4(xch[(0 * K + k) * 64 + lane], xch[(1 * K + k) * 64 + lane], xch[(2 * K + k) * 64 + lane], xch[(3 * K + k) * 64 + lane]);
            __syncthreads();   // (measured: 40 cycles with the waves in step; a variant without it for N > 64 lost bit-identity with the one-wave form)
        };
        // chain c of  sum_d (xs_d - X~_id)^2  for row i
        auto kvec_chain = [&](int i, int c) {
            double acc = 0.0;
            for (int m0 = 0; c + 4 * m0 < D; m0 += 8) {
                double xv[8], tv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int d = c + 4 * (m0 + u), dc = d < D ? d : D - 1;
                    xv[u] = xs[dc];
                    tv[u] = d < D ? xt_at(i, dc) : xv[u];       // beyond D: difference 0
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double df = xv[u] - tv[u];
                    acc += df * df;
                }
            }
            return acc;
        };
        // chain c of  sum_j M(i_r, j) v_j  for this lane's R rows (M: first- or second-pass matrix, v broadcast from LDS)
        auto mat_chain = [&](auto&& M, const double* v, int c, double (&acc)[R]) {
            constexpr int T = R <= 1 ? 16 : (R <= 2 ? 8 : 4);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = 0.0;
            for (int m0 = 0; c + 4 * m0 < N; m0 += T) {
                double vj[T], cv[T][R];
#pragma unroll
                for (int u = 0; u < T; ++u) {
                    const int j = c + 4 * (m0 + u), jc = j < N ? j : N - 1;
                    const double vv = v[jc];
                    vj[u] = j < N ? vv : 0.0;                    // beyond N: term 0
#pragma unroll
                    for (int r = 0; r < R; ++r) cv[u][r] = M(lane + 64 * r, jc);
                }
#pragma unroll
                for (int u = 0; u < T; ++u) {
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] += cv[u][r] * vj[u];
                }
            }
        };
        // chain c of  sum_i X~_id (c_i alpha_i)  and  sum_i X~_id (c_i w_i)  for dimension d
        auto grad_chain = [&](int d, int c, double& gm, double& gs) {
            gm = 0.0;
            gs = 0.0;
            for (int m0 = 0; c + 4 * m0 < N; m0 += 8) {
                double xv[8], av[8], wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = c + 4 * (m0 + u), ic = i < N ? i : N - 1;
                    const double x = xt_at(ic, d);
                    xv[u] = i < N ? x : 0.0;
                    av[u] = cab[ic];
                    wv[u] = cwb[ic];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    gm += xv[u] * av[u];
                    gs += xv[u] * wv[u];
                }
            }
        };

        double kr[R], cr[R], qrow[R];
        double mu = 0.0, ca = 0.0;
        if constexpr (COOP) {
            double part[R];
#pragma unroll
            for (int r = 0; r < R; ++r) part[r] = (lane + 64 * r < N) ? kvec_chain(lane + 64 * r, wave) : 0.0;
            exchange(part, qrow, std::integral_constant<int, R>{});
        } else {
            // one wave, all four chains: term d belongs to chain d & 3; eight reads in flight
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = lane + 64 * r;
                qrow[r] = 0.0;
                if (i < N) {
                    double q4[4] = {0.0, 0.0, 0.0, 0.0};
                    int d = 0;
                    for (; d + 8 <= D; d += 8) {
                        double xv[8], tv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            xv[u] = xs[d + u];
                            tv[u] = xt_at(i, d + u);
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const double df = xv[u] - tv[u];
                            q4[u & 3] += df * df;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)        // d is a multiple of 8 here
                        if (d + u < D) {
                            const double df = xs[d + u] - xt_at(i, d + u);
                            q4[u & 3] += df * df;
                        }
                    qrow[r] = combine4(q4[0], q4[1], q4[2], q4[3]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            kr[r] = 0.0;
            cr[r] = 0.0;
            const int i = lane + 64 * r;
            if (i < N) {
                const double q = qrow[r];
                if (p.matern) {
                    const double s = sqrt(5.0 * q), e = exp(-s);
                    kr[r] = p.a * (1.0 + s + (5.0 / 3.0) * q) * e;
                    cr[r] = p.a * (5.0 / 3.0) * (1.0 + s) * e;
                } else {
                    kr[r] = p.a * exp(-0.5 * q);
                    cr[r] = kr[r];
                }
                const double al = alr[r];
                mu += al * kr[r];
                ca += al * cr[r];
                cab[i] = cr[r] * al;
            }
            kb[i] = kr[r];
        }
        __syncthreads();
        WAVE_T(0);
        // w = K^-1 k for this lane's rows (SOLVE: v = L^-1 k first)
        double w[R];
        auto full_pass = [&](auto&& M, const double* v) {
            if constexpr (COOP) {
                double part[R];
                mat_chain(M, v, wave, part);
                exchange(part, w, std::integral_constant<int, R>{});
            } else {
                // one wave, all four chains: column j belongs to chain j & 3.  G columns in flight (G a multiple of 4); columns
                // N .. G ceil(N / G) - 1 exist (identity padding; the staged copy holds 16 ceil(N / 16) columns) and meet v_j = 0
                constexpr int G = R <= 1 ? 16 : (R <= 2 ? 8 : 4);
                double w4[4][R];
#pragma unroll
                for (int r = 0; r < R; ++r) w4[0][r] = w4[1][r] = w4[2][r] = w4[3][r] = 0.0;
                for (int j0 = 0; j0 < N; j0 += G) {
                    double vj[G], cv[G][R];
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        vj[u] = v[j0 + u];
#pragma unroll
                        for (int r = 0; r < R; ++r) cv[u][r] = M(lane + 64 * r, j0 + u);
                    }
#pragma unroll
                    for (int u = 0; u < G; ++u) {
#pragma unroll
                        for (int r = 0; r < R; ++r) w4[u & 3][r] += cv[u][r] * vj[u];
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) w[r] = combine4(w4[0][r], w4[1][r], w4[2][r], w4[3][r]);
            }
        };
        full_pass(kinv_at, kb);
        double kw = 0.0, cw = 0.0;
        if constexpr (SOLVE) {
            // w holds v = L^-1 k: sigma^2 = a - |v|^2 (k . LLT.solve(k)); then w = L^-T v, with v broadcast from LDS
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = lane + 64 * r;
                const double v = i < N ? w[r] : 0.0;
                kw += v * v;
                cwb[i] = v;
            }
            __syncthreads();
            full_pass(linvT_at, cwb);
            __syncthreads();                          // every lane has read v before c_i w_i overwrites it
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane + 64 * r;
            if (i < N) {
                if constexpr (!SOLVE) kw += kr[r] * w[r];
                const double t = cr[r] * w[r];
                cw += t;
                cwb[i] = t;
            }
        }
        __syncthreads();
        WAVE_T(1);
        mu = wave_sum(mu);
        ca = wave_sum(ca);
        kw = wave_sum(kw);
        cw = wave_sum(cw);
        WAVE_T(2);
        const double s2 = p.a - kw;
        const double sigma = s2 < 0.0 ? 0.0 : sqrt(s2);
        const double inv_sigma = 1.0 / sigma;
        // gradient: lanes over d
        double dm[2] = {0.0, 0.0}, ds[2] = {0.0, 0.0};
        double gsum[4] = {0.0, 0.0, 0.0, 0.0};        // gm, gs of dimensions lane and lane + 64
        if constexpr (COOP) {
            double part[4] = {0.0, 0.0, 0.0, 0.0};
            if (has0) grad_chain(d0, wave, part[0], part[1]);
            if (has1) grad_chain(d1, wave, part[2], part[3]);
            exchange(part, gsum, std::integral_constant<int, 4>{});
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int d = lane + 64 * e;
                if (d < D) {
                    double gm4[4] = {0.0, 0.0, 0.0, 0.0}, gs4[4] = {0.0, 0.0, 0.0, 0.0};
                    int i = 0;
                    for (; i + 8 <= N; i += 8) {
                        double xv[8], av[8], wv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            xv[u] = xt_at(i + u, d);
                            av[u] = cab[i + u];
                            wv[u] = cwb[i + u];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            gm4[u & 3] += xv[u] * av[u];
                            gs4[u & 3] += xv[u] * wv[u];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)            // i is a multiple of 8 here
                        if (i + u < N) {
                            const double xi = xt_at(i + u, d);
                            gm4[u & 3] += xi * cab[i + u];
                            gs4[u & 3] += xi * cwb[i + u];
                        }
                    gsum[2 * e] = combine4(gm4[0], gm4[1], gm4[2], gm4[3]);
                    gsum[2 * e + 1] = combine4(gs4[0], gs4[1], gs4[2], gs4[3]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int d = lane + 64 * e;
            if (d < D) {
                const double il = e == 0 ? il0 : il1;
                dm[e] = -il * (xs[d] * ca - gsum[2 * e]);
                ds[e] = inv_sigma * il * (xs[d] * cw - gsum[2 * e + 1]);
            }
        }
        last_mu = mu; last_sigma = sigma;
        last_dm[0] = dm[0]; last_dm[1] = dm[1]; last_ds[0] = ds[0]; last_ds[1] = ds[1];
        if (p.acq == SLS_ACQ_EXPECTED_IMPROVEMENT) {
            const double diff = mu - p.mu_best;
            const double u = diff / sigma;
            const double Phi = 0.5 * erfc(-u * 0.70710678118654752440);
            const double phi = exp(-0.5 * u * u) * 0.39894228040143267794;
            const double ei = diff * Phi + sigma * phi;
            const double g0 = Phi * dm[0] + phi * ds[0], g1 = Phi * dm[1] + phi * ds[1];
            bool bad = (sigma < 1e-10) || isnan(ei) || (has0 && isnan(g0)) || (has1 && isnan(g1));
            bad = __any(bad);
            val = bad ? 0.0 : ei;
            gr0 = bad ? 0.0 : g0;
            gr1 = bad ? 0.0 : g1;
        } else {
            val = mu + p.ucb_h * sigma;
            gr0 = dm[0] + p.ucb_h * ds[0];
            gr1 = dm[1] + p.ucb_h * ds[1];
        }
        if (!has0) gr0 = 0.0;
        if (!has1) gr1 = 0.0;
        WAVE_T(3);
    };

    auto clamp01 = [](double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); };
    // a lane's two terms of a dot product over d, d + 64 with the contraction written out: left to the compiler, the instantiations
    // of this kernel (one wave per start / cooperative) may fuse different halves and round differently
    auto dot2 = [](double a0, double b0, double a1, double b1) { return fma(a0, b0, a1 * b1); };

    double val, gr0, gr1;
    if (p.n_local == 0) {
        // evaluation-only mode (sls_gp_predict / sls_gp_predict_grad / sls_acq_eval on small problems): the point is used
        // as given (predictions are defined outside [0,1]^D too); outputs are candidate-major with leading dimension ld
        const double q0 = has0 ? p.starts[d0 + (long)n * D] : 0.0, q1 = has1 ? p.starts[d1 + (long)n * D] : 0.0;
        evaluate(q0, q1, val, gr0, gr1);
        if (live) {
            if (lane == 0) {
                if (p.ev_mu) p.ev_mu[n] = last_mu;
                if (p.ev_sigma) p.ev_sigma[n] = last_sigma;
                if (p.ev_val) p.ev_val[n] = val;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int d = lane + 64 * e;
                if (d < D) {
                    if (p.ev_dmu) p.ev_dmu[n + (long)d * p.ld] = last_dm[e];
                    if (p.ev_dsigma) p.ev_dsigma[n + (long)d * p.ld] = last_ds[e];
                    if (p.ev_grad) p.ev_grad[n + (long)d * p.ld] = e == 0 ? gr0 : gr1;
                }
            }
        }
        return;
    }
    // ---- start ----
    double x0 = has0 ? clamp01(p.starts[d0 + (long)n * D]) : 0.0;
    double x1 = has1 ? clamp01(p.starts[d1 + (long)n * D]) : 0.0;
    evaluate(x0, x1, val, gr0, gr1);
    double f = -val, g0 = -gr0, g1 = -gr1;       // minimise phi = -acq
    double dir0 = 0.0, dir1 = 0.0, t = 1.0;
    int hlen = 0, hpos = 0, nbt = 0;
    double sy_last = 0.0, yy_last = 1.0;
    // slot of the h-th newest pair in the circular history (0 <= hpos < m, h < hlen <= m): a conditional add -- the history position
    // lives in a vector register, and an integer remainder by the run-time m there is ~35 instructions, 13 times per direction
    auto hist_slot = [&](int h) {
        const int i = hpos - 1 - h;
        return i < 0 ? i + m : i;
    };
    bool done = false, need_dir = true;
    int n_useful = 1;                            // evaluations this start needed (the first one included)

    for (int ev = 1; ev < p.n_local; ++ev) {
        // all four starts of the workgroup finished (or the shadows of the last one): nothing left that could change
        // (the barriers inside evaluate() need every wave, so the decision is taken for the workgroup as a whole)
        if (__syncthreads_and(done ? 1 : 0)) break;
        WAVE_T0;
        if (!done && need_dir) {
            // projected gradient, two-loop recursion (same statements as the oracle's lb_direction)
            double pg0 = g0, pg1 = g1;
            if ((x0 <= 0.0 && pg0 > 0.0) || (x0 >= 1.0 && pg0 < 0.0)) pg0 = 0.0;
            if ((x1 <= 0.0 && pg1 > 0.0) || (x1 >= 1.0 && pg1 < 0.0)) pg1 = 0.0;
            if (!has0) pg0 = 0.0;
            if (!has1) pg1 = 0.0;
            const bool pg_above = __any(fmax(fabs(pg0), fabs(pg1)) > p.gtol);   // max |pg| > gtol, by vote (no dependent reduction)
            const double pgn2_lane = dot2(pg0, pg0, pg1, pg1);   // |pg|^2 is summed only where it is used (no history, or a direction that does not descend)
            if (!pg_above) {
                done = true;
            } else {
                double q0 = pg0, q1 = pg1;
                double al[8];
                // the whole history into registers first: one LDS round trip for all pairs instead of one per pair on the chain of
                // dependent sums (slots beyond hlen are read from slot 0 and not used)
                double hs0[8], hs1[8], hy0[8], hy1[8], hrho[8];
                const int d0c = has0 ? d0 : 0, d1c = has1 ? d1 : 0;
                const bool wide = D > 64;
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const int idx = h < hlen ? hist_slot(h) : 0;
                    hs0[h] = Sh[idx * p.Dr + d0c];
                    hy0[h] = Yh[idx * p.Dr + d0c];
                    hs1[h] = wide ? Sh[idx * p.Dr + d1c] : 0.0;
                    hy1[h] = wide ? Yh[idx * p.Dr + d1c] : 0.0;
                    hrho[h] = rho[idx];
                }
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    hs0[h] = has0 ? hs0[h] : 0.0;
                    hy0[h] = has0 ? hy0[h] : 0.0;
                    hs1[h] = has1 ? hs1[h] : 0.0;
                    hy1[h] = has1 ? hy1[h] : 0.0;
                }
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    al[h] = 0.0;
                    if (h < hlen) {
                        al[h] = hrho[h] * wave_sum(dot2(hs0[h], q0, hs1[h], q1));
                        q0 -= al[h] * hy0[h];
                        q1 -= al[h] * hy1[h];
                    }
                }
                double gamma;
                if (hlen > 0) {
                    gamma = sy_last / yy_last;   // s.y / y.y of the newest pair: the sums formed when it was stored (same terms, same order)
                } else {
                    const double nn = sqrt(wave_sum(pgn2_lane));
                    gamma = 1.0 / (nn > 1.0 ? nn : 1.0);
                }
                q0 *= gamma;
                q1 *= gamma;
#pragma unroll
                for (int h = 7; h >= 0; --h) {
                    if (h < hlen) {
                        const double beta = hrho[h] * wave_sum(dot2(hy0[h], q0, hy1[h], q1));
                        q0 += hs0[h] * (al[h] - beta);
                        q1 += hs1[h] * (al[h] - beta);
                    }
                }
                dir0 = (pg0 == 0.0) ? 0.0 : -q0;
                dir1 = (pg1 == 0.0) ? 0.0 : -q1;
                double gd = wave_sum(dot2(pg0, dir0, pg1, dir1));
                if (!(gd < 0.0)) {
                    hlen = 0;
                    const double nn = sqrt(wave_sum(pgn2_lane));
                    gamma = 1.0 / (nn > 1.0 ? nn : 1.0);
                    dir0 = -gamma * pg0;
                    dir1 = -gamma * pg1;
                    gd = wave_sum(dot2(pg0, dir0, pg1, dir1));
                    if (!(gd < 0.0)) done = true;
                }
                if (!done) { t = 1.0; nbt = 0; }
            }
            need_dir = false;
        }
        const double xt0 = done ? x0 : clamp01(x0 + t * dir0);
        const double xt1 = done ? x1 : clamp01(x1 + t * dir1);
        WAVE_T(4);
        evaluate(xt0, xt1, val, gr0, gr1);          // every wave evaluates every round (uniform barriers, lock-step count)
        if (tracing) t_prev = wall_clock64();
        if (!done) {
            ++n_useful;
            const double ft = -val;
            const double sd0 = has0 ? xt0 - x0 : 0.0, sd1 = has1 ? xt1 - x1 : 0.0;
            const double gs = wave_sum(dot2(g0, sd0, g1, sd1));
            if (!__any(dot2(sd0, sd0, sd1, sd1) != 0.0)) {   // |step|^2 == 0: a sum of non-negative terms is zero exactly when every term is
                done = true;
            } else if (ft <= f + p.c1 * gs) {
                const double yd0 = has0 ? -gr0 - g0 : 0.0, yd1 = has1 ? -gr1 - g1 : 0.0;
                const double sy = wave_sum(dot2(sd0, yd0, sd1, yd1));
                const double yy = wave_sum(dot2(yd0, yd0, yd1, yd1));
                if (has0) { Sh[hpos * p.Dr + d0] = sd0; Yh[hpos * p.Dr + d0] = yd0; }
                if (has1) { Sh[hpos * p.Dr + d1] = sd1; Yh[hpos * p.Dr + d1] = yd1; }
                if (sy > 1e-10 * yy && sy > 0.0) {
                    if (lane == 0) rho[hpos] = 1.0 / sy;
                    sy_last = sy;
                    yy_last = yy;
                    hpos = hpos + 1 == m ? 0 : hpos + 1;
                    if (hlen < m) hlen += 1;
                }
                // NLopt's relative stopping tests on the accepted step (sls_lbfgs_opts; 0 = off)
                if (lbfgs_f_stalled(f, ft, p.ftol_rel) ||
                    (p.xtol_rel > 0.0 && !__any((has0 && lbfgs_x_moved(x0, xt0, p.xtol_rel) != 0.0) || (has1 && lbfgs_x_moved(x1, xt1, p.xtol_rel) != 0.0))))
                    done = true;
                x0 = xt0; x1 = xt1;
                g0 = -gr0; g1 = -gr1;
                f = ft;
                need_dir = true;
            } else {
                t *= p.shrink;
                nbt += 1;
                if (nbt > p.max_backtracks) done = true;
            }
        }
        WAVE_T(5);
    }
    if (tracing && blockIdx.x == 0 && threadIdx.x == 0) {
        tr[7] = wall_clock64() - tr_begin;
        for (int q = 0; q < 8; ++q) p.trace[q] = tr[q];
        p.trace[8] = clock64() - tr_cyc0;        // shader-clock cycles of the same interval: the clock this lone workgroup ran at
    }
    if (live) {
        if (has0) p.x_out[n + (long)d0 * p.ld] = x0;
        if (has1) p.x_out[n + (long)d1 * p.ld] = x1;
        if (lane == 0) p.f_out[n] = f;
        if (lane == 0 && p.useful) atomicAdd(p.useful, (unsigned long long)n_useful);
    }
}

size_t wave_lds_bytes(int D, int Np, int m, int* lds_per_wave, int* Dr) {
    const int dr = (D + 1) & ~1;                                  // keep every sub-array 16-byte aligned
    const int per = dr + 3 * Np + 2 * m * dr + ((m + 1) & ~1);
    *lds_per_wave = per;
    *Dr = dr;
    return (size_t)4 * per * sizeof(double);
}

void launch_maximize_wave(hipStream_t s, WaveArgs a) {
    size_t bytes = wave_lds_bytes(a.D, a.Np, a.m, &a.lds_per_wave, &a.Dr);
    // staging only for launches that leave the chip idle anyway (<= 64 workgroups): with many starts the occupancy is worth more
    const bool allow = tune_on(TUNE_WAVE_STAGE) && a.S <= 256 && a.n_local > 1;
    // ONE start (the local phase of the DIRECT -> L-BFGS branch) on a problem of at most 128 points: the four waves of the
    // workgroup share every long sum (COOP) instead of shadowing each other
    const int R = a.N <= 64 ? 1 : a.Np / 64;
    const bool coop = a.S == 1 && a.n_local > 1 && R <= 2 && tune_on(TUNE_WAVE_COOP);
    a.xch = coop ? 4 * 64 * 4 : 0;
    bytes += (size_t)a.xch * 8;
    // staged first-pass matrix: 16 ceil(N / 16) columns of K^-1, or of the symmetric image of L^-1 / L^-T (solve-based sigma)
    a.stage_ld = a.Np;
    a.stage_cols = std::min(a.Np, (a.N + 15) & ~15);
    const size_t cap = 160 * 1024, xb = (size_t)(a.Np + 1) * a.D * 8;
    const size_t kb = (size_t)a.Np * a.stage_cols * 8;
    a.stage_kinv = allow && bytes + kb <= cap;
    if (a.stage_kinv) bytes += kb;
    a.stage_xt = allow && a.stage_kinv && bytes + xb <= cap;
    if (a.stage_xt) bytes += xb;
    // opt in to the CU's whole LDS once per device.  The staged forms exist for Np = 128 only (R <= 2).
    const dim3 grid(coop ? 1 : (a.S + 3) / 4), block(256);
#define SLS_WAVE_LAUNCH(ST, RR, SV, CO)                                                              \
    do {                                                                                             \
        ensure_dyn_lds((const void*)maximize_wave_kernel<ST, RR, SV, CO>, 160 * 1024);               \
        hipLaunchKernelGGL((maximize_wave_kernel<ST, RR, SV, CO>), grid, block, bytes, s, a);        \
    } while (0)
#define SLS_WAVE_BY_MODE(ST, RR)                                                  \
    do {                                                                          \
        if (a.solve_sigma && coop) SLS_WAVE_LAUNCH(ST, RR, true, true);           \
        else if (a.solve_sigma) SLS_WAVE_LAUNCH(ST, RR, true, false);             \
        else if (coop) SLS_WAVE_LAUNCH(ST, RR, false, true);                      \
        else SLS_WAVE_LAUNCH(ST, RR, false, false);                               \
    } while (0)
#define SLS_WAVE_BY_MODE_NOCOOP(ST, RR)                                           \
    do {                                                                          \
        if (a.solve_sigma) SLS_WAVE_LAUNCH(ST, RR, true, false);                  \
        else SLS_WAVE_LAUNCH(ST, RR, false, false);                               \
    } while (0)
    if (R <= 2 && a.stage_xt) {
        if (R == 1) SLS_WAVE_BY_MODE(2, 1);
        else SLS_WAVE_BY_MODE(2, 2);
    } else if (R <= 2 && a.stage_kinv && !a.solve_sigma && !coop) {
        if (R == 1) SLS_WAVE_LAUNCH(1, 1, false, false);
        else SLS_WAVE_LAUNCH(1, 2, false, false);
    } else {
        if (a.stage_kinv) bytes -= kb;                // (the first-pass matrix alone is not staged in the solve / cooperative forms)
        a.stage_kinv = a.stage_xt = 0;
        if (R == 1) SLS_WAVE_BY_MODE(0, 1);
        else if (R == 2) SLS_WAVE_BY_MODE(0, 2);
        else if (R == 4) SLS_WAVE_BY_MODE_NOCOOP(0, 4);
        else if (R == 6) SLS_WAVE_BY_MODE_NOCOOP(0, 6);
        else SLS_WAVE_BY_MODE_NOCOOP(0, 8);
    }
#undef SLS_WAVE_BY_MODE_NOCOOP
#undef SLS_WAVE_BY_MODE
#undef SLS_WAVE_LAUNCH
}

}  // namespace slsk
