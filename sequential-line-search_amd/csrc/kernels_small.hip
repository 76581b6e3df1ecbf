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
#include "../../include/sls_hip.h"

namespace slsk {

constexpr int SMALL_OUT_GL = 8;        // out[8 .. 8+D): length-scale gradient (D <= NLL_SMALL_MAX_GRAD_D)
constexpr int SMALL_OUT_ALPHA = 32;    // out[32 .. 32+N): alpha

__device__ __forceinline__ double& small_scratch(double* As, int k) { return As[(k >> 4) * DL + 128 + (k & 15)]; }

__device__ __forceinline__ double small_block_sum(double v, double* As) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) small_scratch(As, 1024 + (threadIdx.x >> 6)) = v;
    __syncthreads();
    return (small_scratch(As, 1024) + small_scratch(As, 1025)) + (small_scratch(As, 1026) + small_scratch(As, 1027));
}

// scratch map: [0,128) alpha, [128,256) y, [256,384) 1/l, [1024,1028) reduction slots
template <bool MATERN>
__global__ __launch_bounds__(256) void nll_small_kernel(const NllSmallArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* As = reinterpret_cast<double*>(smem);
    double* Ts = As + 128 * DL;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, fk = lane >> 4;
    const double* __restrict__ X = args.X;
    double* __restrict__ out = args.out + (long)blockIdx.x * args.out_stride;
    int* __restrict__ info = args.info + blockIdx.x;
    const double* __restrict__ in_dev = args.in_dev ? args.in_dev + (long)blockIdx.x * args.in_stride : nullptr;
    const int D = args.D, N = args.N, want_grad = args.want_grad;
    const int nb16 = (N + 15) >> 4, Nb = 16 * nb16;
    // hyper-parameters and targets travel in the kernel argument block (no upload); D > 32 (NLL_SMALL_MAX_ARG_D) falls back to a device buffer
    const double a = in_dev ? in_dev[0] : args.a, b = in_dev ? in_dev[1] : args.b;
    for (int d = tid; d < D; d += 256) small_scratch(As, 256 + d) = 1.0 / (in_dev ? in_dev[2 + d] : args.ell[d]);
    for (int i = tid; i < 128; i += 256)
        small_scratch(As, 128 + i) = i < N ? (in_dev ? in_dev[2 + D + i] : args.y[i]) : 0.0;
    if (tid == 0) *info = 0;
    __syncthreads();

    auto pair_q = [&](int i, int j) {
        double q = 0.0;
        for (int d = 0; d < D; ++d) {
            const double t = (X[d + (long)i * D] - X[d + (long)j * D]) * small_scratch(As, 256 + d);
            q = fma(t, t, q);
        }
        return q;
    };
    auto kern = [&](double q, double& k, double& c) {
        if (!MATERN) {
            k = a * exp(-0.5 * q);
            c = k;
        } else {
            const double s = sqrt(5.0 * q), e = exp(-s);
            k = a * (1.0 + s + (5.0 / 3.0) * q) * e;
            c = a * (5.0 / 3.0) * (1.0 + s) * e;
        }
    };

    // ---- K_y (lower triangle + full diagonal tiles), identity padding up to the next multiple of 16 ----
    for (int idx = tid; idx < Nb * Nb; idx += 256) {
        const int i = idx % Nb, j = idx / Nb;
        if (i < j) continue;
        double v;
        if (i >= N) v = (i == j) ? 1.0 : 0.0;
        else if (i == j) v = a + b;
        else {
            double k, c;
            kern(pair_q(i, j), k, c);
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

    // ---- alpha = K^-1 y ----
    if (tid < 128) {
        double s = 0.0;
        if (tid < N)
            for (int j = 0; j < N; ++j) s = fma(As[tid + j * DL], small_scratch(As, 128 + j), s);
        small_scratch(As, tid) = s;
        if (tid < N && args.batch <= 1) out[SMALL_OUT_ALPHA + tid] = s;   // batch mode: 8 output words per parameter set
    }
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    if (tid < N) {
        const double al = small_scratch(As, tid);
        s1 = al * al - As[tid + tid * DL];
        s2 = small_scratch(As, 128 + tid) * al;
    }
    const double gb = 0.5 * small_block_sum(s1, As);
    const double quad = small_block_sum(s2, As);

    // ---- gradient contractions over the pairs i >= j ----
    double sa = 0.0;
    double gl[NLL_SMALL_MAX_GRAD_D];
#pragma unroll
    for (int d = 0; d < NLL_SMALL_MAX_GRAD_D; ++d) gl[d] = 0.0;
    if (want_grad) {
        for (int idx = tid; idx < N * N; idx += 256) {
            const int i = idx % N, j = idx / N;
            if (i < j) continue;
            const double w = (i == j ? 0.5 : 1.0) * (small_scratch(As, i) * small_scratch(As, j) - As[i + j * DL]);
            double q = 0.0, dd[NLL_SMALL_MAX_GRAD_D];
#pragma unroll
            for (int d = 0; d < NLL_SMALL_MAX_GRAD_D; ++d) {
                dd[d] = 0.0;
                if (d < D) {
                    const double t = (X[d + (long)i * D] - X[d + (long)j * D]) * small_scratch(As, 256 + d);
                    dd[d] = t * t;
                    q += dd[d];
                }
            }
            if (D > NLL_SMALL_MAX_GRAD_D) q = pair_q(i, j);   // length-scale gradient not requested for such D (host check)
            if (i == j) q = 0.0;
            double k, c;
            kern(q, k, c);
            sa = fma(w, k, sa);
            const double g = w * c;
#pragma unroll
            for (int d = 0; d < NLL_SMALL_MAX_GRAD_D; ++d) gl[d] = fma(g, dd[d], gl[d]);
        }
    }
    const double sa_t = small_block_sum(sa, As);
    if (want_grad) {
#pragma unroll
        for (int d = 0; d < NLL_SMALL_MAX_GRAD_D; ++d) {
            if (d < D) {   // D is uniform: every thread takes part in the block sums
                const double t = small_block_sum(gl[d], As);
                if (tid == 0) out[SMALL_OUT_GL + d] = t * small_scratch(As, 256 + d);
            }
        }
    }
    if (tid == 0) {
        out[0] = sa_t;
        out[1] = gb;
        out[2] = quad;
        out[3] = 2.0 * ld;
        out[4] = (double)__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
