// MAP objective gradient pieces: log marginal likelihood of GaussianProcessRegressor
// (src/gaussian-process-regressor.cpp:66-127, 141-193) and the GP term of the preference objective
// (src/preference-regressor.cpp:53-115).  The reference materialises (D+1) dense N x N derivative matrices
// (src/regressor.cpp:110-134) and takes one N^3 trace per hyper-parameter; here
//     dL/dtheta_p = 1/2 sum_jk W_jk dK_jk/dtheta_p ,  W = alpha alpha^T - K^-1
// is contracted tile by tile: one fused pass rebuilds the kernel tile on the MFMA, forms G = 1/2 W .* c, and the
// D length-scale derivatives come from ONE N x N x D GEMM (Y = G X~) plus a reduction:
//     dL/dl_p = (2 / l_p) sum_j x~_pj ( x~_pj (G 1)_j - Y_jp ).
#include "gemm_f64.hpp"
#include "kernels.hpp"
#include "../../include/sls_hip.h"

namespace slsk {

template <bool MATERN>
__global__ __launch_bounds__(256) void nll_weight_kernel(const double* __restrict__ XT, long ld, int Dp,
                                                            const double* __restrict__ nx, int Np, int N, double a,
                                                            const double* __restrict__ alpha, const double* __restrict__ Kinv,
                                                            double* __restrict__ G, double* __restrict__ wk_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int nt = Np / GEMM_BM;
    const int tm = blockIdx.x % nt, tn = blockIdx.x / nt;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    Acc acc;
    acc.zero();
    gemm_tile<false, false>(acc, XT + m0, ld, XT + n0, ld, 0, Dp, lds);
    double part = 0.0;
    // column by column (acc_tile_by_columns): K^-1 is read and G written with 16-byte accesses, 1 KB per wave instruction
    acc_tile_by_columns<true>(acc, lds, [&](int col, int row, d2_t dot) {
        const int gj = n0 + col;
        const double nj = nx[gj], aj = alpha[gj];
        const d2_t ni = *reinterpret_cast<const d2_t*>(nx + m0 + row);
        const d2_t ai = *reinterpret_cast<const d2_t*>(alpha + m0 + row);
        const long off = (long)(m0 + row) + (long)gj * Np;
        const d2_t kinv = *reinterpret_cast<const d2_t*>(Kinv + off);
        d2_t g;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int gi = m0 + row + e;
            double q = ni[e] + nj - 2.0 * dot[e];
            q = (q < 0.0 || gi == gj) ? 0.0 : q;
            double k, c;
            if (!MATERN) {
                k = a * exp(-0.5 * q);
                c = k;
            } else {
                const double s = sqrt(5.0 * q), ex = exp(-s);
                k = a * (1.0 + s + (5.0 / 3.0) * q) * ex;
                c = a * (5.0 / 3.0) * (1.0 + s) * ex;
            }
            double w = 0.5 * (ai[e] * aj - kinv[e]);
            if (gi >= N || gj >= N) w = 0.0;
            g[e] = w * c;
            part += w * k;
        }
        *reinterpret_cast<d2_t*>(G + off) = g;
    });
    __syncthreads();
    // deterministic block reduction: wave shuffle tree, then 4 waves through LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) wk_part[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

void launch_nll_weight(hipStream_t s, const double* XT, long ld, int Dp, const double* nx, int Np, int N, KernelSpec ks,
                       const double* alpha, const double* Kinv, double* G, double* wk_part) {
    ensure_dyn_lds((const void*)nll_weight_kernel<false>, GEMM_LDS_BYTES);
    ensure_dyn_lds((const void*)nll_weight_kernel<true>, GEMM_LDS_BYTES);
    const int nt = Np / GEMM_BM;
    if (ks.kernel == SLS_KERNEL_ARD_MATERN52)
        hipLaunchKernelGGL(nll_weight_kernel<true>, dim3(nt * nt), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, XT, ld, Dp, nx, Np, N, ks.a,
                           alpha, Kinv, G, wk_part);
    else
        hipLaunchKernelGGL(nll_weight_kernel<false>, dim3(nt * nt), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, XT, ld, Dp, nx, Np, N, ks.a,
                           alpha, Kinv, G, wk_part);
}

__device__ __forceinline__ double block_sum_256(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void nll_scalars_kernel(const double* __restrict__ part, int nparts,
                                                          const double* __restrict__ alpha, const double* __restrict__ y,
                                                          const double* __restrict__ Kinv, int Np, int N, double* __restrict__ out) {
    __shared__ double red[4];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s0 += part[i];
    for (int i = threadIdx.x; i < N; i += 256) {
        s1 += alpha[i] * alpha[i] - Kinv[(long)i * (Np + 1)];
        s2 += y[i] * alpha[i];
    }
    const double t0 = block_sum_256(s0, red);
    const double t1 = block_sum_256(s1, red);
    const double t2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = 0.5 * t1; out[2] = t2; }
}
void launch_nll_scalars(hipStream_t s, const double* part, int nparts, const double* alpha, const double* y,
                        const double* Kinv, int Np, int N, double* out) {
    hipLaunchKernelGGL(nll_scalars_kernel, dim3(1), dim3(256), 0, s, part, nparts, alpha, y, Kinv, Np, N, out);
}

// Y[idx] = sum_c Ypart[c][idx] in chunk order (split-K partial products of launch_gemm_splitk_nt), in place in chunk 0
__global__ __launch_bounds__(256) void sum_chunks_kernel(double* __restrict__ Y, long n, int chunks, long stride) {
    const long i = 2 * (blockIdx.x * 256L + threadIdx.x);
    if (i >= n) return;
    d2_t v = *reinterpret_cast<const d2_t*>(Y + i);
    for (int c = 1; c < chunks; ++c) v += *reinterpret_cast<const d2_t*>(Y + (long)c * stride + i);
    *reinterpret_cast<d2_t*>(Y + i) = v;
}
void launch_sum_chunks(hipStream_t s, double* Y, long n, int chunks, long stride) {
    if (chunks <= 1) return;
    hipLaunchKernelGGL(sum_chunks_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, s, Y, n, chunks, stride);
}

__global__ __launch_bounds__(256) void lengthscale_grad_kernel(const double* __restrict__ XT, const double* __restrict__ Y,
                                                               const double* __restrict__ svec, const double* __restrict__ inv_ell,
                                                               long ld, int N, int D, double* __restrict__ gl) {
    __shared__ double red[4];
    const int p = blockIdx.x;
    double acc = 0.0;
    for (int j = threadIdx.x; j < N; j += 256) {
        const double x = XT[j + (long)p * ld];
        acc += x * (x * svec[j] - Y[j + (long)p * ld]);
    }
    const double t = block_sum_256(acc, red);
    if (threadIdx.x == 0) gl[p] = 2.0 * inv_ell[p] * t;
}
void launch_lengthscale_grad(hipStream_t s, const double* XT, const double* Y, const double* svec, const double* inv_ell,
                             long ld, int N, int D, double* gl) {
    hipLaunchKernelGGL(lengthscale_grad_kernel, dim3(D), dim3(256), 0, s, XT, Y, svec, inv_ell, ld, N, D, gl);
}

}  // namespace slsk
