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
                                                            double* __restrict__ G, double* __restrict__ wk_part,
                                                            double* __restrict__ row_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int nt = Np / GEMM_BM;
    const int tm = blockIdx.x % nt, tn = blockIdx.x / nt;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    Acc acc;
    acc.zero();
    gemm_tile<false, false>(acc, XT + m0, ld, XT + n0, ld, 0, Dp, lds);
    double part = 0.0;
    d2_t rs = {0.0, 0.0};   // this thread's share of the row sums of G: acc_tile_by_columns gives a thread ONE row pair, 32 columns
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
        rs += g;
        *reinterpret_cast<d2_t*>(G + off) = g;
    });
    __syncthreads();
    // deterministic block reduction: wave shuffle tree, then 4 waves through LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = part;
    // (G 1) over this tile's 128 columns: the four waves hold disjoint column sets of the same rows, summed in wave order
    *reinterpret_cast<d2_t*>(lds + 8 + 128 * (threadIdx.x >> 6) + 2 * (threadIdx.x & 63)) = rs;
    __syncthreads();
    if (threadIdx.x == 0) wk_part[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    if (row_part && threadIdx.x < 128) {
        const double* r = lds + 8 + threadIdx.x;
        row_part[(long)tn * Np + m0 + threadIdx.x] = (r[0] + r[128]) + (r[256] + r[384]);
    }
}

void launch_nll_weight(hipStream_t s, const double* XT, long ld, int Dp, const double* nx, int Np, int N, KernelSpec ks,
                       const double* alpha, const double* Kinv, double* G, double* wk_part, double* row_part) {
    ensure_dyn_lds((const void*)nll_weight_kernel<false>, GEMM_LDS_BYTES);
    ensure_dyn_lds((const void*)nll_weight_kernel<true>, GEMM_LDS_BYTES);
    const int nt = Np / GEMM_BM;
    if (ks.kernel == SLS_KERNEL_ARD_MATERN52)
        hipLaunchKernelGGL(nll_weight_kernel<true>, dim3(nt * nt), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, XT, ld, Dp, nx, Np, N, ks.a,
                           alpha, Kinv, G, wk_part, row_part);
    else
        hipLaunchKernelGGL(nll_weight_kernel<false>, dim3(nt * nt), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, XT, ld, Dp, nx, Np, N, ks.a,
                           alpha, Kinv, G, wk_part, row_part);
}

__device__ __forceinline__ double block_sum_256(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// The three sums of an evaluation by ONE workgroup: out[0] = sum_t part[t] (fixed order), out[1] = 1/2 (alpha.alpha - tr Kinv),
// out[2] = y.alpha.  Lfac != nullptr: out[4] = log|K_y| = 2 sum log L_ii as well (256 partial sums, i mod 256, then a binary tree: the order fit_summary_kernel uses);
// info != nullptr: out[5], out[6] = the factorisation's two info words, so that the results of an evaluation sit in one block.
struct NllScalarArgs {
    const double* part; int nparts;
    const double *alpha, *y, *Kinv;
    int Np, N;
    double* out;
    const double* Lfac;
    const int* info;
};
__device__ __forceinline__ void nll_scalars_block(const NllScalarArgs& a, double* red, double* red256) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    // latency-bound: unrolled so that the (independent) loads of eight iterations are in flight together; the additions keep their order
#pragma unroll 8
    for (int i = threadIdx.x; i < a.nparts; i += 256) s0 += a.part[i];
#pragma unroll 8
    for (int i = threadIdx.x; i < a.N; i += 256) {
        s1 += a.alpha[i] * a.alpha[i] - a.Kinv[(long)i * (a.Np + 1)];
        s2 += a.y[i] * a.alpha[i];
    }
    if (a.Lfac) {
        double s = 0.0;
#pragma unroll 8
        for (int i = threadIdx.x; i < a.N; i += 256) s += log(a.Lfac[(long)i * (a.Np + 1)]);
        red256[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red256[threadIdx.x] += red256[threadIdx.x + o];
            __syncthreads();
        }
    }
    const double t0 = block_sum_256(s0, red);
    const double t1 = block_sum_256(s1, red);
    const double t2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
        a.out[0] = t0; a.out[1] = 0.5 * t1; a.out[2] = t2;
        if (a.Lfac) a.out[4] = 2.0 * red256[0];
        if (a.info) { a.out[5] = (double)a.info[0]; a.out[6] = (double)a.info[1]; }
    }
}
__global__ __launch_bounds__(256) void nll_scalars_kernel(NllScalarArgs a) {
    __shared__ double red[4];
    __shared__ double red256[256];
    nll_scalars_block(a, red, red256);
}
void launch_nll_scalars(hipStream_t s, const double* part, int nparts, const double* alpha, const double* y,
                        const double* Kinv, int Np, int N, double* out, const double* Lfac, const int* info) {
    hipLaunchKernelGGL(nll_scalars_kernel, dim3(1), dim3(256), 0, s, NllScalarArgs{part, nparts, alpha, y, Kinv, Np, N, out, Lfac, info});
}

// Two sums "over chunks, in chunk order" in one launch (each tiny launch costs ~10 us of the stream's time at these sizes):
//   blocks [0, nby):  Y[idx] = sum_c Ypart[c][idx], in place in chunk 0 (split-K partial products of launch_gemm_splitk_nt);
//   the rest:         svec_i = sum_c row_part[c][i]: the row sums (G 1) from the per-tile-column partials of nll_weight_kernel.
// (Folding the first sum into lengthscale_grad_kernel -- D workgroups, `chunks` strided reads per element -- cost 63 us instead of 11 + 8.)
__global__ __launch_bounds__(256) void sum_chunks_kernel(double* __restrict__ Y, long n, int chunks, long stride, int nby,
                                                         const double* __restrict__ row_part, int Np, int row_chunks,
                                                         double* __restrict__ svec) {
    if ((int)blockIdx.x < nby) {
        const long i = 2 * (blockIdx.x * 256L + threadIdx.x);
        if (i >= n) return;
        d2_t v = *reinterpret_cast<const d2_t*>(Y + i);
        for (int c = 1; c < chunks; ++c) v += *reinterpret_cast<const d2_t*>(Y + (long)c * stride + i);
        *reinterpret_cast<d2_t*>(Y + i) = v;
    } else {
        const int i = ((int)blockIdx.x - nby) * 256 + threadIdx.x;
        if (i >= Np) return;
        double s = 0.0;
        for (int c = 0; c < row_chunks; ++c) s += row_part[(long)c * Np + i];
        svec[i] = s;
    }
}
void launch_sum_chunks(hipStream_t s, double* Y, long n, int chunks, long stride, const double* row_part, int Np, int row_chunks,
                       double* svec) {
    const int nby = chunks > 1 ? (int)((n / 2 + 255) / 256) : 0;
    hipLaunchKernelGGL(sum_chunks_kernel, dim3((unsigned)(nby + (Np + 255) / 256)), dim3(256), 0, s, Y, n, chunks, stride, nby, row_part, Np,
                       row_chunks, svec);
}

// blocks [0, D): gl[p] = 2 inv_ell[p] sum_j XT[j,p] (XT[j,p] s_j - Y[j,p]);  block D: the sums of the evaluation (nll_scalars_block)
__global__ __launch_bounds__(256) void lengthscale_grad_kernel(const double* __restrict__ XT, const double* __restrict__ Y,
                                                               const double* __restrict__ svec, const double* __restrict__ inv_ell,
                                                               long ld, int N, int D, double* __restrict__ gl, NllScalarArgs sa) {
    __shared__ double red[4];
    __shared__ double red256[256];
    const int p = blockIdx.x;
    if (p == D) {
        nll_scalars_block(sa, red, red256);
        return;
    }
    double acc = 0.0;
    for (int j = threadIdx.x; j < N; j += 256) {
        const double x = XT[j + (long)p * ld];
        acc += x * (x * svec[j] - Y[j + (long)p * ld]);
    }
    const double t = block_sum_256(acc, red);
    if (threadIdx.x == 0) gl[p] = 2.0 * inv_ell[p] * t;
}
void launch_lengthscale_grad(hipStream_t s, const double* XT, const double* Y, const double* svec, const double* inv_ell,
                             long ld, int N, int D, double* gl, const double* part, int nparts, const double* alpha, const double* y,
                             const double* Kinv, int Np, double* out, const double* Lfac, const int* info) {
    hipLaunchKernelGGL(lengthscale_grad_kernel, dim3(D + 1), dim3(256), 0, s, XT, Y, svec, inv_ell, ld, N, D, gl,
                       NllScalarArgs{part, nparts, alpha, y, Kinv, Np, N, out, Lfac, info});
}

}  // namespace slsk
