// Vector and bookkeeping kernels of the fit: matrix-vector products with a fixed summation order, the rank-1 growth of a fitted
// state by one data point (sls_gp_append_point), fills, the summary of a fit (posterior mean at the data points, its first maximum,
// log|K_y|: src/regressor.cpp:29-43 hoisted) and the border row / reduction of the value-only MAP objective.  Split off
// kernels_chol.hip in round 5: nothing here depends on the factorisation kernels.
#include "kernels.hpp"
#include "../../include/sls_hip.h"

namespace slsk {

// y = A x (A column-major Np x Np): columns split into chunks over blockIdx.y so that the whole chip streams the matrix;
// the per-chunk partials are summed in a fixed order by a second tiny kernel (deterministic).
// lower: A is lower triangular (L^-1): a chunk of columns entirely to the right of this block's rows holds zeros only -- its partial sum
// is written as the +0 the products would add up to, without reading the 256 x 128 zeros (half of the matrix: the fit's two products
// with L^-1 streamed 2 x 512 MB at N = 8192 for 2 x 256 MB of content).  Same bits.
__global__ __launch_bounds__(256) void gemv_n_partial_kernel(const double* __restrict__ A, int Np, const double* __restrict__ x,
                                                             double* __restrict__ part, int cols_per_chunk, int lower) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Np) return;
    const int j0 = blockIdx.y * cols_per_chunk;
    if (lower && j0 > (int)blockIdx.x * 256 + 255) {
        part[(long)blockIdx.y * Np + i] = 0.0;
        return;
    }
    double s = 0.0;
    // loads in flight per lane, added in column order (the bits do not depend on the unrolling).  76-80 us at N = 8192 (3.4 TB/s) whether
    // 4 or 16 are in flight and whether a lane takes one row or two (16-byte loads): not the lane's request rate; left as it is
#pragma unroll 16
    for (int j = j0; j < j0 + cols_per_chunk; ++j) s += A[(long)i + (long)j * Np] * x[j];
    part[(long)blockIdx.y * Np + i] = s;
}
__global__ __launch_bounds__(256) void gemv_n_reduce_kernel(const double* __restrict__ part, int Np, int chunks,
                                                            double* __restrict__ y) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Np) return;
    double s = 0.0;
#pragma unroll 16
    for (int c = 0; c < chunks; ++c) s += part[(long)c * Np + i];
    y[i] = s;
}
void launch_gemv_n(hipStream_t s, const double* A, int Np, const double* x, double* y, double* part, bool lower) {
    const int chunks = Np / 128;                 // 128 columns per chunk
    hipLaunchKernelGGL(gemv_n_partial_kernel, dim3((Np + 255) / 256, chunks), dim3(256), 0, s, A, Np, x, part, 128, lower ? 1 : 0);
    hipLaunchKernelGGL(gemv_n_reduce_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, part, Np, chunks, y);
}

// y_j = sum_i A[i,j] x_i : one wave per column, wave-shuffle reduction
// lower: column j of a lower-triangular A is zero above row j: the wave starts at the 64-row group that holds the diagonal (the terms
// left out are +0: same bits)
__global__ __launch_bounds__(256) void gemv_t_kernel(const double* __restrict__ A, int Np, const double* __restrict__ x,
                                                     double* __restrict__ y, int lower) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= Np) return;
    double s = 0.0;
#pragma unroll 8
    for (int i = (lower ? (j & ~63) : 0) + lane; i < Np; i += 64) s += A[(long)i + (long)j * Np] * x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[j] = s;
}
void launch_gemv_t(hipStream_t s, const double* A, int Np, const double* x, double* y, bool lower) {
    hipLaunchKernelGGL(gemv_t_kernel, dim3((Np + 3) / 4), dim3(256), 0, s, A, Np, x, y, lower ? 1 : 0);
}

__device__ __forceinline__ double block_sum256(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void append_dots_kernel(const double* __restrict__ k, const double* __restrict__ u,
                                                          const double* __restrict__ l, const double* __restrict__ y, int N,
                                                          double* __restrict__ out) {
    __shared__ double red[4];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        a += k[i] * u[i];
        b += l[i] * l[i];
        c += u[i] * y[i];
    }
    const double ta = block_sum256(a, red), tb = block_sum256(b, red), tc = block_sum256(c, red);
    if (threadIdx.x == 0) { out[0] = ta; out[1] = tb; out[2] = tc; }
}
void launch_append_dots(hipStream_t s, const double* k, const double* u, const double* l, const double* y, int N, double* scal_out) {
    hipLaunchKernelGGL(append_dots_kernel, dim3(1), dim3(256), 0, s, k, u, l, y, N, scal_out);
}
__global__ __launch_bounds__(256) void append_update_kernel(double* __restrict__ Kinv, double* __restrict__ L, double* __restrict__ Linv,
                                                            double* __restrict__ alpha, int Np, int N, const double* __restrict__ u,
                                                            const double* __restrict__ l, const double* __restrict__ scal, double kappa,
                                                            double eta) {
    // Schur complement s = kappa - k.K^-1 k (from the inverse) and lam^2 = kappa - l.l (from the factor) are the same number
    const double sch = kappa - scal[0];
    const double lam = sqrt(kappa - scal[1]);
    const double uy = scal[2];
    const int i = blockIdx.x * 256 + threadIdx.x;   // row
    const int j = blockIdx.y;                       // column, 0..N
    if (i > N) return;
    if (j < N && i < N) {
        Kinv[(long)i + (long)j * Np] += u[i] * u[j] / sch;
    } else if (j == N) {
        Kinv[(long)i + (long)N * Np] = (i < N) ? -u[i] / sch : 1.0 / sch;
        if (i < N) {
            Kinv[(long)N + (long)i * Np] = -u[i] / sch;
            L[(long)N + (long)i * Np] = l[i];
            Linv[(long)N + (long)i * Np] = -u[i] / lam;
            alpha[i] += u[i] * (uy - eta) / sch;
        } else {
            L[(long)N * (Np + 1)] = lam;
            Linv[(long)N * (Np + 1)] = 1.0 / lam;
            alpha[N] = (eta - uy) / sch;
        }
    }
}
void launch_append_update(hipStream_t s, double* Kinv, double* L, double* Linv, double* alpha, int Np, int N, const double* u,
                          const double* l, const double* scal, double kappa, double eta) {
    hipLaunchKernelGGL(append_update_kernel, dim3((N + 1 + 255) / 256, N + 1), dim3(256), 0, s, Kinv, L, Linv, alpha, Np, N, u, l, scal,
                       kappa, eta);
}

__global__ __launch_bounds__(256) void zero_upper_kernel(double* __restrict__ A, int Np) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= (long)Np * Np) return;
    const int i = idx % Np, j = idx / Np;
    if (i < j) A[idx] = 0.0;
}
void launch_zero_upper(hipStream_t s, double* A, int Np) {
    const long n = (long)Np * Np;
    hipLaunchKernelGGL(zero_upper_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A, Np);
}

__global__ __launch_bounds__(256) void fill_kernel(double* __restrict__ p, long n, double v) {
    long i = blockIdx.x * 256L + threadIdx.x;
    const long stride = gridDim.x * 256L;
    for (; i < n; i += stride) p[i] = v;
}
void launch_fill(hipStream_t s, double* p, long n, double v) {
    if (n <= 0) return;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n, v);
}

// The tail of a fit in ONE single-workgroup launch (it was mu_data + argmax + logdet, ~5-7 us of stream time each, and three blocking
// pageable copies back): mu_data_i = y_i - b alpha_i (regressor.cpp:29-43 at the data points), its FIRST maximum (Eigen maxCoeff
// semantics, argmax_kernel's comparisons and tree), log|K_y| = 2 sum log L_ii (256 partial sums, i mod 256, then a binary tree), and -- when
// `summary` is given -- everything the host wants after the fit in one mapped block: [0] max mu, [1] log|K_y|, [2] arg max,
// [3], [4] the factorisation's two info words.
__global__ __launch_bounds__(1024) void fit_summary_kernel(const double* __restrict__ y, const double* __restrict__ alpha, double b, int N,
                                                           double* __restrict__ mu_data, const double* __restrict__ L, int Np,
                                                           const int* __restrict__ info, double* __restrict__ scal,
                                                           long* __restrict__ d_idx, double* __restrict__ summary) {
    __shared__ double sv[1024];
    __shared__ int si[1024];
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < N; i += 1024) {
        const double v = y[i] - b * alpha[i];
        mu_data[i] = v;
        if (v > bv) { bv = v; bi = i; }   // strictly greater: keeps the earliest index within this thread
    }
    sv[tid] = bv;
    si[tid] = bi;
    if (tid < 256) {
        double s = 0.0;
        for (int i = tid; i < N; i += 256) s += log(L[(long)i * (Np + 1)]);
        red[tid] = s;
    }
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) {
            const double ov = sv[tid + s];
            const int oi = si[tid + s];
            if (ov > sv[tid] || (ov == sv[tid] && oi < si[tid])) {
                sv[tid] = ov;
                si[tid] = oi;
            }
            if (s <= 128) red[tid] += red[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        // all -inf / NaN: Eigen's maxCoeff returns index 0
        const double best = (si[0] == 0x7fffffff) ? (y[0] - b * alpha[0]) : sv[0];
        const long idx = (si[0] == 0x7fffffff) ? 0 : si[0];
        const double ld = 2.0 * red[0];
        scal[0] = best; scal[1] = ld; d_idx[0] = idx;
        if (summary) {
            summary[0] = best; summary[1] = ld; summary[2] = (double)idx;
            summary[3] = (double)info[0]; summary[4] = (double)info[1];
        }
    }
}
void launch_fit_summary(hipStream_t s, const double* y, const double* alpha, double b, int N, double* mu_data, const double* L, int Np,
                        const int* info, double* scal, long* d_idx, double* summary) {
    hipLaunchKernelGGL(fit_summary_kernel, dim3(1), dim3(1024), 0, s, y, alpha, b, N, mu_data, L, Np, info, scal, d_idx, summary);
}

// ---- bordered factorisation: quad = y^T K^-1 y and log|K| from the factor alone --------------------------------------------
// Row N of the (identity-padded) matrix is replaced by (y^T, c): the factorisation then leaves t = L^-1 y in row N of L
// (L_Nk = (y_k - sum_{m<k} L_Nm L_km) / L_kk is the forward substitution), so y^T K^-1 y = |t|^2 comes out of the ONE persistent
// launch -- no inverse, no separate solve.  c only has to keep the last pivot c - |t|^2 positive; nothing else depends on it.
// Workgroup q of the launch handles problem q (A + q strideA).
__global__ __launch_bounds__(256) void border_row_kernel(double* __restrict__ A, long strideA, int Np, int N, const double* __restrict__ y,
                                                         double c) {
    double* Aq = A + blockIdx.x * strideA;
    for (int k = threadIdx.x; k <= N; k += 256) Aq[N + (long)k * Np] = k < N ? y[k] : c;
}
void launch_border_row(hipStream_t s, double* A, long strideA, int nprob, int Np, int N, const double* y, double c) {
    hipLaunchKernelGGL(border_row_kernel, dim3(nprob), dim3(256), 0, s, A, strideA, Np, N, y, c);
}
// out[2 q] = y^T K^-1 y = sum_k L_Nk^2, out[2 q + 1] = log|K| = 2 sum_{i<N} log L_ii  (fixed summation order)
__global__ __launch_bounds__(256) void border_reduce_kernel(const double* __restrict__ L, long strideA, int Np, int N,
                                                            double* __restrict__ out) {
    __shared__ double red[2][256];
    const double* Lq = L + blockIdx.x * strideA;
    double q = 0.0, ld = 0.0;
    for (int k = threadIdx.x; k < N; k += 256) {
        const double t = Lq[N + (long)k * Np];
        q = fma(t, t, q);
        ld += log(Lq[(long)k * (Np + 1)]);
    }
    red[0][threadIdx.x] = q;
    red[1][threadIdx.x] = ld;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = red[0][0];
        out[2 * blockIdx.x + 1] = 2.0 * red[1][0];
    }
}
void launch_border_reduce(hipStream_t s, const double* L, long strideA, int nprob, int Np, int N, double* out) {
    hipLaunchKernelGGL(border_reduce_kernel, dim3(nprob), dim3(256), 0, s, L, strideA, Np, N, out);
}

}  // namespace slsk
