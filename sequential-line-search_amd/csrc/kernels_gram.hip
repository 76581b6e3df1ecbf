// Gram-matrix kernels: CalcLargeKF/KY (src/regressor.cpp:61-89) and batched CalcSmallK (src/regressor.cpp:45-59),
// ARD squared-exponential and ARD Matern-5/2 (mathtoolbox kernel functions).
//
// q_ij = |x~_i|^2 + |x~_j|^2 - 2 x~_i . x~_j with x~ = (x - 0.5)/l; the dot products run on the fp64 MFMA tile
// (gemm_f64.hpp), the transcendental epilogue on the VALU, and the result tile is written once, coalesced
// (HBM-write bound: 8 N^2 bytes).
#include "gemm_f64.hpp"
#include "kernels.hpp"
#include "../../include/sls_hip.h"

namespace slsk {

// k and the first-argument-derivative weight c (dk/dx_d = -c (x_d - x'_d) / l_d^2)
__device__ __forceinline__ void kernel_kc(int kernel, double a, double q, double& k, double& c) {
    if (kernel == SLS_KERNEL_ARD_SQUARED_EXPONENTIAL) {
        k = a * exp(-0.5 * q);
        c = k;
    } else {
        const double s = sqrt(5.0 * q);
        const double e = exp(-s);
        k = a * (1.0 + s + (5.0 / 3.0) * q) * e;
        c = a * (5.0 / 3.0) * (1.0 + s) * e;
    }
}

// 64 points per workgroup.  Pass 1 (all 256 threads, 64 dimensions at a time): scaled coordinates out (64 consecutive doubles per
// dimension) and into LDS; pass 2 (one thread per point): the squared norm, summed over the dimensions in index order -- the order,
// and therefore the bits, of the one-thread-per-point loop this replaces (16 workgroups and 128 dependent strided loads per thread at
// N = 4096, D = 128: 28 us; now 64 workgroups and coalesced accesses).
constexpr int PREP_PTS = 64, PREP_DCH = 64, PREP_LD = PREP_DCH + 1;   // 33 KB of LDS
template <bool CAND_MAJOR>
__global__ __launch_bounds__(256) void prep_kernel(const double* __restrict__ X, long ldr, int D, int M,
                                                   const double* __restrict__ inv_ell, double* __restrict__ XT, long ld, int Mp,
                                                   int Dcols, double* __restrict__ norms) {
    __shared__ double vs[PREP_PTS * PREP_LD];
    __shared__ double ils[PREP_DCH];   // this chunk's inverse length scales: read ONCE per workgroup (they may sit in mapped host memory)
    const int i0 = blockIdx.x * PREP_PTS, tid = threadIdx.x;
    const int p = tid & 63, w = tid >> 6, i = i0 + p;
    double s = 0.0;
    for (int d0 = 0; d0 < Dcols; d0 += PREP_DCH) {
        if (tid < PREP_DCH) ils[tid] = d0 + tid < D ? inv_ell[d0 + tid] : 0.0;
        double raw[PREP_DCH / 4];
        // all 16 loads of a thread in flight together (constant trip count, unrolled), then the arithmetic
        if (CAND_MAJOR) {          // X[i + d ldr]: points contiguous
#pragma unroll
            for (int k = 0; k < PREP_DCH / 4; ++k) {
                const int d = d0 + w + 4 * k;
                raw[k] = (i < M && d < D) ? X[i + (long)d * ldr] : 0.5;
            }
        } else {                   // X[d + i D]: dimensions contiguous -- read along d, written along i through LDS
#pragma unroll
            for (int k = 0; k < PREP_DCH / 4; ++k) {
                const int pp = w + 4 * k, d = d0 + p, ii = i0 + pp;      // lane -> dimension, (wave, k) -> point
                raw[k] = (ii < M && d < D) ? X[d + (long)ii * D] : 0.5;
            }
        }
        __syncthreads();
        if (CAND_MAJOR) {
#pragma unroll
            for (int k = 0; k < PREP_DCH / 4; ++k) {
                const int dd = w + 4 * k, d = d0 + dd;
                const double v = (i < M && d < D) ? (raw[k] - 0.5) * ils[dd] : 0.0;
                if (i < Mp && d < Dcols) XT[i + (long)d * ld] = v;
                vs[p * PREP_LD + dd] = v;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PREP_DCH / 4; ++k) {
                const int pp = w + 4 * k, ii = i0 + pp, d = d0 + p;
                vs[pp * PREP_LD + p] = (ii < M && d < D) ? (raw[k] - 0.5) * ils[p] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PREP_DCH / 4; ++k) {
                const int dd = w + 4 * k, d = d0 + dd;
                if (i < Mp && d < Dcols) XT[i + (long)d * ld] = vs[p * PREP_LD + dd];
            }
        }
        __syncthreads();
        if (tid < PREP_PTS) {
            const int dn = min(PREP_DCH, D - d0);
            for (int dd = 0; dd < dn; ++dd) {
                const double v = vs[tid * PREP_LD + dd];
                s += v * v;
            }
        }
        __syncthreads();
    }
    if (tid < PREP_PTS && i0 + tid < Mp) norms[i0 + tid] = s;
}

void launch_prep_points(hipStream_t s, const double* X, int D, int M, const double* inv_ell, double* XT, long ld, int Mp,
                        int Dcols, double* norms) {
    hipLaunchKernelGGL(prep_kernel<false>, dim3((Mp + PREP_PTS - 1) / PREP_PTS), dim3(256), 0, s, X, 0L, D, M, inv_ell, XT, ld, Mp, Dcols,
                       norms);
}
void launch_prep_cands(hipStream_t s, const double* xr, long ldr, int D, int M, const double* inv_ell, double* XT, long ld,
                       int Mp, int Dcols, double* norms) {
    hipLaunchKernelGGL(prep_kernel<true>, dim3((Mp + PREP_PTS - 1) / PREP_PTS), dim3(256), 0, s, xr, ldr, D, M, inv_ell, XT, ld, Mp, Dcols,
                       norms);
}

// XaT = diag(alpha) XT.  One thread per (row, 16 columns): the launch had Np / 256 workgroups with a Dcols-long loop per thread
// (8 workgroups, 27 us at N = 2048, Dcols = 128).
__global__ __launch_bounds__(256) void scale_rows_kernel(const double* __restrict__ XT, const double* __restrict__ alpha,
                                                         double* __restrict__ XaT, long ld, int Np, int Dcols) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Np) return;
    const double al = alpha[i];
    const int d0 = blockIdx.y * 16;
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = XT[i + (long)(d0 + k) * ld];
#pragma unroll
    for (int k = 0; k < 16; ++k) XaT[i + (long)(d0 + k) * ld] = al * v[k];
}
void launch_scale_rows(hipStream_t s, const double* XT, const double* alpha, double* XaT, long ld, int Np, int Dcols) {
    hipLaunchKernelGGL(scale_rows_kernel, dim3((Np + 255) / 256, Dcols / 16), dim3(256), 0, s, XT, alpha, XaT, ld, Np, Dcols);
}

__global__ __launch_bounds__(256, 2) void gram_sym_kernel(const double* __restrict__ XT, long ld, int Dp,
                                                          const double* __restrict__ nx, int Np, int N, int kernel, double a,
                                                          double b, double* __restrict__ K, int lower_only) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int nt = Np / GEMM_BM;
    int tm, tn;
    if (lower_only) {
        // only the nt (nt + 1) / 2 lower tiles are launched, enumerated row by row; every XCD gets a contiguous run of THAT list
        // (the full nt x nt grid with an early return gave XCD 0 the 484 lower tiles of the first eight tile columns and XCD 7
        // thirty-six: 0.28 ms at N = 8192 where the tiles' own work is half of that)
        const int t = xcd_remap(blockIdx.x, nt * (nt + 1) / 2);
        tm = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while (tm * (tm + 1) / 2 > t) --tm;
        while ((tm + 1) * (tm + 2) / 2 <= t) ++tm;
        tn = t - tm * (tm + 1) / 2;
    } else {
        const int t = xcd_remap(blockIdx.x, nt * nt);
        tm = t % nt;
        tn = t / nt;
    }
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    Acc acc;
    acc.zero();
    gemm_tile<false, false>(acc, XT + m0, ld, XT + n0, ld, 0, Dp, lds);
    // epilogue column by column (acc_tile_by_columns): 16-byte stores, a whole 1 KB column of K per wave instruction
    acc_tile_by_columns<true>(acc, lds, [&](int col, int row, d2_t dot) {
        const int gj = n0 + col;
        const double nj = nx[gj];
        const d2_t ni = *reinterpret_cast<const d2_t*>(nx + m0 + row);
        d2_t out;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int gi = m0 + row + e;
            double q = ni[e] + nj - 2.0 * dot[e];
            q = (q < 0.0 || gi == gj) ? 0.0 : q;
            double k, c;
            kernel_kc(kernel, a, q, k, c);
            if (gi == gj) k += b;
            if (gi >= N || gj >= N) k = (gi == gj) ? 1.0 : 0.0;
            out[e] = k;
        }
        *reinterpret_cast<d2_t*>(K + (long)(m0 + row) + (long)gj * Np) = out;
    });
}

void launch_gram_sym(hipStream_t s, const double* XT, long ld, int Dp, const double* nx, int Np, int N, KernelSpec ks, double b,
                     double* K, bool lower_only) {
    ensure_dyn_lds((const void*)gram_sym_kernel, GEMM_LDS_BYTES);
    const int nt = Np / GEMM_BM;
    hipLaunchKernelGGL(gram_sym_kernel, dim3(lower_only ? nt * (nt + 1) / 2 : nt * nt), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, XT, ld,
                       Dp, nx, Np, N, ks.kernel, ks.a, b, K, (int)lower_only);
}

// One tile = 128 candidates (m) x 128 training points (n').  Writes K*, C* candidate-major and the per-tile partial
// column sums of alpha_i k_in and alpha_i c_in (reduced deterministically by the finalize kernel).
template <bool MATERN>
__global__ __launch_bounds__(256, 2) void cross_gram_kernel(const double* __restrict__ XsT, long lds_, const double* __restrict__ ns,
                                                            int Sp, const double* __restrict__ XT, long ld,
                                                            const double* __restrict__ nx, int Np, int N, int Dp, double a,
                                                            const double* __restrict__ alpha, double* __restrict__ Ks,
                                                            double* __restrict__ Cs, long ldk, double* __restrict__ mu_part,
                                                            double* __restrict__ ca_part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int ntm = Sp / GEMM_BM, ntn = Np / GEMM_BN;
    const int t = xcd_remap(blockIdx.x, ntm * ntn);
    const int tm = t % ntm, tn = t / ntm;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    Acc acc;
    acc.zero();
    gemm_tile<false, false>(acc, XsT + m0, lds_, XT + n0, ld, 0, Dp, lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Epilogue column by column (acc_tile_by_columns): a column = one training point over the tile's 128 candidates, 1 KB
    // contiguous in K* / C*, written with 16-byte stores.  A thread keeps the partial sums of its two candidates over the 32
    // training points it visits (wave + 4 q in each half, ascending); the four waves' sums are added in wave order below.
    // The order depends on nothing but the tile's structure: a candidate's bits do not depend on its position in the batch.
    const d2_t nm = *reinterpret_cast<const d2_t*>(ns + m0 + 2 * lane);
    d2_t smu = {0.0, 0.0}, sca = {0.0, 0.0};
    acc_tile_by_columns<true>(acc, lds, [&](int col, int row, d2_t dot) {
        const int gi = n0 + col;
        const double ni = nx[gi];
        const double al = alpha ? alpha[gi] : 0.0;
        d2_t kv, cv;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            double q = nm[e] + ni - 2.0 * dot[e];
            q = q < 0.0 ? 0.0 : q;
            double k, c;
            kernel_kc(MATERN ? SLS_KERNEL_ARD_MATERN52 : SLS_KERNEL_ARD_SQUARED_EXPONENTIAL, a, q, k, c);
            if (gi >= N) { k = 0.0; c = 0.0; }
            kv[e] = k;
            cv[e] = c;
            smu[e] += al * k;
            sca[e] += al * c;
        }
        // streamed out, read back by acq_gemm from HBM / the Infinity Cache much later: non-temporal stores
        __builtin_nontemporal_store(kv, reinterpret_cast<d2_t*>(Ks + (long)(m0 + row) + (long)gi * ldk));
        if (MATERN) __builtin_nontemporal_store(cv, reinterpret_cast<d2_t*>(Cs + (long)(m0 + row) + (long)gi * ldk));
    });
    if (alpha) {
        __syncthreads();
        double* red = lds;  // [4 waves][2 arrays][128]
        *reinterpret_cast<d2_t*>(red + (wave * 2 + 0) * 128 + 2 * lane) = smu;
        *reinterpret_cast<d2_t*>(red + (wave * 2 + 1) * 128 + 2 * lane) = sca;
        __syncthreads();
        if (threadIdx.x < 128) {
            const int ml = threadIdx.x;
            mu_part[(long)tn * ldk + m0 + ml] = ((red[0 * 128 + ml] + red[2 * 128 + ml]) + red[4 * 128 + ml]) + red[6 * 128 + ml];
            ca_part[(long)tn * ldk + m0 + ml] = ((red[1 * 128 + ml] + red[3 * 128 + ml]) + red[5 * 128 + ml]) + red[7 * 128 + ml];
        }
    }
}

void launch_cross_gram(hipStream_t s, const double* XsT, long lds_, const double* ns, int Sp, const double* XT, long ld,
                       const double* nx, int Np, int N, int Dp, KernelSpec ks, const double* alpha, double* Ks, double* Cs,
                       long ldk, double* mu_part, double* ca_part) {
    ensure_dyn_lds((const void*)cross_gram_kernel<false>, GEMM_LDS_BYTES);
    ensure_dyn_lds((const void*)cross_gram_kernel<true>, GEMM_LDS_BYTES);
    const int nt = (Sp / GEMM_BM) * (Np / GEMM_BN);
    if (ks.kernel == SLS_KERNEL_ARD_MATERN52)
        hipLaunchKernelGGL(cross_gram_kernel<true>, dim3(nt), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, XsT, lds_, ns, Sp, XT, ld, nx,
                           Np, N, Dp, ks.a, alpha, Ks, Cs, ldk, mu_part, ca_part);
    else
        hipLaunchKernelGGL(cross_gram_kernel<false>, dim3(nt), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, XsT, lds_, ns, Sp, XT, ld, nx,
                           Np, N, Dp, ks.a, alpha, Ks, Cs, ldk, mu_part, ca_part);
}

}  // namespace slsk
