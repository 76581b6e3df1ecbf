// Standalone micro-benchmarks (not part of the shipped library):
//   1. v_mfma_f64_16x16x4_f64 issue-rate ceiling (the fp64 matrix peak the
//      rooflines in DESIGN.md are priced against),
//   2. HBM copy bandwidth,
//   3. the 128x128 fp64 GEMM building block (gemm_f64.hpp): correctness on an
//      asymmetric small case + TFLOP/s on the shapes the hot path uses.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_f64.hpp"

using namespace slsk;
#ifndef STAGGER
#define STAGGER 1
#endif

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__global__ __launch_bounds__(256) void mfma_peak(double* out, int iters) {
    d4_t c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = d4_t{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_peak(double* out, int iters) {
    double c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = i;
    double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = fma(c[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void copy_kernel(const d2_t* __restrict__ in, d2_t* __restrict__ out, long n2) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n2; i += stride) out[i] = in[i];
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const double* __restrict__ A, long lda,
                                                      const double* __restrict__ B, long ldb, double* __restrict__ C,
                                                      long ldc, int M, int N, int K, int group_m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int ntm = M / GEMM_BM, ntn = N / GEMM_BN;
    int t = xcd_remap(blockIdx.x, ntm * ntn);
    // grouped order: group_m m-tiles x all n-tiles, n-major inside the group
    const int gmm = group_m < 0 ? -group_m : group_m;
    const int gsz = gmm * ntn;
    const int g = t / gsz, w = t % gsz;
    const int gm = min(gmm, ntm - g * gmm);
    const int tm = g * gmm + (w % gm), tn = w / gm;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    const double* Ap = A_KC ? A + (long)m0 * lda : A + m0;
    const double* Bp = B_KC ? B + (long)n0 * ldb : B + n0;
    Acc acc;
    acc.zero();
    // (round 1 measured a staggered k start here for group_m < 0; gemm_tile no longer takes a start offset)
    gemm_tile<A_KC, B_KC>(acc, Ap, lda, Bp, ldb, 0, K, lds);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)(m0 + acc_m(i)) + (long)(n0 + acc_n(j, r)) * ldc] = acc.v[i][j][r];
}

static double frand() { return (double)rand() / RAND_MAX - 0.5; }

template <bool A_KC, bool B_KC>
static void check_small() {
    const int M = 256, N = 384, K = 64;
    std::vector<double> A(M * K), B(N * K), C(M * N), R(M * N);
    for (auto& v : A) v = frand();
    for (auto& v : B) v = frand();
    auto a = [&](int m, int k) { return A_KC ? A[k + m * K] : A[m + k * M]; };
    auto b = [&](int n, int k) { return B_KC ? B[k + n * K] : B[n + k * N]; };
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += a(m, k) * b(n, k);
            R[m + n * M] = s;
        }
    double *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 8));
    CK(hipMalloc(&dB, B.size() * 8));
    CK(hipMalloc(&dC, C.size() * 8));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
    const int nt = (M / 128) * (N / 128);
    hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC>), dim3(nt), dim3(256), GEMM_LDS_BYTES, 0, dA, (long)(A_KC ? K : M), dB,
                       (long)(B_KC ? K : N), dC, (long)M, M, N, K, 8);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost));
    double err = 0;
    for (size_t i = 0; i < C.size(); ++i) err = fmax(err, fabs(C[i] - R[i]));
    printf("gemm check A_KC=%d B_KC=%d  max abs err %.3e  %s\n", (int)A_KC, (int)B_KC, err, err < 1e-12 ? "OK" : "FAIL");
    CK(hipFree(dA));
    CK(hipFree(dB));
    CK(hipFree(dC));
}

template <bool A_KC, bool B_KC>
static void bench_gemm(int M, int N, int K, int group_m, int reps, int pad = 0) {
    double *dA, *dB, *dC;
    const long lda = (A_KC ? K : M) + pad, ldb = (B_KC ? K : N) + pad, ldc = M + pad;
    CK(hipMalloc(&dA, (size_t)lda * (A_KC ? M : K) * 8));
    CK(hipMalloc(&dB, (size_t)ldb * (B_KC ? N : K) * 8));
    CK(hipMalloc(&dC, (size_t)ldc * N * 8));
    // random fill (uniform [-0.5,0.5)) -- never bench on zeros (DVFS)
    {
        std::vector<double> h((size_t)1 << 22);
        for (auto& v : h) v = frand();
        const size_t na = (size_t)lda * (A_KC ? M : K), nb = (size_t)ldb * (B_KC ? N : K);
        for (size_t off = 0; off < na; off += h.size())
            CK(hipMemcpy(dA + off, h.data(), std::min(h.size(), na - off) * 8, hipMemcpyHostToDevice));
        for (size_t off = 0; off < nb; off += h.size())
            CK(hipMemcpy(dB + off, h.data(), std::min(h.size(), nb - off) * 8, hipMemcpyHostToDevice));
    }
    const int nt = (M / 128) * (N / 128);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&]() {
        hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC>), dim3(nt), dim3(256), GEMM_LDS_BYTES, 0, dA, lda, dB, ldb, dC, ldc, M, N, K,
                           group_m);
    };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("gemm A_KC=%d B_KC=%d M=%d N=%d K=%d group_m=%d pad=%d: %.3f ms  %.2f TFLOP/s\n", (int)A_KC, (int)B_KC, M, N, K,
           group_m, pad, ms, 2.0 * M * N * K / ms * 1e-9);
    CK(hipFree(dA));
    CK(hipFree(dB));
    CK(hipFree(dC));
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  clock %d kHz  L2 %d  mem %.1f GB\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate, prop.l2CacheSize, prop.totalGlobalMem / 1e9);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms;
    {
        double* out;
        const int blocks = 256 * 8, iters = 20000;
        CK(hipMalloc(&out, blocks * 256 * 8));
        hipLaunchKernelGGL(mfma_peak, dim3(blocks), dim3(256), 0, 0, out, 100);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_peak, dim3(blocks), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 4;
        printf("mfma_f64_16x16x4 peak: %.3f ms  %.2f TFLOP/s\n", ms, flops / ms * 1e-9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(fma_peak, dim3(blocks), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double vflops = (double)blocks * 256 * iters * 16 * 2.0;
        printf("v_fma_f64 peak: %.3f ms  %.2f TFLOP/s\n", ms, vflops / ms * 1e-9);
        CK(hipFree(out));
    }
    {
        const long n = 1L << 30;  // 1 GiB each way
        d2_t *a, *b;
        CK(hipMalloc(&a, n));
        CK(hipMalloc(&b, n));
        CK(hipMemset(a, 1, n));
        hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, a, b, n / 16);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, a, b, n / 16);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy 1 GiB: %.3f ms  %.2f TB/s (read+write)\n", ms / 10, 2.0 * n / (ms / 10) * 1e-9);
        CK(hipFree(a));
        CK(hipFree(b));
    }
    CK(hipFuncSetAttribute((const void*)gemm_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)gemm_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)gemm_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)gemm_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));
    check_small<false, false>();
    check_small<true, false>();
    check_small<false, true>();
    check_small<true, true>();
    bench_gemm<false, false>(8192, 8192, 8192, 8, 3);
    bench_gemm<true, true>(8192, 8192, 8192, 8, 3);
    bench_gemm<false, false>(16384, 8192, 8192, 8, 2);   // acquisition main GEMM, 16k candidates
    bench_gemm<false, false>(16384, 8192, 8192, 8, 2, 16);
    bench_gemm<false, false>(16384, 8192, 8192, 8, 2, 32);
    bench_gemm<false, false>(16384, 8192, 8192, 8, 2, 144);
    bench_gemm<false, false>(16384, 8192, 8192, 16, 2, 16);
    bench_gemm<false, false>(16384, 8192, 8192, 4, 2);
    bench_gemm<false, false>(16384, 8192, 8192, 16, 2);
    bench_gemm<false, false>(2048, 2048, 2048, 8, 5);
    bench_gemm<false, false>(8192, 128, 8192, 8, 5);     // panel-shaped
    return 0;
}
