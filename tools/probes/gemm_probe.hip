// tiny probe for PMC / variant runs of the library's 128x128 tile: gemm_probe M N K 0.  (Round 1 also measured staggered k starts
// here, modes 1-4; the k loop of gemm_tile_mc no longer takes a start offset.)  A 2-slab-deep register prefetch was tried and spills (256 VGPRs, 22 TFLOP/s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_f64.hpp"
using namespace slsk;
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe_kernel(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                       double* __restrict__ C, long ldc, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int ntm = M / 128, ntn = N / 128;
    int t = xcd_remap(blockIdx.x, ntm * ntn);
    const int gsz = 8 * ntn;
    const int g = t / gsz, w = t % gsz;
    const int gm = min(8, ntm - g * 8);
    const int tm = g * 8 + (w % gm), tn = w / gm;
    const int m0 = tm * 128, n0 = tn * 128;
    Acc acc;
    acc.zero();
    int ks = ((tm & 7) + (tn & 7)) * GEMM_BK;
    if (MODE == 2) ks = ((tm + tn) & 7) * GEMM_BK;
    if (MODE == 3) ks = ((tm + tn) & 7) * 2 * GEMM_BK;
    if (MODE == 4) ks = ((tm & 7) + (tn & 7)) * 2 * GEMM_BK;
    (void)ks;
    gemm_tile<false, false>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)(m0 + acc_m(i)) + (long)(n0 + acc_n(j, r)) * ldc] = acc.v[i][j][r];
}
template <int MODE>
static void run(int M, int N, int K, const double* dA, const double* dB, double* dC) {
    hipFuncSetAttribute((const void*)probe_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
    const int nt = (M / 128) * (N / 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe_kernel<MODE>, dim3(nt), dim3(256), GEMM_LDS_BYTES, 0, dA, (long)M, dB, (long)N, dC, (long)M, M, N, K);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 2; ++r)
        hipLaunchKernelGGL(probe_kernel<MODE>, dim3(nt), dim3(256), GEMM_LDS_BYTES, 0, dA, (long)M, dB, (long)N, dC, (long)M, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 2;
    std::vector<double> h(8); hipMemcpy(h.data(), dC, 64, hipMemcpyDeviceToHost);
    printf("mode %d M=%d N=%d K=%d: %.3f ms  %.2f TFLOP/s  C[0]=%.12g\n", MODE, M, N, K, ms, 2.0 * M * N * K / ms * 1e-9, h[0]);
}
int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), mode = atoi(argv[4]);
    double *dA, *dB, *dC;
    hipMalloc(&dA, (size_t)M * K * 8); hipMalloc(&dB, (size_t)N * K * 8); hipMalloc(&dC, (size_t)M * N * 8);
    std::vector<double> h((size_t)1 << 22);
    for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
    for (size_t off = 0; off < (size_t)M * K; off += h.size()) hipMemcpy(dA + off, h.data(), std::min(h.size(), (size_t)M * K - off) * 8, hipMemcpyHostToDevice);
    for (size_t off = 0; off < (size_t)N * K; off += h.size()) hipMemcpy(dB + off, h.data(), std::min(h.size(), (size_t)N * K - off) * 8, hipMemcpyHostToDevice);
    if (mode == 0) run<0>(M, N, K, dA, dB, dC);
    if (mode == 1) run<1>(M, N, K, dA, dB, dC);
    if (mode == 2) run<2>(M, N, K, dA, dB, dC);
    if (mode == 3) run<3>(M, N, K, dA, dB, dC);
    if (mode == 4) run<4>(M, N, K, dA, dB, dC);
    return 0;
}
