// probe: ring forms of the 128x128 fp64 tile (gemm_f64.hpp: gemm_tile_ring) against the shipped double-buffered gemm_tile.
// usage: gemm_probe_ring M N K [reps]     -- prints time, TFLOP/s and a checksum of C for every variant (the checksums must agree:
// same k order, same bits)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_f64.hpp"
using namespace slsk;

template <int VAR>
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
    if (VAR == 0) gemm_tile<false, false>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 1) gemm_tile_ring<4, 8, 4>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 2) gemm_tile_ring<4, 16, 2>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 3) gemm_tile_ring<4, 8, 3>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 4) gemm_tile_ring<4, 16, 4>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 6) gemm_tile_pipe<4>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 5) gemm_tile_ring<4, 16, 3>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)(m0 + acc_m(i)) + (long)(n0 + acc_n(j, r)) * ldc] = acc.v[i][j][r];
}

__global__ void checksum_kernel(const double* C, long n, unsigned long long* out) {
    unsigned long long h = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        h += (unsigned long long)__double_as_longlong(C[i]) * (unsigned long long)(2 * i + 1);
    atomicAdd(out, h);
}

template <int VAR>
static void run(int M, int N, int K, int reps, const double* dA, const double* dB, double* dC, int lds_bytes, const char* name) {
    hipFuncSetAttribute((const void*)probe_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int nt = (M / 128) * (N / 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(dC, 0, (size_t)M * N * 8);
    hipLaunchKernelGGL(probe_kernel<VAR>, dim3(nt), dim3(256), lds_bytes, 0, dA, (long)M, dB, (long)N, dC, (long)M, M, N, K);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(probe_kernel<VAR>, dim3(nt), dim3(256), lds_bytes, 0, dA, (long)M, dB, (long)N, dC, (long)M, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    unsigned long long* dh; hipMalloc(&dh, 8); hipMemset(dh, 0, 8);
    hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, 0, dC, (long)M * N, dh);
    unsigned long long h; hipMemcpy(&h, dh, 8, hipMemcpyDeviceToHost); hipFree(dh);
    printf("%-34s M=%d N=%d K=%d: %8.3f ms  %6.2f TFLOP/s  checksum %016llx  (%s)\n", name, M, N, K, ms, 2.0 * M * N * K / ms * 1e-9, h,
           hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    const int reps = argc > 4 ? atoi(argv[4]) : 2;
    double *dA, *dB, *dC;
    hipMalloc(&dA, (size_t)M * K * 8); hipMalloc(&dB, (size_t)N * K * 8); hipMalloc(&dC, (size_t)M * N * 8);
    std::vector<double> h((size_t)1 << 22);
    for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
    for (size_t off = 0; off < (size_t)M * K; off += h.size()) hipMemcpy(dA + off, h.data(), std::min(h.size(), (size_t)M * K - off) * 8, hipMemcpyHostToDevice);
    for (size_t off = 0; off < (size_t)N * K; off += h.size()) hipMemcpy(dB + off, h.data(), std::min(h.size(), (size_t)N * K - off) * 8, hipMemcpyHostToDevice);
    const int stage16 = 2 * 16 * GEMM_LDS_MC_LD * 8, stage8 = stage16 / 2;
    run<0>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "gemm_tile (BK16 x 2, syncthreads)");
    run<2>(M, N, K, reps, dA, dB, dC, 2 * stage16, "ring BK16 x 2 (bare barrier)");
    run<1>(M, N, K, reps, dA, dB, dC, 4 * stage8, "ring BK8 x 4");
    run<6>(M, N, K, reps, dA, dB, dC, 4 * stage8, "pipelined ring BK8 x 4");
    run<3>(M, N, K, reps, dA, dB, dC, 3 * stage8, "ring BK8 x 3");
    run<5>(M, N, K, reps, dA, dB, dC, 3 * stage16, "ring BK16 x 3 (1 WG/CU)");
    run<4>(M, N, K, reps, dA, dB, dC, 4 * stage16, "ring BK16 x 4 (1 WG/CU)");
    return 0;
}
