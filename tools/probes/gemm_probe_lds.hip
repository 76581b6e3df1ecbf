// probe: the shipped 4-wave 128x128 tile, operand slabs brought in with LDS-direct loads (global_load_lds_dwordx4, new
// on gfx950) instead of global -> VGPR -> ds_write.  usage: gemm_probe_lds M N K
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_f64.hpp"
using namespace slsk;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// one k-row (128 doubles = 64 lanes x 16 B) of an M-contiguous operand straight into its LDS row
__device__ __forceinline__ void row_to_lds(const double* g, double* l) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void probe_lds_kernel(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
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
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    Acc acc;
    acc.zero();
    const double* Ap = A + m0 + 2 * lane;
    const double* Bp = B + n0 + 2 * lane;
    auto issue = [&](int k0, double* buf) {
        // wave w brings rows 4w .. 4w+3 of both operands
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * wave + r;
            row_to_lds(Ap + (long)(k0 + row) * lda, buf + row * GEMM_LDS_MC_LD);
            row_to_lds(Bp + (long)(k0 + row) * ldb, buf + GEMM_LDS_TILE + row * GEMM_LDS_MC_LD);
        }
    };
    issue(0, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < K; k0 += GEMM_BK) {
        const bool more = (k0 + GEMM_BK) < K;
        const int nxt = cur ^ (2 * GEMM_LDS_TILE);
#ifndef SPREAD
        if (more) issue(k0 + GEMM_BK, lds + nxt);
#endif
        const double* la = lds + cur;
        const double* lb = lds + cur + GEMM_LDS_TILE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#ifdef SPREAD
            if (more && kk == SPREAD) issue(k0 + GEMM_BK, lds + nxt);
#endif
            double af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = frag_read<false>(la, wm + 16 * i, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = frag_read<false>(lb, wn + 16 * j, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc.v[i][j], 0, 0, 0);
        }
#ifndef NOWAIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        cur = nxt;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)(m0 + acc_m(i)) + (long)(n0 + acc_n(j, r)) * ldc] = acc.v[i][j][r];
}

int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    double *dA, *dB, *dC;
    hipMalloc(&dA, (size_t)M * K * 8); hipMalloc(&dB, (size_t)N * K * 8); hipMalloc(&dC, (size_t)M * N * 8);
    std::vector<double> h((size_t)1 << 22);
    for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
    for (size_t off = 0; off < (size_t)M * K; off += h.size()) hipMemcpy(dA + off, h.data(), std::min(h.size(), (size_t)M * K - off) * 8, hipMemcpyHostToDevice);
    for (size_t off = 0; off < (size_t)N * K; off += h.size()) hipMemcpy(dB + off, h.data(), std::min(h.size(), (size_t)N * K - off) * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
    const int nt = (M / 128) * (N / 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe_lds_kernel, dim3(nt), dim3(256), GEMM_LDS_BYTES, 0, dA, (long)M, dB, (long)N, dC, (long)M, M, N, K);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 2; ++r)
        hipLaunchKernelGGL(probe_lds_kernel, dim3(nt), dim3(256), GEMM_LDS_BYTES, 0, dA, (long)M, dB, (long)N, dC, (long)M, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 2;
    std::vector<double> hc(8); hipMemcpy(hc.data(), dC, 64, hipMemcpyDeviceToHost);
    printf("LDS-direct tile M=%d N=%d K=%d: %.3f ms  %.2f TFLOP/s  C[0]=%.12g (%s)\n", M, N, K, ms, 2.0 * M * N * K / ms * 1e-9, hc[0], hipGetErrorString(hipGetLastError()));
    return 0;
}
