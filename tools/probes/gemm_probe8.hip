// probe: 128x128 tile with EIGHT waves (512 threads, wave tile 64x32, 64 accumulator VGPRs) -> 4 waves per SIMD at 2
// workgroups per CU, against the shipped 4-wave tile (gemm_probe).  usage: gemm_probe8 M N K
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_f64.hpp"
using namespace slsk;

struct Acc8 {
    d4_t v[4][2];
};

__global__ __launch_bounds__(512, 2) void probe8_kernel(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32;
    Acc8 acc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc.v[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
    const double* Ap = A + m0;
    const double* Bp = B + n0;
    // 128 x 16 slab = 1024 16-byte pieces -> 2 per thread per operand
    d2_t sa[2], sb[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * i;
            const int k = idx >> 6, m2 = idx & 63;
            sa[i] = *reinterpret_cast<const d2_t*>(Ap + (long)(2 * m2) + (long)(k0 + k) * lda);
            sb[i] = *reinterpret_cast<const d2_t*>(Bp + (long)(2 * m2) + (long)(k0 + k) * ldb);
        }
    };
    auto store = [&](double* l) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * i;
            const int k = idx >> 6, m2 = idx & 63;
            *reinterpret_cast<d2_t*>(l + k * GEMM_LDS_MC_LD + 2 * m2) = sa[i];
            *reinterpret_cast<d2_t*>(l + GEMM_LDS_TILE + k * GEMM_LDS_MC_LD + 2 * m2) = sb[i];
        }
    };
    load(0);
    store(lds);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < K; k0 += GEMM_BK) {
        const bool more = (k0 + GEMM_BK) < K;
        if (more) load(k0 + GEMM_BK);
        const double* la = lds + cur;
        const double* lb = lds + cur + GEMM_LDS_TILE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[4], bf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = frag_read<false>(la, wm + 16 * i, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = frag_read<false>(lb, wn + 16 * j, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc.v[i][j], 0, 0, 0);
        }
        const int nxt = cur ^ (2 * GEMM_LDS_TILE);
        if (more) store(lds + nxt);
        __syncthreads();
        cur = nxt;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(long)(m0 + wm + 16 * i + (lane & 15)) + (long)(n0 + wn + 16 * j + (lane >> 4) + 4 * r) * ldc] = acc.v[i][j][r];
}

int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    double *dA, *dB, *dC;
    hipMalloc(&dA, (size_t)M * K * 8); hipMalloc(&dB, (size_t)N * K * 8); hipMalloc(&dC, (size_t)M * N * 8);
    std::vector<double> h((size_t)1 << 22);
    for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
    for (size_t off = 0; off < (size_t)M * K; off += h.size()) hipMemcpy(dA + off, h.data(), std::min(h.size(), (size_t)M * K - off) * 8, hipMemcpyHostToDevice);
    for (size_t off = 0; off < (size_t)N * K; off += h.size()) hipMemcpy(dB + off, h.data(), std::min(h.size(), (size_t)N * K - off) * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
    const int nt = (M / 128) * (N / 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe8_kernel, dim3(nt), dim3(512), GEMM_LDS_BYTES, 0, dA, (long)M, dB, (long)N, dC, (long)M, M, N, K);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 2; ++r)
        hipLaunchKernelGGL(probe8_kernel, dim3(nt), dim3(512), GEMM_LDS_BYTES, 0, dA, (long)M, dB, (long)N, dC, (long)M, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 2;
    std::vector<double> hc(8); hipMemcpy(hc.data(), dC, 64, hipMemcpyDeviceToHost);
    printf("8-wave tile M=%d N=%d K=%d: %.3f ms  %.2f TFLOP/s  C[0]=%.12g (%s)\n", M, N, K, ms, 2.0 * M * N * K / ms * 1e-9, hc[0], hipGetErrorString(hipGetLastError()));
    return 0;
}
