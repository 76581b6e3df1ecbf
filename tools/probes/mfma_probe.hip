// v_mfma_f64_16x16x4_f64 sustained issue rate for register-only loops: NACC independent accumulators per wave,
// WAVES waves per SIMD (workgroups of 256 threads = one wave per SIMD each), shared or distinct A/B operands.
// usage: mfma_probe   -> table of TFLOP/s (peak by datasheet: 78.6)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int NACC, bool DISTINCT>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    d4_t c[NACC];
    double a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { c[i] = d4_t{0, 0, 0, 0}; a[i] = threadIdx.x * 1e-3 + i; b[i] = 1.0 + threadIdx.x * 1e-4 - i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(DISTINCT ? a[i] : a[0], DISTINCT ? b[i] : b[0], c[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool DISTINCT>
static void run(double* out, int waves_per_simd, int lds_bytes) {
    const int blocks = 256 * waves_per_simd, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, DISTINCT>), dim3(blocks), dim3(256), lds_bytes, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, DISTINCT>), dim3(blocks), dim3(256), lds_bytes, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * NACC * 2.0 * 16 * 16 * 4;
    printf("acc/wave %2d  %s A/B  %d wave(s)/SIMD: %7.3f ms  %6.2f TFLOP/s\n", NACC, DISTINCT ? "distinct" : "shared  ", waves_per_simd, ms, flops / ms * 1e-9);
}

int main() {
    double* out; hipMalloc(&out, 256 * 8 * 256 * 8);
    // dynamic LDS of 40 KB / 80 KB per workgroup limits residency to 4 / 2 / 1 workgroups per CU
    hipFuncSetAttribute((const void*)k<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    run<8, false>(out, 8, 0);
    run<16, false>(out, 8, 0);
    run<16, true>(out, 8, 0);
    run<16, true>(out, 4, 40 * 1024);
    run<16, true>(out, 2, 80 * 1024);
    run<16, true>(out, 1, 160 * 1024);
    run<4, true>(out, 4, 40 * 1024);
    return 0;
}
