// Same-node vendor yardstick (tools only -- nothing in the product path links rocBLAS / rocSOLVER): rocBLAS dgemm at the
// dominant kernel's shape (K^-1 K*: 8192 x 65536 x 8192) and rocSOLVER dpotrf / dpotri at N = 2048 / 4096 / 8192 / 16384, timed with
// HIP events on the same GPU the library's own kernels are measured on.  Writes one JSON object to stdout
// (-> profiles/r04_vendor_yardstick.json).  Every "fraction of peak" in DESIGN.md is against the 78.6 TFLOP/s model of the fp64
// MFMA pipe; these are the external numbers beside it.
// Build: hipcc --offload-arch=gfx950 -O2 vendor_yardstick.cpp -o bin/vendor_yardstick -lrocblas -lrocsolver
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { fprintf(stderr, "%s failed: %d (line %d)\n", #x, (int)e_, __LINE__); exit(1); } } while (0)

__global__ void fill_spd(double* A, int n, unsigned seed) {
    // symmetric, diagonally dominant: A_ij = small pseudo-random value, A_ii = n
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)n * n) return;
    const int i = idx % n, j = idx / n;
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    unsigned h = (unsigned)(lo * 2654435761u) ^ (unsigned)(hi * 40503u) ^ seed;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    A[idx] = i == j ? (double)n : ((h & 0xffff) / 65536.0 - 0.5);
}
__global__ void fill_rand(double* A, long n, unsigned seed) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= n) return;
    unsigned h = (unsigned)(idx * 2654435761u) ^ seed;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    A[idx] = (h & 0xffff) / 65536.0 - 0.5;
}

template <class F>
static double time_ms(hipStream_t s, int reps, F&& f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    rocblas_handle h;
    CK(rocblas_create_handle(&h));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    CK(rocblas_set_stream(h, s));
    const double peak = 78.6;
    printf("{\n \"peak_model_TFLOPs\": %.1f,\n", peak);
    {   // dgemm: W = K^-1 K*  (M = N_train, N = candidates, K = N_train), column-major NN
        const int M = 8192, K = 8192;
        printf(" \"rocblas_dgemm\": [");
        bool first = true;
        for (int N : {16384, 65536}) {
            double *A, *B, *C;
            CK(hipMalloc(&A, (size_t)M * K * 8)); CK(hipMalloc(&B, (size_t)K * N * 8)); CK(hipMalloc(&C, (size_t)M * N * 8));
            fill_rand<<<(unsigned)(((size_t)M * K + 255) / 256), 256, 0, s>>>(A, (long)M * K, 1u);
            fill_rand<<<(unsigned)(((size_t)K * N + 255) / 256), 256, 0, s>>>(B, (long)K * N, 2u);
            const double one = 1.0, zero = 0.0;
            const double ms = time_ms(s, 3, [&] {
                CK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, M, N, K, &one, A, M, B, K, &zero, C, M));
            });
            const double tf = 2.0 * M * (double)N * K / (ms * 1e-3) / 1e12;
            printf("%s\n  {\"M\": %d, \"N\": %d, \"K\": %d, \"ms\": %.3f, \"TFLOPs\": %.2f, \"frac_of_peak_model\": %.3f}", first ? "" : ",", M, N, K, ms, tf, tf / peak);
            first = false;
            CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
        }
        printf("\n ],\n");
    }
    {
        printf(" \"rocsolver_dpotrf_dpotri\": [");
        bool first = true;
        for (int n : {2048, 4096, 8192, 16384}) {
            double *A, *A0;
            int* info;
            CK(hipMalloc(&A, (size_t)n * n * 8)); CK(hipMalloc(&A0, (size_t)n * n * 8)); CK(hipMalloc(&info, sizeof(int)));
            fill_spd<<<(unsigned)(((size_t)n * n + 255) / 256), 256, 0, s>>>(A0, n, 7u);
            CK(hipStreamSynchronize(s));
            // the copy A0 -> A is inside the timed loop for both (it is subtracted: measured alone)
            const int reps = n >= 8192 ? 3 : 10;
            const double ms_copy = time_ms(s, reps, [&] { CK(hipMemcpyAsync(A, A0, (size_t)n * n * 8, hipMemcpyDeviceToDevice, s)); });
            const double ms_f = time_ms(s, reps, [&] {
                CK(hipMemcpyAsync(A, A0, (size_t)n * n * 8, hipMemcpyDeviceToDevice, s));
                CK(rocsolver_dpotrf(h, rocblas_fill_lower, n, A, n, info));
            }) - ms_copy;
            const double ms_fi = time_ms(s, reps, [&] {
                CK(hipMemcpyAsync(A, A0, (size_t)n * n * 8, hipMemcpyDeviceToDevice, s));
                CK(rocsolver_dpotrf(h, rocblas_fill_lower, n, A, n, info));
                CK(rocsolver_dpotri(h, rocblas_fill_lower, n, A, n, info));
            }) - ms_copy;
            int hinfo = -1;
            CK(hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost));
            const double f3 = (double)n * n * n / 3.0;
            printf("%s\n  {\"N\": %d, \"dpotrf_ms\": %.3f, \"dpotrf_TFLOPs\": %.2f, \"dpotrf_frac_of_peak_model\": %.3f, \"dpotri_ms\": %.3f, "
                   "\"potrf_plus_potri_ms\": %.3f, \"potrf_plus_potri_TFLOPs_on_N3\": %.2f, \"info\": %d}",
                   first ? "" : ",", n, ms_f, f3 / (ms_f * 1e-3) / 1e12, f3 / (ms_f * 1e-3) / 1e12 / peak, ms_fi - ms_f, ms_fi,
                   3.0 * f3 / (ms_fi * 1e-3) / 1e12, hinfo);
            first = false;
            CK(hipFree(A)); CK(hipFree(A0)); CK(hipFree(info));
        }
        printf("\n ]\n}\n");
    }
    rocblas_destroy_handle(h);
    return 0;
}
