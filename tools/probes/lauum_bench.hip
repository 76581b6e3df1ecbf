// lauum tile-kernel experiments (debug tool, not shipped): the K^-1 = U U^T launch of launch_lauum with a chosen leading
// dimension of U (is the distance of lauum from the list-schedule bound a power-of-two-stride effect?), as a full square
// product for reference.  usage: lauum_bench N [pad ...]
#include "../../sequential-line-search_amd/csrc/kernels_chol.hip"
#include "../../sequential-line-search_amd/csrc/kernels_tri.hip"
#include "../../sequential-line-search_amd/csrc/kernels_vec.hip"
#include <cstdio>
#include <vector>

__global__ void fill_u(double* U, long ld, int Np) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= (long)Np * Np) return;
    const int i = idx % Np, j = idx / Np;
    U[i + j * ld] = j >= i ? 1e-3 * ((i * 7 + j * 13) % 97) : 0.0;
}

int main(int argc, char** argv) {
    using namespace slsk;
    const int Np = argc > 1 ? atoi(argv[1]) : 8192;
    std::vector<int> pads;
    for (int i = 2; i < argc; ++i) pads.push_back(atoi(argv[i]));
    if (pads.empty()) pads = {0, 16, 144};
    const int nb = Np / NB;
    hipStream_t s; hipStreamCreate(&s);
    double* K; hipMalloc(&K, (size_t)Np * Np * 8);
    // context experiment: the same lauum launch right behind a persistent potrf (8 ms in which 255 of the 256 workgroups
    // mostly sleep) -- does the chip come out of that in a slower state?
    if (getenv("AFTER_POTRF")) {
        double *A0, *A, *Li, *U; int* info;
        const size_t bytes = (size_t)Np * Np * 8;
        hipMalloc(&A0, bytes); hipMalloc(&A, bytes); hipMalloc(&Li, bytes); hipMalloc(&U, bytes); hipMalloc(&info, 8192);
        hipLaunchKernelGGL(fill_u, dim3((unsigned)(((long)Np * Np + 255) / 256)), dim3(256), 0, s, U, (long)Np, Np);
        // SPD matrix: U^T U + I would need a product; use a diagonally dominant one instead
        std::vector<double> h((size_t)Np * Np, 0.0);
        for (int i = 0; i < Np; ++i) { h[(size_t)i * Np + i] = 4.0; if (i + 1 < Np) { h[(size_t)i * Np + i + 1] = 1.0; h[(size_t)(i + 1) * Np + i] = 1.0; } }
        hipMemcpy(A0, h.data(), bytes, hipMemcpyHostToDevice);
        GemmDesc g = mkdesc(U, Np, U, Np, K, Np, nb, nb, Np, 1.0, 0.0);
        g.tri = 1; g.kmode = 3; g.order = 1;
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemcpyAsync(A, A0, bytes, hipMemcpyDeviceToDevice, s);
                hipMemsetAsync(info, 0, 8192, s);
                hipMemsetAsync(Li, 0, bytes, s);
                hipStreamSynchronize(s);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                if (mode == 1) launch_potrf(s, A, Np, Li, info, 0, info + 64);
                hipEventRecord(e0, s);
                launch_tri_gemm<false, false>(s, g, 1, false);
                hipEventRecord(e1, s); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("N=%d lauum %-28s %8.3f ms\n", Np, mode ? "right behind a potrf" : "behind an idle stream", best);
        }
    }
    for (int pad : pads) {
        const long ld = Np + pad;
        double* U; hipMalloc(&U, (size_t)ld * Np * 8);
        hipLaunchKernelGGL(fill_u, dim3((unsigned)(((long)Np * Np + 255) / 256)), dim3(256), 0, s, U, ld, Np);
        for (int variant = 0; variant < 3; ++variant) {
            GemmDesc g = mkdesc(U, ld, U, ld, K, Np, nb, nb, Np, 1.0, 0.0);
            const char* name = "lauum (tri, k >= 128 tm, rows first)";
            if (variant <= 1) { g.tri = 1; g.kmode = 3; g.order = 1; }
            if (variant == 1) name = "lauum, one workgroup per CU";
            if (variant == 2) name = "full square product (k = 0 .. N)";
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, s);
                launch_tri_gemm<false, false>(s, g, 1, variant == 1);   // (4th argument: one workgroup per CU)
                hipEventRecord(e1, s); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double flops = variant == 2 ? 2.0 * Np * (double)Np * Np : (double)Np * Np * Np / 3.0;
            printf("N=%d ld=%ld %-40s %8.3f ms  %6.2f TFLOP/s (%s)\n", Np, ld, name, best, flops / (best * 1e-3) / 1e12,
                   hipGetErrorString(hipGetLastError()));
        }
        hipFree(U);
    }
    return 0;
}
