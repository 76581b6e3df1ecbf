// cycle stamps of chol_diag_kernel phases + residual check + rsqrt seed accuracy (debug tool, not shipped)
#define SLS_DIAG_TIMING 1
#include "../../sequential-line-search_amd/csrc/kernels_chol.hip"
#include "../../sequential-line-search_amd/csrc/kernels_tri.hip"
#include "../../sequential-line-search_amd/csrc/kernels_vec.hip"
#include <cmath>
#include <cstring>
#include <cstdio>
#include <vector>

__global__ void rsq_probe(const double* x, double* y0, double* y1, double* y2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double y = __builtin_amdgcn_rsq(d);
    y0[i] = y;
    double e = fma(-d * y, y, 1.0);
    y = fma(y * e, fma(0.375, e, 0.5), y);
    y1[i] = y;
    e = fma(-d * y, y, 1.0);
    y = fma(y * e, fma(0.375, e, 0.5), y);
    y2[i] = y;
}

int main() {
    {   // v_rsq_f64 seed accuracy and Newton steps
        const int n = 1 << 20;
        std::vector<double> x(n), y0(n), y1(n), y2(n);
        for (int i = 0; i < n; ++i) x[i] = std::ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 80) - 40);
        double *dx, *d0, *d1, *d2;
        hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
        hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(rsq_probe, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
        hipMemcpy(y0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(y1.data(), d1, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(y2.data(), d2, n * 8, hipMemcpyDeviceToHost);
        double e0 = 0, e1 = 0, e2 = 0;
        for (int i = 0; i < n; ++i) {
            const long double r = 1.0L / sqrtl((long double)x[i]);
            e0 = fmax(e0, (double)fabsl((y0[i] - r) / r)); e1 = fmax(e1, (double)fabsl((y1[i] - r) / r)); e2 = fmax(e2, (double)fabsl((y2[i] - r) / r));
        }
        printf("rsq seed max rel err %.3g ; after 1 step %.3g ; after 2 steps %.3g (eps = 1.1e-16)\n", e0, e1, e2);
    }
    for (int Np : {1024, 4096}) {
        std::vector<double> A((size_t)Np * Np, 0.0), L((size_t)Np * Np), T((size_t)Np * Np);
        for (int i = 0; i < Np; ++i) for (int j = 0; j < Np; ++j) A[i + (size_t)j * Np] = (i == j ? 2.0 : 0.0) + 0.5 / (1.0 + abs(i - j));
        double *dA, *dL; int* info;
        hipMalloc(&dA, A.size() * 8); hipMalloc(&dL, A.size() * 8); hipMalloc(&info, 4096);
        hipMemset(info, 0, 4096); hipMemset(dL, 0, A.size() * 8);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            slsk::launch_potrf(0, dA, Np, dL, info);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long st[8]; hipMemcpy(st, info, 64, hipMemcpyDeviceToHost);
            printf("potrf N=%d: %.3f ms; last diag kernel cycles: load %lld  steps %lld  storeL %lld  storeT %lld  total %lld  phaseA(kb=0) %lld\n", Np, ms,
                   st[2] - st[1], st[3] - st[2], st[4] - st[3], st[6] - st[4], st[6] - st[1], st[7]);
            if (rep == 2) {
                long long f[128]; hipMemcpy(f, info, 1024, hipMemcpyDeviceToHost);
                printf("  storeT split: barrier1 %lld  transpose %lld  barrier2 %lld  store loop %lld\n", f[8] - f[4], f[9] - f[8], f[10] - f[9], f[6] - f[10]);
                for (int kb = 0; kb < 8; ++kb) {
                    const long long* g = f + 16 + 8 * kb;
                    printf("  kb=%d: diag16 %lld  wait+sync1 %lld  B+sync %lld  C(wave0) %lld | wave1: S done at +%lld, C done at +%lld (rel. A start)\n", kb,
                           g[1] - g[0], g[2] - g[1], g[3] - g[2], g[4] - g[3], g[5] - g[0], g[6] - g[0]);
                }
            }
            hipMemset(info, 0, 64);
        }
        hipMemcpy(L.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(T.data(), dL, A.size() * 8, hipMemcpyDeviceToHost);
        {   // order-independent fingerprints of the factor and of the diagonal-block inverses (A/B of diag16 variants: same bits?)
            unsigned long long hL = 0, hT = 0;
            for (size_t q = 0; q < L.size(); ++q) { unsigned long long u; memcpy(&u, &L[q], 8); hL += u * (2 * q + 1); memcpy(&u, &T[q], 8); hT += u * (2 * q + 1); }
            printf("fingerprints N=%d: L %016llx  T %016llx\n", Np, hL, hT);
        }
        if (Np == 1024) {
            double r1 = 0, r2 = 0;
            for (int i = 0; i < Np; ++i) for (int j = 0; j <= i; ++j) {
                double s = 0; for (int k = 0; k <= j; ++k) s += L[i + (size_t)k * Np] * L[j + (size_t)k * Np];
                r1 = fmax(r1, fabs(s - A[i + (size_t)j * Np]));
            }
            for (int b = 0; b < Np / 128; ++b)   // diagonal blocks: T_bb L_bb = I
                for (int i = 0; i < 128; ++i) for (int j = 0; j < 128; ++j) {
                    double s = 0; for (int k = 0; k < 128; ++k) s += T[(b * 128 + i) + (size_t)(b * 128 + k) * Np] * L[(b * 128 + k) + (size_t)(b * 128 + j) * Np];
                    r2 = fmax(r2, fabs(s - (i == j ? 1.0 : 0.0)));
                }
            printf("residuals: max|L L^T - A| = %.3g ; max|T_bb L_bb - I| = %.3g\n", r1, r2);
        }
        hipFree(dA); hipFree(dL); hipFree(info);
    }
    return 0;
}
