// cycle stamps of chol_diag_kernel phases (debug tool, not shipped)
#define SLS_DIAG_TIMING 1
#include "kernels_chol.hip"
#include <cstdio>
#include <vector>
int main() {
    const int Np = 1024;
    std::vector<double> A((size_t)Np * Np, 0.0);
    for (int i = 0; i < Np; ++i) for (int j = 0; j < Np; ++j) A[i + (size_t)j * Np] = (i == j ? 2.0 : 0.0) + 0.5 / (1.0 + abs(i - j));
    double *dA, *dL; int* info;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dL, A.size() * 8); hipMalloc(&info, 64);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemset(info, 0, 64); hipMemset(dL, 0, A.size() * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        slsk::launch_potrf(0, dA, Np, dL, info);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long st[8]; hipMemcpy(st, info, 64, hipMemcpyDeviceToHost);
        printf("potrf N=%d: %.3f ms; last diag kernel cycles: load %lld  factor %lld  storeL %lld  inverse %lld  storeT %lld  total %lld  diag16(kb=0) %lld\n", Np, ms,
               st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[6] - st[1], st[7]);
    }
    return 0;
}
