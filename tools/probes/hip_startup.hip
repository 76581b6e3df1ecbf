// The floor of any HIP process on the box: runtime initialisation + one trivial kernel (the wall time of config C1,
// bayesian_optimization_1d 1 20, is compared with it in DESIGN.md 8).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* p) { *p = 1; }
int main() {
    int* d = nullptr;
    if (hipMalloc(&d, 4) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d);
    int h = 0;
    if (hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    std::printf("ok %d\n", h);
    return 0;
}
