// prints which XCD each workgroup of a 1-D grid lands on (debug tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = x & 0xf;
    }
    // keep the block alive a little so that residency matters
    for (volatile int i = 0; i < 2000; ++i) {}
}
int main() {
    const int n = 8192;
    int* d; hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 73728, 0, d);
    std::vector<int> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    printf("first 48 blocks -> xcd: "); for (int i = 0; i < 48; ++i) printf("%d ", h[i]); printf("\n");
    int ok = 0; for (int i = 0; i < n; ++i) ok += (h[i] == i % 8);
    printf("blocks with xcd == b %% 8: %d / %d\n", ok, n);
    return 0;
}
