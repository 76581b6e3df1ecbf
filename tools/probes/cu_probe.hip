// which CU does slot s = b >> 3 of XCD b & 7 land on, for a 512-workgroup resident grid (2 per CU)?  (debug tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(int* out) {
    if (threadIdx.x == 0) {
        unsigned x, h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        out[2 * blockIdx.x] = x & 0xf;
        out[2 * blockIdx.x + 1] = h;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) __builtin_amdgcn_s_sleep(32);   // stay resident 200 us
}
int main() {
    const int n = 512;
    int* d; hipMalloc(&d, n * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 73728, 0, d);
    std::vector<int> h(2 * n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    for (int x = 0; x < 2; ++x) {
        printf("XCD %d: slot -> (se,sh,cu): ", x);
        for (int s = 0; s < 64; ++s) {
            const int b = s * 8 + x; const unsigned v = h[2 * b + 1];
            printf("%d:%u.%u.%u ", s, (v >> 13) & 7, (v >> 12) & 1, (v >> 8) & 15);
        }
        printf("\n");
    }
    // co-residency: for every slot find the other slot on the same CU
    int same32 = 0, same1 = 0, total = 0;
    for (int x = 0; x < 8; ++x)
        for (int s = 0; s < 64; ++s)
            for (int t = s + 1; t < 64; ++t) {
                const unsigned a = h[2 * (s * 8 + x) + 1] & 0xff00, b = h[2 * (t * 8 + x) + 1] & 0xff00;
                if (a == b) { ++total; same32 += (t == s + 32); same1 += (t == s + 1 && (s & 1) == 0); }
            }
    printf("co-resident pairs %d: partner = slot+32 in %d, partner = slot^1 in %d\n", total, same32, same1);
    return 0;
}
