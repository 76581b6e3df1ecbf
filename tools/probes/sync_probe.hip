// Round trip of a small kernel: launch + hipStreamSynchronize against launch + polling a word the kernel's last workgroup writes into
// mapped (page-locked) host memory.  Prices the ten batched evaluations of a DIRECT run (28 us of wall for 11 us of kernel each).
// Build: make -C tools/probes bin/sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void work(double* out, int spin, unsigned* counter, volatile long* flag, long seq) {
    double v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = fma(v, 0.999, 1e-3);
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;   // results into mapped memory, like the evaluation kernels
    if (flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned prev = atomicAdd(counter, 1u);
            if (prev == gridDim.x - 1) {
                *counter = 0;
                __threadfence_system();
                *flag = seq;
            }
        }
    }
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double* host; CK(hipHostMalloc(&host, 40 * 256 * 8 + 64, hipHostMallocMapped));
    double* dev;  CK(hipHostGetDevicePointer((void**)&dev, host, 0));
    volatile long* flag = (volatile long*)(host + 40 * 256);
    long* dflag = (long*)(dev + 40 * 256);
    unsigned* counter; CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
    *flag = 0;
    for (int spin : {100, 2000}) {
        for (int mode = 0; mode < 2; ++mode) {
            double best = 1e30;
            long seq = 1000 * (spin + mode);
            for (int rep = 0; rep < 5; ++rep) {
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < 200; ++i) {
                    ++seq;
                    hipLaunchKernelGGL(work, dim3(40), dim3(256), 0, s, dev, spin, counter, mode ? dflag : nullptr, seq);
                    if (mode) { while (*flag != seq) {} }
                    else CK(hipStreamSynchronize(s));
                }
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
                if (us < best) best = us;
            }
            CK(hipStreamSynchronize(s));
            printf("kernel of %4d dependent fma per thread, 40 x 256 threads: %-28s %6.2f us per round trip\n", spin, mode ? "poll a mapped word" : "hipStreamSynchronize", best);
        }
    }
    return 0;
}
