// Latencies that price the one-workgroup kernels (kernels_small.hip, kernels_wave.hip, chol_diag.hpp): ONE workgroup of four waves
// on an otherwise idle chip, one wave per SIMD -- nothing hides a dependent chain there, so the cost of a phase is the sum of the
// latencies on its longest path, not its instruction count.  Cycles of the shader clock (s_memtime) per operation, thread 0 of wave 0.
// Build: make -C tools/probes bin/lat_probe     Run: ./bin/lat_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../sequential-line-search_amd/csrc/wave_reduce.hpp"

using namespace slsk;
typedef double d4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NSLOT = 32;
__device__ __forceinline__ long long tick() {
    long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__global__ __launch_bounds__(256) void lat_kernel(long long* out, double* sink, int reps) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 128 * 144; i += 256) lds[i] = 1.0 / (1.0 + i);
    __syncthreads();
    long long t0, t1;
    double acc_sink = 0.0;
    int slot = 0;
#define REC() do { if (tid == 0) out[slot] = t1 - t0; ++slot; } while (0)

    // 0: dependent chain of matrix products (one accumulator)
    {
        d4_t c = {0, 0, 0, 0};
        double a = 1e-3 * lane, b = 1.0 + 1e-4 * lane;
        t0 = tick();
        for (int r = 0; r < reps; ++r) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        acc_sink += c[0] + c[1] + c[2] + c[3];
        asm volatile("" ::"v"(acc_sink));
        t1 = tick();
        REC();
    }
    // 1: two accumulators alternating
    {
        d4_t c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
        double a = 1e-3 * lane, b = 1.0 + 1e-4 * lane;
        t0 = tick();
        for (int r = 0; r < reps; r += 2) {
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, d, 0, 0, 0);
        }
        acc_sink += c[0] + d[1];
        asm volatile("" ::"v"(acc_sink));
        t1 = tick();
        REC();
    }
    // 2: dependent LDS reads (pointer chase: the value read selects the next address)
    {
        int idx = lane;
        t0 = tick();
        for (int r = 0; r < reps; ++r) {
            const double v = lds[idx];
            idx = (idx + 64 + (v > 2.0 ? 1 : 0)) & 8191;
        }
        acc_sink += idx;
        asm volatile("" ::"v"(acc_sink));
        t1 = tick();
        REC();
    }
    // 3: dependent wave-wide sums
    {
        double v = 1.0 + lane;
        t0 = tick();
        for (int r = 0; r < reps; ++r) v = wave_sum(v) * 1e-2;
        acc_sink += v;
        asm volatile("" ::"v"(acc_sink));
        t1 = tick();
        REC();
    }
    // 4: dependent exp
    {
        double v = -1e-3 * lane;
        t0 = tick();
        for (int r = 0; r < reps; ++r) v = exp(v) - 1.0;
        acc_sink += v;
        asm volatile("" ::"v"(acc_sink));
        t1 = tick();
        REC();
    }
    // 5: four independent exp chains (per exp)
    {
        double v0 = -1e-3 * lane, v1 = v0 - 0.1, v2 = v0 - 0.2, v3 = v0 - 0.3;
        t0 = tick();
        for (int r = 0; r < reps; r += 4) {
            v0 = exp(v0) - 1.0;
            v1 = exp(v1) - 1.0;
            v2 = exp(v2) - 1.0;
            v3 = exp(v3) - 1.0;
        }
        acc_sink += v0 + v1 + v2 + v3;
        asm volatile("" ::"v"(acc_sink));
        t1 = tick();
        REC();
    }
    // 6: workgroup barriers back to back (four waves)
    {
        t0 = tick();
        for (int r = 0; r < reps; ++r) __syncthreads();
        t1 = tick();
        REC();
    }
    // 7: dependent fused multiply-adds
    {
        double v = 1.0 + 1e-3 * lane;
        t0 = tick();
        for (int r = 0; r < reps; ++r) v = fma(v, 0.999, 1e-3);
        acc_sink += v;
        asm volatile("" ::"v"(acc_sink));
        t1 = tick();
        REC();
    }
    // 8: dependent sqrt
    {
        double v = 2.0 + lane;
        t0 = tick();
        for (int r = 0; r < reps; ++r) v = sqrt(v) + 1.5;
        acc_sink += v;
        asm volatile("" ::"v"(acc_sink));
        t1 = tick();
        REC();
    }
    // 9: a matrix-core step as the small kernels run it: eight fragment reads from LDS, then four dependent products (per step)
    {
        d4_t c = {0, 0, 0, 0};
        const int fl = lane & 15, fk = lane >> 4;
        t0 = tick();
        for (int r = 0; r < reps; ++r) {
            double af[4], bf[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                af[kk] = lds[((r & 7) * 16 + 4 * kk + fk) * 144 + fl];
                bf[kk] = lds[((r & 7) * 16 + 4 * kk + fk) * 144 + 16 + fl];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[kk], af[kk], c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc_sink += c[0] + c[3];
        asm volatile("" ::"v"(acc_sink));
        t1 = tick();
        REC();
    }
    // 10: LDS read-modify-write by one lane (dependent)
    {
        t0 = tick();
        for (int r = 0; r < reps; ++r)
            if (lane == 0) lds[r & 63] += 1.0;
        t1 = tick();
        REC();
    }
    // 11: dependent division
    {
        double v = 2.0 + lane;
        t0 = tick();
        for (int r = 0; r < reps; ++r) v = 3.0 / v + 1.0;
        acc_sink += v;
        asm volatile("" ::"v"(acc_sink));
        t1 = tick();
        REC();
    }
    // 12: dependent log
    {
        double v = 2.0 + lane;
        t0 = tick();
        for (int r = 0; r < reps; ++r) v = log(v) + 3.0;
        acc_sink += v;
        asm volatile("" ::"v"(acc_sink));
        t1 = tick();
        REC();
    }
    // 13: independent LDS reads, 8 in flight (per read)
    {
        double s = 0.0;
        t0 = tick();
        for (int r = 0; r < reps; r += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = lds[((r + u) & 127) * 144 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        acc_sink += s;
        asm volatile("" ::"v"(acc_sink));
        t1 = tick();
        REC();
    }
    if (tid == 0) out[NSLOT - 1] = slot;
    sink[tid] = acc_sink;
}

int main() {
    long long* d_out;
    double* d_sink;
    CK(hipMalloc(&d_out, NSLOT * sizeof(long long)));
    CK(hipMalloc(&d_sink, 256 * sizeof(double)));
    CK(hipFuncSetAttribute((const void*)lat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int reps = 256;
    long long h[NSLOT];
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(lat_kernel, dim3(1), dim3(256), 128 * 144 * 8, 0, d_out, d_sink, reps);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    const char* names[] = {"matrix product, dependent chain", "matrix product, two accumulators alternating", "LDS read, dependent", "wave_sum, dependent",
                           "exp, dependent", "exp, four independent chains", "workgroup barrier (4 waves)", "fma f64, dependent", "sqrt, dependent",
                           "matrix-core step: 8 LDS reads + 4 dependent products", "LDS read-modify-write by one lane", "division, dependent",
                           "log, dependent", "LDS read, 8 in flight"};
    printf("shader-clock cycles per operation (s_memtime; 2.4 GHz when the chip is otherwise idle)\n");
    for (int i = 0; i < (int)h[NSLOT - 1]; ++i) printf("%-56s %8.1f cycles  (%6.1f ns)\n", names[i], (double)h[i] / reps, (double)h[i] / reps / 2.4);
    return 0;
}
