// Acquisition hot loop: batched  w = K^-1 k  for every candidate (the dominant fp64 MFMA GEMM), the gradient
// contractions, the EI / GP-UCB finalisation and the lock-step bounded L-BFGS update.
//
// Replaces, for S candidates at once, the per-point call chain
//   objective (src/acquisition-function.cpp:38-56) -> CalcAcquisitionValue(:170-198) / ...Derivative(:200-230)
//   -> PredictMu/Sigma/MuDerivative/SigmaDerivative (gaussian-process-regressor.cpp:234-272,
//      preference-regressor.cpp:293-330) -> CalcSmallK / CalcSmallKSmallXDerivative (regressor.cpp:45-59,91-108)
// and the multi-start loop (src/acquisition-function.cpp:121-153).
#include <cstdlib>

#include "gemm_f64.hpp"
#include "kernels.hpp"
#include "../../include/sls_hip.h"

namespace slsk {

// ---------------------------------------------------------------------------------------------------------
// acq_gemm: tile = 128 candidates (m) x 128 rows of K^-1 (n').  acc = W = (K^-1 K*)^T tile.
// Epilogue: P = C* .* W stored candidate-major; per-tile partial sums over n' of K*.*W and C*.*W.
// ---------------------------------------------------------------------------------------------------------
// HALF = false: the whole tile.  HALF = true: its 64 K^-1 rows [64 nhalf, 64 nhalf + 64) only (see gemm_tile NJ = 2).
// Partial sums over the K^-1 rows are formed in ONE fixed order whichever form computes them: per lane a running sum over
// its 16 elements of a 64-row half in (j, r) order, then the two cross-lane steps; the sum of a tile is half 0 + half 1 --
// added in the tile kernel (whole tile: slot 2 tn holds the sum, slot 2 tn + 1 zero) or by finalize (half tiles: one slot
// each), the same addition either way: a candidate's sums do not depend on whether its tile ran whole or as two halves.
template <bool MATERN, bool HALF>
__device__ __forceinline__ void acq_tile(int t, int nhalf, int ntm, int ntn, const double* __restrict__ Ks, const double* __restrict__ Cs,
                                         long ldk, const double* __restrict__ Kinv, int Np, double* __restrict__ P,
                                         double* __restrict__ kw_part, double* __restrict__ cw_part, double* lds) {
    // grouped order: 8 candidate tiles x all K^-1 row tiles, so the 64 tiles resident on one XCD share panels in L2
    const int GM = 8;
    const int gsz = GM * ntn;
    const int g = t / gsz, w = t % gsz;
    const int gm = min(GM, ntm - g * GM);
    const int tm = g * GM + (w % gm), tn = w / gm;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    Acc acc;
    acc.zero();
    // The k loop runs 0 .. Np in the same order for every tile: a candidate's summation order does not depend on the tile
    // position it occupies (the active-set compaction of the maximiser moves candidates between tiles).
    gemm_tile<false, false, HALF ? 2 : 4, true>(acc, Ks + m0, ldk, Kinv + n0, (long)Np, 0, Np, lds, nhalf);   // Np % 128 == 0
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NJ = HALF ? 2 : 4;
    const int nbase = HALF ? 64 * nhalf + (wave >> 1) * 32 : (wave >> 1) * 64;
    double* red = lds;
    double* ex = lds + 1024;                          // half tile: [which][wm wave][i][lane] running sums handed lo -> hi
    // Per lane ONE running sum over the 16 elements of a 64-row half in (j, r) order.  In a half tile those 16 elements sit in
    // two waves (rows 0..31: wave >> 1 = 0, rows 32..63: wave >> 1 = 1, same lane positions): the second continues the first's
    // running sums, handed over through LDS, so the order of additions is the whole tile's.
    if (HALF && (wave >> 1) == 1) __syncthreads();    // wait for the lo waves' sums (they hit the matching barrier below)
    double skw[4], scw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long gm_ = m0 + acc_m(i);
        double pk = 0.0, pc = 0.0;
        if (HALF && (wave >> 1) == 1) {
            pk = ex[((0 * 2 + (wave & 1)) * 4 + i) * 64 + lane];
            pc = ex[((1 * 2 + (wave & 1)) * 4 + i) * 64 + lane];
        }
        // all K* / C* values of this row block first (up to 32 loads in flight), then the arithmetic and the stores: written
        // load - use - store per element the compiler emits 64 dependent memory round trips per lane
        double kv[NJ][4], cv[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long off = gm_ + (long)(n0 + nbase + 16 * j + (lane >> 4) + 4 * r) * ldk;
                kv[j][r] = Ks[off];
                if (MATERN) cv[j][r] = Cs[off];
            }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long off = gm_ + (long)(n0 + nbase + 16 * j + (lane >> 4) + 4 * r) * ldk;
                const double wv = acc.v[i][j][r];
                const double k = kv[j][r];
                const double c = MATERN ? cv[j][r] : k;
                const double p = c * wv;
                P[off] = p;
                pc += p;
                if (MATERN) pk += k * wv;
            }
        if (HALF && (wave >> 1) == 0) {
            ex[((0 * 2 + (wave & 1)) * 4 + i) * 64 + lane] = pk;
            ex[((1 * 2 + (wave & 1)) * 4 + i) * 64 + lane] = pc;
        }
        scw[i] = pc;
        skw[i] = MATERN ? pk : pc;
    }
    if (HALF && (wave >> 1) == 0) __syncthreads();    // publish the lo sums
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        skw[i] += __shfl_xor(skw[i], 16);
        skw[i] += __shfl_xor(skw[i], 32);
        scw[i] += __shfl_xor(scw[i], 16);
        scw[i] += __shfl_xor(scw[i], 32);
    }
    if (lane < 16 && (!HALF || (wave >> 1) == 1)) {   // half tile: the hi waves hold the half's sums
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ml = (wave & 1) * 64 + 16 * i + lane;
            const int slot = HALF ? 0 : (wave >> 1);
            red[(slot * 2 + 0) * 128 + ml] = skw[i];
            red[(slot * 2 + 1) * 128 + ml] = scw[i];
        }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int ml = threadIdx.x;
        if (HALF) {
            kw_part[(long)(2 * tn + nhalf) * ldk + m0 + ml] = red[0 * 128 + ml];
            cw_part[(long)(2 * tn + nhalf) * ldk + m0 + ml] = red[1 * 128 + ml];
        } else {
            // whole tile: the two halves' sums added here, into the even slot (finalize ignores the odd slot of a whole tile)
            kw_part[(long)(2 * tn + 0) * ldk + m0 + ml] = red[0 * 128 + ml] + red[2 * 128 + ml];
            cw_part[(long)(2 * tn + 0) * ldk + m0 + ml] = red[1 * 128 + ml] + red[3 * 128 + ml];
        }
    }
}

// ntiles: the tiles this launch runs as whole tiles (the first ntiles of the grouped order); a partially filled last
// generation goes to acq_gemm_half_kernel instead (launch_acq_gemm).
template <bool MATERN>
__global__ __launch_bounds__(256, 1) void acq_gemm_kernel(const double* __restrict__ Ks, const double* __restrict__ Cs, long ldk,
                                                          int Sp, const double* __restrict__ Kinv, int Np,
                                                          double* __restrict__ P, double* __restrict__ kw_part,
                                                          double* __restrict__ cw_part, int* __restrict__ sync,
                                                          int phase, int ntiles, int prio, int gate_every, int nx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int ntm = Sp / GEMM_BM, ntn = Np / GEMM_BN;
    if (sync == nullptr) {
        acq_tile<MATERN, false>(xcd_remap(blockIdx.x, ntiles), 0, ntm, ntn, Ks, Cs, ldk, Kinv, Np, P, kw_part, cw_part, lds);
        return;
    }
    // nx: the device's XCDs (launch_acq_gemm reads the geometry from the device; workgroups go round-robin over them)
    const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx, slots = gridDim.x / nx;
    const int nchunks = (ntiles + slots - 1) / slots;
    // Slots s and s + slots/2 of an XCD share a CU (cu_probe.hip: all 256 pairs).  With `phase` the two halves are gated
    // separately and the upper half starts `phase` ticks (100 MHz) late, so the two workgroups of a CU never run their
    // epilogues at the same moment (one tile's epilogue hides behind the other's MFMA loop again) while sharers of a
    // panel stay a few slabs apart, inside the L2 window.
    const int half = slots >> 1;
    const int grp = phase > 0 ? (slot >= half ? 1 : 0) : 0;
    // The two workgroups of a CU (slots s, s + half) at different wave priorities: two waves that alternate on a SIMD's
    // MFMA pipe lose ~3 % to the switches (MFMA-only loop: 77.2 TFLOP/s with one wave per SIMD, 75.1 with two); with a
    // strict order the favoured wave issues back to back and the other one fills its bubbles.
    if (prio && slot >= half) __builtin_amdgcn_s_setprio(1);
    int* gate = sync + 2 * xcd + grp;
    const int per_gen = phase > 0 ? half : slots;
    if (grp == 1) {
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            while (wall_clock64() - t0 < phase) __builtin_amdgcn_s_sleep(16);
        }
        __syncthreads();
    }
    int gen = 0;
    for (int c = xcd; c < nchunks; c += nx, ++gen) {
        if (gen > 0 && phase >= 0 && gen % gate_every == 0) {   // phase < 0: persistent but ungated (SLS_PERSIST=2)
            if (threadIdx.x == 0) {
                const int target = per_gen * gen;
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target &&
                       wall_clock64() - t0 < 20000)   // 100 MHz counter: give up after 200 us
                    __builtin_amdgcn_s_sleep(16);
            }
            __syncthreads();
        }
        const int t = c * slots + slot;
        if (t < ntiles) acq_tile<MATERN, false>(t, 0, ntm, ntn, Ks, Cs, ldk, Kinv, Np, P, kw_part, cw_part, lds);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(gate, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The partially filled LAST generation (at most half of the 512 workgroup slots would hold a tile): its tiles
// [tail_first, tail_first + gridDim.x / 2) run as half tiles on twice as many workgroups, unit u = tile u / 2, half u & 1.
template <bool MATERN>
__global__ __launch_bounds__(256, 1) void acq_gemm_half_kernel(const double* __restrict__ Ks, const double* __restrict__ Cs, long ldk,
                                                               int Sp, const double* __restrict__ Kinv, int Np,
                                                               double* __restrict__ P, double* __restrict__ kw_part,
                                                               double* __restrict__ cw_part, int tail_first) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int u = xcd_remap(blockIdx.x, gridDim.x);
    acq_tile<MATERN, true>(tail_first + (u >> 1), u & 1, Sp / GEMM_BM, Np / GEMM_BN, Ks, Cs, ldk, Kinv, Np, P, kw_part, cw_part, lds);
}

int launch_acq_gemm(hipStream_t s, const double* Ks, const double* Cs, long ldk, int Sp, const double* Kinv, int Np, double* P,
                    double* kw_part, double* cw_part, int* sync) {
    const int nt = (Sp / GEMM_BM) * (Np / GEMM_BN);
    // SLS_PERSIST (0: one tile per workgroup, 1: persistent workgroups with generation gates, 2: persistent without gates;
    // default 1 with two workgroups per CU, 2 with one, see below) is read per call so that tests and A/B runs can switch
    // within one process
    const int persist_env = (int)tune(TUNE_PERSIST, 1);
    // SLS_GATE_PHASE: start offset (ticks of the 100 MHz clock) between the two gate groups of an XCD; 0: one gate per XCD.
    // Measured per 65 536-candidate launch: phase 0 124.8 ms / 77 GB (hit rate 0.856); phase 2000..8000 123.2-123.3 ms /
    // 112 GB (0.795: each group of 8 x 4 tiles shares 12 panels); ungated 124.9 ms / 333 GB (0.42).
    const int phase = (int)tune(TUNE_GATE_PHASE, tune(TUNE_ACQ_WG_PER_CU, 1) == 2 ? 2000 : 0);
    // SLS_GATE_EVERY: the workgroups of an XCD meet at a gate only every n-th generation.  Round 5, one workgroup per CU, per
    // 65 536-candidate launch (bench.py's own measurement, tools/ab_acq_gemm_forms.sh): ungated 4198.6 ms per step / 164 GB of
    // fabric traffic; gated every generation 4235 / 116; every 4th 4230 / 116; every 16th 4199.5 / 116.2; every 32nd 4200 / 118;
    // every 64th 4202 / 128.  The sharers of a panel drift apart slowly: meeting every 16 tiles keeps them inside the L2 window
    // (-29 % traffic) and costs nothing measurable.
    const int gate_every = std::max(1, (int)tune(TUNE_GATE_EVERY, tune(TUNE_ACQ_WG_PER_CU, 1) == 2 ? 1 : 16));
    const int prio = 0;   // (wave priority for one of a CU's two workgroups was a switch in rounds 2-3: no measurable effect, removed)
    // persistent, generation-gated form when there are at least two generations of tiles (MI355X: 256 CUs x 2 = 512 slots)
    // SLS_ACQ_WG_PER_CU (default 1): one workgroup per CU (a 96 KB LDS request keeps a second one out).  One wave per SIMD has
    // the MFMA pipe to itself -- two waves alternating on it lose ~3 % to the switches (gemm_probe_ring, 16384 x 8192 x 8192:
    // 76.8 TFLOP/s = the MFMA-only loop's rate, against 75.0 with two workgroups per CU).  The tile epilogues are then hidden by
    // nobody, which is why a gate in front of EVERY tile costs this form 0.3-1.4 % (rounds 3-4 therefore ran it ungated, at
    // 156-164 instead of 116 GB of fabric traffic per 65 536-candidate launch); round 5: a gate every 16th tile (SLS_GATE_EVERY)
    // keeps the traffic of the gated form at the speed of the ungated one.  Persistent workgroups walking their tile lists without gates are another
    // 0.25 % faster than one workgroup per tile (4293 / 4298 -> 4285 / 4282 ms): no workgroup launch between tiles.
    // 2 = two per CU, generation-gated (the round-1 form).
    const int wg_per_cu = tune(TUNE_ACQ_WG_PER_CU, 1) == 2 ? 2 : 1;
    const ChipGeometry chip = chip_geometry();                       // from the device, not literals: a partitioned device has fewer CUs / XCDs
    const int cap = chip.n_cu * wg_per_cu;                           // tiles the chip holds at a time (MI355X, SPX: 256 or 512)
    const int nx = (chip.n_xcd <= 8 && cap % chip.n_xcd == 0) ? chip.n_xcd : 1;   // the gate table has 2 x 8 words
    const int lds_bytes = wg_per_cu == 1 ? 96 * 1024 : GEMM_LDS_BYTES;
    ensure_dyn_lds((const void*)acq_gemm_kernel<false>, lds_bytes);
    ensure_dyn_lds((const void*)acq_gemm_kernel<true>, lds_bytes);
    ensure_dyn_lds((const void*)acq_gemm_half_kernel<false>, lds_bytes);
    ensure_dyn_lds((const void*)acq_gemm_half_kernel<true>, lds_bytes);
    const bool persist = persist_env && sync && nt >= 2 * cap && Np >= 2048;
    const int phase_k = persist_env == 2 ? -1 : phase;               // SLS_PERSIST=2: persistent workgroups without the gates
    // Tail split (SLS_TAIL_SPLIT=0 disables): the chip holds 512 tiles at a time; if the last such generation is at most half
    // full its tiles run as half tiles on twice the workgroups in a second launch (same bits, half the time for that
    // generation: 663 -> 642 ms per step on the 8 192-start shard of an 8-GPU run).
    const bool tail_ok = tune_on(TUNE_TAIL_SPLIT);
    const int rem = nt % cap;
    const int tail = (tail_ok && rem > 0 && rem <= cap / 2) ? rem : 0;
    const int nmain = nt - tail;
    int* sy = nullptr;
    int grid = nmain;
    if (persist && nmain >= 2 * cap) {
        (void)hipMemsetAsync(sync, 0, 16 * sizeof(int), s);
        sy = sync;
        grid = cap;
    }
    const bool matern = Cs != Ks;
    if (nmain > 0) {
        if (matern)
            hipLaunchKernelGGL(acq_gemm_kernel<true>, dim3(grid), dim3(GEMM_THREADS), lds_bytes, s, Ks, Cs, ldk, Sp, Kinv, Np, P,
                               kw_part, cw_part, sy, phase_k, nmain, prio, gate_every, nx);
        else
            hipLaunchKernelGGL(acq_gemm_kernel<false>, dim3(grid), dim3(GEMM_THREADS), lds_bytes, s, Ks, Cs, ldk, Sp, Kinv, Np, P,
                               kw_part, cw_part, sy, phase_k, nmain, prio, gate_every, nx);
    }
    if (tail > 0) {
        if (matern)
            hipLaunchKernelGGL(acq_gemm_half_kernel<true>, dim3(2 * tail), dim3(GEMM_THREADS), lds_bytes, s, Ks, Cs, ldk, Sp, Kinv,
                               Np, P, kw_part, cw_part, nmain);
        else
            hipLaunchKernelGGL(acq_gemm_half_kernel<false>, dim3(2 * tail), dim3(GEMM_THREADS), lds_bytes, s, Ks, Cs, ldk, Sp, Kinv,
                               Np, P, kw_part, cw_part, nmain);
    }
    return nmain;   // tiles [nmain, nt) ran as two halves
}

// ---------------------------------------------------------------------------------------------------------
// var_gemm: prediction without gradients.  sigma^2 = a - |L^-1 k*|^2 needs only V = L^-1 K* with the TRIANGULAR factor:
// tile (128 candidates, 128 rows n' of L^-1) contracts k < 128 (tn + 1) only, i.e. N^2 flops per candidate instead of the
// 2 N^2 of the K^-1 form (SURVEY.md 8d "Batched predict": N^2 M*).  Epilogue: per-tile partial sums of v^2 into kw_part
// (finalize then forms a - sum as for the K^-1 path); no P, no C* traffic.  Long tiles (large tn) are issued first.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void var_tile(int tm, int tn, const double* __restrict__ Ks, long ldk, const double* __restrict__ Linv,
                                         int Np, double* __restrict__ kw_part, double* __restrict__ cw_part, double* lds) {
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    Acc acc;
    acc.zero();
    gemm_tile<false, false, 4, true>(acc, Ks + m0, ldk, Linv + n0, (long)Np, 0, GEMM_BN * (tn + 1), lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double sv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double p = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) p = fma(acc.v[i][j][r], acc.v[i][j][r], p);
        p += __shfl_xor(p, 16);
        p += __shfl_xor(p, 32);
        sv[i] = p;
    }
    __syncthreads();   // the k loop's last LDS reads are done before the buffer is reused
    double* red = lds;
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(wave >> 1) * 128 + (wave & 1) * 64 + 16 * i + lane] = sv[i];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int ml = threadIdx.x;
        kw_part[(long)(2 * tn) * ldk + m0 + ml] = red[ml] + red[128 + ml];   // even slot of acq_gemm's layout (whole tile)
        cw_part[(long)(2 * tn) * ldk + m0 + ml] = 0.0;
    }
    __syncthreads();
}

// pair == 0: one tile per workgroup, long tiles (large tn) first.  pair == 1 (fewer than two rounds of tiles): workgroup
// (tm, p) runs tiles tn = ntn-1-p and tn = p back to back, so every workgroup contracts 128 (ntn + 1) columns in total;
// with one tile each the launch would last as long as its longest tile, i.e. as long as the K^-1 form.
__global__ __launch_bounds__(256, 2) void var_gemm_kernel(const double* __restrict__ Ks, long ldk, int Sp,
                                                          const double* __restrict__ Linv, int Np,
                                                          double* __restrict__ kw_part, double* __restrict__ cw_part, int pair) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int ntm = Sp / GEMM_BM, ntn = Np / GEMM_BN;
    if (!pair) {
        const int T = ntm * ntn;
        int tm = blockIdx.x % ntm, tn = ntn - 1 - blockIdx.x / ntm;
        if ((ntm & 7) == 0 && (ntn & 7) == 0) {
            // 8 x 8 tile groups (64 workgroups resident on one XCD share 16 operand panels through its L2), groups ordered
            // longest first and dealt round-robin to the XCDs: a contiguous range per XCD would hand all the long tiles
            // (large tn) to XCD 0 and the launch would last almost as long as the untriangular product.
            const int t = xcd_remap(blockIdx.x, T);
            const int per = T >> 3;
            const int x = t / per, l = t - x * per;
            const int gid = (l >> 6) * 8 + x, w = l & 63;
            const int ngm = ntm >> 3;
            tn = ntn - 1 - (8 * (gid / ngm) + (w >> 3));
            tm = 8 * (gid % ngm) + (w & 7);
        }
        var_tile(tm, tn, Ks, ldk, Linv, Np, kw_part, cw_part, lds);
        return;
    }
    const int tm = blockIdx.x % ntm, p = blockIdx.x / ntm;
    const int hi = ntn - 1 - p;
    var_tile(tm, hi, Ks, ldk, Linv, Np, kw_part, cw_part, lds);
    if (p < hi) var_tile(tm, p, Ks, ldk, Linv, Np, kw_part, cw_part, lds);
}

void launch_var_gemm(hipStream_t s, const double* Ks, long ldk, int Sp, const double* Linv, int Np, double* kw_part,
                     double* cw_part) {
    ensure_dyn_lds((const void*)var_gemm_kernel, GEMM_LDS_BYTES);
    const int ntm = Sp / GEMM_BM, ntn = Np / GEMM_BN;
    const int pair = (ntm * ntn < 1024 && ntn > 1) ? 1 : 0;
    const int grid = pair ? ntm * ((ntn + 1) / 2) : ntm * ntn;
    hipLaunchKernelGGL(var_gemm_kernel, dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, Ks, ldk, Sp, Linv, Np, kw_part, cw_part,
                       pair);
}

// ---------------------------------------------------------------------------------------------------------
// grad_gemm: z = 0: Gs = P * X~ ; z = 1: Gm = C* * (alpha .* X~).  Tile = 128 candidates x 128 dims.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void grad_gemm_kernel(const double* __restrict__ P, const double* __restrict__ Cs, long ldk,
                                                           int Sp, const double* __restrict__ XT, const double* __restrict__ XaT,
                                                           long ld, int Np, int Dcols, double* __restrict__ Gs,
                                                           double* __restrict__ Gm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int ntm = Sp / GEMM_BM;
    const int tm = blockIdx.x % ntm, tn = blockIdx.x / ntm;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    const double* A = blockIdx.y == 0 ? P : Cs;
    const double* B = blockIdx.y == 0 ? XT : XaT;
    double* C = blockIdx.y == 0 ? Gs : Gm;
    Acc acc;
    acc.zero();
    // B operand: elem(n' = d, k = i) = B[i + d*ld]  -> K-contiguous
    gemm_tile<false, true>(acc, A + m0, ldk, B + (long)n0 * ld, ld, 0, Np, lds);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)(m0 + acc_m(i)) + (long)(n0 + acc_n(j, r)) * ldk] = acc.v[i][j][r];
}

constexpr int GRAD64_A_LD = 136;   // A slab rows of 136 doubles: 53 248 B of LDS per workgroup, three workgroups per CU (gemm_tile_n64)
// D <= 64: 128 x 64 tiles (no wasted MFMA columns), 3 workgroups per CU.
// The contraction over the N training points is ALWAYS summed as four quarter ranges, ((q0 + q1) + q2) + q3, each quarter
// accumulated from zero: one workgroup runs the four quarters back to back (gridDim.z == 1), or -- when the launch has fewer
// tiles than the chip has workgroup slots (small active sets, the per-GPU shard of a multi-GPU run) -- four workgroups
// take one quarter each and grad_reduce4_kernel adds the partials in that same order.  Either way the same bits, so a
// candidate's gradient does not depend on how many other candidates share its launch.
__global__ __launch_bounds__(256, 3) void grad_gemm64_kernel(const double* __restrict__ P, const double* __restrict__ Cs, long ldk,
                                                             int Sp, const double* __restrict__ XT, const double* __restrict__ XaT,
                                                             long ld, int Np, double* __restrict__ Gs, double* __restrict__ Gm,
                                                             double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int m0 = blockIdx.x * GEMM_BM;
    const double* A = blockIdx.y == 0 ? P : Cs;
    const double* B = blockIdx.y == 0 ? XT : XaT;
    const bool split = gridDim.z > 1;
    double* C = split ? part + ((long)blockIdx.z * 2 + blockIdx.y) * (long)Sp * 64 : (blockIdx.y == 0 ? Gs : Gm);
    const long ldc = split ? (long)Sp : ldk;
    const int q = Np / 4;                                   // Np is a multiple of 128: quarters are multiples of 32
    Acc64 tot;
    tot.zero();
    const int c0 = split ? blockIdx.z : 0, c1 = split ? blockIdx.z + 1 : 4;
    for (int c = c0; c < c1; ++c) {
        Acc64 acc;
        acc.zero();
        gemm_tile_n64<false, GRAD64_A_LD>(acc, A + m0, ldk, B, ld, c * q, (c + 1) * q, lds);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) tot.v[i][j] = (c == c0) ? acc.v[i][j] : tot.v[i][j] + acc.v[i][j];
    }
    const int lane = threadIdx.x & 63, wm = (threadIdx.x >> 6) * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(long)(m0 + wm + 16 * i + (lane & 15)) + (long)(16 * j + (lane >> 4) + 4 * r) * ldc] = tot.v[i][j][r];
}

__global__ __launch_bounds__(256) void grad_reduce4_kernel(const double* __restrict__ part, int Sp, long ldk, double* __restrict__ Gs,
                                                           double* __restrict__ Gm) {
    const long idx = blockIdx.x * 256L + threadIdx.x;       // n + d * Sp, d < 64
    if (idx >= (long)Sp * 64) return;
    const long n = idx % Sp, d = idx / Sp;
    const long stride = (long)Sp * 64;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const double* p = part + (long)y * stride + idx;
        const double v = ((p[0] + p[2 * stride]) + p[4 * stride]) + p[6 * stride];
        (y == 0 ? Gs : Gm)[n + d * ldk] = v;
    }
}

void launch_grad_gemm(hipStream_t s, const double* P, const double* Cs, long ldk, int Sp, const double* XT, const double* XaT,
                      long ld, int Np, int Dcols, double* Gs, double* Gm, double* part) {
    ensure_dyn_lds((const void*)grad_gemm_kernel, GEMM_LDS_BYTES);
    ensure_dyn_lds((const void*)grad_gemm64_kernel, gemm_n64_lds_bytes<GRAD64_A_LD>());
    if (Dcols < 0) {   // caller signals D <= 64 by passing -Dcols
        const bool split = part != nullptr && grad_gemm_wants_split(Sp);
        hipLaunchKernelGGL(grad_gemm64_kernel, dim3(Sp / GEMM_BM, 2, split ? 4 : 1), dim3(GEMM_THREADS), gemm_n64_lds_bytes<GRAD64_A_LD>(), s, P, Cs,
                           ldk, Sp, XT, XaT, ld, Np, Gs, Gm, part);
        if (split)
            hipLaunchKernelGGL(grad_reduce4_kernel, dim3((unsigned)(((long)Sp * 64 + 255) / 256)), dim3(256), 0, s, part, Sp, ldk, Gs, Gm);
        return;
    }
    const int nt = (Sp / GEMM_BM) * (Dcols / GEMM_BN);
    hipLaunchKernelGGL(grad_gemm_kernel, dim3(nt, 2), dim3(GEMM_THREADS), GEMM_LDS_BYTES, s, P, Cs, ldk, Sp, XT, XaT, ld, Np, Dcols,
                       Gs, Gm);
}

// ---------------------------------------------------------------------------------------------------------
// finalize: one thread per candidate.  mathtoolbox EI / GP-UCB on the reduced sums.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs p) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= p.S) return;
    double mu = 0.0, ca = 0.0, kw = 0.0, cw = 0.0;
    const int tm_ = n >> 7, grp_ = tm_ >> 3;
    const int gm_ = min(8, p.ntm - 8 * grp_);
    const int tile_base = grp_ * 8 * p.nbt + (tm_ - 8 * grp_);       // tile index = tile_base + t * gm_ (acq_tile's order)
    #pragma unroll 8
    for (int t = 0; t < p.nbt; ++t) {
        mu += p.mu_part[(long)t * p.ldk + n];
        ca += p.ca_part[(long)t * p.ldk + n];
        // tile (tm, t) in acq_gemm's grouped order; the tiles from split_first on ran as two halves (one slot each)
        double kt = p.kw_part[(long)(2 * t) * p.ldk + n], ct = p.cw_part[(long)(2 * t) * p.ldk + n];
        if (tile_base + t * gm_ >= p.split_first) {
            kt += p.kw_part[(long)(2 * t + 1) * p.ldk + n];
            ct += p.cw_part[(long)(2 * t + 1) * p.ldk + n];
        }
        kw += kt;
        cw += ct;
    }
    if (p.kw_solve_part) {                            // k . LLT.solve(k) = |L^-1 k|^2 (preference-regressor.cpp:299-313): var_gemm's sums
        kw = 0.0;
        for (int t = 0; t < p.nbt; ++t) kw += p.kw_solve_part[(long)(2 * t) * p.ldk + n];
    }
    const double s2 = p.a - kw;
    const double sigma = s2 < 0.0 ? 0.0 : sqrt(s2);   // gaussian-process-regressor.cpp:253-254
    if (p.mu) p.mu[n] = mu;
    if (p.sigma) p.sigma[n] = sigma;
    const bool need_g = p.dmu || p.dsigma || p.grad;
    double Phi = 0.0, phi = 0.0;
    bool bad = false;
    if (p.val || p.grad) {
        if (p.acq == SLS_ACQ_EXPECTED_IMPROVEMENT) {
            const double diff = mu - p.mu_best;
            const double u = diff / sigma;
            Phi = 0.5 * erfc(-u * 0.70710678118654752440);
            phi = exp(-0.5 * u * u) * 0.39894228040143267794;
            const double ei = diff * Phi + sigma * phi;
            bad = (sigma < 1e-10) || isnan(ei);
            if (p.val) p.val[n] = bad ? 0.0 : ei;
        } else {
            if (p.val) p.val[n] = mu + p.ucb_h * sigma;
        }
    }
    if (!need_g) return;
    const double inv_sigma = 1.0 / sigma;
    if (p.grad && p.acq == SLS_ACQ_EXPECTED_IMPROVEMENT) {
        // NaN anywhere in the gradient -> zero vector (mathtoolbox guard); needs a scan before the write
        for (int d = 0; d < p.D && !bad; ++d) {
            const double xt = p.XsT[n + (long)d * p.ldk];
            const double il = p.inv_ell[d];
            const double dm = -il * (xt * ca - p.Gm[n + (long)d * p.ldk]);
            const double ds = inv_sigma * il * (xt * cw - p.Gs[n + (long)d * p.ldk]);
            if (isnan(Phi * dm + phi * ds)) bad = true;
        }
    }
    #pragma unroll 4
    for (int d = 0; d < p.D; ++d) {
        const double xt = p.XsT[n + (long)d * p.ldk];
        const double il = p.inv_ell[d];
        const double dm = -il * (xt * ca - p.Gm[n + (long)d * p.ldk]);                 // PredictMuDerivative
        const double ds = inv_sigma * il * (xt * cw - p.Gs[n + (long)d * p.ldk]);      // PredictSigmaDerivative
        if (p.dmu) p.dmu[n + (long)d * p.ldo] = dm;
        if (p.dsigma) p.dsigma[n + (long)d * p.ldo] = ds;
        if (p.grad) {
            double g;
            if (p.acq == SLS_ACQ_EXPECTED_IMPROVEMENT) g = bad ? 0.0 : Phi * dm + phi * ds;
            else g = dm + p.ucb_h * ds;
            p.grad[n + (long)d * p.ldo] = g;
        }
    }
}

void launch_finalize(hipStream_t s, const FinalizeArgs& a) {
    hipLaunchKernelGGL(finalize_kernel, dim3((a.S + 255) / 256), dim3(256), 0, s, a);
}

__global__ __launch_bounds__(256) void combine_kernel(int S, int D, long ld, const double* __restrict__ mu,
                                                     const double* __restrict__ sigma, const double* __restrict__ dmu,
                                                     const double* __restrict__ dsigma, int acq, double mu_best, double ucb_h,
                                                     double* __restrict__ val, double* __restrict__ grad) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= S) return;
    const double m = mu[n], sg = sigma[n];
    if (acq == SLS_ACQ_EXPECTED_IMPROVEMENT) {
        const double diff = m - mu_best;
        const double u = diff / sg;
        const double Phi = 0.5 * erfc(-u * 0.70710678118654752440);
        const double phi = exp(-0.5 * u * u) * 0.39894228040143267794;
        const double ei = diff * Phi + sg * phi;
        bool bad = (sg < 1e-10) || isnan(ei);
        val[n] = bad ? 0.0 : ei;
        if (grad) {
            for (int d = 0; d < D && !bad; ++d)
                if (isnan(Phi * dmu[n + d * ld] + phi * dsigma[n + d * ld])) bad = true;
            for (int d = 0; d < D; ++d) grad[n + d * ld] = bad ? 0.0 : Phi * dmu[n + d * ld] + phi * dsigma[n + d * ld];
        }
    } else {
        val[n] = m + ucb_h * sg;
        if (grad)
            for (int d = 0; d < D; ++d) grad[n + d * ld] = dmu[n + d * ld] + ucb_h * dsigma[n + d * ld];
    }
}
void launch_combine(hipStream_t s, int S, int D, long ld, const double* mu, const double* sigma, const double* dmu,
                    const double* dsigma, int acq, double mu_best, double ucb_h, double* val, double* grad) {
    hipLaunchKernelGGL(combine_kernel, dim3((S + 255) / 256), dim3(256), 0, s, S, D, ld, mu, sigma, dmu, dsigma, acq, mu_best,
                       ucb_h, val, grad);
}

// ---------------------------------------------------------------------------------------------------------
// Lock-step bounded L-BFGS (the oracle's slso_acq_maximize is the same algorithm, statement by
// statement).  Four lanes per start (see lbfgs_step_kernel); every per-start vector is candidate-major so all accesses coalesce.
// Minimises phi = -acq on [0,1]^D.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void clamp_starts_kernel(const double* __restrict__ starts, int D, int S, double* __restrict__ xt,
                                                           long ld, int Sp) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= Sp) return;
    for (int d = 0; d < D; ++d) {
        double v = 0.5;
        if (n < S) {
            v = starts[d + (long)n * D];
            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        }
        xt[n + (long)d * ld] = v;
    }
}
void launch_clamp_starts(hipStream_t s, const double* starts, int D, int S, double* xt, long ld, int Sp) {
    hipLaunchKernelGGL(clamp_starts_kernel, dim3((Sp + 255) / 256), dim3(256), 0, s, starts, D, S, xt, ld, Sp);
}

// FOUR lanes per start (q = 0..3 own the dimensions q, q + 4, ...), 16 starts per 64-thread workgroup.  With one thread per
// start the 8 192-start shard of an 8-GPU run put one wave on 128 of the chip's 1024 SIMDs, each walking a serial chain of ~20
// passes over D (16.6 ms per step).  Sums over the dimensions are formed per lane and combined as (q0 + q1) + (q2 + q3) in
// every lane (xor shuffles: the same bits in all four), so the four lanes of a start take the same branches; the scalar state of
// a start is read once, carried in registers and written back by lane q = 0.  The group size is fixed: a start's arithmetic must
// not depend on how many starts are alive (active-set compaction).
__global__ __launch_bounds__(64) void lbfgs_step_kernel(LbfgsState st, const double* __restrict__ val,
                                                         const double* __restrict__ grad, int first) {
    const int q = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + (threadIdx.x & 15);     // column of (val, grad)
    if (j >= st.nlive) return;
    auto gsum = [](double v) {
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return v;
    };
    auto gmax = [](double v) {
        v = fmax(v, __shfl_xor(v, 16));
        v = fmax(v, __shfl_xor(v, 32));
        return v;
    };
    const int n = st.live ? st.live[j] : j;          // the start it belongs to
    const long ld = st.ld, ldv = st.ldv;
    const int D = st.D, m = st.m;
    // the state arrays are distinct buffers: restrict-qualified views let the compiler issue a pass's loads together instead
    // of ordering every load behind the previous store
    double* __restrict__ x_ = st.x; double* __restrict__ g_ = st.g; double* __restrict__ dir_ = st.dir;
    double* __restrict__ xt_ = st.xt; double* __restrict__ scr_ = st.scr;
    double* __restrict__ ShA = st.Sh; double* __restrict__ YhA = st.Yh;
    // scalar state of the start (identical in its four lanes)
    double f_v = 0.0, t_v = 1.0;
    int hlen_v = 0, hpos_v = 0, nbt_v = 0;
    bool done = false;
    int idx_new = -1;            // history slot written in this call (its rho is not in memory yet for the other lanes)
    double rho_new = 0.0;
    bool need_dir = false;
    if (first) {
        f_v = -val[j];
        #pragma unroll 4
        for (int d = q; d < D; d += 4) {
            x_[n + d * ld] = xt_[n + d * ld];
            g_[n + d * ld] = -grad[j + d * ldv];
        }
        need_dir = true;
    } else {
        if (st.done[n]) return;
        f_v = st.f[n]; t_v = st.t[n]; hlen_v = st.hlen[n]; hpos_v = st.hpos[n]; nbt_v = st.nbt[n];
        const double ft = -val[j];
        double gs = 0.0, ss = 0.0;
        #pragma unroll 4
        for (int d = q; d < D; d += 4) {
            const double sd = xt_[n + d * ld] - x_[n + d * ld];
            gs += g_[n + d * ld] * sd;
            ss += sd * sd;
        }
        gs = gsum(gs);
        ss = gsum(ss);
        if (ss == 0.0) {
            if (q == 0) st.done[n] = 1;
            return;
        }
        if (ft <= f_v + st.c1 * gs) {
            double sy = 0.0, yy = 0.0, xmoved = 0.0;
            const int idx = hpos_v;
            double* __restrict__ Sh = ShA + (long)idx * D * ld;
            double* __restrict__ Yh = YhA + (long)idx * D * ld;
            #pragma unroll 4
            for (int d = q; d < D; d += 4) {
                const double xtd = xt_[n + d * ld], gtd = -grad[j + d * ldv];
                const double sd = xtd - x_[n + d * ld];
                const double yd = gtd - g_[n + d * ld];
                xmoved += lbfgs_x_moved(x_[n + d * ld], xtd, st.xtol_rel);
                Sh[n + d * ld] = sd;
                Yh[n + d * ld] = yd;
                sy += sd * yd;
                yy += yd * yd;
                x_[n + d * ld] = xtd;
                g_[n + d * ld] = gtd;
            }
            sy = gsum(sy);
            yy = gsum(yy);
            if (sy > 1e-10 * yy && sy > 0.0) {
                idx_new = idx;
                rho_new = 1.0 / sy;
                if (q == 0) st.rho[(long)idx * ld + n] = rho_new;
                hpos_v = (idx + 1) % m;
                if (hlen_v < m) hlen_v += 1;
            }
            if (lbfgs_f_stalled(f_v, ft, st.ftol_rel) || (st.xtol_rel > 0.0 && gsum(xmoved) == 0.0)) done = true;   // NLopt's relative tests
            f_v = ft;
            need_dir = true;
        } else {
            t_v *= st.shrink;
            nbt_v += 1;
            if (nbt_v > st.max_backtracks) done = true;
        }
    }
    if (!done && need_dir) {
        // projected gradient -> scr (pg), two-loop recursion in dir
        double pgmax = 0.0, pgn2 = 0.0;
        #pragma unroll 4
        for (int d = q; d < D; d += 4) {
            double v = g_[n + d * ld];
            const double xv = x_[n + d * ld];
            if ((xv <= 0.0 && v > 0.0) || (xv >= 1.0 && v < 0.0)) v = 0.0;
            scr_[n + d * ld] = v;
            dir_[n + d * ld] = v;
            pgmax = fmax(pgmax, fabs(v));
            pgn2 += v * v;
        }
        pgmax = gmax(pgmax);
        pgn2 = gsum(pgn2);
        if (!(pgmax > st.gtol)) {
            done = true;
        } else {
            const int hlen = hlen_v;
            const int hpos = hpos_v;
            auto rho_of = [&](int idx) { return idx == idx_new ? rho_new : st.rho[(long)idx * ld + n]; };
            double al[8];
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                al[h] = 0.0;
                if (h < hlen) {
                    const int idx = (hpos - 1 - h + 2 * m) % m;
                    const double* __restrict__ Sh = ShA + (long)idx * D * ld;
                    const double* __restrict__ Yh = YhA + (long)idx * D * ld;
                    double dot = 0.0;
                    #pragma unroll 4
                    for (int d = q; d < D; d += 4) dot += Sh[n + d * ld] * dir_[n + d * ld];
                    al[h] = rho_of(idx) * gsum(dot);
                    #pragma unroll 4
                    for (int d = q; d < D; d += 4) dir_[n + d * ld] -= al[h] * Yh[n + d * ld];
                }
            }
            double gamma;
            if (hlen > 0) {
                const int idx = (hpos - 1 + m) % m;
                const double* __restrict__ Sh = ShA + (long)idx * D * ld;
                const double* __restrict__ Yh = YhA + (long)idx * D * ld;
                double sy = 0.0, yy = 0.0;
                #pragma unroll 4
                for (int d = q; d < D; d += 4) {
                    sy += Sh[n + d * ld] * Yh[n + d * ld];
                    yy += Yh[n + d * ld] * Yh[n + d * ld];
                }
                gamma = gsum(sy) / gsum(yy);
            } else {
                const double nn = sqrt(pgn2);
                gamma = 1.0 / (nn > 1.0 ? nn : 1.0);
            }
            #pragma unroll 4
            for (int d = q; d < D; d += 4) dir_[n + d * ld] *= gamma;
#pragma unroll
            for (int h = 7; h >= 0; --h) {
                if (h < hlen) {
                    const int idx = (hpos - 1 - h + 2 * m) % m;
                    const double* __restrict__ Sh = ShA + (long)idx * D * ld;
                    const double* __restrict__ Yh = YhA + (long)idx * D * ld;
                    double dot = 0.0;
                    #pragma unroll 4
                    for (int d = q; d < D; d += 4) dot += Yh[n + d * ld] * dir_[n + d * ld];
                    const double beta = rho_of(idx) * gsum(dot);
                    #pragma unroll 4
                    for (int d = q; d < D; d += 4) dir_[n + d * ld] += Sh[n + d * ld] * (al[h] - beta);
                }
            }
            double gd = 0.0;
            #pragma unroll 4
            for (int d = q; d < D; d += 4) {
                const double pg = scr_[n + d * ld];
                const double dv = (pg == 0.0) ? 0.0 : -dir_[n + d * ld];
                dir_[n + d * ld] = dv;
                gd += pg * dv;
            }
            gd = gsum(gd);
            if (!(gd < 0.0)) {
                hlen_v = 0;
                const double nn = sqrt(pgn2);
                gamma = 1.0 / (nn > 1.0 ? nn : 1.0);
                gd = 0.0;
                #pragma unroll 4
                for (int d = q; d < D; d += 4) {
                    const double pg = scr_[n + d * ld];
                    const double dv = -gamma * pg;
                    dir_[n + d * ld] = dv;
                    gd += pg * dv;
                }
                gd = gsum(gd);
                if (!(gd < 0.0)) done = true;
            }
            if (!done) { t_v = 1.0; nbt_v = 0; }
        }
    }
    // propose the next trial point.  A trial that coincides with x (the step vanished in the clamp / in rounding) would be
    // evaluated once and then stop the start with x, f unchanged ("ss == 0" above): it is retired here, one evaluation
    // earlier, with the same end state.
    double moved = 0.0;
    #pragma unroll 4
    for (int d = q; d < D; d += 4) {
        const double xv = x_[n + d * ld];
        double v = xv;
        if (!done) {
            v = v + t_v * dir_[n + d * ld];
            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            if (v != xv) moved = 1.0;
        }
        xt_[n + d * ld] = v;
    }
    moved = gmax(moved);
    if (!done && moved == 0.0) done = true;
    if (q == 0) {
        st.f[n] = f_v; st.t[n] = t_v; st.hlen[n] = hlen_v; st.hpos[n] = hpos_v; st.nbt[n] = nbt_v;
        st.done[n] = done ? 1 : 0;
    }
}

// The same step with the start's vectors in REGISTERS (D <= 4 DPL; lane q holds the dimensions q + 4 e, e < DPL).  The kernel
// above walks ~20 + 4 m passes over D through global memory, each pass loading what the previous one stored (dir is read and
// rewritten 2 m + 3 times): all of a launch's waves are resident at once, so its duration is ONE wave's chain of store -> load
// round trips (505 us per round at 45 000 live starts, 6x what its bytes need).  Here x, g, the direction and the newest (s, y)
// pair stay in registers from the first load to the last store; the history is only read, so the loads of a pair do not wait
// for anything and those of the next pair are issued while the current one is used (the loads are unconditional -- every slot
// of the ring is valid memory -- and the arithmetic of a slot beyond hlen is skipped).  Expressions, summation order and
// branches are those of lbfgs_step_kernel: the bits are the same (tests run both, SLS_LBFGS_REG=0 selects the memory form).
template <int DPL>
__global__ __launch_bounds__(64) void lbfgs_step_reg_kernel(LbfgsState st, const double* __restrict__ val,
                                                             const double* __restrict__ grad, int first) {
    const int q = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + (threadIdx.x & 15);     // column of (val, grad)
    if (j >= st.nlive) return;
    auto gsum = [](double v) {
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return v;
    };
    auto gmax = [](double v) {
        v = fmax(v, __shfl_xor(v, 16));
        v = fmax(v, __shfl_xor(v, 32));
        return v;
    };
    const int n = st.live ? st.live[j] : j;
    const long ld = st.ld, ldv = st.ldv;
    const int D = st.D, m = st.m;
    double* __restrict__ x_ = st.x; double* __restrict__ g_ = st.g; double* __restrict__ dir_ = st.dir;
    double* __restrict__ xt_ = st.xt;
    double* __restrict__ ShA = st.Sh; double* __restrict__ YhA = st.Yh;
    double xr[DPL], gr[DPL], dr[DPL], sn[DPL], yn[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) xr[e] = gr[e] = dr[e] = sn[e] = yn[e] = 0.0;
    // history pair `idx` of start n: one row of 4 DPL doubles, lane q's DPL values (dimensions q, q + 4, ...) contiguous -- read
    // and written with 16-byte accesses, and only the rows of LIVE starts are touched (lbfgs_step_kernel's [h][d][n] arrays are
    // fetched line by line whatever is alive: 800 MB per round at 65 536 starts, the whole cost of that kernel)
    auto hrow = [&](double* base, int idx) { return base + (((long)n * m + idx) * 4 + q) * DPL; };
    auto hload = [&](const double* row, double (&v)[DPL]) {
#pragma unroll
        for (int e = 0; e < DPL; e += 2) {
            const d2_t t = *reinterpret_cast<const d2_t*>(row + e);
            v[e] = t[0]; v[e + 1] = t[1];
        }
    };
    double f_v = 0.0, t_v = 1.0;
    int hlen_v = 0, hpos_v = 0, nbt_v = 0;
    bool done = false;
    bool have_new = false;       // (sn, yn, rho_new) is the newest history pair, written to slot hpos - 1 in this call
    double rho_new = 0.0;
    bool need_dir = false;
    if (first) {
        f_v = -val[j];
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            const int d = q + 4 * e;
            if (d < D) {
                xr[e] = xt_[n + d * ld];
                gr[e] = -grad[j + d * ldv];
                x_[n + d * ld] = xr[e];
                g_[n + d * ld] = gr[e];
            }
        }
        need_dir = true;
    } else {
        if (st.done[n]) return;
        f_v = st.f[n]; t_v = st.t[n]; hlen_v = st.hlen[n]; hpos_v = st.hpos[n]; nbt_v = st.nbt[n];
        const double ft = -val[j];
        double xtr[DPL];
        double gs = 0.0, ss = 0.0;
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            const int d = q + 4 * e;
            if (d < D) { xtr[e] = xt_[n + d * ld]; xr[e] = x_[n + d * ld]; gr[e] = g_[n + d * ld]; }
        }
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            const int d = q + 4 * e;
            if (d < D) {
                const double sd = xtr[e] - xr[e];
                gs += gr[e] * sd;
                ss += sd * sd;
            }
        }
        gs = gsum(gs);
        ss = gsum(ss);
        if (ss == 0.0) {
            if (q == 0) st.done[n] = 1;
            return;
        }
        if (ft <= f_v + st.c1 * gs) {
            double sy = 0.0, yy = 0.0, xmoved = 0.0;
            const int idx = hpos_v;
#pragma unroll
            for (int e = 0; e < DPL; ++e) {
                const int d = q + 4 * e;
                if (d < D) {
                    const double xtd = xtr[e], gtd = -grad[j + d * ldv];
                    const double sd = xtd - xr[e];
                    const double yd = gtd - gr[e];
                    xmoved += lbfgs_x_moved(xr[e], xtd, st.xtol_rel);
                    sn[e] = sd;
                    yn[e] = yd;
                    sy += sd * yd;
                    yy += yd * yd;
                    xr[e] = xtd;
                    gr[e] = gtd;
                    x_[n + d * ld] = xtd;
                    g_[n + d * ld] = gtd;
                }
            }
            {
                double* __restrict__ Sh = hrow(ShA, idx);
                double* __restrict__ Yh = hrow(YhA, idx);
#pragma unroll
                for (int e = 0; e < DPL; e += 2) {
                    *reinterpret_cast<d2_t*>(Sh + e) = d2_t{sn[e], sn[e + 1]};
                    *reinterpret_cast<d2_t*>(Yh + e) = d2_t{yn[e], yn[e + 1]};
                }
            }
            sy = gsum(sy);
            yy = gsum(yy);
            if (sy > 1e-10 * yy && sy > 0.0) {
                have_new = true;
                rho_new = 1.0 / sy;
                if (q == 0) st.rho[(long)idx * ld + n] = rho_new;
                hpos_v = (idx + 1) % m;
                if (hlen_v < m) hlen_v += 1;
            }
            if (lbfgs_f_stalled(f_v, ft, st.ftol_rel) || (st.xtol_rel > 0.0 && gsum(xmoved) == 0.0)) done = true;   // NLopt's relative tests
            f_v = ft;
            need_dir = true;
        } else {
            t_v *= st.shrink;
            nbt_v += 1;
            if (nbt_v > st.max_backtracks) done = true;
        }
    }
    // projected gradient of dimension e (recomputed where the memory form re-reads its scratch copy: the same value)
    auto pg_of = [&](int e) {
        double v = gr[e];
        const double xv = xr[e];
        if ((xv <= 0.0 && v > 0.0) || (xv >= 1.0 && v < 0.0)) v = 0.0;
        return v;
    };
    bool have_dir = false;       // dr holds the new direction (to be stored); otherwise the stored one is used below
    if (!done && need_dir) {
        double pgmax = 0.0, pgn2 = 0.0;
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            const int d = q + 4 * e;
            if (d < D) {
                const double v = pg_of(e);
                dr[e] = v;
                pgmax = fmax(pgmax, fabs(v));
                pgn2 += v * v;
            }
        }
        pgmax = gmax(pgmax);
        pgn2 = gsum(pgn2);
        if (!(pgmax > st.gtol)) {
            done = true;
        } else {
            const int hlen = hlen_v;
            const int hpos = hpos_v;
            double al[8];
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                al[h] = 0.0;
                if (h < m) {                                       // uniform; the ring has m valid slots
                    const int idx = (hpos - 1 - h + 2 * m) % m;
                    const bool newest = h == 0 && have_new;
                    double sh[DPL], yh[DPL];
                    hload(hrow(ShA, idx), sh);
                    hload(hrow(YhA, idx), yh);
                    const double rho = st.rho[(long)idx * ld + n];
                    if (h < hlen) {
                        double dot = 0.0;
#pragma unroll
                        for (int e = 0; e < DPL; ++e)
                            if (q + 4 * e < D) dot += (newest ? sn[e] : sh[e]) * dr[e];
                        al[h] = (newest ? rho_new : rho) * gsum(dot);
#pragma unroll
                        for (int e = 0; e < DPL; ++e)
                            if (q + 4 * e < D) dr[e] -= al[h] * (newest ? yn[e] : yh[e]);
                    }
                }
            }
            double gamma;
            if (hlen > 0) {
                const int idx = (hpos - 1 + m) % m;
                double sh[DPL], yh[DPL];
                hload(hrow(ShA, idx), sh);
                hload(hrow(YhA, idx), yh);
                double sy = 0.0, yy = 0.0;
#pragma unroll
                for (int e = 0; e < DPL; ++e) {
                    if (q + 4 * e < D) {
                        const double sv = have_new ? sn[e] : sh[e];
                        const double yv = have_new ? yn[e] : yh[e];
                        sy += sv * yv;
                        yy += yv * yv;
                    }
                }
                gamma = gsum(sy) / gsum(yy);
            } else {
                const double nn = sqrt(pgn2);
                gamma = 1.0 / (nn > 1.0 ? nn : 1.0);
            }
#pragma unroll
            for (int e = 0; e < DPL; ++e)
                if (q + 4 * e < D) dr[e] *= gamma;
#pragma unroll
            for (int h = 7; h >= 0; --h) {
                if (h < m) {
                    const int idx = (hpos - 1 - h + 2 * m) % m;
                    const bool newest = h == 0 && have_new;
                    double sh[DPL], yh[DPL];
                    hload(hrow(ShA, idx), sh);
                    hload(hrow(YhA, idx), yh);
                    const double rho = st.rho[(long)idx * ld + n];
                    if (h < hlen) {
                        double dot = 0.0;
#pragma unroll
                        for (int e = 0; e < DPL; ++e)
                            if (q + 4 * e < D) dot += (newest ? yn[e] : yh[e]) * dr[e];
                        const double beta = (newest ? rho_new : rho) * gsum(dot);
#pragma unroll
                        for (int e = 0; e < DPL; ++e)
                            if (q + 4 * e < D) dr[e] += (newest ? sn[e] : sh[e]) * (al[h] - beta);
                    }
                }
            }
            double gd = 0.0;
#pragma unroll
            for (int e = 0; e < DPL; ++e) {
                if (q + 4 * e < D) {
                    const double pg = pg_of(e);
                    const double dv = (pg == 0.0) ? 0.0 : -dr[e];
                    dr[e] = dv;
                    gd += pg * dv;
                }
            }
            gd = gsum(gd);
            if (!(gd < 0.0)) {
                hlen_v = 0;
                const double nn = sqrt(pgn2);
                gamma = 1.0 / (nn > 1.0 ? nn : 1.0);
                gd = 0.0;
#pragma unroll
                for (int e = 0; e < DPL; ++e) {
                    if (q + 4 * e < D) {
                        const double pg = pg_of(e);
                        const double dv = -gamma * pg;
                        dr[e] = dv;
                        gd += pg * dv;
                    }
                }
                gd = gsum(gd);
                if (!(gd < 0.0)) done = true;
            }
            if (!done) { t_v = 1.0; nbt_v = 0; }
            have_dir = true;
        }
    }
    double moved = 0.0;
#pragma unroll
    for (int e = 0; e < DPL; ++e) {
        const int d = q + 4 * e;
        if (d < D) {
            if (have_dir) dir_[n + d * ld] = dr[e];
            const double xv = xr[e];
            double v = xv;
            if (!done) {
                v = v + t_v * (have_dir ? dr[e] : dir_[n + d * ld]);
                v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
                if (v != xv) moved = 1.0;
            }
            xt_[n + d * ld] = v;
        }
    }
    moved = gmax(moved);
    if (!done && moved == 0.0) done = true;
    if (q == 0) {
        st.f[n] = f_v; st.t[n] = t_v; st.hlen[n] = hlen_v; st.hpos[n] = hpos_v; st.nbt[n] = nbt_v;
        st.done[n] = done ? 1 : 0;
    }
}

void launch_lbfgs_step(hipStream_t s, const LbfgsState& st, const double* val, const double* grad, bool first) {
    if (st.nlive <= 0) return;
    const bool use_reg = tune_on(TUNE_LBFGS_REG);
    const dim3 grid((st.nlive + 15) / 16), block(64);
    if (use_reg && st.D <= 16) hipLaunchKernelGGL(lbfgs_step_reg_kernel<4>, grid, block, 0, s, st, val, grad, (int)first);
    else if (use_reg && st.D <= 64) hipLaunchKernelGGL(lbfgs_step_reg_kernel<16>, grid, block, 0, s, st, val, grad, (int)first);
    else hipLaunchKernelGGL(lbfgs_step_kernel, grid, block, 0, s, st, val, grad, (int)first);
}

// One workgroup: thread t owns the contiguous segment [t*per, (t+1)*per) of the input list, counts its survivors, an
// exclusive scan over the 1024 counts gives its output offset (stable, increasing order).
// Stable compaction of the starts still moving, in two small multi-block launches (the single-workgroup form walked 64
// dependent gathers per thread: 158 us per round at 65 536 starts, 1.3 % of an 8-GPU shard's step): blocks of 1024 entries,
// (1) live entries per block, (2) every block adds up the counts of the blocks before it (at most 64 words) and scatters its
// entries behind that offset by an in-block exclusive scan.  Output order = input order, as before (same live lists).
__device__ __forceinline__ int compact_flags(const int* __restrict__ live_in, int n_in, const int* __restrict__ done, int base, int (&n)[4]) {
    int c = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = base + e;
        n[e] = -1;
        if (i < n_in) {
            const int idx = live_in ? live_in[i] : i;
            if (!done[idx]) { n[e] = idx; ++c; }
        }
    }
    return c;
}
__global__ __launch_bounds__(256) void compact_count_kernel(const int* __restrict__ live_in, int n_in, const int* __restrict__ done,
                                                            int* __restrict__ block_counts) {
    __shared__ int red[4];
    int n[4];
    int c = compact_flags(live_in, n_in, done, blockIdx.x * 1024 + 4 * threadIdx.x, n);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void compact_scatter_kernel(const int* __restrict__ live_in, int n_in, const int* __restrict__ done,
                                                              const int* __restrict__ block_counts, int* __restrict__ live_out,
                                                              int* __restrict__ count_out) {
    __shared__ int wsum[4];
    __shared__ int boff;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (wave == 0) {                                         // offset of this block = entries of the blocks before it
        int v = 0;
        for (int q = lane; q < (int)blockIdx.x; q += 64) v += block_counts[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) boff = v;
    }
    int n[4];
    const int c = compact_flags(live_in, n_in, done, blockIdx.x * 1024 + 4 * t, n);
    int incl = c;                                            // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int pos = boff + incl - c;
    for (int w = 0; w < wave; ++w) pos += wsum[w];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (n[e] >= 0) live_out[pos++] = n[e];
    if (blockIdx.x == gridDim.x - 1 && t == 255) count_out[0] = pos;
}
void launch_compact_live(hipStream_t s, const int* live_in, int n_in, const int* done, int* live_out, int* count_out, int* block_counts) {
    const int nb = (n_in + 1023) / 1024;
    hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(256), 0, s, live_in, n_in, done, block_counts);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(nb), dim3(256), 0, s, live_in, n_in, done, block_counts, live_out, count_out);
}

__global__ __launch_bounds__(256) void gather_trials_kernel(const double* __restrict__ xt, long ld, int D, const int* __restrict__ live,
                                                            const int* __restrict__ count_dev, double* __restrict__ xc, long ldc) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int cnt = count_dev[0];
    const int cntp = (cnt + 127) & ~127;
    if (j >= cntp) return;
    if (j < cnt) {
        const int n = live[j];
        for (int d = 0; d < D; ++d) xc[j + d * ldc] = xt[n + d * ld];
    } else {
        for (int d = 0; d < D; ++d) xc[j + d * ldc] = 0.5;
    }
}
void launch_gather_trials(hipStream_t s, const double* xt, long ld, int D, const int* live, const int* count_dev, int n_max,
                          double* xc, long ldc) {
    if (n_max <= 0) return;
    const int np = (n_max + 127) & ~127;
    hipLaunchKernelGGL(gather_trials_kernel, dim3((np + 255) / 256), dim3(256), 0, s, xt, ld, D, live, count_dev, xc, ldc);
}

// first maximum: Eigen maxCoeff semantics (src/acquisition-function.cpp:146-153)
// The end of a maximiser run in one launch: first minimum of f (= first maximum of the acquisition value, strictly-greater scan per
// thread, then a tree that prefers the lower index on ties), and the winner's coordinates x[bi + d ldx] gathered behind it -- into `best` (mapped host memory):
// [0] value, [1] index, [2] *counter (a statistics word of the run, or 0), [8 .. 8 + D) coordinates.  It was a launch, two or three
// blocking copies, two or three synchronisations and a strided copy.
__global__ __launch_bounds__(1024) void argmax_neg_gather_kernel(const double* __restrict__ f, int n, const double* __restrict__ x, long ldx,
                                                                 int D, double* __restrict__ best,
                                                                 const unsigned long long* __restrict__ counter) {
    __shared__ double sv[1024];
    __shared__ int si[1024];
    double bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double v = -f[i];
        if (v > bv) { bv = v; bi = i; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double ov = sv[threadIdx.x + s];
            const int oi = si[threadIdx.x + s];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const int win = (si[0] == 0x7fffffff) ? 0 : si[0];
    if (threadIdx.x == 0) {
        best[0] = (si[0] == 0x7fffffff) ? -f[0] : sv[0];
        best[1] = (double)win;
        best[2] = counter ? (double)counter[0] : 0.0;
    }
    for (int d = threadIdx.x; d < D; d += 1024) best[8 + d] = x[win + (long)d * ldx];
}
void launch_argmax_neg_gather(hipStream_t s, const double* f, int S, const double* x, long ldx, int D, double* best,
                              const unsigned long long* counter) {
    hipLaunchKernelGGL(argmax_neg_gather_kernel, dim3(1), dim3(1024), 0, s, f, S, x, ldx, D, best, counter);
}

}  // namespace slsk
