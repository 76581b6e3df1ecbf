// probe: ring forms of the 128x128 fp64 tile (gemm_f64.hpp: gemm_tile_ring) against the shipped double-buffered gemm_tile.
// usage: gemm_probe_ring M N K [reps]     -- prints time, TFLOP/s and a checksum of C for every variant (the checksums must agree:
// same k order, same bits)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_f64.hpp"
using namespace slsk;

namespace slsk {
// ---- variants measured and rejected (round 2); they lived in gemm_f64.hpp while being measured ----
// Ring form of the M-contiguous x M-contiguous tile: slabs of BK k-rows in a ring of STAGES LDS stages, the LDS-direct
// loads of slab s + STAGES - 1 issued as soon as the barrier of slab s has retired slab s - 1.  Same fragment layout and
// the same k order as gemm_tile, hence the same bits; what changes is how long a load may take before somebody waits for
// it: gemm_tile waits for slab s + 1 at the end of slab s (~48 MFMAs = 1.3 us after the issue), the ring with BK = 8 and
// four stages waits ~2.75 slabs = 88 MFMAs = 2.4 us after it, with the same 73 728 bytes of LDS (two workgroups per CU).
// The barrier is a bare s_barrier behind an explicit s_waitcnt: __syncthreads() carries a workgroup fence, for which the
// compiler emits s_waitcnt vmcnt(0) -- that drains the ring (it is why the first 4-stage attempt measured "no change").
// lgkmcnt(0): this wave's LDS reads of the retiring slab have returned before anybody's load may overwrite it.
template <int NJ = 4, int BK = 8, int STAGES = 4>
__device__ __forceinline__ void gemm_tile_ring(Acc& acc, const double* __restrict__ A, long lda, const double* __restrict__ B,
                                               long ldb, int kb, int ke, double* lds, int kfirst = -1, int nhalf = 0) {
    static_assert(BK == 8 || BK == 16, "slab depth");
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
    constexpr int SLAB = BK * GEMM_LDS_MC_LD;      // doubles per operand slab
    constexpr int STAGE = 2 * SLAB;                // A slab | B slab
    constexpr int RPW = BK / 4;                    // k-rows each wave brings per slab and operand
    constexpr int LPW = 2 * RPW;                   // loads per wave and slab
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave & 1) * 64;
    const int wn = NJ == 4 ? (wave >> 1) * 64 : 64 * nhalf + (wave >> 1) * 32;
    if (kb >= ke) return;
    if (kfirst < kb || kfirst >= ke) kfirst = kb;
    const int nst = (ke - kb) / BK;
    const double* Ap = A + 2 * lane;
    const double* Bp = B + 2 * lane;
    int kiss = kfirst;                             // k of the next slab to issue
    int siss = 0;                                  // its ring position
    auto issue = [&]() {
        double* base = lds + siss * STAGE;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = RPW * wave + r;
            slab_row_to_lds(Ap + (long)(kiss + row) * lda, base + row * GEMM_LDS_MC_LD);
            slab_row_to_lds(Bp + (long)(kiss + row) * ldb, base + SLAB + row * GEMM_LDS_MC_LD);
        }
        kiss += BK;
        if (kiss >= ke) kiss = kb;
        siss = (siss + 1 == STAGES) ? 0 : siss + 1;
    };
    for (int s = 0; s < STAGES - 1 && s < nst; ++s) issue();
    int scur = 0;
    for (int s = 0; s < nst; ++s) {
        // slabs s + 1 .. s + STAGES - 2 may stay in flight
        const int later = nst - 1 - s;
        if (later >= STAGES - 2) ring_wait_barrier<(STAGES - 2) * LPW>();
        else if (STAGES == 4 && later == 1) ring_wait_barrier<LPW>();
        else ring_wait_barrier<0>();
        if (s + STAGES - 1 < nst) issue();         // into the stage slab s - 1 occupied
        const double* la = lds + scur * STAGE;
        const double* lb = la + SLAB;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double af[4], bf[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = frag_read<false>(la, wm + 16 * i, kk, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = frag_read<false>(lb, wn + 16 * j, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc.v[i][j], 0, 0, 0);
        }
        scur = (scur + 1 == STAGES) ? 0 : scur + 1;
    }
    ring_wait_barrier<0>();                        // the callers reuse the LDS block in their epilogues
}

// Software-pipelined ring (BK = 8, four stages): the two k-groups of a slab alternate with the barrier in between,
//     12 MFMAs on F0 | read F1 (slab s, k-group 1) | 4 MFMAs on F0 | 8 MFMAs on F1 | wait + barrier (slab s + 1 landed,
//     slab s read by everybody) | issue slab s + 4 into slab s's stage | read F0 (slab s + 1, k-group 0) | 8 MFMAs on F1
// so every fragment read is issued 4-8 MFMAs (256-512 cycles) before its first use and the instructions behind a barrier
// are MFMAs whose operands are already in registers: the wave has no LDS-latency bubble per slab (gemm_tile and the plain ring
// expose one after every barrier, which only the co-resident workgroup's wave can fill).
template <int NJ = 4>
__device__ __forceinline__ void gemm_tile_pipe(Acc& acc, const double* __restrict__ A, long lda, const double* __restrict__ B,
                                               long ldb, int kb, int ke, double* lds, int kfirst = -1, int nhalf = 0) {
    constexpr int BK = 8, STAGES = 4;
    constexpr int SLAB = BK * GEMM_LDS_MC_LD;
    constexpr int STAGE = 2 * SLAB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave & 1) * 64;
    const int wn = NJ == 4 ? (wave >> 1) * 64 : 64 * nhalf + (wave >> 1) * 32;
    if (kb >= ke) return;
    if (kfirst < kb || kfirst >= ke) kfirst = kb;
    const int nst = (ke - kb) / BK;
    const double* Ap = A + 2 * lane;
    const double* Bp = B + 2 * lane;
    int kiss = kfirst, siss = 0;
    auto issue = [&]() {
        double* base = lds + siss * STAGE;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = 2 * wave + r;
            slab_row_to_lds(Ap + (long)(kiss + row) * lda, base + row * GEMM_LDS_MC_LD);
            slab_row_to_lds(Bp + (long)(kiss + row) * ldb, base + SLAB + row * GEMM_LDS_MC_LD);
        }
        kiss += BK;
        if (kiss >= ke) kiss = kb;
        siss = (siss + 1) & (STAGES - 1);
    };
    auto wait_for = [&](int later) {   // `later` slabs behind the awaited one may stay in flight (4 loads per wave each)
        if (later >= 3) ring_wait_barrier<12>();
        else if (later == 2) ring_wait_barrier<8>();
        else if (later == 1) ring_wait_barrier<4>();
        else ring_wait_barrier<0>();
    };
    auto read = [&](double (&af)[4], double (&bf)[NJ], int stage, int kk) {
        const double* la = lds + stage * STAGE;
        const double* lb = la + SLAB;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = frag_read<false>(la, wm + 16 * i, kk, lane);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[j] = frag_read<false>(lb, wn + 16 * j, kk, lane);
    };
    // part 0: rows i < 3, part 1: row i = 3; parts 2 / 3: the (i + j) even / odd MFMAs (both halves use every fragment, so
    // no fragment read can be sunk past the first half towards the barrier)
    auto mfma = [&](const double (&af)[4], const double (&bf)[NJ], int part) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bool on = part == 0 ? i < 3 : part == 1 ? i == 3 : ((i + j) & 1) == (part & 1);
                if (on) acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc.v[i][j], 0, 0, 0);
            }
    };
    for (int s = 0; s < STAGES && s < nst; ++s) issue();
    wait_for(min(STAGES, nst) - 1);
    double a0[4], b0[NJ], a1[4], b1[NJ];
    read(a0, b0, 0, 0);
    int scur = 0;
    for (int s = 0; s < nst; ++s) {
        // (the compiler puts s_waitcnt lgkmcnt(0) in front of each k-group: every read is placed a few MFMAs before that)
        mfma(a0, b0, 0);
        __builtin_amdgcn_sched_barrier(0);
        read(a1, b1, scur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma(a0, b0, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma(a1, b1, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nst) {
            wait_for(min(STAGES - 2, nst - 2 - s));
            if (s + STAGES < nst) issue();         // into slab s's stage
            read(a0, b0, (scur + 1) & (STAGES - 1), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma(a1, b1, 3);
        __builtin_amdgcn_sched_barrier(0);
        scur = (scur + 1) & (STAGES - 1);
    }
    ring_wait_barrier<0>();                        // the callers reuse the LDS block in their epilogues
}

}  // namespace slsk


// The shipped double buffer (BK = 16 x 2) with the eight LDS-direct loads of the next slab SPREAD between the MFMAs (one
// load per LSTEP MFMAs) instead of issued in one burst behind the barrier, fragment reads software-pipelined by hand and
// the barrier in the middle of the last k-group (8 MFMAs follow it, then the next slab's first fragments are there).
template <int LSTEP>
__device__ __forceinline__ void gemm_tile_spread(Acc& acc, const double* __restrict__ A, long lda, const double* __restrict__ B,
                                                 long ldb, int K, double* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    const double* Ap = A + 2 * lane + (long)(4 * wave) * lda;
    const double* Bp = B + 2 * lane + (long)(4 * wave) * ldb;
    // load l of a slab: l < 4 -> A row 4 wave + l, else B row 4 wave + l - 4
    auto issue1 = [&](int l, int k0, int bufoff) {
        const int r = l & 3;
        if (l < 4) slab_row_to_lds(Ap + (long)(k0 + r) * lda, lds + bufoff + (4 * wave + r) * GEMM_LDS_MC_LD);
        else slab_row_to_lds(Bp + (long)(k0 + r) * ldb, lds + bufoff + GEMM_LDS_TILE + (4 * wave + r) * GEMM_LDS_MC_LD);
    };
    auto read = [&](double (&f)[8], int bufoff, int kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = frag_read<false>(lds + bufoff, wm + 16 * i, kk, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) f[4 + j] = frag_read<false>(lds + bufoff + GEMM_LDS_TILE, wn + 16 * j, kk, lane);
    };
#pragma unroll
    for (int l = 0; l < 8; ++l) issue1(l, 0, 0);
    ring_wait_barrier<0>();
    double f0[8], f1[8];
    read(f0, 0, 0);
    int cur = 0;
    const int nslab = K / GEMM_BK;
    for (int s = 0; s < nslab; ++s) {
        const bool more = s + 1 < nslab;
        const int nxt = cur ^ (2 * GEMM_LDS_TILE);
        const int k1 = (s + 1) * GEMM_BK;
        int nl = 0;   // loads issued so far (compile-time after unrolling)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double(&f)[8] = (kk & 1) ? f1 : f0;
            double(&g)[8] = (kk & 1) ? f0 : f1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[4 + j], f[i], acc.v[i][j], 0, 0, 0);
                    const int m = kk * 16 + i * 4 + j + 1;          // MFMAs issued in this slab so far
                    if (m % LSTEP == 0 && m / LSTEP <= 8) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) issue1(m / LSTEP - 1, k1, nxt);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (kk < 3 && i == 2) {                             // next k-group's fragments, 4 MFMAs ahead
                    __builtin_amdgcn_sched_barrier(0);
                    read(g, cur, kk + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kk == 3 && i == 1) {                            // slab s + 1 landed and slab s read by everybody
                    __builtin_amdgcn_sched_barrier(0);
                    ring_wait_barrier<0>();
                    if (more) read(g, nxt, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        (void)nl;
        cur = nxt;
    }
    ring_wait_barrier<0>();
}

// spread, second form: load addresses = uniform row pointer (SGPR pair, advanced by scalar adds) + a constant 32-bit lane
// offset (saddr form of global_load_lds: no VALU work per load), last slab peeled (no branch around the loads).
template <int LSTEP, bool MORE, int RI = 2, int BI = 1>
__device__ __forceinline__ void spread2_slab(Acc& acc, double (&f0)[8], double (&f1)[8], const double* __restrict__ Arow,
                                             const double* __restrict__ Brow, long lda, long ldb, unsigned loff, double* lds, int cur,
                                             int wave, int wm, int wn, int lane) {
    const int nxt = cur ^ (2 * GEMM_LDS_TILE);
    auto issue1 = [&](int l) {
        const int r = l & 3;
        const char* g = l < 4 ? reinterpret_cast<const char*>(Arow + (long)r * lda) : reinterpret_cast<const char*>(Brow + (long)r * ldb);
        double* d = lds + nxt + (l < 4 ? 0 : GEMM_LDS_TILE) + (4 * wave + r) * GEMM_LDS_MC_LD;
        slab_row_to_lds(reinterpret_cast<const double*>(g + loff), d);
    };
    auto read = [&](double (&f)[8], int bufoff, int kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = frag_read<false>(lds + bufoff, wm + 16 * i, kk, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) f[4 + j] = frag_read<false>(lds + bufoff + GEMM_LDS_TILE, wn + 16 * j, kk, lane);
    };
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        double(&f)[8] = (kk & 1) ? f1 : f0;
        double(&g)[8] = (kk & 1) ? f0 : f1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[4 + j], f[i], acc.v[i][j], 0, 0, 0);
                const int m = kk * 16 + i * 4 + j + 1;
                if (MORE && m % LSTEP == 0 && m / LSTEP <= 8) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue1(m / LSTEP - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (kk < 3 && i == RI) {
                __builtin_amdgcn_sched_barrier(0);
                read(g, cur, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kk == 3 && i == BI) {
                __builtin_amdgcn_sched_barrier(0);
                ring_wait_barrier<0>();
                if (MORE) read(g, nxt, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <int LSTEP, int RI = 2, int BI = 1>
__device__ __forceinline__ void gemm_tile_spread2(Acc& acc, const double* __restrict__ A, long lda, const double* __restrict__ B,
                                                  long ldb, int K, double* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    const unsigned loff = 16u * lane;
    const double* Arow = A + (long)(4 * wave) * lda;   // uniform: this wave's first k-row of the slab being loaded
    const double* Brow = B + (long)(4 * wave) * ldb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        slab_row_to_lds(reinterpret_cast<const double*>(reinterpret_cast<const char*>(Arow + (long)r * lda) + loff), lds + (4 * wave + r) * GEMM_LDS_MC_LD);
        slab_row_to_lds(reinterpret_cast<const double*>(reinterpret_cast<const char*>(Brow + (long)r * ldb) + loff), lds + GEMM_LDS_TILE + (4 * wave + r) * GEMM_LDS_MC_LD);
    }
    ring_wait_barrier<0>();
    double f0[8], f1[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) f0[i] = frag_read<false>(lds, wm + 16 * i, 0, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) f0[4 + j] = frag_read<false>(lds + GEMM_LDS_TILE, wn + 16 * j, 0, lane);
    int cur = 0;
    const int nslab = K / GEMM_BK;
    for (int s = 0; s + 1 < nslab; ++s) {
        Arow += (long)GEMM_BK * lda;
        Brow += (long)GEMM_BK * ldb;
        spread2_slab<LSTEP, true, RI, BI>(acc, f0, f1, Arow, Brow, lda, ldb, loff, lds, cur, wave, wm, wn, lane);
        cur ^= 2 * GEMM_LDS_TILE;
    }
    spread2_slab<LSTEP, false, RI, BI>(acc, f0, f1, Arow, Brow, lda, ldb, loff, lds, cur, wave, wm, wn, lane);
    ring_wait_barrier<0>();
}

// Ablations of the shipped loop (timing only, results are wrong): which ingredient costs the MFMA pipe its idle cycles?
//   bit 0: no s_waitcnt vmcnt(0)   bit 1: no barrier   bit 2: no global -> LDS loads   bit 3: no LDS fragment reads (registers)
template <int ABL>
__device__ __forceinline__ void gemm_tile_ablate(Acc& acc, const double* __restrict__ A, long lda, const double* __restrict__ B,
                                                 long ldb, int K, double* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    auto issue = [&](int k0, int bufoff) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * wave + r;
            slab_row_to_lds(A + 2 * lane + (long)(k0 + row) * lda, lds + bufoff + row * GEMM_LDS_MC_LD);
            slab_row_to_lds(B + 2 * lane + (long)(k0 + row) * ldb, lds + bufoff + GEMM_LDS_TILE + row * GEMM_LDS_MC_LD);
        }
    };
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    double af[4], bf[4];
    unsigned dummy32 = lane;
    unsigned long long dummy64 = lane, dummy64b = 3;
    unsigned dummys = wave;
    if (ABL & 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { af[i] = frag_read<false>(lds, wm + 16 * i, 0, lane); bf[i] = frag_read<false>(lds + GEMM_LDS_TILE, wn + 16 * i, 0, lane); }
    }
    for (int k0 = 0; k0 < K; k0 += GEMM_BK) {
        const bool more = (k0 + GEMM_BK) < K;
        if (more && !(ABL & 4)) issue(k0 + GEMM_BK, cur ^ (2 * GEMM_LDS_TILE));
        const double* la = lds + cur;
        const double* lb = lds + cur + GEMM_LDS_TILE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (!(ABL & 8)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = frag_read<false>(la, wm + 16 * i, kk, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = frag_read<false>(lb, wn + 16 * j, kk, lane);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc.v[i][j], 0, 0, 0);
            if (ABL & 16) {
#pragma unroll
                for (int q = 0; q < 6; ++q) asm volatile("v_add_u32 %0, 1, %0" : "+v"(dummy32));
            }
            if (ABL & 32) {
#pragma unroll
                for (int q = 0; q < 6; ++q) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(dummy64) : "v"(dummy64b));
            }
            if (ABL & 64) {
#pragma unroll
                for (int q = 0; q < 6; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(dummys));
            }
        }
        if (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(ABL & 2)) __syncthreads();
        else asm volatile("" ::: "memory");
        cur ^= 2 * GEMM_LDS_TILE;
    }
    if (dummy32 == 0xdeadbeefu && dummy64 == 77 && dummys == 0xdeadbeefu) acc.v[0][0][0] += 1.0;
}

template <int VAR>
__global__ __launch_bounds__(256, 2) void probe_kernel(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                       double* __restrict__ C, long ldc, int M, int N, int K, int hot, int* cu_cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int ntm = M / 128, ntn = N / 128;
    int t = xcd_remap(blockIdx.x, ntm * ntn);
    const int gsz = 8 * ntn;
    const int g = t / gsz, w = t % gsz;
    const int gm = min(8, ntm - g * 8);
    const int tm = g * 8 + (w % gm), tn = w / gm;
    const int m0 = tm * 128, n0 = tn * 128;
    if (hot) {   // every tile reads the same two panels (always in L2): what is left of the loss is not memory latency
        A -= m0;
        B -= n0;
    }
    Acc acc;
    acc.zero();
    if (VAR == 0) gemm_tile<false, false>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 1) gemm_tile_ring<4, 8, 4>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 2) gemm_tile_ring<4, 16, 2>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 3) gemm_tile_ring<4, 8, 3>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 4) gemm_tile_ring<4, 16, 4>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 6) gemm_tile_pipe<4>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 50 || VAR == 53) {   // the two workgroups of a CU at different priorities (which one arrived first: per-CU counter)
        __shared__ int s_par;
        if (threadIdx.x == 0) {
            unsigned x, h;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
            s_par = atomicAdd(&cu_cnt[(x & 7) * 256 + ((h >> 8) & 0xff)], 1) & 1;
        }
        __syncthreads();
        if (s_par) {
            if (VAR == 50) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(1);
        }
    }
    if (VAR == 40 || VAR == 50 || VAR == 51 || VAR == 53) gemm_tile_mc<4>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 41) gemm_tile_mc<4, true>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
    if (VAR == 52) gemm_tile_ablate<15>(acc, A + m0, lda, B + n0, ldb, K, lds);
    if (VAR == 20) gemm_tile_spread<2>(acc, A + m0, lda, B + n0, ldb, K, lds);
    if (VAR == 21) gemm_tile_spread<4>(acc, A + m0, lda, B + n0, ldb, K, lds);
    if (VAR == 22) gemm_tile_spread<1>(acc, A + m0, lda, B + n0, ldb, K, lds);
    if (VAR == 23) gemm_tile_spread<6>(acc, A + m0, lda, B + n0, ldb, K, lds);
    if (VAR == 30) gemm_tile_spread2<2>(acc, A + m0, lda, B + n0, ldb, K, lds);
    if (VAR == 31) gemm_tile_spread2<4>(acc, A + m0, lda, B + n0, ldb, K, lds);
    if (VAR == 32) gemm_tile_spread2<1>(acc, A + m0, lda, B + n0, ldb, K, lds);
    if (VAR >= 1000) gemm_tile_spread2<2, (VAR / 10) % 10, VAR % 10>(acc, A + m0, lda, B + n0, ldb, K, lds);
    else if (VAR >= 100) gemm_tile_ablate<VAR - 100>(acc, A + m0, lda, B + n0, ldb, K, lds);
    if (VAR == 5) gemm_tile_ring<4, 16, 3>(acc, A + m0, lda, B + n0, ldb, 0, K, lds);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)(m0 + acc_m(i)) + (long)(n0 + acc_n(j, r)) * ldc] = acc.v[i][j][r];
}

__global__ void checksum_kernel(const double* C, long n, unsigned long long* out) {
    unsigned long long h = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        h += (unsigned long long)__double_as_longlong(C[i]) * (unsigned long long)(2 * i + 1);
    atomicAdd(out, h);
}

static int g_hot = 0;
static int* g_cnt = nullptr;
template <int VAR>
static void run(int M, int N, int K, int reps, const double* dA, const double* dB, double* dC, int lds_bytes, const char* name) {
    hipFuncSetAttribute((const void*)probe_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int nt = (M / 128) * (N / 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(dC, 0, (size_t)M * N * 8);
    hipLaunchKernelGGL(probe_kernel<VAR>, dim3(nt), dim3(256), lds_bytes, 0, dA, (long)M, dB, (long)N, dC, (long)M, M, N, K, g_hot, g_cnt);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(probe_kernel<VAR>, dim3(nt), dim3(256), lds_bytes, 0, dA, (long)M, dB, (long)N, dC, (long)M, M, N, K, g_hot, g_cnt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    unsigned long long* dh; hipMalloc(&dh, 8); hipMemset(dh, 0, 8);
    hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, 0, dC, (long)M * N, dh);
    unsigned long long h; hipMemcpy(&h, dh, 8, hipMemcpyDeviceToHost); hipFree(dh);
    printf("%-34s M=%d N=%d K=%d: %8.3f ms  %6.2f TFLOP/s  checksum %016llx  (%s)\n", name, M, N, K, ms, 2.0 * M * N * K / ms * 1e-9, h,
           hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    const int reps = argc > 4 ? atoi(argv[4]) : 2;
    g_hot = getenv("L2HOT") ? 1 : 0;
    hipMalloc(&g_cnt, 2048 * 4); hipMemset(g_cnt, 0, 2048 * 4);
    double *dA, *dB, *dC;
    hipMalloc(&dA, (size_t)M * K * 8); hipMalloc(&dB, (size_t)N * K * 8); hipMalloc(&dC, (size_t)M * N * 8);
    std::vector<double> h((size_t)1 << 22);
    for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
    for (size_t off = 0; off < (size_t)M * K; off += h.size()) hipMemcpy(dA + off, h.data(), std::min(h.size(), (size_t)M * K - off) * 8, hipMemcpyHostToDevice);
    for (size_t off = 0; off < (size_t)N * K; off += h.size()) hipMemcpy(dB + off, h.data(), std::min(h.size(), (size_t)N * K - off) * 8, hipMemcpyHostToDevice);
    const int stage16 = 2 * 16 * GEMM_LDS_MC_LD * 8, stage8 = stage16 / 2;
    if (const char* o = getenv("ONLY")) {   // one variant (PMC runs)
        const int v = atoi(o);
        if (v == 0) run<0>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "gemm_tile");
        if (v == 40) run<40>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "gemm_tile_mc");
        if (v == 1021) run<1021>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2");
        if (v == 104) run<104>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate: no global loads");
        if (v == 108) run<108>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate: no LDS reads");
        if (v == 115) run<115>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate: MFMA only");
        return 0;
    }
    run<0>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "gemm_tile (BK16 x 2, syncthreads)");
    if (!getenv("SHORT")) {
    run<2>(M, N, K, reps, dA, dB, dC, 2 * stage16, "ring BK16 x 2 (bare barrier)");
    run<1>(M, N, K, reps, dA, dB, dC, 4 * stage8, "ring BK8 x 4");
    run<6>(M, N, K, reps, dA, dB, dC, 4 * stage8, "pipelined ring BK8 x 4");
    run<3>(M, N, K, reps, dA, dB, dC, 3 * stage8, "ring BK8 x 3");
    run<5>(M, N, K, reps, dA, dB, dC, 3 * stage16, "ring BK16 x 3 (1 WG/CU)");
    run<4>(M, N, K, reps, dA, dB, dC, 4 * stage16, "ring BK16 x 4 (1 WG/CU)");
    }
    if (getenv("SPREADV")) {
        run<20>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread: 1 load / 2 MFMAs");
        run<21>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread: 1 load / 4 MFMAs");
        run<22>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread: 1 load / 1 MFMA");
        run<23>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread: 1 load / 6 MFMAs");
        run<30>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2 (saddr, peeled): 1 / 2");
        run<31>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2 (saddr, peeled): 1 / 4");
        run<32>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2 (saddr, peeled): 1 / 1");
        run<40>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "gemm_tile_mc (saddr asm)");
        run<0>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "gemm_tile again");
    }
    if (getenv("PRIO")) {
        run<40>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "gemm_tile_mc");
        run<41>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "gemm_tile_mc<EVEN> (2 slabs / iteration)");
        run<50>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "gemm_tile_mc, CU partner at prio 3");
        run<53>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "gemm_tile_mc, CU partner at prio 1");
        run<51>(M, N, K, reps, dA, dB, dC, 160 * 1024, "gemm_tile_mc, 1 WG/CU");
        run<41>(M, N, K, reps, dA, dB, dC, 160 * 1024, "gemm_tile_mc<EVEN>, 1 WG/CU");
        run<52>(M, N, K, reps, dA, dB, dC, 160 * 1024, "MFMA only, 1 WG/CU");
        run<115>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "MFMA only, 2 WG/CU");
        run<40>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "gemm_tile_mc again");
    }
    if (getenv("SWEEP")) {
        run<1021>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2 read@2 barrier@1 (base)");
        run<1011>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2 read@1 barrier@1");
        run<1001>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2 read@0 barrier@1");
        run<1020>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2 read@2 barrier@0");
        run<1010>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2 read@1 barrier@0");
        run<1000>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2 read@0 barrier@0");
        run<1022>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2 read@2 barrier@2");
        run<1012>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2 read@1 barrier@2");
        run<1021>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "spread2 read@2 barrier@1 (again)");
    }
    if (getenv("ABLATE")) {
        run<100>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate 0 (= shipped loop)");
        run<101>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate: no vmcnt wait");
        run<102>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate: no barrier");
        run<103>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate: no wait, no barrier");
        run<104>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate: no global loads");
        run<108>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate: no LDS reads");
        run<112>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate: no loads, no LDS reads");
        run<106>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate: no loads, no barrier");
        run<115>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate: MFMA only");
        run<131>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "MFMA only + 24 v_add_u32 / slab");
        run<147>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "MFMA only + 24 v_lshl_add_u64 / slab");
        run<179>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "MFMA only + 24 s_add_u32 / slab");
        run<115>(M, N, K, reps, dA, dB, dC, GEMM_LDS_BYTES, "ablate: MFMA only (again)");
    }
    return 0;
}
