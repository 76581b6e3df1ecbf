// LDS-resident Cholesky + triangular inverse of a (<=) 128x128 SPD block: device code shared by chol_diag_kernel
// (kernels_chol.hip) and the fused small-problem MAP kernel (kernels_small.hip).
#pragma once
#include <type_traits>

#include "gemm_f64.hpp"
#include "wave_reduce.hpp"

namespace slsk {

#ifdef SLS_DIAG_TIMING
#define DIAG_STAMP(slot) do { if (threadIdx.x == 0 && info) ((long long*)info)[slot] = clock64(); } while (0)
#define DIAG_STAMP_T(t_, slot) do { if (threadIdx.x == (t_) && info) ((long long*)info)[slot] = clock64(); } while (0)
#else
#define DIAG_STAMP_T(t_, slot) do {} while (0)
#define DIAG_STAMP(slot) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------------
// Diagonal block: Cholesky + inverse of one 128x128 block by ONE workgroup (4 waves), matrix resident in LDS.
//   LDS: As[128 x 144] (column-major, ld 144 -> MFMA fragment reads conflict-free) + Ts[8][16 x 16] = exactly 160 KiB.
//   Phase 1 (factor), per 16-column panel:  wave 0 factors the 16x16 diagonal tile in registers (lane = row, pivots and
//     columns broadcast with wave shuffles) and inverts it; all waves then form the panel L = A T16^T and the trailing
//     update A_ij -= L_i L_j^T with v_mfma_f64_16x16x4 on 16x16 tiles read straight from LDS.
//   The inverse is built block row by block row WHILE wave 0 is busy with the next diagonal tile: waves 1-3 form
//     S = L[kb][:kb] Linv[:kb][:kb] during the pivot chain of step kb and T[kb][j] = -T16 S joins the panel phase; the tiles
//     go to the unused strictly-upper part of As (as Linv^T), so no LDS beyond the matrix itself is needed.
// The serial chain is 128 pivot steps of ~one rsqrt + one readlane each instead of 256 barrier-separated steps.
// ---------------------------------------------------------------------------------------------------------
constexpr int DL = 144;
constexpr int DIAG_LDS_BYTES = (128 * DL + 8 * 256) * 8;   // 163840

__device__ __forceinline__ d4_t mfma16(double bfrag, double afrag, d4_t acc) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(bfrag, afrag, acc, 0, 0, 0);
}

// value of lane k of each 16-lane row in every lane of that row: one v_mov_b64_dpp row_newbcast:k (the only DPP control
// gfx90a+ allows on 64-bit operands).  The diagonal-tile code keeps row i of the tile in lanes i, 16+i, 32+i, 48+i, so
// this is a wave-wide broadcast; it sits on the serial pivot chain, where the alternatives cost two v_readlane + SGPR
// pressure (the unrolled code spilled 146 SGPRs) or a ~100-cycle ds_bpermute.  k is a constant after unrolling.
__device__ __forceinline__ double bcast(double x, int k) {
    const long v = __builtin_bit_cast(long, x);
    long r = v;
    switch (k) {
#define SLS_BC(K) case K: r = __builtin_amdgcn_update_dpp(0L, v, 0x150 + K, 0xf, 0xf, true); break;
        SLS_BC(0) SLS_BC(1) SLS_BC(2) SLS_BC(3) SLS_BC(4) SLS_BC(5) SLS_BC(6) SLS_BC(7)
        SLS_BC(8) SLS_BC(9) SLS_BC(10) SLS_BC(11) SLS_BC(12) SLS_BC(13) SLS_BC(14) SLS_BC(15)
#undef SLS_BC
        default: break;
    }
    return __builtin_bit_cast(double, r);
}

// rsqrt for the pivots of diag16 below.
// 1/sqrt(d) to full fp64 accuracy: hardware v_rsq_f64 seed (measured max rel. error 5.2e-8) + one third-order Newton step
// on the residual (-> 1.4e-16, identical to a second step; diag_timing.hip).  The correctly-rounded sqrt()/division pair
// of the math library costs ~350 dependent cycles on the pivot chain.
__device__ __forceinline__ double rsqrt_nr(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);          // 1 - d y^2
    y = fma(y * e, fma(0.375, e, 0.5), y);          // y (1 + e/2 + 3 e^2/8)
    return y;
}

// workgroup barrier that waits for this wave's LDS traffic only (not for outstanding global stores)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16x16 diagonal tile at (c0, c0): optional in-register Cholesky, then inverse.  Executed by one full wave; lane & 15 = row.
// Measured (tools/probes/diag_timing, cycles per call on MI355X): 4650 as two consecutive loops with a branch per pivot for
// the failure report; 4400 in this form (one basic block: step k of the inverse issued right behind pivot k, failure recorded
// in a register and reported once); 4850 with the running diagonal kept in every lane (pivot one fma behind the broadcast
// instead of fma -> broadcast); 5300 with v_fmac_f64_dpp fusing broadcast + fma.  The wave is ISSUE-bound (~660 fp64 / DPP
// instructions at ~7 cycles), not latency-bound: shortening the dependency chain buys nothing, fewer instructions would.
// Also measured and not kept (round 3): wave 0 as a pure pivot wave (only its own panel tile + the next diagonal tile's update
// from registers, waves 1-3 meeting on an LDS counter instead of the second barrier): 57 500 vs 57 900 cycles per block --
// waves 1-3 (trailing update ~600 cycles per 16 x 16 tile, S tiles) are then the longer path; three tiles in flight per wave
// did not shorten them (not latency-bound either), and a k-interleaved S phase compiled into a branch per MFMA (80 000).
// Round 5 form of the factoring diagonal tile.  Former form: every lane keeps its whole row (16 columns, the same in all four
// 16-lane groups) and every pivot updates all later columns with a broadcast + fma pair each: ~660 instructions on the serial path
// of a 16-column step.  Now only the CURRENT group of four columns is kept that way (cur[]); the later columns are DEALT to the
// four lane groups -- lane (row, g) keeps columns g + 4 q -- which is the operand layout of v_mfma_f64_16x16x4: after the four
// pivots of a group the rank-4 update of all later columns is ONE matrix instruction with the group's columns as both operands
// (every lane receives the updates of exactly its own columns).  The next group's four columns reach all lanes through the LDS
// tile itself (one dealt store, four reads).  Values above the diagonal are garbage exactly as before and never reach a result
// (an output of the matrix instruction depends on row m and row n of its operands only).  The factor differs from the former
// one in the last bits (four rank-1 terms are summed before they are subtracted).
// (First attempt, measured and dropped: all 16 columns dealt and the pivot column sent to all four lane groups with
// v_permlane16/32_swap -- 12 instructions per pivot with the copies the swaps need: 4600 cycles against the former 4400.)
__device__ __forceinline__ void diag16_dealt(double* As, double* Tk, int c0, int lane, int* info, int global_off) {
    const int row = lane & 15, g = lane >> 4;
    double cur[4], sl[4], bq[4];
    double own_inv = 1.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        cur[r] = As[c0 + row + (c0 + r) * DL];
        sl[r] = r > 0 ? As[c0 + row + (c0 + 4 * r + g) * DL] : 0.0;
        bq[r] = (row == 4 * r + g) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int G = 0; G < 4; ++G) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = 4 * G + c;
            const double d = bcast(cur[c], k);              // the pivot
            const double inv = rsqrt_nr(d);
            const double lik = cur[c] * inv;                // L[row][k] (rows < k: unused values)
            cur[c] = lik;
#pragma unroll
            for (int c2 = c + 1; c2 < 4; ++c2) cur[c2] -= lik * bcast(lik, 4 * G + c2);
            own_inv = (row == k) ? inv : own_inv;
            const double ck = (row > k) ? lik * inv : 0.0;
#pragma unroll
            for (int r = 0; r <= k / 4; ++r) bq[r] = fma(-ck, bcast(bq[r], k), bq[r]);
        }
        // lane (row, g) takes column 4 G + g of the finished group: the matrix instruction's operand and what goes back to LDS
        double op = cur[0];
        op = (g == 1) ? cur[1] : op;
        op = (g == 2) ? cur[2] : op;
        op = (g == 3) ? cur[3] : op;
        As[c0 + row + (c0 + 4 * G + g) * DL] = (4 * G + g <= row) ? op : 0.0;
        if (G < 3) {
            d4_t u = {0.0, 0.0, 0.0, 0.0};
            u = mfma16(op, op, u);                          // u[q] at lane (row, g): sum_k L[row][4G+k] L[4q+g][4G+k]
#pragma unroll
            for (int q = G + 1; q < 4; ++q) sl[q] -= u[q];
            // the next group's columns to every lane, through the tile's own LDS image (DS operations of a wave execute in order)
            As[c0 + row + (c0 + 4 * (G + 1) + g) * DL] = sl[G + 1];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int c = 0; c < 4; ++c) cur[c] = As[c0 + row + (c0 + 4 * (G + 1) + c) * DL];
        }
    }
    // The failure report is taken off the pivot chain (the wave is alone on its SIMD: every instruction there is paid in full --
    // three per pivot for a flag that is almost never set): a pivot that is not positive makes its own L_kk and everything behind
    // it NaN (rsq of a non-positive number times that number), so the first row whose L_rr, read back from the finished tile, is
    // not positive IS the first failed pivot.  (1 / L_rr for the inverse stays the chain's own 1 / sqrt(d): taking it from the tile
    // as well changed last bits that the ill-conditioned sigma-gradient cases of the test-suite amplify to their tolerance.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const double lrr = As[c0 + row + (c0 + row) * DL];
    const unsigned long long failed = __ballot(!(lrr > 0.0)) & 0xffffull;   // lanes 0 .. 15: rows 0 .. 15
    if (failed && lane == 0 && info) atomicCAS(info, 0, global_off + c0 + (__ffsll((long long)failed) - 1) + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col = 4 * r + g;
        Tk[row + 16 * col] = (col <= row) ? bq[r] * own_inv : 0.0;
    }
}

template <bool FACTOR>
__device__ __forceinline__ void diag16(double* As, double* Tk, int c0, int lane, int* info, int global_off) {
    if constexpr (FACTOR) {
        diag16_dealt(As, Tk, c0, lane, info, global_off);
        return;
    }
    const int row = lane & 15, g = lane >> 4;
    double a[16];
    double own_inv = 1.0;
    int bad = 16;                                   // first non-positive pivot of this tile (16: none)
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = As[c0 + row + (c0 + j) * DL];
    // inverse: B = I; for k: B[i,:] -= (L[i,k] / L_kk) B[k,:] (i > k); finally B[i,:] /= L_ii.  Row k of B is final before step
    // k, so its scaling waits until the end (no select on the chain).  The columns of B are independent and are dealt to
    // the four 16-lane rows of the wave: lane (row, g) keeps columns g, g+4, g+8, g+12, the broadcast of row k stays inside
    // each 16-lane row, and step k costs k/4 + 1 broadcast + fma pairs instead of k + 1 (columns > k see B[k,j] = 0).
    double bq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bq[r] = (row == 4 * r + g) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double d = bcast(a[k], k);
        double inv;
        if (FACTOR) {
            bad = (!(d > 0.0) && bad == 16) ? k : bad;      // off the chain: reported once, after the loop
            inv = rsqrt_nr(d);
            const double lik = a[k] * inv;   // lane k: d * rsqrt(d) = L_kk; rows < k hold unused upper-triangle values
            a[k] = lik;
#pragma unroll
            for (int j = k + 1; j < 16; ++j) a[j] -= lik * bcast(lik, j);
        } else {
            inv = 1.0 / d;
        }
        own_inv = (row == k) ? inv : own_inv;
        const double ck = (row > k) ? a[k] * inv : 0.0;
#pragma unroll
        for (int r = 0; r <= k / 4; ++r) bq[r] = fma(-ck, bcast(bq[r], k), bq[r]);
    }
    if (FACTOR && bad < 16 && lane == 0 && info) atomicCAS(info, 0, global_off + c0 + bad + 1);
    if (FACTOR && lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) As[c0 + row + (c0 + j) * DL] = (j <= row) ? a[j] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col = 4 * r + g;
        Tk[row + 16 * col] = (col <= row) ? bq[r] * own_inv : 0.0;
    }
}

// Linv^T tiles in the strictly-upper part of As -> natural strictly-lower positions: the 28 tiles are dealt 7 per wave
// (tile t -> wave t & 3) with compile-time coordinates (one branch on the wave id instead of 28), reads before writes.
// Each 16-lane group walks a wrapped diagonal of the tile (column fl, row fl + fk + 4q), so both the row-major reads and
// the column-major writes touch 16 different banks (a straight transpose makes one side a 16-way conflict).
template <int W>
__device__ __forceinline__ void transpose_inverse_tiles(double* As, int fl, int fk) {
    double tv[7][4];
#pragma unroll
    for (int n = 0; n < 7; ++n) {
        const int t = 4 * n + W;
        const int ti = t < 1 ? 1 : t < 3 ? 2 : t < 6 ? 3 : t < 10 ? 4 : t < 15 ? 5 : t < 21 ? 6 : 7;
        const int tj = t - ti * (ti - 1) / 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) tv[n][q] = As[(16 * ti + ((fl + fk + 4 * q) & 15)) * DL + 16 * tj + fl];
    }
#pragma unroll
    for (int n = 0; n < 7; ++n) {
        const int t = 4 * n + W;
        const int ti = t < 1 ? 1 : t < 3 ? 2 : t < 6 ? 3 : t < 10 ? 4 : t < 15 ? 5 : t < 21 ? 6 : 7;
        const int tj = t - ti * (ti - 1) / 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) As[(16 * tj + fl) * DL + 16 * ti + ((fl + fk + 4 * q) & 15)] = tv[n][q];
    }
}

// The 16-column steps of the LDS-resident factorisation + inverse (see chol_diag_kernel), restricted to the leading nb16
// 16x16 block rows/columns (nb16 = 8: the whole 128x128 block; smaller for the fused small-problem kernels, whose
// identity padding needs no work).  On return (after the caller's barrier) the lower triangle of As holds L, the
// strictly-upper tiles hold Linv^T and Ts the inverses of the diagonal tiles.
// HAVE_T16 (with !FACTOR): Ts already holds the inverses of the diagonal tiles (the caller loaded the ones the factorisation
// produced); wave 0 has no pivot chain to run and the block inverse is built around exactly those tiles.
// idle0: work of the caller's that needs neither the image nor a barrier, run by waves 1-3 during the pivot chain of the FIRST
// diagonal tile, when they have no S tiles to form (map_opt_kernel: the Bradley-Terry-Luce terms of the trial point).
struct NoIdleWork {
    __device__ __forceinline__ void operator()() const {}
};
template <bool FACTOR, bool HAVE_T16 = false, class Idle = NoIdleWork>
__device__ __forceinline__ void chol_diag_steps(double* As, double* Ts, int* __restrict__ info, int global_off, int nb16, Idle&& idle0 = Idle{}) {
    static_assert(!(FACTOR && HAVE_T16), "the factorisation produces the 16 x 16 inverses itself");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, fk = lane >> 4;
    // Per 16-column step kb (three barrier-separated phases; the strictly-upper tiles of As, unused by the factorisation,
    // receive the inverse: slot (j, kb) holds T[kb][j] transposed, i.e. the upper triangle of As becomes Linv^T):
    //   A  wave 0: Cholesky + inverse of the 16x16 diagonal tile (serial pivot chain, the critical path);
    //      waves 1-3, meanwhile: S[kb][j] = sum_{j <= k < kb} L[kb][k] T[k][j] for j < kb  (block row kb of -T16^-1 Linv)
    //   B  T[kb][j] = -T16 S[kb][j]  and (FACTOR) the panel L_r = A_r T16^T, r > kb: 7 tiles over the 4 waves
    //   C  (FACTOR) trailing update A_ij -= L_i L_j^T, 7 >= i >= j > kb
    for (int kb = 0; kb < nb16; ++kb) {
        const int c0 = 16 * kb;
#ifdef SLS_DIAG_TIMING
        long long t_a = clock64();
#endif
        DIAG_STAMP_T(0, 16 + 8 * kb + 0);
        if (wave == 0) {
            if constexpr (!HAVE_T16) diag16<FACTOR>(As, Ts + 256 * kb, c0, lane, info, global_off);
            DIAG_STAMP_T(0, 16 + 8 * kb + 1);
        } else {
            if (kb == 0) idle0();
            // tile j costs kb - j MFMA groups: waves 1..3 take j = {0, 5, 6}, {1, 4}, {2, 3} (10 / 9 / 9 groups at kb = 7)
            const int js[3] = {wave - 1, 6 - wave, wave == 1 ? 6 : 8};
            for (int n = 0; n < 3; ++n) {
                const int j = js[n];
                if (j >= kb) continue;
                d4_t c = {0.0, 0.0, 0.0, 0.0};
                for (int k = j; k < kb; ++k) {
                    // the eight fragment reads of a block issued together, then the four products (round 5: the scheduler paired
                    // every read with its use: read - wait - product, four times over; the sums keep their order)
                    double af[4], bf[4];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int kq = 4 * kk + fk;
                        af[kk] = As[(16 * k + kq) * DL + c0 + fl];   // L[kb][k] (row m = fl, col kq)
                        // T[k][j] (row kq, col n = fl): diagonal inverse in Ts for k == j, transposed upper slot otherwise
                        bf[kk] = (k == j) ? Ts[256 * j + kq + 16 * fl] : As[(16 * k + kq) * DL + 16 * j + fl];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) c = mfma16(af[kk], bf[kk], c);   // operands swapped: lane holds S (m = fk + 4q, n = fl), stores run along n
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) As[(c0 + fk + 4 * q) * DL + 16 * j + fl] = c[q];   // S -> slot (j, kb)
            }
            DIAG_STAMP_T(64, 16 + 8 * kb + 5);
        }
        __syncthreads();
        DIAG_STAMP_T(0, 16 + 8 * kb + 2);
#ifdef SLS_DIAG_TIMING
        if (tid == 0 && kb == 0 && info) ((long long*)info)[7] = clock64() - t_a;
#endif
        const double* Tk = Ts + 256 * kb;
        for (int t = wave; t < nb16 - 1; t += 4) {
            if (t < kb) {
                const int j = t;
                d4_t c = {0.0, 0.0, 0.0, 0.0};
                double af[4], bf[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int kq = 4 * kk + fk;
                    af[kk] = -Tk[fl + 16 * kq];                        // -T16 (row m = fl, col kq)
                    bf[kk] = As[(c0 + kq) * DL + 16 * j + fl];         // S (row kq, col n = fl)
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) c = mfma16(af[kk], bf[kk], c);   // swapped as for S
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) As[(c0 + fk + 4 * q) * DL + 16 * j + fl] = c[q];
            } else if (FACTOR) {
                const int r = t + 1;   // panel: L_r = A_r T16^T
                d4_t acc = {0.0, 0.0, 0.0, 0.0};
                double af[4], bf[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = 4 * kk + fk;
                    af[kk] = As[(c0 + k) * DL + 16 * r + fl];   // A_r[m = fl][k]
                    bf[kk] = Tk[fl + 16 * k];                    // T[n = fl][k]
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = mfma16(bf[kk], af[kk], acc);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) As[(c0 + fk + 4 * q) * DL + 16 * r + fl] = acc[q];
            }
        }
        __syncthreads();
        DIAG_STAMP_T(0, 16 + 8 * kb + 3);
        if (!FACTOR) continue;
        // trailing update: A_ij -= L_i L_j^T for 7 >= i >= j > kb.  Wave 0 takes only the next diagonal tile and runs straight
        // on into its pivot chain (no barrier here: nobody else touches that tile); waves 1-3 share the other tiles and then
        // the S tiles of the next step, all hidden behind the chain.
        auto update_tile = [&](int i, int j) {
            d4_t acc;
            double af[4], bf[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = As[(16 * j + fk + 4 * q) * DL + 16 * i + fl];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = c0 + 4 * kk + fk;
                af[kk] = -As[k * DL + 16 * i + fl];
                bf[kk] = As[k * DL + 16 * j + fl];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = mfma16(bf[kk], af[kk], acc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) As[(16 * j + fk + 4 * q) * DL + 16 * i + fl] = acc[q];
        };
        if (wave == 0) {
            if (kb < nb16 - 1) update_tile(kb + 1, kb + 1);
        } else {
            // tiles in column-major order after the first, (i, j) advanced 3 at a time
            int i = kb + 1, j = kb + 1;
            auto advance = [&](int n) {
                for (; n > 0; --n) {
                    if (++i == nb16) { ++j; i = j; }
                }
            };
            advance(wave);
            while (j < nb16) {
                update_tile(i, j);
                advance(3);
            }
        }
        DIAG_STAMP_T(0, 16 + 8 * kb + 4);
        DIAG_STAMP_T(64, 16 + 8 * kb + 6);
    }
}

// Factor ONLY: the Cholesky factor of the 128 x 128 LDS image (lower tiles of As) and the inverses of its eight 16 x 16 diagonal
// tiles (Ts); the block inverse is NOT built (no S / T phases).  For callers that solve against L_jj with the small inverses
// instead of multiplying by T_jj (the dataflow Cholesky's panel tiles), the block inverse leaves the serial chain altogether
// and is formed afterwards, for all diagonal blocks at once.
// With the inverse gone waves 1-3 only carry panel and update tiles, and wave 0 can be a pure PIVOT wave: per step it takes
// the one panel tile it needs itself, L_{kb+1,kb}, applies it to the next diagonal tile straight from its accumulators (the
// accumulator layout is the MFMA operand layout) and runs on into the next pivot chain -- one workgroup barrier per step on
// its path instead of two.  Waves 1-3 meet each other and wave 0's tile on an LDS counter (s_barrier would need wave 0).
// (The same restructuring WITH the inverse phases gained nothing: waves 1-3 were then the longer path, see diag16.)
// Same operations per element as chol_diag_steps<true>: L and the 16 x 16 inverses have the same bits.
// publish(s): optional hook run by waves 1-3 (uniformly, 192 threads) as soon as column block s of L and T16_s are final --
// right behind the panel tiles of step s (the two-workgroup chain of the dataflow Cholesky streams them to the workgroup that
// solves the next panel tile while this one is still factoring); the last block (s = 7) is published behind the last barrier.
struct NoPublish {
    __device__ __forceinline__ void operator()(int) const {}
};
template <class Publish = NoPublish>
__device__ __forceinline__ void chol_factor_steps(double* As, double* Ts, int* __restrict__ info, int global_off,
                                                  Publish&& publish = NoPublish{}) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, fk = lane >> 4;
    constexpr int nb16 = 8;
    // arrival counter in the padding rows of the last LDS column (rows 128 .. 143 of a column are never part of the matrix)
    int* sync_word = reinterpret_cast<int*>(As + 127 * DL + 128);
    if (tid == 0) *sync_word = 0;                    // visible to everybody behind the first step's barrier
    for (int kb = 0; kb < nb16; ++kb) {
        const int c0 = 16 * kb;
        if (wave == 0) diag16<true>(As, Ts + 256 * kb, c0, lane, info, global_off);
        __syncthreads();                              // T16(kb), L16(kb) visible; waves 1-3 have finished the update of step kb - 1
        if (kb == nb16 - 1) {
            if (wave != 0) publish(kb);
            break;
        }
        const double* Tk = Ts + 256 * kb;
        auto panel_tile = [&](int r) {                // L_r = A_r T16^T, in place
            d4_t acc = {0.0, 0.0, 0.0, 0.0};
            double af[4], bf[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = 4 * kk + fk;
                af[kk] = As[(c0 + k) * DL + 16 * r + fl];   // A_r[m = fl][k]
                bf[kk] = Tk[fl + 16 * k];                    // T[n = fl][k]
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = mfma16(bf[kk], af[kk], acc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) As[(c0 + fk + 4 * q) * DL + 16 * r + fl] = acc[q];
            return acc;
        };
        if (wave == 0) {
            const d4_t p = panel_tile(kb + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the tile is in LDS
            if (lane == 0) __hip_atomic_fetch_add(sync_word, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int d = kb + 1;
            d4_t acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = As[(16 * d + fk + 4 * q) * DL + 16 * d + fl];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = mfma16(p[kk], -p[kk], acc);   // bf = L[n = fl][k], af = -L[m = fl][k]: both p[kk]
#pragma unroll
            for (int q = 0; q < 4; ++q) As[(16 * d + fk + 4 * q) * DL + 16 * d + fl] = acc[q];
        } else {
            for (int r = kb + 1 + wave; r < nb16; r += 3) panel_tile(r);       // rows kb + 2 .. 7
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(sync_word, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(sync_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * (kb + 1)) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            publish(kb);                                  // column block kb of L is final
            // trailing update A_ij -= L_i L_j^T, 7 >= i >= j > kb, tiles in column-major order after the first (wave 0's)
            int i = kb + 1, j = kb + 1;
            auto advance = [&](int n) {
                for (; n > 0; --n) {
                    if (++i == nb16) { ++j; i = j; }
                }
            };
            advance(wave);
            while (j < nb16) {
                d4_t acc;
                double af[4], bf[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = As[(16 * j + fk + 4 * q) * DL + 16 * i + fl];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = c0 + 4 * kk + fk;
                    af[kk] = -As[k * DL + 16 * i + fl];
                    bf[kk] = As[k * DL + 16 * j + fl];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = mfma16(bf[kk], af[kk], acc);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) As[(16 * j + fk + 4 * q) * DL + 16 * i + fl] = acc[q];
                advance(3);
            }
        }
    }
}

}  // namespace slsk
