// Dense SPD factorisation on the fp64 MFMA tile: blocked Cholesky (Eigen::LLT stand-in,
// src/preference-regressor.cpp:162,290), triangular inverse and K^-1 = L^-T L^-1 (replaces MatrixXd::inverse(),
// src/gaussian-process-regressor.cpp:159,211,231), block triangular solves (LLT::solve), GEMV helpers.
//
// Right-looking, NB = 128:
//   chol_diag   one workgroup factors the 128x128 diagonal block resident in LDS (16-column steps: register Cholesky of the
//               16x16 tile on one wave, MFMA panel + trailing update) and produces its inverse T_jj on the way;
//   panel       L_ij = A_ij T_jj^T            (MFMA GEMM, in place)
//   syrk        A_ik -= L_ij L_kj^T, i>=k>j   (MFMA GEMM, lower tiles only)
// Triangular inverse: recursive doubling over block pairs, X21 = -X22 (L21 X11), every level two batched GEMMs.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>

#include "chol_diag.hpp"
#include "gemm_f64.hpp"
#include "kernels.hpp"

#ifndef SLS_POTRF_MODE_DEFAULT
#define SLS_POTRF_MODE_DEFAULT 3
#endif
#ifndef SLS_POTRF_STREAM_DEFAULT
#define SLS_POTRF_STREAM_DEFAULT 1
#endif

namespace slsk {

typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

// Cholesky factor + inverse of ONE 128 x 128 diagonal block by the calling workgroup (256 threads, `smem` = DIAG_LDS_BYTES of
// LDS): A (lower triangle read) -> L in place (FACTOR) and T = L^-1 into Tout.  Shared by the one-block launch
// (chol_diag_kernel) and the single-launch persistent factorisation (potrf_persistent_kernel).
// LOAD = false: the caller has already built the LDS image As (lower tiles of A, upper tiles zero) and passed a barrier.
// WT: write-through (sc1) stores for L and T -- the caller publishes them with a drained flag instead of a release fence.
// HAVE_T16 (!FACTOR, LOAD): the diagonal 16 x 16 tiles of Tout already hold the inverses of L's diagonal tiles (written by the
// factorisation, and possibly still being read by its panel solves): they are loaded, not recomputed, and not stored again.
template <bool FACTOR, bool LOAD = true, bool WT = false, bool HAVE_T16 = false>
__device__ __forceinline__ void diag_block(double* __restrict__ A, long lda, double* __restrict__ Tout, long ldt,
                                           int* __restrict__ info, int global_off, char* smem) {
    double* As = reinterpret_cast<double*>(smem);   // [i + j*DL]
    double* Ts = As + 128 * DL;                     // [tile][i + 16 j]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): the role branches below become scalar branches
    const int fl = lane & 15, fk = lane >> 4;       // fragment index / k sub-index

    DIAG_STAMP(1);
    if (LOAD) {   // thread -> (row pair, column): 64 threads cover one column with 16-byte loads, 4 columns per pass
        const int i2 = 2 * (tid & 63), jc = tid >> 6;
#pragma unroll 8
        for (int p = 0; p < 32; ++p) {
            const int j = 4 * p + jc;
            d2_t v = *reinterpret_cast<const d2_t*>(A + (long)i2 + (long)j * lda);
            if ((i2 >> 4) < (j >> 4)) v = d2_t{0.0, 0.0};
            *reinterpret_cast<d2_t*>(As + i2 + j * DL) = v;
        }
        if (HAVE_T16) {
            for (int q2 = tid; q2 < 1024; q2 += 256) {            // 8 tiles x 16 columns x 8 row pairs
                const int t16i = q2 >> 7, c = (q2 >> 3) & 15, r2 = 2 * (q2 & 7);
                *reinterpret_cast<d2_t*>(Ts + 256 * t16i + r2 + 16 * c) =
                    *reinterpret_cast<const d2_t*>(Tout + (16 * t16i + r2) + (long)(16 * t16i + c) * ldt);
            }
        }
        __syncthreads();
    }
    DIAG_STAMP(2);
    auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(A, 0, 0x7fffffff, 0x00020000);
    auto rsrcT = __builtin_amdgcn_make_buffer_rsrc(Tout, 0, 0x7fffffff, 0x00020000);

    chol_diag_steps<FACTOR, HAVE_T16>(As, Ts, info, global_off, 8);
    __syncthreads();
    DIAG_STAMP(3);
#ifdef SLS_DIAG_SKIP_STOREL
    if (false) {
#else
    if (FACTOR) {
#endif
        const int i2 = 2 * (tid & 63), jc = tid >> 6;
#pragma unroll 8
        for (int p = 0; p < 32; ++p) {
            const int j = 4 * p + jc;
            d2_t v = *reinterpret_cast<const d2_t*>(As + i2 + j * DL);
            if (i2 < j) v[0] = 0.0;
            if (i2 + 1 < j) v[1] = 0.0;
            if (WT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrcA, (int)((i2 + (long)j * lda) * 8), 0, 16);
            else *reinterpret_cast<d2_t*>(A + (long)i2 + (long)j * lda) = v;
        }
    }
    DIAG_STAMP(4);
    // inverse out: T[ti][tj] sits transposed in the upper slot (tj, ti).  L has left the lower triangle (FACTOR) or is not
    // written back at all (!FACTOR), so the tiles are first transposed LDS -> LDS into their natural lower positions and
    // then leave with the same coalesced 16-byte pattern as L.  The two barriers order LDS accesses only: a __syncthreads()
    // here would also drain the 128 KB of global stores of L just issued (~10k cycles).
    lds_barrier();
    DIAG_STAMP(8);
    if (wave == 0) transpose_inverse_tiles<0>(As, fl, fk);
    else if (wave == 1) transpose_inverse_tiles<1>(As, fl, fk);
    else if (wave == 2) transpose_inverse_tiles<2>(As, fl, fk);
    else transpose_inverse_tiles<3>(As, fl, fk);
    DIAG_STAMP(9);
    lds_barrier();
    DIAG_STAMP(10);
    {
        const int i2 = 2 * (tid & 63), jc = tid >> 6;
        const int ti = i2 >> 4;
#pragma unroll 8
        for (int p = 0; p < 32; ++p) {
            const int j = 4 * p + jc, tj = j >> 4;
            d2_t v = {0.0, 0.0};
            if (ti > tj) v = *reinterpret_cast<const d2_t*>(As + i2 + j * DL);
            else if (ti == tj) {
                if (HAVE_T16) continue;                           // already there, bit for bit
                v = *reinterpret_cast<const d2_t*>(Ts + 256 * ti + (i2 & 15) + 16 * (j & 15));
            }
            if (WT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrcT, (int)((i2 + (long)j * ldt) * 8), 0, 16);
            else *reinterpret_cast<d2_t*>(Tout + (long)i2 + (long)j * ldt) = v;
        }
    }
    DIAG_STAMP(6);
}

template <bool FACTOR>
__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ A, long lda, double* __restrict__ Tout, long ldt,
                                                        int* __restrict__ info, int global_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    diag_block<FACTOR>(A, lda, Tout, ldt, info, global_off, smem);
}

static void diag_attr() {
    ensure_dyn_lds((const void*)chol_diag_kernel<true>, DIAG_LDS_BYTES);
    ensure_dyn_lds((const void*)chol_diag_kernel<false>, DIAG_LDS_BYTES);
}
void launch_chol_diag(hipStream_t s, double* A, long lda, double* Tout, long ldt, int* info, int global_off) {
    diag_attr();
    hipLaunchKernelGGL(chol_diag_kernel<true>, dim3(1), dim3(256), DIAG_LDS_BYTES, s, A, lda, Tout, ldt, info, global_off);
}

// ---------------------------------------------------------------------------------------------------------
// Single-launch factorisation (dataflow form, below): arguments and the flag / wait primitives.
// Rounds 2-3 also carried a form with grid barriers (potrf_persistent_kernel) and a hybrid of persistent panel kernels with
// side-stream updates; both were superseded by the dataflow form (DESIGN.md 8a keeps their measurements) and were removed in
// round 4 together with the probe switches of the dataflow form that measured "no gain" (cyclic owner grid, products with T_jj
// instead of triangular solves, chain tiles through global memory, acquire / sleep variants).
// ---------------------------------------------------------------------------------------------------------
struct PersistArgs {
    double* A;
    long ld;
    int nb;
    double* Linv;
    int* info;        // [0] first non-positive pivot + 1, [1] abort
    int* sync;        // [1] set-up counter, [8 + x] workers resident on XCD x, [DF_FACT ..] the flag tables (see DF_FACT)
    int nbo;          // 128-columns per update chunk (1: every step is applied on its own)
    long long timeout;
    long long* trace;   // optional (probes): 16 wall-clock stamps per step of the chain, then 16 words per worker
    int near;           // columns at the start of an outer block that take the previous block step by step
    // several independent factorisations in ONE launch (the points of a DIRECT iteration of the MAP fit): problem q uses the
    // workgroups [q gridDim.x / nprob, (q + 1) gridDim.x / nprob) and A + q strideA, Linv + q strideA, sync + q stride_sync,
    // info + 2 q
    int nprob;
    long strideA;
    long stride_sync;
    // fused inverse (potri_team below; nprob == 1): workgroups [g1, gridDim.x) build X = L^-1 (Linv), U = X^T and K^-1 = U U^T
    // behind the factorisation, inside the same launch.  g1 = 0: factorisation only.
    int g1;
    double* U;
    double* Kinv;
    int inv_cx, inv_ck;   // 128-blocks of the contraction per X / K^-1 accumulation task (fixed chunking)
    int inv_plast;        // 1: the term of the row just above is split off (P_i, one product per row on the column wavefront); 0: every
                          // term is accumulated, then U_ji = M T_ii^T (two products per row, no P task: shorter tail for few rows)
    // streamed panel tiles (stream_trsm): nchain = 2: two chain workgroups take turns -- while one factors diagonal block j, the other
    // solves the panel tile (j+1, j) against the column blocks it publishes, then multiplies and factors block j+1; the workers'
    // panel tiles are solved the same way.  nchain = 1: the round-3 chain (one workgroup, one thing after the other).
    int nchain;
    // split_sub: the sub-diagonal tiles (k+1, k) -- whose last update sits between the chain's solve of step k-1 and its solve
    // of step k -- have one owner per 64-column half (gemm_tile_mc<2>: same slabs, same k order, same bits).  split_band: so do
    // the tiles (i, k) with 1 <= i - k <= split_band: every row runs the cycle "panel tile solved behind diagonal block k-1 ->
    // update of (i, k) with it -> panel tile (i, k) solved behind diagonal block k", which is as long as the chain's own step when
    // the update is a whole-tile product (14 + 3 + 21 us against 39): a row that falls behind once never catches up, and sooner
    // or later it is the sub-diagonal one.  The first half's owner solves the tile once the second half has reported (upd_done).
    int split_sub, split_band;
    // fused inverse, DYNAMIC pools (round 6): the items of both teams in ONE global order (the factorisation's tiles column by column,
    // then the inverse's items in potri_team's order), item n in pool n mod pool_nx = the workgroups of one XCD; any workgroup of the
    // pool claims the first item of the pool's list whose next task has its inputs (compare-and-swap on the item's state word) and
    // runs that ONE task.  pool_items: {type, i, k or j, half or initial progress} per item (host-built, cached per device);
    // pool_state: (progress << 2) | {0 free, 1 claimed, 2 done}, zeroed with the flag tables.  0 items: static ownership.
    const int4* pool_items;
    int pool_n, pool_nx, pool_keep;
    // the tiles within pool_near blocks of the diagonal (the chain's feeders: their tasks are the ones whose start latency the chain
    // feels) stay OUT of the pools: the first pool_near_w workgroups behind the chain own them statically and take nothing else
    // (-1 / 0: everything is pooled)
    int pool_near, pool_near_w;
    int* pool_state;
};
#define PK_STAMP(slot) do { if (a.trace && threadIdx.x == 0) a.trace[16 * j + (slot)] = wall_clock64(); } while (0)

__device__ __forceinline__ int df_flag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane: spin (RELAXED polls -- an acquire load at agent scope would invalidate the XCD's L2 on every poll) until
// *p >= target; false = aborted / timed out
__device__ __forceinline__ bool pk_spin(int* p, int target, int* abort_flag, long long timeout) {
    const long long t0 = wall_clock64();
    int n = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if ((++n & 63) == 0) {
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (wall_clock64() - t0 > timeout) {
                __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}
// this CU's vector L1 only (the L2 is handled once per XCD by the barrier leader).  `buffer_inv sc1`: the agent-scope form.  Round 2
// used `buffer_inv sc0` here, which is a WORKGROUP-scope invalidate and drops nothing another CU's stores could have made stale
// (MI355X_MICROARCH.md, "inter-workgroup visibility"): the barrier form then relied on its tiles' streaming reads evicting the
// 32 KB L1 by themselves -- true in every test, but not a guarantee, and the likely reason why the fused chain step of round 2
// (which re-read small, recently touched regions) gave build-dependent factors.  The dataflow form needs no such invalidate for
// its update tiles at all (one owner per tile) and takes a real agent-scope acquire before every task.
__device__ __forceinline__ void pk_inv_l1() { asm volatile("buffer_inv sc1" ::: "memory"); }

// wait until the counter *p reaches `target` (the chain watching the workers' barrier counter)
__device__ __forceinline__ bool pk_wait_count(int* p, int target, const PersistArgs& a) {
    int ok = 1;
    if ((threadIdx.x & 63) == 0) {
        ok = pk_spin(p, target, a.info + 1, a.timeout) ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ok = __builtin_amdgcn_readfirstlane(ok);
    __syncthreads();
    return ok != 0;
}
// The workers' and the chain's full 128 x 128 x K products: gemm_tile_mc (gemm_f64.hpp).  Round 2 first used a four-stage ring
// of BK = 16 slabs here (147 KB of LDS) -- whose __syncthreads() drained the ring with the compiler's s_waitcnt vmcnt(0), so
// it never ran deeper than a double buffer; the shared tile is faster and needs half the LDS.
__device__ __forceinline__ void gemm_tile_deep(Acc& acc, const double* __restrict__ A, long lda, const double* __restrict__ B,
                                               long ldb, int K, double* lds) {
    gemm_tile_mc<4, true>(acc, A, lda, B, ldb, 0, K, lds);   // K: multiples of 128
}

// Accumulator tile -> global memory through a column-major LDS image [128][DL] (147 KB: workgroups that own a CU's whole LDS):
// the accumulator layout gives every lane 8-byte accesses 64 KB apart (16 lanes per 128-byte run), and a 128 x 128 read-modify-
// write issued that way takes 16 us per tile (measured in the dataflow Cholesky: as long as the K = 128 product itself).
// From the image every wave moves whole 1 KB columns with 16-byte accesses, all loads of a half tile in flight at once.
// SUB: C -= acc, otherwise C = acc.  WT: write-through (sc1) stores -- the tile is read by other CUs next (the caller drains
// and raises a flag), otherwise plain stores.  Values are only moved.
// MODE: 0 C = acc, 1 C -= acc, 2 C = -acc, 3 C += acc.  KEEP: the values stored are also written back into the image (the caller
// then transposes it in place for the mirror tile: image_transpose_inplace).
template <int MODE, bool WT, bool KEEP = false>
__device__ __forceinline__ void tile_commit(double* __restrict__ C, long ld, const Acc& acc, double* img) {
    constexpr bool SUB = MODE == 1 || MODE == 3;      // C is read
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) img[acc_m(i) + acc_n(jj, r) * DL] = acc.v[i][jj][r];
    lds_barrier();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        d2_t cv[16];
        if (SUB) {
#pragma unroll
            for (int q = 0; q < 16; ++q) cv[q] = *reinterpret_cast<const d2_t*>(C + 2 * lane + (long)(w + 4 * (16 * h + q)) * ld);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = w + 4 * (16 * h + q);
            d2_t v = *reinterpret_cast<const d2_t*>(img + 2 * lane + c * DL);
            if (MODE == 1) v = cv[q] - v;
            else if (MODE == 2) v = -v;
            else if (MODE == 3) v = cv[q] + v;
            if (KEEP) *reinterpret_cast<d2_t*>(img + 2 * lane + c * DL) = v;
            if (WT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrc, (int)((2 * lane + (long)c * ld) * 8), 0, 16);
            else *reinterpret_cast<d2_t*>(C + 2 * lane + (long)c * ld) = v;
        }
    }
}

// The same for HALF a tile: the 64 columns [64 nhalf, 64 nhalf + 64) that gemm_tile_mc<2> leaves in acc.v[.][0..1] (wave (wm, sub):
// 64 rows x 32 columns at column 64 nhalf + 32 sub).  Sub-diagonal tiles of the streamed factorisation have one owner per half.
template <int MODE, bool WT>
__device__ __forceinline__ void tile_commit_half(double* __restrict__ C, long ld, const Acc& acc, double* img, int nhalf) {
    constexpr bool SUB = MODE == 1 || MODE == 3;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = 64 * nhalf + (w >> 1) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) img[acc_m(i) + (n0 + 16 * jj + (lane >> 4) + 4 * r) * DL] = acc.v[i][jj][r];
    lds_barrier();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0x7fffffff, 0x00020000);
    d2_t cv[16];
    if (SUB) {
#pragma unroll
        for (int q = 0; q < 16; ++q) cv[q] = *reinterpret_cast<const d2_t*>(C + 2 * lane + (long)(64 * nhalf + w + 4 * q) * ld);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = 64 * nhalf + w + 4 * q;
        d2_t v = *reinterpret_cast<const d2_t*>(img + 2 * lane + c * DL);
        if (MODE == 1) v = cv[q] - v;
        else if (MODE == 2) v = -v;
        else if (MODE == 3) v = cv[q] + v;
        if (WT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrc, (int)((2 * lane + (long)c * ld) * 8), 0, 16);
        else *reinterpret_cast<d2_t*>(C + 2 * lane + (long)c * ld) = v;
    }
}

// 128 x 128 LDS image [m + n DL] -> its transpose, in place.  The 36 units (28 pairs of 16 x 16 tiles (ti, tj) / (tj, ti) and the 8
// diagonal tiles) are dealt to the four waves; inside a tile every 16-lane group walks a wrapped diagonal (row fl, column
// fl + fk + 4 q mod 16), so the reads (column-major) and the writes (of the transposed positions) both touch 16 different banks
// -- a straight transpose makes one side a 16-way conflict (transpose_inverse_tiles, chol_diag.hpp, uses the same walk).  A unit
// touches only its own two tiles and reads both completely before it writes: no barrier inside; the caller brackets the call
// with barriers.
__device__ __forceinline__ void image_transpose_inplace(double* img) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fl = lane & 15, fk = lane >> 4;
    for (int n = 0; n < 9; ++n) {
        const int u = 4 * n + wave;
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= u) ++ti;
        const int tj = u - ti * (ti + 1) / 2;
        double va[4], vb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int wr = (fl + fk + 4 * q) & 15;
            va[q] = img[(16 * ti + fl) + (16 * tj + wr) * DL];      // tile (ti, tj), element (fl, wr)
            vb[q] = img[(16 * tj + fl) + (16 * ti + wr) * DL];      // tile (tj, ti), element (fl, wr)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every read of the unit before its first write
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int wr = (fl + fk + 4 * q) & 15;
            img[(16 * tj + wr) + (16 * ti + fl) * DL] = va[q];      // tile (tj, ti), element (wr, fl)
            img[(16 * ti + wr) + (16 * tj + fl) * DL] = vb[q];      // tile (ti, tj), element (wr, fl)
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The chain workgroup's 128 x 128 tiles: wave w owns the 16-row blocks {w, 7 - w} and all eight 16-column blocks (ChainAcc),
// which balances a triangle exactly.  Operand slabs (16 k x 128, M-contiguous) go global -> LDS with LDS-direct loads.
// ---------------------------------------------------------------------------------------------------------
struct ChainAcc {
    d4_t v[2][8];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < 8; ++n) v[a][n] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
};

// wait until at most N of this wave's loads are outstanding, then the workgroup barrier -- a bare s_barrier: __syncthreads()
// carries a fence for which the compiler waits for vmcnt(0), i.e. for EVERY slab in flight (each of the 8 steps then paid a
// full load latency: 17.5 us per product where the MFMAs need 7.7)
template <int N>
__device__ __forceinline__ void chain_wait_barrier() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    ring_wait_barrier<N>();
}

// ---- panel tiles as triangular solves ----------------------------------------------------------------------
// X = A L^-T for a 128 x 128 tile A and the lower-triangular diagonal block L = L_jj, by 16-column blocks:
//     X_s = (A_s - sum_{k<s} X_k L_sk^T) T16_s^T ,   T16_s = (L_ss)^-1 (the eight 16 x 16 inverses diag16 produces anyway).
// Right-looking over a slab stream (slab s = columns 16 s .. 16 s + 15 of A and of L, both M-contiguous, LDS-direct loads, ring
// of four slab pairs): when slab s lands, X_s = (A_s - S_s) T16_s^T closes block s and every
// later block receives S_c += X_s L_cs^T.  288 MFMAs per wave -- the count of the product with the full inverse T_jj restricted
// to its non-zero blocks -- but the 128 x 128 inverse itself is no longer needed by any panel tile: the chain stops building it
// (no S / T phases in the diagonal block, no transposes, no 128 KB store per step); launch_diag_inverse forms all T_jj after
// the factorisation, in parallel.  Wave w owns the 16-row blocks w and 7 - w; V[a][c] holds S_c until step c and X_c afterwards.
template <class T16>
__device__ __forceinline__ void chain_trsm(ChainAcc& V, const double* __restrict__ A, long lda, const double* __restrict__ L, long ldl,
                                           T16&& t16, double* lds) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fl = lane & 15, fk = lane >> 4;
    const int mi[2] = {wave, 7 - wave};
    constexpr int SLAB = GEMM_LDS_TILE;
    auto issue = [&](int s) {                            // wave w brings k-rows 4w .. 4w+3 of slab s
        double* base = lds + (s & 3) * 2 * SLAB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * wave + r;
            slab_row_to_lds(A + 2 * lane + (long)(16 * s + row) * lda, base + row * GEMM_LDS_MC_LD);
            slab_row_to_lds(L + 2 * lane + (long)(16 * s + row) * ldl, base + SLAB + row * GEMM_LDS_MC_LD);
        }
    };
    V.zero();
    issue(0); issue(1); issue(2);
    auto step = [&](auto S) {
        constexpr int s = decltype(S)::value;
        chain_wait_barrier<8 * (s <= 5 ? 2 : 7 - s)>();  // every wave's part of slab s is in LDS; slab s-1 is no longer read
        if (s + 3 < 8) issue(s + 3);                     // into the buffer of slab s - 1
        const double* la = lds + (s & 3) * 2 * SLAB;     // A[:, 16 s ..]: element (m, kk) at la[kk * LD + m]
        const double* lb = la + SLAB;                    // L[:, 16 s ..]
        double tf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) tf[kk] = t16(s, fl, 4 * kk + fk);   // T16_s[n = fl][k] (LDS in both users)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            d4_t r;                                       // A_s - S_s in the accumulator layout (m = fl, n = fk + 4 q)
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = la[(fk + 4 * q) * GEMM_LDS_MC_LD + 16 * mi[a] + fl] - V.v[a][s][q];
            d4_t x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[kk], r[kk], x, 0, 0, 0);   // r as the A fragment: (m = fl, k = fk + 4 kk)
            V.v[a][s] = x;
#pragma unroll
            for (int c = s + 1; c < 8; ++c) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double bf = lb[(fk + 4 * kk) * GEMM_LDS_MC_LD + 16 * c + fl];   // L[16 c + n][16 s + k]
                    V.v[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf, x[kk], V.v[a][c], 0, 0, 0);
                }
            }
        }
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
    __syncthreads();                                     // LDS free for the next user
}

// ---- streamed panel tiles ----------------------------------------------------------------------------------
// The diagonal block publishes its column blocks WHILE it factors (StreamPublish below: column block s of L_jj and T16_s leave
// right behind the panel tiles of its step s, `flag` counts 3 arrivals per block); a panel tile that is already complete can be
// solved against them block by block instead of waiting for the whole diagonal block: X_s only needs the column blocks <= s.
// Same arithmetic as chain_trsm (same slabs, same MFMA order: same bits); what differs is where the L slabs come from: slab s
// is requested when its block is published (A slabs: three ahead, as before), a slab that is already published when the
// previous step starts is requested one step early.  LDS: A ring 4 x 18 KB, L slabs 2 x 18 KB, T16 2 x 2 KB (behind the image area).
// false: the wait was aborted (another workgroup gave up or the bounded wait expired); the caller leaves the kernel.
__device__ __forceinline__ bool stream_wait(int* flag, int target, int* abort_flag, long long timeout) {
    int ok = 1;
    if ((threadIdx.x & 63) == 0) ok = pk_spin(flag, target, abort_flag, timeout) ? 1 : 0;
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}
__device__ __forceinline__ bool stream_trsm(ChainAcc& V, const double* __restrict__ A, long lda, const double* __restrict__ L, long ldl,
                                            const double* __restrict__ T, long ldt, int* flag, int* abort_flag, long long timeout,
                                            double* lds) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fl = lane & 15, fk = lane >> 4;
    const int mi[2] = {wave, 7 - wave};
    constexpr int SLAB = GEMM_LDS_TILE;
    double* Lbuf = lds + 4 * SLAB;                       // [2][SLAB]
    double* Tbuf = lds + 128 * DL;                       // [2][256]: behind the image area (the workers keep scheduler state in the image's padding rows)
    auto issueA = [&](int s) {
        double* base = lds + (s & 3) * SLAB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * wave + r;
            slab_row_to_lds(A + 2 * lane + (long)(16 * s + row) * lda, base + row * GEMM_LDS_MC_LD);
        }
    };
    auto issueL = [&](int s) {                           // this wave's four k-rows of column block s + (waves 0, 1) half of T16_s
        double* base = Lbuf + (s & 1) * SLAB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * wave + r;
            slab_row_to_lds_sc1(L + 2 * lane + (long)(16 * s + row) * ldl, base + row * GEMM_LDS_MC_LD);
        }
        if (wave < 2)                                    // 8 columns x 128 bytes: lane -> (column 8 wave + lane / 8, row pair lane % 8)
            slab_row_to_lds_sc1(T + (16 * s + 2 * (lane & 7)) + (long)(16 * s + 8 * wave + (lane >> 3)) * ldt, Tbuf + (s & 1) * 256 + 128 * wave);
    };
    V.zero();
    issueA(0); issueA(1); issueA(2);
    int issuedL = 0;                                     // column blocks this wave has requested
    bool ok = true;
    auto step = [&](auto S) {
        constexpr int s = decltype(S)::value;
        if (!ok) return;
        if (issuedL <= s) {
            if (!stream_wait(flag, 3 * (s + 1), abort_flag, timeout)) { ok = false; return; }
            issueL(s);                                   // agent-scope loads: no invalidate needed in front of them
            issuedL = s + 1;
        }
        ring_wait_barrier<0>();                          // slab s of A and of L in LDS (every wave's part); slab s - 1 no longer read
        if (s + 3 < 8) issueA(s + 3);                    // into the buffer of slab s - 1
        if (s + 1 < 8) {                                 // the next block is already published: request it now, behind this step's compute
            int have = 0;
            if (lane == 0) have = df_flag(flag) >= 3 * (s + 2) ? 1 : 0;
            if (__builtin_amdgcn_readfirstlane(have)) {
                issueL(s + 1);
                issuedL = s + 2;
            }
        }
        const double* la = lds + (s & 3) * SLAB;         // A[:, 16 s ..]: element (m, kk) at la[kk * LD + m]
        const double* lb = Lbuf + (s & 1) * SLAB;        // L[:, 16 s ..]
        const double* tb = Tbuf + (s & 1) * 256;
        double tf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) tf[kk] = tb[fl + 16 * (4 * kk + fk)];   // T16_s[n = fl][k]
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            d4_t r;
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = la[(fk + 4 * q) * GEMM_LDS_MC_LD + 16 * mi[a] + fl] - V.v[a][s][q];
            d4_t x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[kk], r[kk], x, 0, 0, 0);
            V.v[a][s] = x;
#pragma unroll
            for (int c = s + 1; c < 8; ++c) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double bf = lb[(fk + 4 * kk) * GEMM_LDS_MC_LD + 16 * c + fl];
                    V.v[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf, x[kk], V.v[a][c], 0, 0, 0);
                }
            }
        }
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
    if (!ok) return false;
    __syncthreads();                                     // LDS free for the next user
    return true;
}

// chol_factor_steps' publish hook for the streamed panel tiles: run by waves 1-3 (192 threads) as soon as column block kb of the
// LDS-resident factor is final.  Column block kb of L (16 columns, zeros above the diagonal) and T16_kb go to global memory
// write-through; `flag` += 1 per wave and block once that wave's stores have drained.  The first three blocks are flagged one
// step late (their stores have drained by then without a stall: waves 1-3 are the longer path in the first steps), the others
// at once.
struct StreamPublish {
    double* A; long lda;
    double* T; long ldt;
    const double* As;
    const double* Ts;
    int* flag;
    int flagged;
    __device__ __forceinline__ void raise(int upto) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (flagged < upto) {
            if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(flag, upto - flagged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flagged = upto;
        }
    }
    __device__ __forceinline__ void operator()(int kb) {
        const int t = (int)threadIdx.x - 64;             // 0 .. 191
        raise(kb);                                       // the blocks before this one: issued at least a step ago
        auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(A, 0, 0x7fffffff, 0x00020000);
        auto rsrcT = __builtin_amdgcn_make_buffer_rsrc(T, 0, 0x7fffffff, 0x00020000);
        for (int idx = t; idx < 1024; idx += 192) {      // 16 columns x 64 row pairs
            const int j = 16 * kb + (idx >> 6), i2 = 2 * (idx & 63);
            d2_t v = *reinterpret_cast<const d2_t*>(As + i2 + j * DL);
            if (i2 < j) v[0] = 0.0;
            if (i2 + 1 < j) v[1] = 0.0;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrcA, (int)((i2 + (long)j * lda) * 8), 0, 16);
        }
        if (t < 128) {                                   // T16_kb: 16 columns x 8 row pairs
            const int c = t >> 3, r2 = 2 * (t & 7);
            const d2_t v = *reinterpret_cast<const d2_t*>(Ts + 256 * kb + r2 + 16 * c);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrcT, (int)(((16 * kb + r2) + (long)(16 * kb + c) * ldt) * 8), 0, 16);
        }
        if (kb >= 3) raise(kb + 1);
    }
};

// Diagonal block of the TRSM form: factor only (chol_factor_steps), L stored write-through, and the eight 16 x 16 inverses
// into the diagonal tiles of Tout (= their final place inside T_jj; the rest of T_jj comes from launch_diag_inverse).
// stream_flag != nullptr: the column blocks are published as they become final (StreamPublish; *stream_flag = 24 on return)
// instead of stored after the last step.
template <bool LOAD>
__device__ __forceinline__ void diag_block_factor(double* __restrict__ A, long lda, double* __restrict__ Tout, long ldt,
                                                  int* __restrict__ info, int global_off, char* smem, int* stream_flag = nullptr) {
    double* As = reinterpret_cast<double*>(smem);
    double* Ts = As + 128 * DL;
    const int tid = threadIdx.x;
    if (LOAD) {
        const int i2 = 2 * (tid & 63), jc = tid >> 6;
#pragma unroll 8
        for (int p = 0; p < 32; ++p) {
            const int j = 4 * p + jc;
            d2_t v = *reinterpret_cast<const d2_t*>(A + (long)i2 + (long)j * lda);
            if ((i2 >> 4) < (j >> 4)) v = d2_t{0.0, 0.0};
            *reinterpret_cast<d2_t*>(As + i2 + j * DL) = v;
        }
        __syncthreads();
    }
    if (stream_flag) {
        StreamPublish pub{A, lda, Tout, ldt, As, Ts, stream_flag, 0};
        chol_factor_steps(As, Ts, info, global_off, pub);
        __syncthreads();
        return;
    }
    chol_factor_steps(As, Ts, info, global_off);
    __syncthreads();
    auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(A, 0, 0x7fffffff, 0x00020000);
    auto rsrcT = __builtin_amdgcn_make_buffer_rsrc(Tout, 0, 0x7fffffff, 0x00020000);
    {
        const int i2 = 2 * (tid & 63), jc = tid >> 6;
#pragma unroll 8
        for (int p = 0; p < 32; ++p) {
            const int j = 4 * p + jc;
            d2_t v = *reinterpret_cast<const d2_t*>(As + i2 + j * DL);
            if (i2 < j) v[0] = 0.0;
            if (i2 + 1 < j) v[1] = 0.0;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrcA, (int)((i2 + (long)j * lda) * 8), 0, 16);
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {                        // 8 tiles x 16 columns x 8 row pairs = 1024 16-byte stores
        const int idx = tid + 256 * p;
        const int t = idx >> 7, c = (idx >> 3) & 15, r2 = 2 * (idx & 7);
        const d2_t v = *reinterpret_cast<const d2_t*>(Ts + 256 * t + r2 + 16 * c);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrcT, (int)(((16 * t + r2) + (long)(16 * t + c) * ldt) * 8), 0, 16);
    }
}

// ---- the dataflow chain's products with their results kept in LDS ----------------------------------------
// ChainAcc (wave w: 16-row blocks w and 7 - w) -> image img[m + n DL]: the layout of diag_block's As, and of eight
// consecutive operand slabs [16][DL] when the image is read as an operand with k = n.
template <bool TRI>
__device__ __forceinline__ void chain_acc_to_image(const ChainAcc& acc, double* img) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int mi = a == 0 ? wave : 7 - wave;
#pragma unroll
        for (int nj = 0; nj < 8; ++nj) {
            if (!TRI && nj > mi) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(16 * nj + (lane >> 4) + 4 * r) * DL + 16 * mi + (lane & 15)] = acc.v[a][nj][r];
        }
    }
}
// image -> global tile, whole 1 KB columns per wave, write-through (the panel tile L_{j+1,j} the workers wait for)
__device__ __forceinline__ void chain_image_store_wt(double* __restrict__ C, long ld, const double* img) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0x7fffffff, 0x00020000);
#pragma unroll 8
    for (int q = 0; q < 32; ++q) {
        const int c = w + 4 * q;
        const d2_t v = *reinterpret_cast<const d2_t*>(img + 2 * lane + c * DL);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrc, (int)((2 * lane + (long)c * ld) * 8), 0, 16);
    }
}
// A A^T on the lower 16 x 16 blocks with A (128 x 128, element (m, k) at img[m + k DL]) resident in LDS: the MFMA sequence of
// chain_gemm<false> per block (slab by slab, k ascending) without its loads and barriers -- same bits -- but with FEWER FRAGMENT
// READS: the 36 lower blocks are dealt to the waves as 3 x 3 groups (wave 0: rows 5-7 x
// columns 0-2, wave 1: rows 5-7 x columns 3-5, wave 2: rows 2-4 x columns 0-2, wave 3: the three 2 x 2 triangles on the
// diagonal), so a wave reads 5-6 fragments per k step for its nine MFMAs instead of 10-11 (rows w and 7 - w against every column
// block) -- the A and B fragments of a row block are the same LDS words.  Measured with the fine stamps (POTRF_BENCH_FINE=1):
// the chain's products are bounded by their LDS reads (11.0 us for 7.7 us of MFMA work; 12.4 us with 18 reads per nine MFMAs).
// Every block still sums its k in ascending order: same bits.  Result -> image in place (one barrier inside: every wave has
// read the operand before anybody overwrites it).
struct NoSlabHook {
    __device__ __forceinline__ void operator()(int) const {}
};
// before_slab(s): called by every thread before slab s (columns 16 s .. 16 s + 15 of the operand) is read.
// Csub != nullptr: the image receives Csub - A A^T instead (Csub: the 128 x 128 tile the product is subtracted from, in global
// memory; its lower 16 x 16 blocks are loaded in the ACCUMULATOR layout -- agent-scope loads, in flight behind the product -- so
// the subtraction happens in the write-back, without a pass of its own over the image).  The same single subtraction of the
// complete sum: same bits as image <- product, image <- Csub - image.
// The pieces of the product, shared by the in-place form below and by the form that accumulates it from published blocks (stream_syrk_image):
// the 3 x 3 deal, the tile the product is subtracted from in the accumulator layout, ONE slab of the sum, the write-back.
template <int W>
struct SyrkDeal {
    static constexpr int NB9[4][9][2] = {
        {{5, 0}, {5, 1}, {5, 2}, {6, 0}, {6, 1}, {6, 2}, {7, 0}, {7, 1}, {7, 2}},
        {{5, 3}, {5, 4}, {5, 5}, {6, 3}, {6, 4}, {6, 5}, {7, 3}, {7, 4}, {7, 5}},
        {{2, 0}, {2, 1}, {2, 2}, {3, 0}, {3, 1}, {3, 2}, {4, 0}, {4, 1}, {4, 2}},
        {{0, 0}, {1, 0}, {1, 1}, {3, 3}, {4, 3}, {4, 4}, {6, 6}, {7, 6}, {7, 7}}};
    static constexpr int row(int p) { return NB9[W][p][0]; }
    static constexpr int col(int p) { return NB9[W][p][1]; }
    static constexpr bool used(int b) {
        bool u = false;
        for (int p = 0; p < 9; ++p) u = u || NB9[W][p][0] == b || NB9[W][p][1] == b;
        return u;
    }
};
template <int W>
__device__ __forceinline__ void syrk_load_c0(d4_t (&c0)[9], const double* __restrict__ Csub, long ldc, int lane) {
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Csub), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long off = (16 * SyrkDeal<W>::row(p) + (lane & 15)) + (long)(16 * SyrkDeal<W>::col(p) + (lane >> 4) + 4 * r) * ldc;
            c0[p][r] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(off * 8), 0, 16));
        }
}
// acc += (slab)(slab)^T on this wave's nine blocks; slab: 16 k-rows of the operand, element (m, k) at slab[k GEMM_LDS_MC_LD + m]
template <int W>
__device__ __forceinline__ void syrk_slab(d4_t (&acc)[9], const double* slab, int lane) {
    const int l0 = (lane >> 4) * GEMM_LDS_MC_LD + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double* la = slab + 4 * kk * GEMM_LDS_MC_LD + l0;
        double f[8];
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (SyrkDeal<W>::used(b)) f[b] = la[16 * b];
#pragma unroll
        for (int p = 0; p < 9; ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[SyrkDeal<W>::col(p)], f[SyrkDeal<W>::row(p)], acc[p], 0, 0, 0);
    }
}
template <int W, bool SUB>
__device__ __forceinline__ void syrk_to_image(double* img, const d4_t (&acc)[9], const d4_t (&c0)[9], int lane) {
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            img[(16 * SyrkDeal<W>::col(p) + (lane >> 4) + 4 * r) * DL + 16 * SyrkDeal<W>::row(p) + (lane & 15)] = SUB ? c0[p][r] - acc[p][r] : acc[p][r];
}
template <int W, class F>
__device__ __forceinline__ void chain_syrk_inplace_wave(double* img, F&& before_slab, const double* __restrict__ Csub = nullptr, long ldc = 0) {
    const int lane = threadIdx.x & 63;
    d4_t acc[9], c0[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) acc[p] = d4_t{0.0, 0.0, 0.0, 0.0};
    if (Csub) syrk_load_c0<W>(c0, Csub, ldc, lane);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        before_slab(s);
        syrk_slab<W>(acc, img + s * GEMM_LDS_TILE, lane);
    }
    lds_barrier();
    if (Csub) syrk_to_image<W, true>(img, acc, c0, lane);
    else syrk_to_image<W, false>(img, acc, c0, lane);
}
template <class F = NoSlabHook>
__device__ __forceinline__ void chain_syrk_inplace(double* img, F&& before_slab = NoSlabHook{}, const double* __restrict__ Csub = nullptr,
                                                   long ldc = 0) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) chain_syrk_inplace_wave<0>(img, before_slab, Csub, ldc);
    else if (wave == 1) chain_syrk_inplace_wave<1>(img, before_slab, Csub, ldc);
    else if (wave == 2) chain_syrk_inplace_wave<2>(img, before_slab, Csub, ldc);
    else chain_syrk_inplace_wave<3>(img, before_slab, Csub, ldc);
}
// image <- (global tile) - image on the lower 16 x 16 blocks, zero above: diag_block's input, built in place
__device__ __forceinline__ void chain_image_rsub(const double* __restrict__ C, long ld, double* img) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ti = lane >> 3;                                    // 16-row block of rows 2 lane, 2 lane + 1
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        d2_t cv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = w + 4 * (16 * h + q);
            cv[q] = d2_t{0.0, 0.0};
            if (ti >= (c >> 4)) cv[q] = *reinterpret_cast<const d2_t*>(C + 2 * lane + (long)c * ld);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = w + 4 * (16 * h + q);
            d2_t v = d2_t{0.0, 0.0};
            if (ti >= (c >> 4)) v = cv[q] - *reinterpret_cast<const d2_t*>(img + 2 * lane + c * DL);
            *reinterpret_cast<d2_t*>(img + 2 * lane + c * DL) = v;
        }
    }
}

// ---- the streamed solve that PUBLISHES its blocks (round 6: the three-workgroup chain) -----------------------------------------
// stream_trsm, with every finished 16-column block X_s leaving at once: to an LDS slab (two of them, behind the solve's rings), from
// there to global memory (whole 1 KB columns, write-through: the tile is stored BY the solve, in place) and counted in `xprog` (+1 per
// wave and block once that wave's four columns of it have drained: 4 (s + 1) = block s is in global memory).  Two consumers run
// behind the counter: the workgroup that forms the next diagonal tile's A - X X^T slab by slab (stream_syrk_image), and the last
// update of the tile below (stream_update_half).  Same arithmetic as stream_trsm: same bits.
// Order of a step, behind its barrier: the next column block of the diagonal block (if the poll of the step before saw it
// published), the poll for the step after (an inline-assembly load: the compiler puts a vmcnt(0) in front of a tracked load's
// result while LDS-direct loads are in flight), the stores of X_{s-1} (formed a step ago), the A slab three steps ahead; block
// s - 2 is counted at the top of step s.  Every wait is a FULL drain: counted waits (vmcnt(8): "everything older than the four
// stores and the four slab rows") made 1 of 300 launches at N = 2560 / 3072 produce a different factor (POTRF_BENCH_STRESS) --
// stores and loads of a wave do not retire in one common order on this chip, so a count behind a mix of both says nothing about
// WHICH operations are still out.  Measured: 17-20 us per tile when the solve runs unthrottled (7.7 us of products).
// LDS: A ring slabs 0-3, L slabs 4-5, X slabs 6-7 of the image area, T16 buffers behind it -- the whole 160 KB.
// The caller drains (raise(8) has) and raises the tile's own flag.
__device__ __forceinline__ bool stream_trsm_publish(double* __restrict__ A, long lda, const double* __restrict__ L, long ldl,
                                                    const double* __restrict__ T, long ldt, int* flag, int* abort_flag, long long timeout,
                                                    int* xprog, double* lds) {
    // `lane` is made opaque: the per-lane LDS / global offsets derived from it are otherwise hoisted out of the caller's loop over the
    // diagonal blocks and kept (spilled) across it
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fl = lane & 15, fk = lane >> 4;
    const int mi[2] = {wave, 7 - wave};
    constexpr int SLAB = GEMM_LDS_TILE;
    double* Lbuf = lds + 4 * SLAB;                       // [2][SLAB]
    double* Xbuf = lds + 6 * SLAB;                       // [2][SLAB]
    double* Tbuf = lds + 128 * DL;                       // [2][256]
    auto issueA = [&](int s) {
        double* base = lds + (s & 3) * SLAB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * wave + r;
            slab_row_to_lds(A + 2 * lane + (long)(16 * s + row) * lda, base + row * GEMM_LDS_MC_LD);
        }
    };
    auto issueL = [&](int s) {
        double* base = Lbuf + (s & 1) * SLAB;
        if (s < 7) {                                     // block 7 has no later block to update: only T16_7 is needed
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * wave + r;
                slab_row_to_lds_sc1(L + 2 * lane + (long)(16 * s + row) * ldl, base + row * GEMM_LDS_MC_LD);
            }
        }
        if (wave < 2)
            slab_row_to_lds_sc1(T + (16 * s + 2 * (lane & 7)) + (long)(16 * s + 8 * wave + (lane >> 3)) * ldt, Tbuf + (s & 1) * 256 + 128 * wave);
    };
    auto rsrcX = __builtin_amdgcn_make_buffer_rsrc(A, 0, 0x7fffffff, 0x00020000);
    auto store_block = [&](int sb) {                     // block sb from its slab: column 16 sb + 4 q + wave of the tile, 64 lanes x 16 bytes
        const double* xs_ = Xbuf + (sb & 1) * SLAB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * q + wave;
            const d2_t v = *reinterpret_cast<const d2_t*>(xs_ + c * GEMM_LDS_MC_LD + 2 * lane);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrcX, (int)((2 * lane + (long)(16 * sb + c) * lda) * 8), 0, 16);
        }
    };
    auto count_block = [&]() {                           // every operation of this wave has completed: one more block of it is out
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(xprog, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    ChainAcc V;
    V.zero();
    issueA(0); issueA(1); issueA(2);
    int issuedL = 0;
    int seen_raw = 0;                                    // the diagonal block's publication count as last polled (lane 0)
    bool ok = true;
    auto step = [&](auto S) {
        constexpr int s = decltype(S)::value;
        if (!ok) return;
        if (s >= 2) count_block();                       // block s - 2: stored in the step before
        if (issuedL <= s) {
            if (!stream_wait(flag, 3 * (s + 1), abort_flag, timeout)) { ok = false; return; }
            issueL(s);
            issuedL = s + 1;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(seen_raw)::"memory");   // slab s of A and of L in LDS; X_{s-1} in its slab
        const int seen = __builtin_amdgcn_readfirstlane(seen_raw);          // the diagonal block's count as polled a step ago
        if (s + 1 < 8) {
            if (seen >= 3 * (s + 2)) {
                issueL(s + 1);
                issuedL = s + 2;
            }
            if (lane == 0) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(seen_raw) : "v"(flag) : "memory");
        }
        if (s > 0) store_block(s - 1);
        if (s + 3 < 8) issueA(s + 3);
        const double* la = lds + (s & 3) * SLAB;
        const double* lb = Lbuf + (s & 1) * SLAB;
        const double* tb = Tbuf + (s & 1) * 256;
        double* xb = Xbuf + (s & 1) * SLAB;
        double tf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) tf[kk] = tb[fl + 16 * (4 * kk + fk)];
        // both row blocks' X_s first (into the slab), then the later blocks' sums
        d4_t xs[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            d4_t r;
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = la[(fk + 4 * q) * GEMM_LDS_MC_LD + 16 * mi[a] + fl] - V.v[a][s][q];
            d4_t x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[kk], r[kk], x, 0, 0, 0);
            xs[a] = x;
#pragma unroll
            for (int q = 0; q < 4; ++q) xb[(fk + 4 * q) * GEMM_LDS_MC_LD + 16 * mi[a] + fl] = x[q];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int c = s + 1; c < 8; ++c) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double bf = lb[(fk + 4 * kk) * GEMM_LDS_MC_LD + 16 * c + fl];
                    V.v[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf, xs[a][kk], V.v[a][c], 0, 0, 0);
                }
            }
        }
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
    if (!ok) return false;
    count_block();                                       // block 6: stored in step 7
    lds_barrier();                                       // X_7 complete in its slab
    store_block(7);
    count_block();                                       // block 7: the tile is complete in global memory
    return true;
}

// The other half of the three-workgroup chain: A_{j,j} - X X^T for the NEXT diagonal block, accumulated from the 16-column blocks of
// X = L_{j,j-1} as the solving workgroup publishes them (xprog: 4 (s + 1) = block s is in global memory), one slab of
// chain_syrk_inplace's sum per block -- same deal, same k order, same single subtraction: same bits.  On return the image holds the
// difference on its lower 16 x 16 blocks (diag_block_factor<false>'s input, behind the caller's barrier).  Behind the last block remain
// one slab (36 products per wave) and the write-back.
template <class Ready>
__device__ __forceinline__ bool stream_syrk_image(const double* __restrict__ X, long ldx, int* xprog, int* abort_flag, long long timeout,
                                                  const double* __restrict__ Csub, long ldc, Ready&& ready, long long* stamp, double* lds) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int SLAB = GEMM_LDS_TILE;
    auto issue = [&](int s) {
        double* base = lds + (s & 3) * SLAB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * wave + r;
            slab_row_to_lds_sc1(X + 2 * lane + (long)(16 * s + row) * ldx, base + row * GEMM_LDS_MC_LD);
        }
    };
    d4_t acc[9], c0[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) acc[p] = d4_t{0.0, 0.0, 0.0, 0.0};
    int issued = 0;
    bool ok = true;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (issued <= s) {
            if (!stream_wait(xprog, 4 * (s + 1), abort_flag, timeout)) { ok = false; break; }
            issue(s);
            issued = s + 1;
        }
        if (s == 7 && stamp) stamp[0] = wall_clock64();    // probes: the last block of the tile has arrived
        if (s == 6) {                                      // the tile the product is subtracted from: in flight behind the last two slabs
            if (!ready()) { ok = false; break; }
            if (wave == 0) syrk_load_c0<0>(c0, Csub, ldc, lane);
            else if (wave == 1) syrk_load_c0<1>(c0, Csub, ldc, lane);
            else if (wave == 2) syrk_load_c0<2>(c0, Csub, ldc, lane);
            else syrk_load_c0<3>(c0, Csub, ldc, lane);
        }
        int seen = 0;
        if (s + 1 < 8 && lane == 0) seen = df_flag(xprog);
        ring_wait_barrier<0>();                          // slab s in LDS (every wave's rows); slab s - 1 no longer read
        if (s + 1 < 8) {
            if (__builtin_amdgcn_readfirstlane(seen) >= 4 * (s + 2)) {
                issue(s + 1);
                issued = s + 2;
            }
        }
        const double* xb = lds + (s & 3) * SLAB;
        if (wave == 0) syrk_slab<0>(acc, xb, lane);
        else if (wave == 1) syrk_slab<1>(acc, xb, lane);
        else if (wave == 2) syrk_slab<2>(acc, xb, lane);
        else syrk_slab<3>(acc, xb, lane);
    }
    if (!ok) return false;
    lds_barrier();                                       // every wave has read the last slab: the image may be written
    if (wave == 0) syrk_to_image<0, true>(lds, acc, c0, lane);
    else if (wave == 1) syrk_to_image<1, true>(lds, acc, c0, lane);
    else if (wave == 2) syrk_to_image<2, true>(lds, acc, c0, lane);
    else syrk_to_image<3, true>(lds, acc, c0, lane);
    return true;
}

// The LAST update of half a sub-diagonal tile, C_{k+1,k}[:, 64 nhalf ..] -= L_{k+1,k-1} L_{k,k-1}^T, run BEHIND the chain's solve of
// L_{k,k-1}: its K loop walks the 16-column blocks of both operands, and block s of L_{k,k-1} is in global memory when the solve
// has finished its step s (xprog).  The tile is then complete one block's product + its write-back after the chain's tile -- not
// a whole K = 128 product later (13 us: with the product L L^T folded into the solve that product had become the chain's step).
// Same slabs, same fragment layout, same k order as gemm_tile_mc<2>: same bits; the tile's values are in registers before the
// last block arrives.
__device__ __forceinline__ bool stream_update_half(double* __restrict__ C, long ld, const double* __restrict__ A, const double* __restrict__ B,
                                                   int nhalf, int* xprogA, int* xprog, int* abort_flag, long long timeout, double* lds) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fl = lane & 15, fk = lane >> 4;
    const int wm = (wave & 1) * 64, wn = 64 * nhalf + (wave >> 1) * 32;
    constexpr int SLAB = GEMM_LDS_TILE;
    // BOTH operands belong to the column the chain is busy with: L_{k,k-1} is the chain's own tile, L_{k+1,k-1} the tile right below it,
    // solved by a worker behind the same diagonal block -- each publishes its 16-column blocks as they become final (xprog, xprogA)
    // Four slots of (A slab, B slab): every block that is published is requested at once, up to two ahead of the one in work (the
    // slot of block s + 3 is the one a slower wave may still be reading in step s - 1's place: this request runs in front of the barrier).  The
    // two progress counters are polled ONCE per step, by loads issued behind the step's barrier and read at the top of the next step
    // (their round trip, ~1 us each, stands behind the step's products: a loop that polled, loaded and multiplied one after the other
    // took 4 us per block and ended 9 us behind its producers, which publish a block every 2.1 us).
    auto issueAB = [&](int s) {
        double* base = lds + (s & 3) * 2 * SLAB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * wave + r;
            slab_row_to_lds_sc1(A + 2 * lane + (long)(16 * s + row) * ld, base + row * GEMM_LDS_MC_LD);
            slab_row_to_lds_sc1(B + 2 * lane + (long)(16 * s + row) * ld, base + SLAB + row * GEMM_LDS_MC_LD);
        }
    };
    // (agent-scope loads: with the last update as an item of its own the tile's earlier updates were stored by another CU)
    d2_t cv[16];
    auto rsrcC = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int q = 0; q < 16; ++q)
        cv[q] = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(rsrcC, (int)((2 * lane + (long)(64 * nhalf + wave + 4 * q) * ld) * 8), 0, 16));
    d4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc[i][jj] = d4_t{0.0, 0.0, 0.0, 0.0};
    int issued = 0;
    int pa = 0, pb = 0;                                  // the counters as polled (lane 0)
    bool ok = true;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        // everything this wave has in flight -- the polls of the step before and the slabs requested so far -- has landed
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pa), "+v"(pb)::"memory");
        int pub = min(__builtin_amdgcn_readfirstlane(pa), __builtin_amdgcn_readfirstlane(pb)) >> 2;   // blocks of both tiles in global memory
        bool fresh = false;
        if (pub <= s) {                                  // block s itself is not there yet: wait for it (both counters)
            if (!stream_wait(xprog, 4 * (s + 1), abort_flag, timeout) || !stream_wait(xprogA, 4 * (s + 1), abort_flag, timeout)) { ok = false; break; }
            pub = s + 1;
        }
        for (; issued < min(pub, s + 3); ++issued) {
            issueAB(issued);
            fresh = fresh || issued == s;
        }
        if (fresh) ring_wait_barrier<0>();               // block s was requested just now
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // landed (the wait above); younger slabs stay in flight
        if (s + 1 < 8 && lane == 0) {
            asm volatile("global_load_dword %0, %1, off sc1" : "=v"(pa) : "v"(xprogA) : "memory");
            asm volatile("global_load_dword %0, %1, off sc1" : "=v"(pb) : "v"(xprog) : "memory");
        }
        const double* la = lds + (s & 3) * 2 * SLAB;
        const double* lb = la + SLAB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double fa[4], fb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = la[(4 * kk + fk) * GEMM_LDS_MC_LD + wm + 16 * i + fl];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) fb[jj] = lb[(4 * kk + fk) * GEMM_LDS_MC_LD + wn + 16 * jj + fl];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[jj], fa[i], acc[i][jj], 0, 0, 0);
        }
    }
    if (!ok) return false;
    lds_barrier();                                       // the rings are no longer read: the image may be written
    // tile_commit_half<1, true> with the tile's values already in registers
    const int n0 = 64 * nhalf + (wave >> 1) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[(wm + 16 * i + fl) + (n0 + 16 * jj + fk + 4 * r) * DL] = acc[i][jj][r];
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = 64 * nhalf + wave + 4 * q;
        const d2_t v = cv[q] - *reinterpret_cast<const d2_t*>(lds + 2 * lane + c * DL);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrcC, (int)((2 * lane + (long)c * ld) * 8), 0, 16);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// Single-launch factorisation, DATAFLOW form (default): no grid barriers.
//
// The barrier form above keeps chain and workers in lock step: the chain waits until every worker has left step j-1, the
// workers wait twice per step for the slowest tile, and at the end of an outer block the chain stands still for the whole
// K = 128 nbo update (0.9 ms per block at N = 8192).  Here every lower tile (i, k) has ONE owner for its whole life -- worker
// (i mod PR) + PR (k mod PC), the workers numbered XCD by XCD so that an XCD's workers form a compact PR x ~32/PR patch of the
// cyclic grid and share their row / column panels in the XCD's L2 -- and the owner applies the tile's updates in the fixed
// chunks of the two-level schedule (one K = 128 nbo accumulation per finished outer block, single steps inside the tile's own
// block: the chunking does not depend on timing, so the bits do not either), then turns it into L_ik = A_ik T_kk^T.  What is
// ready is decided from flags: panel_done[i][j] (L_ij stored), factored[j] (L_jj, T_jj stored), chain_ready[j] (tiles
// (j+1, j) and (j+1, j+1) carry every update the chain does not apply itself).  A worker scans its tiles in column order
// (column k is needed at step k), runs the first task whose inputs exist, and only sleeps when none does; a task never
// waits inside, so there is no circular wait, and a task's inputs were produced on tiles of smaller column index.
// Ownership also removes the coherence traffic of the barrier form: a tile under update is read and written by one CU only
// (no release / acquire per step), panels are written once -- write-through (sc1) stores + drained flag -- and read by CUs
// that never touched those addresses before.  The chain (workgroup 0) is the barrier form's chain with its waits replaced by
// chain_ready[j]; at a block boundary it applies only step j (K = 128) to the next diagonal tile, the owner having applied
// the block's earlier steps as one shorter chunk (the barrier form ran the whole K = 128 nbo product on the chain there).
// Bit-identical to the one-level multi-launch schedule for nbo = 1; agreement to rounding for nbo > 1.
// ---------------------------------------------------------------------------------------------------------
constexpr int DF_FACT = 32;          // sync: [DF_FACT + j] factored, [DF_FACT + nb + j] chain_ready, [DF_FACT + 2 nb + i + j nb] panel_done,
                                     // then xdone[nb][nb] (fused inverse) and diag_ready[nb]
constexpr int DF_MAXT = 120;         // owned tiles per worker (launcher checks; state lives in the LDS padding of columns 0 .. DF_MAXT - 1): N <= ~31 000
constexpr int DF_WIN = 16;           // tiles examined per scheduling round (16 lanes each)

// end of the update chunk of tile (., k) that starts at step j0: a finished outer block in one accumulation (K = 128 nbo),
// single steps inside the tile's own block -- and, for the first `near` columns of a block, also for the block just before
// it: those tiles are needed right after that block's last step, a K = 128 nbo product (110 us) would sit on the chain's path.
// A function of (k, j0) only: the summation order never depends on timing.
__device__ __forceinline__ int df_chunk_end(int k, int j0, int target, int nbo, int near) {
    const int bj = j0 / nbo, bk = k / nbo;
    int j1 = (bj < bk && !(bj == bk - 1 && k - bk * nbo < near)) ? (bj + 1) * nbo : j0 + 1;
    return min(j1, target);
}
// every wave drains its write-through stores, then ONE flag operation
__device__ __forceinline__ void df_publish_store(int* p) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void df_publish_add(int* p) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------
// Fused inverse: K^-1 = (L L^T)^-1 inside the factorisation's launch (N <= 4096).
//
// A factorisation of this size is bound by its serial chain (48 us per 128 columns) and leaves most of the chip idle; the
// separate trtri (recursive doubling: 3 launches per level) and lauum launches that followed it cost 0.43 ms at N = 2048 and
// 1.0 ms at N = 4096 -- more than half of the factorisation itself.  Here a second team of workgroups (blockIdx >= g1) consumes the
// factorisation's own flags and builds the inverse BEHIND the chain, tile by tile:
//   T(j)     T_jj = L_jj^-1 (diag_block<false>), U_jj = T_jj^T                                   needs factored[j]
//   P(i)     P_i = T_ii L_{i,i-1}  (L^T through an LDS transpose; kept in block (i, i-1) of U: its strictly-lower blocks are unused)   needs T(i), panel_done[i][i-1]
//   X(i, j)  M = -(sum_{k=j}^{i-2} U_jk L_ik^T) accumulated in the (unused) block (j, i) of U in fixed chunks of inv_cx blocks,
//            Q = M T_ii^T (needs T(i)), and LAST  U_ji = Q - U_{j,i-1} P_i^T,  X_ij = U_ji^T        needs xdone[i-1][j], P(i)
//   K(i, j)  K^-1_ij = sum_{k >= i} U_ik U_jk^T in fixed chunks of inv_ck blocks (+ the mirror tile)   needs xdone[k][i], xdone[k][j]
// (block forward substitution X_ij = -T_ii sum_k L_ik X_kj, written for the transposes so that every product has both operands
// M-contiguous: gemm_tile_mc).  The term of the row just above is split off -- X_ij = -T_ii S'_ij - P_i X_{i-1,j} with everything
// but the last product available a row earlier --, so the column wavefront advances by ONE product per row (~22 us) instead of
// two (~45 us, as slow as the chain itself: the first form finished 0.7 ms behind the factorisation at N = 4096).  One owner per item, fixed chunk boundaries: the bits do not depend on timing.  Items are dealt
// round-robin in ONE global order -- T / X by row, then K by row -- that is consistent with the dependencies, so the globally
// first unfinished item is always the first unfinished item of its owner: no circular wait.  The inverse's wavefront follows the
// chain about one step behind; after the last diagonal block remain T(nb-1), P(nb-1), the last row's products and the last chunk
// of every K^-1 tile (~75 us).
// ---------------------------------------------------------------------------------------------------------
// One accumulation task of K^-1_ij = sum_{k >= i} U_ik U_jk^T: the blocks [i + d, min(i + d + ck, nb)) of the contraction; behind the
// last one the mirror tile (j, i) is written too.  Returns the end of the range.
__device__ __forceinline__ int potri_k_task(const PersistArgs& a, int i, int j, int d, int ck, double* lds) {
    const int nb = a.nb;
    const long ld = a.ld;
    const int k0 = i + d, k1 = min(k0 + ck, nb);
    double* Kij = a.Kinv + (long)i * NB + (long)j * NB * ld;
    Acc acc;
    acc.zero();
    gemm_tile_mc<4, true>(acc, a.U + (long)i * NB + (long)k0 * NB * ld, ld, a.U + (long)j * NB + (long)k0 * NB * ld, ld, 0,
                          (k1 - k0) * NB, lds);
    if (k1 < nb) {
        if (d == 0) tile_commit<0, false>(Kij, ld, acc, lds);
        else tile_commit<3, false>(Kij, ld, acc, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        if (d == 0) tile_commit<0, false, true>(Kij, ld, acc, lds);
        else tile_commit<3, false, true>(Kij, ld, acc, lds);
        if (i != j) {                                                  // the mirror tile (K^-1 is used as a full matrix)
            lds_barrier();
            image_transpose_inplace(lds);
            lds_barrier();
            chain_image_store_wt(a.Kinv + (long)j * NB + (long)i * NB * ld, ld, lds);
        }
    }
    return k1;
}
// One item of the fused inverse (type 0 T(j), 4 P(i), 1 X(i, j), 2 K(i, j); d = its progress): are the inputs of its next task there?
// Called by the 16 lanes of a scheduling slot, one flag per lane (l = lane in the slot); the caller combines the answers.
__device__ __forceinline__ bool potri_item_ready(const PersistArgs& a, int type, int i, int j, int d, int l) {
    const int nb = a.nb;
    const int* factored = a.sync + DF_FACT;
    const int* panel_done = a.sync + DF_FACT + 2 * nb;
    const int* xdone = a.sync + DF_FACT + 2 * nb + nb * nb;     // [k + j nb], k >= j: X_kj / U_jk stored (k == j: T_jj / U_jj)
    const int cx = a.inv_cx, ck = a.inv_ck;
    bool ok = true;
    if (type == 0) {
        if (l == 0) ok = df_flag(factored + i) >= (a.nchain >= 2 ? 24 : 1);
    } else if (type == 1) {
        // d: terms applied (k = j .. i - 1 - plast), then nterms -> Q pending, nterms + 1 -> last step pending (plast)
        const int nterms = i - j - a.inv_plast;
        if (d < nterms) {
            const int k0 = j + d, k1 = min(k0 + cx, j + nterms), k = k0 + (l & 7);
            if (k < k1) ok = df_flag(l < 8 ? xdone + k + (long)j * nb : panel_done + i + (long)k * nb) != 0;
        } else if (d == nterms) {
            if (l == 0) ok = df_flag(xdone + i + (long)i * nb) != 0;
        } else {
            if (l == 0) ok = df_flag(xdone + (i - 1) + (long)j * nb) != 0;
            else if (l == 1) ok = df_flag(xdone + (i - 1) + (long)i * nb) != 0;        // P(i): the unused entry (i-1, i)
        }
    } else if (type == 4) {
        if (l == 0) ok = df_flag(xdone + i + (long)i * nb) != 0;
        else if (l == 1) ok = df_flag(panel_done + i + (long)j * nb) != 0;
    } else {
        const int k0 = i + d, k1 = min(k0 + ck, nb), k = k0 + (l & 7);
        if (k < k1) ok = df_flag(xdone + k + (long)(l < 8 ? i : j) * nb) != 0;
    }
    return ok;
}
struct PotriStep {
    int d;          // the item's progress after the task
    bool done;      // the item is finished
};
// ... and the task itself (whole workgroup).
__device__ __forceinline__ PotriStep potri_item_task(const PersistArgs& a, int type, int i, int j, int d, double* lds, char* smem) {
    const int nb = a.nb;
    const long ld = a.ld;
    int* xdone = a.sync + DF_FACT + 2 * nb + nb * nb;
    const int tid = threadIdx.x;
    const int cx = a.inv_cx, ck = a.inv_ck;
    PotriStep res{d, false};
    if (type == 0) {
        // T_jj around the eight 16 x 16 diagonal inverses the chain left (loaded, not recomputed: panel solves of the
        // factorisation may still be reading them, and the bits must not depend on who comes first) and U_jj = T_jj^T.  After
        // diag_block the strictly-upper tiles of the LDS image still hold the transposed inverse tiles, Ts the diagonal ones.
        double* Ljj = a.A + (long)i * NB * (ld + 1);
        double* Tjj = a.Linv + (long)i * NB * (ld + 1);
        double* Ujj = a.U + (long)i * NB * (ld + 1);
        diag_block<false, true, true, true>(Ljj, ld, Tjj, ld, nullptr, 0, smem);
        const double* As = lds;
        const double* Ts = lds + 128 * DL;
        auto rsrcU = __builtin_amdgcn_make_buffer_rsrc(Ujj, 0, 0x7fffffff, 0x00020000);
        const int i2 = 2 * (tid & 63), jc = tid >> 6, tr = i2 >> 4;
#pragma unroll 8
        for (int p = 0; p < 32; ++p) {
            const int c = 4 * p + jc, tc = c >> 4;
            d2_t v = {0.0, 0.0};
            if (tr < tc) v = *reinterpret_cast<const d2_t*>(As + i2 + c * DL);
            else if (tr == tc) {                           // U[r][c] = T16[c & 15][r & 15]: zero for r > c by construction
                v[0] = Ts[256 * tc + (c & 15) + 16 * (i2 & 15)];
                v[1] = Ts[256 * tc + (c & 15) + 16 * ((i2 + 1) & 15)];
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rsrcU, (int)((i2 + (long)c * ld) * 8), 0, 16);
        }
        df_publish_store(xdone + i + (long)i * nb);
        res.done = true;
    } else if (type == 4) {
        // P_i = T_ii L_{i,i-1}.  L's contraction index is its contiguous one, so the tile goes through an LDS transpose first
        // (image -> in place -> scratch block, read back as an M-contiguous operand); scratch = block (i, i-1) of U: nothing
        // else reads or writes U's strictly-lower blocks.  (Not a block of K^-1: its mirror tiles are written as soon as the two
        // columns they depend on are complete, while another column may still need P_i.)
        const double* Lsub = a.A + (long)i * NB + (long)j * NB * ld;
        const double* Tii = a.Linv + (long)i * NB * (ld + 1);
        double* Pi = a.U + (long)i * NB + (long)j * NB * ld;
        {
            const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll 8
            for (int q = 0; q < 32; ++q) {
                const int c = w + 4 * q;
                slab_row_to_lds(Lsub + 2 * lane + (long)c * ld, lds + c * DL);
            }
            ring_wait_barrier<0>();
        }
        image_transpose_inplace(lds);
        lds_barrier();
        chain_image_store_wt(Pi, ld, lds);                                 // L_{i,i-1}^T
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // this CU reads the block back through its L1
        __syncthreads();
        Acc acc;
        acc.zero();
        gemm_tile_mc<4, true>(acc, Tii, ld, Pi, ld, 0, NB, lds);           // P[m][n] = sum_k T_ii[m][k] L^T[n][k]
        tile_commit<0, true>(Pi, ld, acc, lds);
        df_publish_store(xdone + j + (long)i * nb);                        // entry (i-1, i)
        res.done = true;
    } else if (type == 1) {
        double* Mji = a.U + (long)j * NB + (long)i * NB * ld;              // block (j, i) of U: M, then Q, then U_ji
        const int nterms = i - j - a.inv_plast;
        if (d < nterms) {
            const int k0 = j + d, k1 = min(k0 + cx, j + nterms);
            Acc acc;
            acc.zero();
            gemm_tile_mc<4, true>(acc, a.U + (long)j * NB + (long)k0 * NB * ld, ld, a.A + (long)i * NB + (long)k0 * NB * ld, ld, 0,
                                  (k1 - k0) * NB, lds);
            if (d == 0) tile_commit<2, false>(Mji, ld, acc, lds);
            else tile_commit<1, false>(Mji, ld, acc, lds);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            res.d = k1 - j;
        } else if (d == nterms) {
            const double* Tii = a.Linv + (long)i * NB * (ld + 1);
            Acc acc;
            acc.zero();
            gemm_tile_mc<4, true>(acc, Mji, ld, Tii, ld, 0, NB, lds);      // Q = M T_ii^T
            if (a.inv_plast) {
                tile_commit<0, false>(Mji, ld, acc, lds);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                res.d = nterms + 1;
            } else {                                                       // every term is in: Q is U_ji
                tile_commit<0, true>(Mji, ld, acc, lds);
                df_publish_store(xdone + i + (long)j * nb);
                image_transpose_inplace(lds);
                lds_barrier();
                chain_image_store_wt(a.Linv + (long)i * NB + (long)j * NB * ld, ld, lds);
                res.done = true;
            }
        } else {
            const double* Pi = a.U + (long)i * NB + (long)(i - 1) * NB * ld;
            Acc acc;
            acc.zero();
            gemm_tile_mc<4, true>(acc, a.U + (long)j * NB + (long)(i - 1) * NB * ld, ld, Pi, ld, 0, NB, lds);   // U_{j,i-1} P_i^T
            if (j == i - 1) tile_commit<2, true, true>(Mji, ld, acc, lds);                                     // no Q: U_ji = -acc
            else tile_commit<1, true, true>(Mji, ld, acc, lds);                                                // U_ji = Q - acc
            df_publish_store(xdone + i + (long)j * nb);                    // U_ji is what the other tasks read
            image_transpose_inplace(lds);                                  // X_ij = U_ji^T: an output only
            lds_barrier();
            chain_image_store_wt(a.Linv + (long)i * NB + (long)j * NB * ld, ld, lds);
            res.done = true;
        }
    } else {
        const int k1 = potri_k_task(a, i, j, d, ck, lds);
        res.d = k1 - i;
        res.done = k1 == nb;
    }
    return res;
}

// Dynamic pools: "has the NEXT task of this item its inputs?" for a scanner that looks at one item per THREAD: what fac_ready_lane /
// potri_item_ready ask lane by lane, with all of an item's flag loads (at most four for chunks of up to two blocks) in flight at once
// -- no arrays (dynamic indexing would put them into scratch memory), every case written out.  *lanes = true: the task's chunk is
// longer than two blocks, the caller asks lane by lane.  it = {type, i, k or j, half}; type 0: a tile of the factorisation.
template <bool FUSE>
__device__ __forceinline__ bool pool_item_ready(const PersistArgs& a, const int4 it, int d, bool* lanes) {
    const int nb = a.nb;
    const int* factored = a.sync + DF_FACT;
    const int* panel_done = a.sync + DF_FACT + 2 * nb;
    const int* xdone = a.sync + DF_FACT + 2 * nb + nb * nb;
    const int* upd_done = a.sync + DF_FACT + 3 * nb + 2 * nb * nb;
    const int* xprog = upd_done + nb * nb;
    const int* xprog2 = xprog + nb;
    const int i = it.y, j = it.z;
    *lanes = false;
    // up to four flags, each with the value it must reach; unused ones point at a word that always passes
    const int* p0 = nullptr; const int* p1 = nullptr; const int* p2 = nullptr; const int* p3 = nullptr;
    int t0 = 1, t1 = 1, t2 = 1, t3 = 1;
    if (it.x == 0) {
        const int k = j, typ = it.w;
        const int target = i == k ? k - 1 : k;
        if (d < target) {
            const int j0 = d, j1 = df_chunk_end(k, j0, target, a.nbo, a.near);
            if (FUSE && a.nchain == 3 && i == k + 1 && typ != 0 && j1 == target && j1 - j0 == 1) {
                p0 = xprog + j0; t0 = 4;
                p1 = xprog2 + j0; t1 = 4;
            } else if (j1 - j0 <= 2) {
                p0 = panel_done + i + (long)j0 * nb;
                if (i != k) p1 = panel_done + k + (long)j0 * nb;
                if (j1 - j0 == 2) {
                    p2 = panel_done + i + (long)(j0 + 1) * nb;
                    if (i != k) p3 = panel_done + k + (long)(j0 + 1) * nb;
                }
            } else if (j1 - j0 <= 4) {                      // the update chunks of the large sizes (nbo 3, 4): eight flags, unrolled
                int va[4], vb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool use = q < j1 - j0;
                    va[q] = use ? df_flag(panel_done + i + (long)(j0 + q) * nb) : 1;
                    vb[q] = (use && i != k) ? df_flag(panel_done + k + (long)(j0 + q) * nb) : 1;
                }
                bool r = true;
#pragma unroll
                for (int q = 0; q < 4; ++q) r = r && va[q] != 0 && vb[q] != 0;
                return r;
            } else {
                *lanes = true;
                return false;
            }
        } else if (i > k + 1 && typ != 2) {
            p0 = factored + k; t0 = a.nchain >= 2 ? 3 : 1;
            if (typ == 1) p1 = upd_done + i + (long)k * nb;
        }
    } else if (it.x == 10) {
        p0 = factored + i; t0 = a.nchain >= 2 ? 24 : 1;
    } else if (it.x == 11) {
        const int nterms = i - j - a.inv_plast;
        if (d < nterms) {
            const int k0 = j + d, k1 = min(k0 + a.inv_cx, j + nterms);
            if (k1 - k0 > 2) { *lanes = true; return false; }
            p0 = xdone + k0 + (long)j * nb;
            p1 = panel_done + i + (long)k0 * nb;
            if (k1 - k0 == 2) {
                p2 = xdone + (k0 + 1) + (long)j * nb;
                p3 = panel_done + i + (long)(k0 + 1) * nb;
            }
        } else if (d == nterms) {
            p0 = xdone + i + (long)i * nb;
        } else {
            p0 = xdone + (i - 1) + (long)j * nb;
            p1 = xdone + (i - 1) + (long)i * nb;
        }
    } else if (it.x == 14) {
        p0 = xdone + i + (long)i * nb;
        p1 = panel_done + i + (long)j * nb;
    } else {
        const int k0 = i + d, k1 = min(k0 + a.inv_ck, nb);
        if (k1 - k0 > 2) { *lanes = true; return false; }
        p0 = xdone + k0 + (long)i * nb;
        p1 = xdone + k0 + (long)j * nb;
        if (k1 - k0 == 2) {
            p2 = xdone + (k0 + 1) + (long)i * nb;
            p3 = xdone + (k0 + 1) + (long)j * nb;
        }
    }
    const int v0 = p0 ? df_flag(p0) : 0x7fffffff, v1 = p1 ? df_flag(p1) : 0x7fffffff, v2 = p2 ? df_flag(p2) : 0x7fffffff,
              v3 = p3 ? df_flag(p3) : 0x7fffffff;
    return v0 >= t0 && v1 >= t1 && v2 >= t2 && v3 >= t3;
}

__device__ __forceinline__ void potri_team(const PersistArgs& a, int b2, int G2, double* lds, char* smem) {
    const int nb = a.nb;
    auto SW = [&](int arr, int k) -> int& { return reinterpret_cast<int*>(lds + k * DL + 128)[arr]; };
    const int tid = threadIdx.x;
    if (tid == 0) {
        int nt = 0, turn = 0;                             // turn: position of the next item in the round-robin deal
        auto deal = [&](int type, int i, int j) {
            if (turn == b2 && nt < DF_MAXT) {
                SW(0, nt) = i; SW(1, nt) = j; SW(2, nt) = (type == 1 && j == i - 1 && a.inv_plast) ? 1 : 0; SW(3, nt) = 0; SW(6, nt) = type;
                ++nt;
            }
            if (++turn == G2) turn = 0;
        };
        for (int r = 0; r < nb; ++r) {
            deal(0, r, r);
            if (r > 0 && a.inv_plast) deal(4, r, r - 1);
            for (int j = 0; j < r; ++j) deal(1, r, j);
        }
        for (int r = 0; r < nb; ++r)
            for (int j = 0; j <= r; ++j) deal(2, r, j);
        SW(5, 0) = nt;
    }
    __syncthreads();
    const int nt = SW(5, 0);
    if (nt <= 0) return;
    int first = 0;
    long long t_progress = wall_clock64();
    const int slot = tid >> 4, l = tid & 15;
    long long st_task = 0, st_t0 = t_progress, st_n = 0, st_last = 0;       // probes: ticks in tasks, task count, end of the last task
    for (;;) {
        while (first < nt && SW(3, first)) ++first;
        if (first >= nt) break;
        {
            const int t = first + slot;
            const bool valid = t < nt && !SW(3, t);
            bool ok = true;
            if (valid) {
                const int i = SW(0, t), j = SW(1, t), d = SW(2, t), type = SW(6, t);
                ok = potri_item_ready(a, type, i, j, d, l);
            }
            const unsigned long long m = __ballot(ok);
            const unsigned grp = (unsigned)(m >> (16 * ((tid >> 4) & 3))) & 0xffffu;
            if (l == 0) SW(4, slot) = (valid && grp == 0xffffu) ? 1 : 0;
        }
        __syncthreads();
        int sel = -1;
#pragma unroll
        for (int q = DF_WIN - 1; q >= 0; --q)
            if (SW(4, q)) sel = q;
        __syncthreads();
        if (sel < 0) {
            if (__hip_atomic_load(a.info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            if (wall_clock64() - t_progress > a.timeout) {
                if (tid == 0) __hip_atomic_store(a.info + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __builtin_amdgcn_s_sleep(16);
            continue;
        }
        if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        const long long st_task0 = a.trace ? wall_clock64() : 0;
        const int t = first + sel;
        const int i = SW(0, t), j = SW(1, t), d = SW(2, t), type = SW(6, t);
        const PotriStep step = potri_item_task(a, type, i, j, d, lds, smem);
        if (tid == 0) {
            SW(2, t) = step.d;
            if (step.done) SW(3, t) = 1;
        }
        __syncthreads();
        t_progress = wall_clock64();
        if (a.trace) { st_task += t_progress - st_task0; ++st_n; st_last = t_progress; }
    }
    if (a.trace && tid == 0) {
        long long* o = a.trace + 16 * (long)nb + 16 * (long)(a.g1 - 1 + b2);
        o[0] = st_task; o[2] = st_last - st_t0; o[3] = st_n; o[6] = -nt; o[9] = st_t0; o[10] = st_last;
    }
}

// FUSE: the instantiation that carries the three-workgroup chain and the streamed last updates (SLS_POTRF_FUSE_SYRK; two kernels
// rather than a run-time branch: the two chain forms' hoisted per-lane addresses together did not fit the registers)
template <bool FUSE>
__global__ __launch_bounds__(256) void potrf_dataflow_kernel(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const bool inv_wg = a.g1 > 0 && (int)blockIdx.x >= a.g1;      // a workgroup of the inverse's team (nprob == 1)
    const bool pool = a.pool_n > 0;
    const int b_lin = (int)blockIdx.x - a.nchain;                  // index among the workgroups outside the chain
    const bool near_wg = pool && a.pool_near >= 0 && b_lin >= 0 && b_lin < a.pool_near_w;   // owns tiles next to the diagonal, statically
    const bool pool_wg = pool && !near_wg;
    if (inv_wg && !pool) {
        potri_team(a, (int)blockIdx.x - a.g1, (int)gridDim.x - a.g1, lds, smem);
        return;
    }
    // several independent problems share the launch: this workgroup's problem, its rank inside it, the problem's buffers
    const int G = a.g1 > 0 ? a.g1 : gridDim.x / a.nprob, q = inv_wg ? 0 : blockIdx.x / G, b = inv_wg ? G : blockIdx.x - q * G, nb = a.nb, nbo = a.nbo;
    const int b2 = (int)blockIdx.x - a.g1;   // inverse team: this workgroup
    a.A += q * a.strideA;
    a.Linv += q * a.strideA;
    a.sync += q * a.stride_sync;
    a.info += 2 * q;
    if (a.trace) a.trace += q * (16L * nb + 16L * G);
    const long ld = a.ld;
    int* factored = a.sync + DF_FACT;
    int* chain_ready = a.sync + DF_FACT + nb;
    int* panel_done = a.sync + DF_FACT + 2 * nb;        // [i + j nb]
    // chain_ready[j]: tile (j+1, j) carries every update its owner(s) apply (1, or 2 with one owner per half); diag_ready[j]: tile
    // (j+1, j+1) does.  Separate words: in the streamed form the solve only needs the first -- the diagonal tile's last update
    // comes ~6 us later and is only needed behind the solve
    int* diag_ready = a.sync + DF_FACT + 2 * nb + 2 * nb * nb;
    int* upd_done = diag_ready + nb;                    // [i + k nb]: the second half of tile (i, k) carries all its updates
    int* xprog = upd_done + nb * nb;                    // [j]: 4 (s + 1) = the chain's tile (j+1, j) is in global memory up to column block s (FUSE)
    int* xprog2 = xprog + nb;                           // [j]: the same for the tile (j+2, j) right below it (a worker's)
    const int nchain = a.nchain;
    if constexpr (FUSE) {
        if (b < 3 && nchain == 3) {
            // ---- the chain, THREE workgroups in rotation (round 6) ----
            // While workgroup F factors diagonal block j (publishing its column blocks), S solves the panel tile (j+1, j) behind it
            // block by block and stores every finished block at once (stream_trsm_publish), and Y accumulates A_{j+1,j+1} - X X^T
            // from those blocks as they land (stream_syrk_image): when block j ends, what remains before block j+1 can start is the
            // last block of X (eight products + its store), its arrival at Y and ONE slab of the product -- not the whole product
            // (10.7 us on one CU in the two-workgroup form).  Y then factors block j+1 from its own LDS image, F becomes the next S, S the
            // next Y.  Workgroup c: blocks c, c + 3, ... (Y, then F), then the solve of tile (j+2, j+1) behind block j+1.
            for (int j = (b == 2 ? -1 : b); j <= nb - 1; j += 3) {
                if (j >= 0) {
                    double* Ajj = a.A + (long)j * NB * (ld + 1);
                    double* Tjj = a.Linv + (long)j * NB * (ld + 1);
                    if (j == 0) diag_block_factor<true>(Ajj, ld, Tjj, ld, a.info, 0, smem, factored + 0);
                    else {
                        long long* stamp = (a.trace && threadIdx.x == 0) ? a.trace + 16 * (j - 1) : nullptr;
                        if (!stream_syrk_image(Ajj - (long)NB * ld, ld, xprog + (j - 1), a.info + 1, a.timeout, Ajj, ld,
                                               [&]() { return stream_wait(diag_ready + (j - 1), 1, a.info + 1, a.timeout); }, stamp, lds))
                            return;
                        __syncthreads();
                        if (stamp) stamp[3] = wall_clock64();
                        diag_block_factor<false>(Ajj, ld, Tjj, ld, a.info, j * NB, smem, factored + j);
                        if (stamp) stamp[4] = wall_clock64();
                    }
                }
                const int k = j + 1;                                   // the solve of tile (k+1, k) behind diagonal block k
                if (k + 1 <= nb - 1) {
                    double* Akk = a.A + (long)k * NB * (ld + 1);
                    double* Tkk = a.Linv + (long)k * NB * (ld + 1);
                    long long* stamp = (a.trace && threadIdx.x == 0) ? a.trace + 16 * k : nullptr;
                    if (stamp) stamp[10] = wall_clock64();
                    if (!pk_wait_count(chain_ready + k, 1 + a.split_sub, a)) return;   // tile (k+1, k) carries its owners' updates
                    if (stamp) stamp[11] = wall_clock64();
                    if (!stream_trsm_publish(Akk + NB, ld, Akk, ld, Tkk, ld, factored + k, a.info + 1, a.timeout, xprog + k, lds)) return;
                    df_publish_store(panel_done + (k + 1) + (long)k * nb);    // the last block's stores have drained (raise(8)); the barrier frees the LDS
                    if (stamp) stamp[13] = wall_clock64();
                }
            }
            return;
        }
    }
    if (b < 2 && nchain == 2) {
        // ---- the chain, streamed form: TWO workgroups taking turns ----
        // Workgroup c factors the diagonal blocks j = c, c + 2, ... and publishes their column blocks while it does (StreamPublish).
        // Meanwhile the OTHER one solves the panel tile (j+1, j) block by block behind it (stream_trsm): when diagonal block j ends,
        // seven eighths of that solve are done; it then stores the tile for the workers, forms A_{j+1,j+1} - L L^T in place in its
        // LDS image (the tile's owner-updated values wait in registers) and factors diagonal block j+1 from there -- the next
        // diagonal block never leaves the CU that built it.  (First form of this session: a chain that only factored and
        // multiplied, and a follower that only solved: the panel tile went follower -> global memory -> chain, 3.9 us for the
        // store, its drain and the flag + 1.4 us for the load, per step.)
        if (b == 0) diag_block_factor<true>(a.A, ld, a.Linv, ld, a.info, 0, smem, factored + 0);
        for (int j = 1 - b; j <= nb - 2; j += 2) {
            double* Ajj = a.A + (long)j * NB * (ld + 1);
            double* Tjj = a.Linv + (long)j * NB * (ld + 1);
            double* Asub = Ajj + NB;                           // tile (j+1, j)
            double* Anext = Ajj + (long)NB * (ld + 1);         // tile (j+1, j+1)
            PK_STAMP(10);
            if (!pk_wait_count(chain_ready + j, 1 + a.split_sub, a)) return;   // tile (j+1, j) carries its owners' updates (steps < j)
            PK_STAMP(11);
            int* xflag = panel_done + (j + 1) + (long)j * nb;
            {
                ChainAcc ca;
                if (!stream_trsm(ca, Asub, ld, Ajj, ld, Tjj, ld, factored + j, a.info + 1, a.timeout, lds)) return;
                PK_STAMP(12);
                PK_STAMP(0);                                   // diagonal block j has ended (on the other workgroup) a moment ago
                chain_acc_to_image<true>(ca, lds);
            }
            int* ctl = reinterpret_cast<int*>(lds + 128 * DL);               // a word behind the image (the T16 buffers of the solve: dead)
            if (threadIdx.x == 0) *ctl = df_flag(diag_ready + j) >= 1 ? 1 : 0;   // one decision for the workgroup, see below
            lds_barrier();
            chain_image_store_wt(Asub, ld, lds);
            {
                // The diagonal tile's owner finished its updates long since (its last one comes ~6 us after the sub-diagonal
                // tile's, i.e. ~10 us before this point): a relaxed look at its flag, the full wait only if it is not up yet.
                const int up = *ctl;
                if (!up && !pk_wait_count(diag_ready + j, 1, a)) return;
            }
            // The product starts at once: the stores have left the image (it is only overwritten behind the product), and their
            // drain -- 3.3 us for 128 KB of write-through stores + the 64 KB of loads above -- would sit on the chain's path.  The
            // flag for the workers goes up behind the product's second slab, when the drain is over anyway.
            PK_STAMP(1);
            PK_STAMP(2);
            // image <- A_{j+1,j+1} - L L^T on the lower blocks (the tile comes in the accumulator layout, agent-scope loads in flight
            // behind the product; what stands above the diagonal blocks is never read by the factorisation)
            chain_syrk_inplace(lds, [&](int sblk) {
                if (sblk == 2) {
                    df_publish_store(xflag);
                    PK_STAMP(13);
                }
            }, Anext, ld);
            PK_STAMP(8);
            PK_STAMP(9);
            __syncthreads();
            PK_STAMP(3);
            diag_block_factor<false>(Anext, ld, Tjj + (long)NB * (ld + 1), ld, a.info, (j + 1) * NB, smem, factored + j + 1);
            PK_STAMP(4);
        }
        return;
    }
    if (b == 0) {
        // ---- the chain ----
        // Everything between two diagonal blocks stays in LDS: the panel tile L_{j+1,j} = A_{j+1,j} T_jj^T goes accumulators ->
        // LDS image -> global (write-through, coalesced) and is the LDS-resident operand of the next product; A_{j+1,j+1} -
        // L L^T is formed in place in the image, which is diag_block's input.  (The barrier form stores and re-loads both
        // tiles through global memory with 8-byte accesses 64 KB apart: 34 us per step for 15 us of MFMA work.)
        diag_block_factor<true>(a.A, ld, a.Linv, ld, a.info, 0, smem);
        df_publish_store(factored + 0);
        for (int j = 0; j <= nb - 2; ++j) {
            double* Ajj = a.A + (long)j * NB * (ld + 1);
            double* Tjj = a.Linv + (long)j * NB * (ld + 1);
            double* Asub = Ajj + NB;                           // tile (j+1, j)
            double* Anext = Ajj + (long)NB * (ld + 1);         // tile (j+1, j+1)
            PK_STAMP(0);
            if (!pk_wait_count(chain_ready + j, 1, a)) return; // both tiles carry their owners' updates (steps < j)
            if (!pk_wait_count(diag_ready + j, 1, a)) return;
            PK_STAMP(1);
            {
                ChainAcc ca;
                const double* Ts = lds + 128 * DL;                                    // the 16 x 16 inverses of block j, still in LDS
                chain_trsm(ca, Asub, ld, Ajj, ld, [&](int s, int n, int k) { return Ts[256 * s + n + 16 * k]; }, lds);
                PK_STAMP(5);                                                          // (5 .. 9: fine stamps, POTRF_BENCH_FINE=1 in the probe)
                chain_acc_to_image<true>(ca, lds);
            }
            lds_barrier();
            PK_STAMP(6);
            chain_image_store_wt(Asub, ld, lds);
            PK_STAMP(7);
            df_publish_store(panel_done + (j + 1) + (long)j * nb);                    // drains every wave's stores
            PK_STAMP(2);
            chain_syrk_inplace(lds);                                                  // image <- L_{j+1,j} L_{j+1,j}^T, operand = the image
            PK_STAMP(8);
            lds_barrier();
            PK_STAMP(9);
            chain_image_rsub(Anext, ld, lds);                                         // image = A_{j+1,j+1} - L L^T, zero above
            __syncthreads();
            PK_STAMP(3);
            diag_block_factor<false>(Anext, ld, Tjj + (long)NB * (ld + 1), ld, a.info, (j + 1) * NB, smem);
            df_publish_store(factored + j + 1);
            PK_STAMP(4);
        }
        return;
    }
    // ---- the workers ----
    // scheduler state in the padding rows of the LDS image (rows 128 .. 143 of column k: 32 ints that no tile image, operand slab
    // or staging buffer ever touches): word `arr` of column k.  The 16 KB behind the image hold the diagonal block's 16 x 16 inverses.
    auto SW = [&](int arr, int k) -> int& { return reinterpret_cast<int*>(lds + k * DL + 128)[arr]; };
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int W = G - nchain;
        int ok = 1, widx = -1, nt = 0;
        if (pool) {
            // every workgroup outside the chain is a pool worker: which XCD it sits on decides its pool; all of them meet once so that
            // every pool is known to have workers (a pool without any would never finish: give up, the caller repeats on the
            // multi-launch schedule)
            unsigned x;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
            const int mypool = (int)(x & 7) % a.pool_nx;
            if (pool_wg) __hip_atomic_fetch_add(a.sync + 8 + mypool, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a.sync + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            ok = pk_spin(a.sync + 1, (int)gridDim.x - nchain, a.info + 1, a.timeout) ? 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int q = 0; ok && q < a.pool_nx; ++q)
                if (__hip_atomic_load(a.sync + 8 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __hip_atomic_store(a.info + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                }
            SW(7, 0) = mypool;
        } else if (!inv_wg) {
            // worker index, XCD by XCD
            unsigned x;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
            const int xcc = (int)(x & 7);
            // the tile -> owner map must be a bijection: every worker's count increment is ordered before its arrival (release) and
            // the counts are read behind the rendezvous (acquire)
            const int rank = __hip_atomic_fetch_add(a.sync + 8 + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a.sync + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            ok = pk_spin(a.sync + 1, G - nchain, a.info + 1, a.timeout) ? 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            widx = rank;
            for (int q = 0; q < xcc; ++q) widx += __hip_atomic_load(a.sync + 8 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int band = a.split_band;
        if (pool_wg) {
            // nothing is dealt: the lists are the pool's
        } else if (near_wg) {
            // the tiles next to the diagonal (and their second halves), round-robin over the near workers in the usual order
            int cn = 0;
            for (int k = 0; ok && k < nb; ++k) {
                const int c = nb - k + min(band, nb - 1 - k);
                for (int e = k == 0 ? 1 : 0; e < c; ++e) {
                    const bool second = e >= nb - k;
                    const int i = second ? k + 1 + (e - (nb - k)) : k + e;
                    if (i - k > a.pool_near) continue;
                    if (cn == b_lin && nt < DF_MAXT) {
                        SW(0, nt) = i; SW(1, nt) = k; SW(2, nt) = 0; SW(3, nt) = 0;
                        SW(6, nt) = second ? 2 : (e >= 1 && e <= band ? 1 : 0);
                        ++nt;
                    }
                    if (++cn == a.pool_near_w) cn = 0;
                }
            }
        } else if (ok) {
            // tiles in column-major order (without (0, 0)) dealt round-robin: even in total work and at every stage (a cyclic
            // PR x PC owner grid left 1.4x the mean work on some owners: 5.7 vs 5.0 ms at N = 8192, round 3)
            // split_band: column k carries extra items, the second halves of its tiles (k+1, k) .. (k+band, k)
            int k = 0;
            long off = 0;                                    // items before column k
            auto cnt = [&](int kk) { return nb - kk + min(band, nb - 1 - kk); };
            for (long t = widx; nt < DF_MAXT; t += W) {
                const long tt = t + 1;                       // skip (0, 0)
                while (k < nb && tt >= off + cnt(k)) { off += cnt(k); ++k; }
                if (k >= nb) break;
                const int e = (int)(tt - off);
                const bool second = e >= nb - k;             // an extra item: second half of tile (k + 1 + (e - (nb - k)), k)
                SW(0, nt) = second ? k + 1 + (e - (nb - k)) : k + e; SW(1, nt) = k; SW(2, nt) = 0; SW(3, nt) = 0;
                SW(6, nt) = second ? 2 : (e >= 1 && e <= band ? 1 : 0);        // 0 whole tile, 1 / 2: columns 0-63 / 64-127
                ++nt;
            }
        }
        SW(5, 0) = ok ? nt : -1;
        SW(13, 0) = 0;   // the inverse's item tables (arrays 8 .. 14): slot 0 carries a pool's claimed item; nothing is dealt statically here
    }
    __syncthreads();
    int nt = SW(5, 0), nt2 = SW(13, 0);
    if (nt < 0 || (!pool_wg && nt == 0 && nt2 == 0)) return;
    int first = 0, first2 = 0;
    // dynamic pool: this workgroup's pool, the length of its list, the first entry not yet seen finished, the item in hand
    const int mypool = pool_wg ? SW(7, 0) : 0;
    const int pool_len = pool_wg ? (a.pool_n - mypool + a.pool_nx - 1) / a.pool_nx : 0;
    int pool_first = 0, claimed = -1;
    long long t_progress = wall_clock64();
    const int slot = tid >> 4, l = tid & 15;
    // optional statistics (probes): ticks of the 100 MHz clock in tasks / in scheduling rounds that found work / idle, task counts
    long long st_task = 0, st_idle = 0, st_t0 = t_progress, st_n_upd = 0, st_n_panel = 0, st_rounds = 0, st_gemm = 0, st_rmw = 0;
    long long st2_task = 0, st2_n = 0, st2_last = 0;           // probes: the inverse's tasks of this workgroup
    long long st_lost = 0, st_lost_t = 0, st_cas = 0;          // probes (pools): rounds that saw ready items and lost every race, their ticks, claim attempts
    // one lane's share of "has the next task of the factorisation's item (i, k) at progress d its inputs?" (16 lanes cover an item)
    auto fac_ready_lane = [&](int i, int k, int d, int typ, int l) -> bool {
        bool ok = true;
        const int target = i == k ? k - 1 : k;   // steps the owner applies (the chain applies step k-1 to (k, k))
        if (d < target) {
            const int j0 = d;
            const int j1 = df_chunk_end(k, j0, target, nbo, a.near);
            const int j = j0 + (l & 7);
            // the last update of a sub-diagonal half tile runs behind the chain's solve of its second operand (FUSE): the
            // first block of that tile instead of the whole
            const bool streamed = FUSE && nchain == 3 && i == k + 1 && typ != 0 && j1 == target && j1 - j0 == 1;
            if (streamed) {
                if (l == 8) ok = df_flag(xprog + j0) >= 4;
                else if (l == 0) ok = df_flag(xprog2 + j0) >= 4;
            } else if (j < j1 && !(l >= 8 && i == k)) ok = df_flag(panel_done + (l < 8 ? i : k) + (long)j * nb) != 0;
        } else if (i > k + 1 && typ != 2) {
            // the diagonal block (round-3 chain), or its first column block (3 = one arrival per publishing wave) in the streamed
            // form, where the tile is solved block by block behind it; a tile with two owners: the other half
            const int need = nchain >= 2 ? 3 : 1;
            if (l == 0) ok = df_flag(factored + k) >= need;
            else if (l == 1 && typ == 1) ok = df_flag(upd_done + i + (long)k * nb) != 0;
        }
        return ok;
    };
    for (;;) {
        int sel = -1, sel2 = -1;
        const long long st_round0 = a.trace ? wall_clock64() : 0;
        ++st_rounds;
        if (pool_wg) {
            // ---- dynamic pool: hand back the item of the last task, then claim the next one ----
            if (claimed >= 0) {
                // The item of the last task.  If it is not finished and its NEXT task has its inputs already, the workgroup keeps it
                // and goes on (no hand-back, no scan, no claim: 4-5 us per task; the accumulation chains of the inverse run many
                // chunks in a row late in the launch).  Otherwise: every store of the task has reached the XCD's L2 (the next
                // workgroup to take the item sits on this XCD and reads through an invalidated L1), then ONE word says how far
                // the item is.
                const bool fac_item = nt == 1;
                const int fin = fac_item ? SW(3, 0) : SW(11, 0), dd = fac_item ? SW(2, 0) : SW(10, 0);
                bool go_on = false;
                if (!fin && a.pool_keep) {
                    if (tid == 0) {                          // (its flag loads travel while the task's last stores drain)
                        const int4 it = fac_item ? int4{0, SW(0, 0), SW(1, 0), SW(6, 0)} : int4{SW(14, 0) + 10, SW(8, 0), SW(9, 0), 0};
                        bool lanes = false;
                        const bool r = pool_item_ready<FUSE>(a, it, dd, &lanes);
                        SW(18, 1) = (r && !lanes) ? 1 : 0;
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next task, here or elsewhere, may read what any wave of this one stored
                    __syncthreads();
                    go_on = SW(18, 1) != 0;
                    __syncthreads();
                }
                if (go_on) {
                    if (fac_item) sel = 0; else sel2 = 0;
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(a.pool_state + claimed, fin ? 2 : (dd << 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    claimed = -1;
                }
            }
            if (claimed < 0) {
            // one thread per entry of the pool's list, 256 entries from the first unfinished one: state word and item record, then ALL the
            // flags the item's next task needs, in flight together
            const int m = pool_first + tid;
            int code = 2, w = 0, n = -1;                     // 2 finished (or beyond the end), 1 free and ready, 0 neither
            int4 it = {0, 0, 0, 0};
            if (m < pool_len) {
                n = mypool + m * a.pool_nx;
                w = __hip_atomic_load(a.pool_state + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                it = a.pool_items[n];
                code = (w & 3) == 2 ? 2 : 0;
                if ((w & 3) == 0) {
                    const int d = w ? (w >> 2) : (it.x == 11 ? it.w : 0);      // an untouched X item starts at its initial progress
                    bool lanes = false;
                    bool ok = pool_item_ready<FUSE>(a, it, d, &lanes);
                    if (!lanes) {
                    } else if (it.x == 0) {
                        ok = true;
                        for (int q = 0; q < 16 && ok; ++q) ok = fac_ready_lane(it.y, it.z, d, it.w, q);
                    } else {
                        ok = true;
                        for (int q = 0; q < 16 && ok; ++q) ok = potri_item_ready(a, it.x - 10, it.y, it.z, d, q);
                    }
                    code = ok ? 1 : 0;
                }
            }
            // first entry that is not finished (the list's new start): per wave by ballot, over the waves in LDS
            const unsigned long long b_nf = __ballot(code != 2);
            unsigned long long b_rd = __ballot(code == 1);
            const int wbase = tid & ~63;
            if ((tid & 63) == 0) SW(16, tid >> 6) = b_nf ? wbase + __builtin_ctzll(b_nf) : 1 << 20;
            __syncthreads();
            const int f_nf = min(min(SW(16, 0), SW(16, 1)), min(SW(16, 2), SW(16, 3)));
            if (f_nf >= (1 << 20)) {                         // the whole window is finished
                __syncthreads();
                pool_first += 256;
                if (pool_first >= pool_len) break;           // ... and it was the last one: this pool is done
                continue;
            }
            // claim the first ready entry; if another workgroup of the pool was faster, the next ready one of the same scan
            bool got = false, any_ready = false;
            for (;;) {
                if ((tid & 63) == 0) SW(17, tid >> 6) = b_rd ? wbase + __builtin_ctzll(b_rd) : 1 << 20;
                __syncthreads();
                const int f_rd = min(min(SW(17, 0), SW(17, 1)), min(SW(17, 2), SW(17, 3)));
                if (f_rd >= (1 << 20)) { __syncthreads(); break; }
                any_ready = true;
                ++st_cas;
                if (tid == f_rd) {                           // the thread that looked at the entry claims it, with the word it saw
                    int expect = w;
                    const bool won = __hip_atomic_compare_exchange_strong(a.pool_state + n, &expect, w | 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                          __HIP_MEMORY_SCOPE_AGENT);
                    SW(18, 0) = won ? n : -1;
                    if (won) {
                        const int d = w ? (w >> 2) : (it.x == 11 ? it.w : 0);
                        if (it.x == 0) { SW(0, 0) = it.y; SW(1, 0) = it.z; SW(2, 0) = d; SW(3, 0) = 0; SW(6, 0) = it.w; SW(19, 0) = 1; }
                        else { SW(8, 0) = it.y; SW(9, 0) = it.z; SW(10, 0) = d; SW(11, 0) = 0; SW(14, 0) = it.x - 10; SW(19, 0) = 0; }
                    }
                }
                if ((f_rd & ~63) == wbase) b_rd &= ~(1ull << (f_rd & 63));   // not again in this scan
                __syncthreads();
                claimed = SW(18, 0);
                __syncthreads();
                if (claimed >= 0) { got = true; break; }
            }
            pool_first += f_nf;
            if (got) {
                if (SW(19, 0)) { nt = 1; nt2 = 0; first = 0; sel = 0; }
                else { nt = 0; nt2 = 1; first2 = 0; sel2 = 0; }
            } else {
                if (__hip_atomic_load(a.info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                if (wall_clock64() - t_progress > a.timeout) {
                    if (tid == 0) __hip_atomic_store(a.info + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return;
                }
                if (!any_ready) __builtin_amdgcn_s_sleep(32);   // nothing was ready (after lost races: look again at once)
                if (a.trace) {
                    const long long dt = wall_clock64() - st_round0;
                    st_idle += dt;
                    if (any_ready) { ++st_lost; st_lost_t += dt; }
                }
                continue;
            }
            }   // claimed < 0
        } else {
        while (first < nt && SW(3, first)) ++first;            // uniform: every thread reads the same LDS words
        if (first >= nt) break;
        // ---- which tasks have their inputs? 16 lanes per tile, one flag per lane ----
        if (first < nt) {
            const int t = first + slot;
            bool valid = t < nt && !SW(3, t);
            bool ok = true;
            if (valid) ok = fac_ready_lane(SW(0, t), SW(1, t), SW(2, t), SW(6, t), l);
            const unsigned long long m = __ballot(ok);
            const unsigned grp = (unsigned)(m >> (16 * ((tid >> 4) & 3))) & 0xffffu;
            if (l == 0) SW(4, slot) = (valid && grp == 0xffffu) ? 1 : 0;
            __syncthreads();
#pragma unroll
            for (int q = DF_WIN - 1; q >= 0; --q)
                if (SW(4, q)) sel = q;
            __syncthreads();
        }
        if (sel < 0 && sel2 < 0) {
            if (__hip_atomic_load(a.info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            if (wall_clock64() - t_progress > a.timeout) {
                if (tid == 0) __hip_atomic_store(a.info + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __builtin_amdgcn_s_sleep(16);
            if (a.trace) st_idle += wall_clock64() - st_round0;
            continue;
        }
        }   // static lists
        if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // correct by the memory model, not only by first-touch reasoning
        __syncthreads();
        const long long st_task0 = a.trace ? wall_clock64() : 0;
        if (sel < 0) {
            const int t = first2 + sel2;
            const PotriStep step = potri_item_task(a, SW(14, t), SW(8, t), SW(9, t), SW(10, t), lds, smem);
            if (tid == 0) {
                SW(10, t) = step.d;
                if (step.done) SW(11, t) = 1;
            }
            __syncthreads();
            t_progress = wall_clock64();
            if (a.trace) { st2_task += t_progress - st_task0; ++st2_n; st2_last = t_progress; }
            continue;
        }
        const int t = first + sel;
        const int i = SW(0, t), k = SW(1, t), d = SW(2, t);
        const int target = i == k ? k - 1 : k;
        double* Cik = a.A + (long)i * NB + (long)k * NB * ld;
        if (d < target) {
            const int j0 = d;
            const int j1 = df_chunk_end(k, j0, target, nbo, a.near);
            const int half = SW(6, t);
            if (FUSE && nchain == 3 && half != 0 && i == k + 1 && j1 == target && j1 - j0 == 1) {
                if (!stream_update_half(Cik, ld, a.A + (long)i * NB + (long)j0 * NB * ld, a.A + (long)k * NB + (long)j0 * NB * ld, half - 1,
                                        xprog2 + j0, xprog + j0, a.info + 1, a.timeout, lds))
                    return;
                df_publish_add(chain_ready + k);
                if (a.trace && tid == 0 && k >= 1) {
                    a.trace[16 * (k - 1) + 14 + 0] = half == 1 ? st_task0 : a.trace[16 * (k - 1) + 14];
                    if (half == 1) a.trace[16 * (k - 1) + 15] = wall_clock64();
                }
                if (tid == 0) {
                    SW(2, t) = j1;
                    SW(3, t) = 1;
                }
                ++st_n_upd;
                __syncthreads();
                t_progress = wall_clock64();
                if (a.trace) st_task += t_progress - st_task0;
                continue;
            }
            Acc acc;
            acc.zero();
            if (half == 0)
                gemm_tile_deep(acc, a.A + (long)i * NB + (long)j0 * NB * ld, ld, a.A + (long)k * NB + (long)j0 * NB * ld, ld, (j1 - j0) * NB, lds);
            else
                gemm_tile_mc<2, true>(acc, a.A + (long)i * NB + (long)j0 * NB * ld, ld, a.A + (long)k * NB + (long)j0 * NB * ld, ld, 0,
                                      (j1 - j0) * NB, lds, half - 1);
            const long long st_g = a.trace ? wall_clock64() : 0;
            st_gemm += st_g - st_task0;
            const bool last = j1 == target;
            const bool to_chain = last && i <= k + 1;         // (k+1, k) and (k, k) go to the chain after their last update
            if (half != 0) {                                  // a half of a sub-diagonal tile
                if (to_chain) {
                    tile_commit_half<1, true>(Cik, ld, acc, lds, half - 1);
                    df_publish_add(chain_ready + k);
                    if (a.trace && tid == 0 && k >= 1) {      // probes: the last update of tile (k+1, k), first half: begin / end
                        a.trace[16 * (k - 1) + 14 + 0] = half == 1 ? st_task0 : a.trace[16 * (k - 1) + 14];
                        if (half == 1) a.trace[16 * (k - 1) + 15] = wall_clock64();
                    }
                } else if (last && half == 2) {               // second half of a tile that is solved by the first half's owner
                    tile_commit_half<1, true>(Cik, ld, acc, lds, 1);
                    df_publish_store(upd_done + i + (long)k * nb);
                } else {
                    tile_commit_half<1, false>(Cik, ld, acc, lds, half - 1);
                    __syncthreads();
                }
            } else if (to_chain) {
                tile_commit<1, true>(Cik, ld, acc, lds);
                df_publish_add(i == k ? diag_ready + k - 1 : chain_ready + k);
            } else {
                tile_commit<1, false>(Cik, ld, acc, lds);
                if (a.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_rmw += wall_clock64() - st_g; }
                __syncthreads();
            }
            if (tid == 0) {
                SW(2, t) = j1;
                if (to_chain || (last && half == 2)) SW(3, t) = 1;
            }
            ++st_n_upd;
        } else if (i > k + 1 && SW(6, t) == 2) {
            // second half with no update to apply (column 0): report at once
            if (tid == 0) {
                __hip_atomic_store(upd_done + i + (long)k * nb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                SW(3, t) = 1;
            }
        } else if (i > k + 1) {
            ++st_n_panel;
            if (FUSE && nchain == 3 && i == k + 2) {
                // the tile right below the chain's: its blocks leave as they become final and are counted (xprog2): the last update of
                // tile (k+2, k+1) runs behind them and behind the chain's own tile (stream_update_half)
                if (!stream_trsm_publish(Cik, ld, a.A + (long)k * NB * (ld + 1), ld, a.Linv + (long)k * NB * (ld + 1), ld, factored + k, a.info + 1,
                                         a.timeout, xprog2 + k, lds))
                    return;
            } else if (nchain >= 2) {
                const bool stamp = a.trace && tid == 0 && i == k + 2 && !FUSE;      // probes: the panel tile right below the chain (FUSE: the slots carry the solve.s own stamps)
                if (stamp) a.trace[16 * k + 5] = st_task0;
                ChainAcc ca;
                if (!stream_trsm(ca, Cik, ld, a.A + (long)k * NB * (ld + 1), ld, a.Linv + (long)k * NB * (ld + 1), ld, factored + k,
                                 a.info + 1, a.timeout, lds))
                    return;
                if (stamp) a.trace[16 * k + 6] = wall_clock64();
                chain_acc_to_image<true>(ca, lds);
                lds_barrier();
                chain_image_store_wt(Cik, ld, lds);
                if (stamp) a.trace[16 * k + 7] = wall_clock64();
            } else {
                const double* Lkk = a.A + (long)k * NB * (ld + 1);
                const double* Tkk = a.Linv + (long)k * NB * (ld + 1);              // its diagonal 16 x 16 tiles hold the small inverses
                double* Ts = lds + 128 * DL;                                       // the eight small inverses -> LDS (16 KB)
                for (int q2 = tid; q2 < 1024; q2 += 256) {
                    const int t16i = q2 >> 7, c = (q2 >> 3) & 15, r2 = 2 * (q2 & 7);
                    *reinterpret_cast<d2_t*>(Ts + 256 * t16i + r2 + 16 * c) = *reinterpret_cast<const d2_t*>(Tkk + (16 * t16i + r2) + (long)(16 * t16i + c) * ld);
                }
                __syncthreads();
                ChainAcc ca;
                chain_trsm(ca, Cik, ld, Lkk, ld, [&](int s, int n, int kq) { return Ts[256 * s + n + 16 * kq]; }, lds);
                chain_acc_to_image<true>(ca, lds);
                lds_barrier();
                chain_image_store_wt(Cik, ld, lds);
            }
            df_publish_store(panel_done + i + (long)k * nb);
            if (tid == 0) SW(3, t) = 1;
        } else {
            // no update to apply at all: tiles (1, 0) and (1, 1)
            if (tid == 0) {
                __hip_atomic_fetch_add(i == k ? diag_ready + k - 1 : chain_ready + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                SW(3, t) = 1;
            }
        }
        __syncthreads();
        t_progress = wall_clock64();
        if (a.trace) st_task += t_progress - st_task0;
    }
    if (a.trace && tid == 0) {
        if (inv_wg) {                                           // potri_team's record (o[6] < 0), busy time of both kinds of tasks
            long long* o = a.trace + 16 * (long)nb + 16 * (long)(a.g1 - 1 + b2);
            const long long t_end = wall_clock64();
            o[0] = st_task + st2_task; o[2] = t_end - st_t0; o[3] = st2_n + st_n_upd + st_n_panel; o[6] = -(nt + nt2); o[9] = st_t0; o[10] = st2_last ? st2_last : t_end;
            o[11] = st_task; o[12] = st_n_upd + st_n_panel;
        } else {
            long long* o = a.trace + 16 * (long)nb + 16 * (long)(b - 1);
            o[0] = st_task; o[1] = st_idle; o[2] = wall_clock64() - st_t0; o[3] = st_n_upd; o[4] = st_n_panel; o[5] = st_rounds; o[6] = nt; o[7] = st_gemm; o[8] = st_rmw;
            o[11] = st_lost; o[12] = st_lost_t; o[13] = st_cas;
        }
    }
}

// The single-launch factorisation needs ALL its workgroups resident at once (one per CU, the CU's whole LDS).  Two of them in
// flight on one device -- two contexts on the same GPU (sls_multi with a repeated device, one context per host thread) -- would
// each get part of the chip, spin for the rest until the bounded waits expire and fall back to the multi-launch schedule.  They
// are therefore ordered per device ON THE DEVICE: a launch on another stream than the previous one first makes its stream wait
// for that launch's completion event (no host synchronisation; the mutex only covers wait + launch + record).  Processes
// sharing a GPU are not covered; there the fallback + re-arm path takes over.
namespace {
struct PersistSerial {
    std::mutex m;
    hipEvent_t done = nullptr;
    hipStream_t last = nullptr;
    bool recorded = false;
    int n_cu = 0, resident_per_cu = -1;
};
PersistSerial& persist_serial_of_current_device() {
    static std::mutex mtx;
    static std::map<int, std::unique_ptr<PersistSerial>> table;   // one entry per device ordinal, never aliased
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mtx);
    auto& e = table[dev];
    if (!e) {
        e.reset(new PersistSerial());
        int v = 0;
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        e->n_cu = v > 0 ? v : 64;
    }
    return *e;
}
struct PersistSerialScope {
    PersistSerial& p;
    hipStream_t s;
    PersistSerialScope(PersistSerial& p_, hipStream_t s_) : p(p_), s(s_) {
        p.m.lock();
        if (p.recorded && p.last != s) (void)hipStreamWaitEvent(s, p.done, 0);   // same stream: already in order
    }
    ~PersistSerialScope() {
        if (!p.done && hipEventCreateWithFlags(&p.done, hipEventDisableTiming) != hipSuccess) p.done = nullptr;
        if (p.done && hipEventRecord(p.done, s) == hipSuccess) {
            p.recorded = true;
            p.last = s;
        } else {
            // no event: the next launch cannot be ordered behind this one on the device -- wait for it here instead
            (void)hipStreamSynchronize(s);
            p.recorded = false;
        }
        p.m.unlock();
    }
};
}  // namespace

// 0: multi-launch schedule, 3: single persistent launch, dataflow form (default; the numbering is historical: 1 was the form
// with grid barriers, 2 a hybrid of persistent panel kernels and side-stream updates -- both removed in round 4).
// Read per call: tests and A/B runs switch within one process.
int potrf_default_mode(int Np) {
    (void)Np;
    return tune(TUNE_POTRF_MODE, SLS_POTRF_MODE_DEFAULT) == 0 ? 0 : SLS_POTRF_MODE_DEFAULT;
}

int potrf_dataflow_nbo(int Np) {
    if (tune(TUNE_POTRF_DNBO, 0) >= 1) return (int)std::min(8L, tune(TUNE_POTRF_DNBO, 0));
    // measured (tools/probes/df_scan2.sh, TRSM chain; ms): N = 8192: nbo 2: 4.31, 3: 4.83, 4: 5.15 (4.36 with near = 4), 8 + near 4:
    // 4.71; N = 16384 (the workers are the limit: fewer read-modify-write passes win): 2: 26.1, 4: 24.0, 8 + near 3: 23.6 (62
    // TFLOP/s = 0.79 of peak); N = 4096: 1: 1.69, 2: 1.83
    // round 6 (profiles/r06_potrf8192_scan.log): with the streamed chain at N = 8192: nbo 2: 4.09, 3 + near 2: 3.97, 4 + near 2: 4.24
    return Np >= 16384 ? 8 : Np >= 8192 ? 3 : 1;
}
// ints of device scratch the dataflow form needs per problem
size_t potrf_dataflow_sync_ints(int Np) {
    const size_t nb = Np / NB;
    return DF_FACT + 7 * nb + 5 * nb * nb;           // factored, chain_ready, panel_done[nb][nb], xdone[nb][nb] (fused inverse), diag_ready, upd_done[nb][nb], xprog, xprog2, pool_state[<= 2 nb^2 + 2 nb]
}
// How many independent Np x Np factorisations share one launch well: a problem keeps the chip busy with about nb^2 / 14.5 workers
// (its ~nb^3 / 6 tile tasks of ~20 us against a chain of nb steps of ~48 us); the rest of the CUs can factor other problems.
// Measured (tools/probes/potrf_bench, POTRF_BENCH_BATCH=1; ms per problem): N = 1024: 0.394 alone, 0.051 with 8 per launch;
// N = 2048: 0.781 / 0.110 (8); N = 4096: 1.60 alone, 0.63 (3), 0.59 (4), 0.54 (8) -- a little oversubscription still pays, hence
// the divisor 24.
int potrf_dataflow_max_problems(int Np) {
    PersistSerial& ps = persist_serial_of_current_device();
    const int nb = Np / NB;
    const int per_problem = 1 + std::max(1, (int)(nb * (double)nb / 24.0));
    return std::max(1, std::min(8, ps.n_cu / per_problem));
}
// The dynamic pools' item table for an nb-block problem: the factorisation's tiles in the workers' order (column by column, halves
// behind the whole tiles of their column), then the inverse's items in potri_team's order.  Built on the host once per (device, nb,
// band, plast) and kept on the device.
static const int4* pool_item_table(int nb, int band, int plast, int near, bool with_inverse, int* n_items) {
    static std::mutex mtx;
    static std::map<std::tuple<int, int, int, int, int, int>, std::pair<int4*, int>> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mtx);
    const auto key = std::make_tuple(dev, nb, band, plast, near, with_inverse ? 1 : 0);
    auto it = cache.find(key);
    if (it != cache.end()) {
        *n_items = it->second.second;
        return it->second.first;
    }
    std::vector<int4> v;
    for (int k = 0; k < nb; ++k) {
        const int c = nb - k + std::min(band, nb - 1 - k);
        for (int e = k == 0 ? 1 : 0; e < c; ++e) {
            const bool second = e >= nb - k;
            const int i = second ? k + 1 + (e - (nb - k)) : k + e;
            if (i - k <= near) continue;                     // owned statically by the near workers
            v.push_back(int4{0, i, k, second ? 2 : (e >= 1 && e <= band ? 1 : 0)});
        }
    }
    for (int r = 0; with_inverse && r < nb; ++r) {
        v.push_back(int4{10, r, r, 0});
        if (r > 0 && plast) v.push_back(int4{14, r, r - 1, 0});
        for (int j = 0; j < r; ++j) v.push_back(int4{11, r, j, (j == r - 1 && plast) ? 1 : 0});
    }
    for (int r = 0; with_inverse && r < nb; ++r)
        for (int j = 0; j <= r; ++j) v.push_back(int4{12, r, j, 0});
    int4* d = nullptr;
    if (hipMalloc(&d, v.size() * sizeof(int4)) != hipSuccess) {
        *n_items = 0;
        return nullptr;
    }
    (void)hipMemcpy(d, v.data(), v.size() * sizeof(int4), hipMemcpyHostToDevice);   // once per key: synchronous
    cache[key] = {d, (int)v.size()};
    *n_items = (int)v.size();
    return d;
}

// nprob problems (A + q strideA, Linv + q strideA, sync + q stride_sync ints, info + 2 q) in one launch.
// false: not applicable (too few blocks / too many tiles per worker / the kernel cannot be resident once per CU) -- the caller
// uses another schedule.
// inv != nullptr (nprob == 1): the fused inverse (potri_team) -- Linv, inv->U and inv->Kinv are complete when the launch ends;
// false if the matrix is too large for it (N > 4096: the chip is busy with the factorisation itself) or too few CUs remain.
static bool launch_potrf_dataflow_impl(hipStream_t s, double* A, int Np, double* Linv, int* info, int* sync, int nprob, long strideA,
                                       long stride_sync, bool block_inverses, long long* trace, const PotriFused* inv) {
    // SLS_POTRF_FUSE_SYRK: the chain as THREE workgroups in rotation (factor / solve / product), the product of the next diagonal tile
    // accumulated from the solve's blocks as they are published instead of in one piece behind it.  Round 6, same box, two-workgroup
    // chain -> three (ms, factor alone / factor + inverse; profiles/r06_potrf_chain3_ab.log): N = 1536: 0.447 / 0.523 -> 0.443 / 0.512,
    // 2048: 0.584 / 0.693 -> 0.570 / 0.666, 2560: 0.757 / 1.046 -> 0.740 / 1.046, 3072: 0.917 / 1.280 -> 0.900 / 1.283, 4096: 1.289 /
    // 2.160 -> 1.254 / 2.19.  The step's serial tail falls from 13 us to 2.3, but the solve now runs unthrottled for 17-20 us per tile
    // and starts 9 us behind the diagonal block (the tile below the chain's needs the column before it complete): 33 us per step against
    // 35.7.  On for 12 <= N / 128 <= 20, where it wins.  (The first form of this round, solve + product in ONE workgroup, lost: 37-49 us.)
    const int nb_ = Np / NB;
    const bool fuse = tune_on(TUNE_POTRF_FUSE_SYRK, nb_ >= 12 && nb_ <= 20);
    ensure_dyn_lds((const void*)potrf_dataflow_kernel<true>, DIAG_LDS_BYTES);
    ensure_dyn_lds((const void*)potrf_dataflow_kernel<false>, DIAG_LDS_BYTES);
    const int nb = Np / NB;
    PersistSerial& ps = persist_serial_of_current_device();
    if (ps.resident_per_cu < 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, potrf_dataflow_kernel<true>, 256, DIAG_LDS_BYTES) != hipSuccess) n = 0;
        ps.resident_per_cu = n;
    }
    if (ps.resident_per_cu < 1 || nprob < 1 || nprob > 8) return false;
    const int n_cu = ps.n_cu;
    const int tiles_whole = nb * (nb + 1) / 2 - 1;
    // (a second chain workgroup that followed the factorisation with a streamed solve was measured again in round 4 -- 0.384 / 0.767 /
    // 1.661 / 4.37 ms against 0.395 / 0.781 / 1.599 / 4.09 ms at N = 1024 / 2048 / 4096 / 8192, profiles/r04_potrf_chain2.log -- and
    // removed: the step is bound by the owners' panel + update path, not by the chain alone)
    // streamed panel tiles: two chain workgroups taking turns (SLS_POTRF_STREAM=0: the round-3 chain, one workgroup that factors, then
    // solves its panel tile, then multiplies).
    // Measured (tools/probes/stream_scan.sh, ms; round-3 chain -> streamed chain + streamed worker solves + half-tile owners):
    // N = 512: 0.202 -> 0.164, 1024: 0.396 -> 0.309, 2048: 0.789 -> 0.61, 3072: 1.199 -> 0.96, 4096: 1.645 -> 1.29; at N = 8192
    // the workers are as busy as the chain (4.16 -> 4.10-4.17): the round-3 form stays from N > 5120
    // Several problems per launch are bound by their workers, not by their chains (8 value-only evaluations at N = 4096: 6.4 ms on
    // the round-3 chain, 7.3 ms streamed): the round-3 form there too.
    // Round 6: N = 8192 .. 16 256 streamed as well, with WHOLE tiles (split 0: the half-tile owners were what made it slower there, 4.63 ms)
    // and three-step update chunks: 4.10 -> 3.97 ms (profiles/r06_potrf8192_scan.log).  Dedicated owners for the band next to the
    // diagonal (a switch since removed) were measured too and lose at every size of the band (4.2-9.4 ms, r06_potrf8192_band_scan.log).
    const bool stream_dflt = nprob == 1 && (nb <= 40 || (Np >= 8192 && Np < 16384));
    // fuse (SLS_POTRF_FUSE_SYRK): three chain workgroups in rotation (factor / solve / product), see potrf_dataflow_kernel
    const int nchain = (int)tune(TUNE_POTRF_STREAM, stream_dflt ? SLS_POTRF_STREAM_DEFAULT : 0) != 0 && nb >= 4 ? (fuse && nprob == 1 ? 3 : 2) : 1;
    // SLS_POTRF_SPLIT = band: the tiles (i, k) with 1 <= i - k <= band have one owner per 64-column half (0: whole tiles only)
    // Measured (tools/probes: ms at N = 2048 / 3072 / 4096): factorisation alone: band 1: 0.655 / 0.992 / 1.363, 4: 0.645 / 0.999 / 1.353,
    // all: 0.654 / 1.000 / 1.345 -- what matters is that every row's cycle is shorter than the chain's step, the solving workgroup then finds
    // its tile 14-22 us before the diagonal block ends at EVERY step (before: -4 .. +6 us at every third one).  With the fused
    // inverse sharing the chip: band 1: 0.742 / 1.284 / 2.158, all: 0.733 / 1.344 / 2.518 (CUs are short from N = 3072).
    const int split_band = nchain >= 2 ? std::max(0, std::min(nb - 1, (int)tune(TUNE_POTRF_SPLIT, nb > 40 ? 0 : (nb <= 16 || !inv) ? nb : 1))) : 0;
    const int split_sub = split_band >= 1 ? 1 : 0;
    int n_second = 0;
    for (int kk = 0; kk < nb; ++kk) n_second += std::min(split_band, nb - 1 - kk);
    const int tiles = tiles_whole + n_second;                                 // items dealt to the workers
    int Gp = std::max(nchain + 1, std::min(n_cu / nprob, nchain + tiles));   // workgroups per problem
    int G2 = 0;
    if (inv) {
        if (nprob != 1 || nb < 3 || nb > 32 || !inv->U || !inv->Kinv) return false;
        // the factorisation's team: the chain bounds a factorisation of this size, ~nb^2 / 14.5 workers keep up with it; the rest of
        // the chip builds the inverse.  Measured (tools/probes/potri_scan.sh; ms, factor + inverse): N = 4096: 96 workers 2.12,
        // 80: 2.22, 112: 2.26, 136: 2.56, 48: 3.05 (separate launches: 2.62); N = 3072: 1.46-1.50 for 80-112 (1.92);
        // N = 2048: 0.84-0.86 for 80-112, 0.89 for 135 (1.18); same bits for every split
        const int w1_dflt = std::min(tiles, std::max(3 * n_cu / 8, 3 * nb));   // (round-4 split)
        const int W1 = std::max(1, std::min(tiles, (int)tune(TUNE_POTRI_W1, w1_dflt)));
        Gp = nchain + W1;
        G2 = n_cu * ps.resident_per_cu - Gp;
        const int items = 2 * nb - 1 + nb * nb;                                         // T, P, X and K^-1 items
        if (G2 < 8 || (items + G2 - 1) / G2 > DF_MAXT) return false;
        G2 = std::min(G2, items);
    }
    const int W = Gp - nchain;
    if (Gp * nprob + G2 > n_cu * ps.resident_per_cu || (tiles + W - 1) / W > DF_MAXT) return false;
    const size_t sync_ints = potrf_dataflow_sync_ints(Np);
    if (nprob > 1 && (size_t)stride_sync < sync_ints) return false;
    for (int q = 0; q < nprob; ++q) (void)hipMemsetAsync(sync + q * stride_sync, 0, sync_ints * sizeof(int), s);
    PersistArgs a;
    a.A = A; a.ld = Np; a.nb = nb; a.Linv = Linv; a.info = info; a.sync = sync;
    a.timeout = (int)tune(TUNE_POTRF_TIMEOUT_TICKS, 20000000);   // 0.2 s of the 100 MHz wall clock (test hook: 1 makes every wait expire)
    a.trace = trace;
    a.nbo = potrf_dataflow_nbo(Np);
    a.near = (int)tune(TUNE_POTRF_DNEAR, Np >= 16384 ? 3 : Np >= 8192 ? 2 : 0);
    a.nprob = nprob; a.strideA = strideA; a.stride_sync = stride_sync;
    a.g1 = inv ? Gp : 0;
    a.U = inv ? inv->U : nullptr;
    a.Kinv = inv ? inv->Kinv : nullptr;
    a.nchain = nchain;
    a.split_sub = split_sub;
    a.split_band = split_band;
    a.pool_items = nullptr; a.pool_n = 0; a.pool_nx = 1; a.pool_state = nullptr; a.pool_near = -1; a.pool_near_w = 0; a.pool_keep = (int)tune(TUNE_POTRI_POOL_KEEP, 1);
    // measured (ms, factor + inverse; two products per row -> one): N = 384: 0.209 -> 0.234, 1024: 0.405 -> 0.411, 2048: 0.759 -> 0.753,
    // 3072: 1.466 -> 1.32, 4096: 2.15 -> 2.15 (there the two teams are short of CUs, not of time on the wavefront)
    a.inv_plast = (int)tune(TUNE_POTRI_PLAST, nb >= 12 ? 1 : 0) != 0 ? 1 : 0;
    a.inv_cx = std::max(1, std::min(8, (int)tune(TUNE_POTRI_CX, nb > 16 ? 2 : 1)));
    a.inv_ck = std::max(1, std::min(8, (int)tune(TUNE_POTRI_CK, nb > 16 ? 2 : 1)));
    const bool pool_inv = inv && tune_on(TUNE_POTRI_POOL, nb >= 19) && W + G2 >= 128;
    // (a pool is served by the workgroups of ITS XCD only: a launch too small to put workers on every XCD -- found by the schedule test at
    // nb = 3: six workgroups, the item of XCD 0's pool never ran and the launch gave up -- keeps the static deal)
    const bool pool_fac = !inv && nprob == 1 && tune_on(TUNE_POTRF_POOL, false) && W >= 128;
    if (pool_inv || pool_fac) {
        // dynamic pools, one per XCD (an item's tiles are then only ever touched through ONE L2: no coherence traffic beyond what the
        // static form has).  Modelled before it was built (tools/potri_sched_sim.py, the measured task durations): static ownership
        // 2017 us at N = 4096 (measured 2150), one pool per XCD 1330-1540 depending on the claim's cost.  Measured (ms, factor + inverse,
        // static -> pools; profiles/r06_potri_pool.log): N = 2048: 0.67 -> 0.77, 2560: 1.06 -> 1.05, 3072: 1.29 -> 1.21, 3584: 1.65 -> 1.39,
        // 4096: 2.16 -> 1.65-1.75.  A scan + claim costs 4-5 us per task, which the short chains of the small sizes cannot hide: on
        // from N = 2432.  Up to N = 3456 the DIAGONAL tiles (whose last update the chain's next block waits for) stay out of the
        // pools, with 32 workgroups that own them statically and nothing else (ms, static / pools / pools + these; same box,
        // profiles/r06_potri_pool.log): 2304: 0.880 / 0.912 / 0.879, 2560: 1.052 / 1.015 / 0.973, 3072: 1.300 / 1.209 / 1.170,
        // 3584: 1.644 / 1.370 / 1.368, 4096: 2.159 / 1.653 / 1.758.
        int n_items = 0;
        a.pool_near = (int)tune(TUNE_POTRI_POOL_NEAR, (inv && nb <= 27) ? 0 : -1);
        a.pool_near_w = a.pool_near >= 0 ? std::max(1, std::min(64, (int)tune(TUNE_POTRI_POOL_NEAR_W, 32))) : 0;
        if (a.pool_near >= 0) {
            // the statically owned tiles must fit their owners' tables (forced switches: a wide band over few owners would not)
            long near_items = 0;
            for (int kk = 0; kk < nb; ++kk) near_items += std::min(a.pool_near, nb - 1 - kk) + 1 + std::min(std::min(split_band, a.pool_near), nb - 1 - kk);
            if ((near_items + a.pool_near_w - 1) / a.pool_near_w > DF_MAXT) { a.pool_near = -1; a.pool_near_w = 0; }
        }
        const int4* tab = pool_item_table(nb, split_band, a.inv_plast, a.pool_near, inv != nullptr, &n_items);
        const ChipGeometry chip = chip_geometry();
        if (tab && n_items > 0 && n_items <= 2 * nb * nb + 2 * nb) {
            a.pool_items = tab;
            a.pool_n = n_items;
            a.pool_nx = std::max(1, std::min(8, chip.n_xcd));
            a.pool_state = sync + DF_FACT + 5 * nb + 3 * nb * nb;   // behind xprog2: 2 nb^2 + 2 nb words (potrf_dataflow_sync_ints)
        }
    }
    {
        PersistSerialScope serial(ps, s);
        // the <true> instantiation carries ONLY the three-workgroup chain: a forced SLS_POTRF_FUSE_SYRK=1 where the chain is not streamed
        // (fewer than four blocks, SLS_POTRF_STREAM=0, several problems per launch) takes the other one
        if (nchain == 3) hipLaunchKernelGGL(potrf_dataflow_kernel<true>, dim3(Gp * nprob + G2), dim3(256), DIAG_LDS_BYTES, s, a);
        else hipLaunchKernelGGL(potrf_dataflow_kernel<false>, dim3(Gp * nprob + G2), dim3(256), DIAG_LDS_BYTES, s, a);
    }
    // T_jj for every diagonal block, off the factorisation's serial chain (callers that only need the factor skip it; the fused
    // inverse builds them itself)
    if (block_inverses && !inv)
        for (int q = 0; q < nprob; ++q) launch_diag_inverse(s, A + q * strideA, Np, Linv + q * strideA);
    return true;
}
bool launch_potrf_dataflow_batch(hipStream_t s, double* A, int Np, double* Linv, int* info, int* sync, int nprob, long strideA,
                                 long stride_sync, bool block_inverses, long long* trace) {
    return launch_potrf_dataflow_impl(s, A, Np, Linv, info, sync, nprob, strideA, stride_sync, block_inverses, trace, nullptr);
}
bool launch_potrf_dataflow(hipStream_t s, double* A, int Np, double* Linv, int* info, int* sync, long long* trace) {
    return launch_potrf_dataflow_impl(s, A, Np, Linv, info, sync, 1, 0, 0, true, trace, nullptr);
}
bool launch_potri_dataflow(hipStream_t s, double* A, int Np, double* Linv, double* U, double* Kinv, int* info, int* sync, long long* trace) {
    const PotriFused inv{U, Kinv};
    return launch_potrf_dataflow_impl(s, A, Np, Linv, info, sync, 1, 0, 0, false, trace, &inv);
}

__global__ __launch_bounds__(256) void diag_inverse_batched_kernel(const double* __restrict__ L, long ld, double* __restrict__ Linv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long off = (long)blockIdx.x * NB * (ld + 1);
    diag_block<false>(const_cast<double*>(L) + off, ld, Linv + off, ld, nullptr, 0, smem);   // FACTOR = false never writes A
}
void launch_diag_inverse(hipStream_t s, const double* L, int Np, double* Linv) {
    ensure_dyn_lds((const void*)diag_inverse_batched_kernel, DIAG_LDS_BYTES);
    hipLaunchKernelGGL(diag_inverse_batched_kernel, dim3(Np / NB), dim3(256), DIAG_LDS_BYTES, s, L, (long)Np, Linv);
}

}  // namespace slsk
