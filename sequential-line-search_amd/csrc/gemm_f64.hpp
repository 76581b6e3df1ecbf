// fp64 MFMA GEMM building block for gfx950 (MI355X / CDNA4).
//
// One workgroup (256 threads = 4 waves, 2x2) computes a 128x128 tile of
//     C[m][n] (+)= sum_k opA[m][k] * opB[n][k]
// with v_mfma_f64_16x16x4_f64.  Each wave owns a 64x64 sub-tile = 4x4 MFMA
// tiles (128 accumulator VGPRs).  Operand slabs of BK=16 k-rows are double
// buffered in LDS, one barrier per slab (64 MFMAs per wave between barriers):
// LDS-direct loads for M-contiguous operands (gemm_tile_mc), global -> regs ->
// LDS for K-contiguous ones.
//
// Both operands are "M x K" style matrices; each may be stored either
//   M-contiguous ("MC"):  elem(m,k) = P[m + k*ld]   (column-major M x K)
//   K-contiguous ("KC"):  elem(m,k) = P[k + m*ld]   (column-major K x M)
// so NT / NN / TN / TT products of column-major matrices are all covered
// without a transposing copy.  LDS images:
//   MC: [16 k][144]  (128 + 16 pad doubles: row stride 1152 B == 128 mod 256 ->
//        the two 16-lane k-rows of a ds_read_b64 half-wave hit disjoint banks)
//   KC: [128 m][18]  (16 + 2 pad doubles: m*18 mod 32 distinct even numbers)
// both 18432 B, conflict-free for the MFMA fragment reads.
//
// The MFMA is issued as D = Bfrag x Afrag so that the 16 lanes sharing
// lane>>4 hold 16 consecutive m (the memory-contiguous direction of C):
//   acc[i][j][r] = C[m = 16 i + (lane & 15)][n = 16 j + (lane >> 4) + 4 r]
// (f64 16x16x4 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg).
//
// All dimensions are multiples of the tile (callers pad).
#pragma once
#include <hip/hip_runtime.h>

namespace slsk {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_BK = 16;
constexpr int GEMM_THREADS = 256;
constexpr int GEMM_LDS_MC_LD = 144;                       // doubles per k-row (MC image)
constexpr int GEMM_LDS_KC_LD = 18;                        // doubles per m-row (KC image)
constexpr int GEMM_LDS_TILE = 16 * 144;                   // doubles per operand slab (= 128*18)
constexpr int GEMM_LDS_BYTES = 4 * GEMM_LDS_TILE * 8;     // A,B x 2 buffers = 73728 B

struct Acc {
    d4_t v[4][4];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
};

// Staging registers for one 128x16 operand slab: 8 doubles per thread.
struct Stage {
    d2_t r[4];
};

template <bool KC>
__device__ __forceinline__ void stage_load(Stage& s, const double* __restrict__ P, long ld, int k0, int tid) {
    // P points at (m0, k = 0) of the operand panel.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        if (!KC) {
            const int k = idx >> 6, m2 = idx & 63;
            s.r[i] = *reinterpret_cast<const d2_t*>(P + (long)(2 * m2) + (long)(k0 + k) * ld);
        } else {
            const int m = idx >> 3, k2 = idx & 7;
            s.r[i] = *reinterpret_cast<const d2_t*>(P + (long)(k0 + 2 * k2) + (long)m * ld);
        }
    }
}

template <bool KC>
__device__ __forceinline__ void stage_store(const Stage& s, double* lds, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        if (!KC) {
            const int k = idx >> 6, m2 = idx & 63;
            *reinterpret_cast<d2_t*>(lds + k * GEMM_LDS_MC_LD + 2 * m2) = s.r[i];
        } else {
            const int m = idx >> 3, k2 = idx & 7;
            *reinterpret_cast<d2_t*>(lds + m * GEMM_LDS_KC_LD + 2 * k2) = s.r[i];
        }
    }
}

template <bool KC>
__device__ __forceinline__ double frag_read(const double* lds, int mbase, int kk, int lane) {
    // fragment element (row = mbase + (lane & 15), k = 4 kk + (lane >> 4))
    if (!KC)
        return lds[(4 * kk + (lane >> 4)) * GEMM_LDS_MC_LD + mbase + (lane & 15)];
    else
        return lds[(mbase + (lane & 15)) * GEMM_LDS_KC_LD + 4 * kk + (lane >> 4)];
}

// 64 lanes x 16 B from global memory straight into 1024 contiguous LDS bytes at l (wave-uniform): global_load_lds_dwordx4
__device__ __forceinline__ void slab_row_to_lds(const double* g, double* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0,
                                     0);
}

// The same with an agent-scope load (sc1): coherent across the XCDs' L2s without an invalidate before it -- for data another
// workgroup published a moment ago (the streamed column blocks of the Cholesky's diagonal block): a `buffer_inv sc1` per block and
// wave instead drops the whole L2 of the XCD every time.
__device__ __forceinline__ void slab_row_to_lds_sc1(const double* g, double* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0,
                                     16);
}

// ---------------------------------------------------------------------------------------------------------------------
// gemm_tile_mc: both operands M-contiguous (the acquisition GEMM, the variance GEMM, the Gram products, most of the
// Cholesky chain).  A k-row of a slab is 128 contiguous doubles = 64 lanes x 16 B, exactly what one LDS-direct load
// (global_load_lds_dwordx4, gfx950) deposits at M0 + 16 B * lane: the slab never passes through VGPRs.  Double buffered
// BK = 16 slabs as before; what round 2 changed, from measurements with tools/probes/gemm_probe_ring.hip (16384 x 8192 x
// 8192: 70.3 -> 73 TFLOP/s, same bits):
//   * fp64 MFMA and the vector ALU do not co-execute on gfx950 (SQ_VALU_MFMA_COEXEC_CYCLES = 0; 24 dependent
//     v_lshl_add_u64 per slab cost the MFMA-only loop 3.2 %): every VALU instruction in the k loop is paid in MFMA
//     cycles.  The loads therefore use the SADDR form -- a uniform row pointer in SGPRs, advanced with scalar adds, plus
//     one constant 32-bit lane offset -- instead of a 64-bit per-lane address computed with VALU adds per load (the
//     compiler only emits that form outside loops, hence the inline assembly), the last slab is peeled (no branch around
//     the loads) and the two LDS buffers alternate at compile time (their offsets fold into the ds_read immediates).
//   * the fragments of the next k-group are read four MFMAs before they are needed, and the barrier that publishes the
//     next slab sits in the MIDDLE of the last k-group: eight MFMAs whose operands are already in registers follow it,
//     behind which the first fragment reads of the new slab complete.
//   * the barrier is a bare s_barrier behind an explicit s_waitcnt (ring_wait_barrier): __syncthreads() adds a workgroup
//     fence for which the compiler drains vmcnt AND places no MFMA across.
// Deeper rings (BK = 8 x 4 stages, loads three slabs ahead) were measured slower (68.4): the loss was never load latency
// (an L2-resident operand set gives the same rate) but issue cycles; see DESIGN.md 8a.
__device__ __forceinline__ void slab_row_to_lds_saddr(const double* srow, unsigned lane_off, unsigned lds_byte_addr) {
    // srow: wave-uniform address of the k-row (SGPR pair); lane_off = 16 * lane; lds_byte_addr: LDS address for lane 0 (M0)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(srow), "s"(lds_byte_addr)
                 : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void ring_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// One BK = 16 slab out of the LDS buffer at cur_off (doubles) while the next slab is loaded into the buffer at nxt_off.  f0
// holds the fragments of k-group 0 on entry and those of the next slab's k-group 0 on exit; Arow / Brow: this wave's first
// k-row of the NEXT slab (wave-uniform).
// CUR = 0 / 1: the buffer is known at compile time (its offsets fold into the ds_read immediates: no VALU address update
// per slab); CUR = -1: cur_rt / nxt_rt say which.
template <int NJ, int CUR>
__device__ __forceinline__ void mc_slab(Acc& acc, double (&f0)[4 + NJ], double (&f1)[4 + NJ], const double* Arow, const double* Brow,
                                        long lda, long ldb, unsigned lane_off, unsigned lds_base, const double* lds, int cur_rt,
                                        int nxt_rt, int wave, int wm, int wn, int lane) {
    const int cur_off = CUR < 0 ? cur_rt : CUR * 2 * GEMM_LDS_TILE;
    const int nxt_off = CUR < 0 ? nxt_rt : (1 - CUR) * 2 * GEMM_LDS_TILE;
    auto issue = [&](int l) {   // load l of the next slab: l < 4 -> A row 4 wave + l, else B row 4 wave + l - 4
        const int r = l & 3;
        const double* g = l < 4 ? Arow + (long)r * lda : Brow + (long)r * ldb;
        const unsigned d = lds_base + 8u * (unsigned)(nxt_off + (l < 4 ? 0 : GEMM_LDS_TILE) + (4 * wave + r) * GEMM_LDS_MC_LD);
        slab_row_to_lds_saddr(g, lane_off, d);
    };
    auto read = [&](double (&f)[4 + NJ], int bufoff, int kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = frag_read<false>(lds + bufoff, wm + 16 * i, kk, lane);
#pragma unroll
        for (int j = 0; j < NJ; ++j) f[4 + j] = frag_read<false>(lds + bufoff + GEMM_LDS_TILE, wn + 16 * j, kk, lane);
    };
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        double(&f)[4 + NJ] = (kk & 1) ? f1 : f0;
        double(&g)[4 + NJ] = (kk & 1) ? f0 : f1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[4 + j], f[i], acc.v[i][j], 0, 0, 0);
                const int m = (kk * 4 + i) * NJ + j + 1;            // MFMAs of this slab issued so far
                if ((m & 1) == 0 && m <= 16) {                      // one load per two MFMAs: all eight within the first 16
                    __builtin_amdgcn_sched_barrier(0);
                    issue(m / 2 - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (kk < 3 && i == 2) {                                 // next k-group's fragments, one row of MFMAs ahead
                __builtin_amdgcn_sched_barrier(0);
                read(g, cur_off, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kk == 3 && i == 1) {                                // next slab landed, this one read by everybody
                __builtin_amdgcn_sched_barrier(0);
                ring_wait_barrier<0>();
                read(g, nxt_off, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// acc += A[m0.., kb..ke) * B[n0.., kb..ke)^T for M-contiguous A, B (pointing at row m0 / n0, k = 0); kb, ke multiples of 16.
// NJ = 4: the whole 128 x 128 tile (wave (wm, wn) owns a 64 x 64 quadrant).  NJ = 2: only the 64 columns [64 nhalf, 64 nhalf
// + 64) of it -- wave (wm, sub) owns 64 x 32 at column 64 nhalf + 32 sub, accumulators acc.v[i][0..1] -- with the same slabs,
// the same fragment layout and the same k order, so every element gets the bits the full tile would give it at half the
// MFMA work per workgroup (used to split the tiles of a partially filled last generation over twice as many workgroups).
// Every slab runs the same code: the last one "prefetches" the first slab again (1/nslab extra traffic, from L2) instead of
// being a peeled copy of the loop body -- peeled tails made the register allocator spill the accumulators.
// `lds`: the 73 728-byte, 16-byte aligned dynamic LDS block; free for reuse on return (all waves have passed a barrier).
// EVEN: the caller guarantees an even number of slabs (ke - kb a multiple of 32): the loop then runs two slabs per iteration
// with compile-time buffers.
template <int NJ = 4, bool EVEN = false>
__device__ __forceinline__ void gemm_tile_mc(Acc& acc, const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                             int kb, int ke, double* lds, int nhalf = 0) {
    if (kb >= ke) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave & 1) * 64;
    const int wn = NJ == 4 ? (wave >> 1) * 64 : 64 * nhalf + (wave >> 1) * 32;
    const unsigned lane_off = 16u * lane;
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) double*)lds;
    const double* A0 = A + (long)(kb + 4 * wave) * lda;       // wave-uniform: this wave's rows of the first slab
    const double* B0 = B + (long)(kb + 4 * wave) * ldb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        slab_row_to_lds_saddr(A0 + (long)r * lda, lane_off, lds_base + 8u * (unsigned)((4 * wave + r) * GEMM_LDS_MC_LD));
        slab_row_to_lds_saddr(B0 + (long)r * ldb, lane_off, lds_base + 8u * (unsigned)(GEMM_LDS_TILE + (4 * wave + r) * GEMM_LDS_MC_LD));
    }
    ring_wait_barrier<0>();
    double f0[4 + NJ], f1[4 + NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i) f0[i] = frag_read<false>(lds, wm + 16 * i, 0, lane);
#pragma unroll
    for (int j = 0; j < NJ; ++j) f0[4 + j] = frag_read<false>(lds + GEMM_LDS_TILE, wn + 16 * j, 0, lane);
    const int nslab = (ke - kb) / GEMM_BK;
    const double* Arow = A0;
    const double* Brow = B0;
    if constexpr (EVEN) {
        for (int s = 0; s < nslab; s += 2) {
            Arow += (long)GEMM_BK * lda;
            Brow += (long)GEMM_BK * ldb;
            mc_slab<NJ, 0>(acc, f0, f1, Arow, Brow, lda, ldb, lane_off, lds_base, lds, 0, 0, wave, wm, wn, lane);
            const bool last = s + 2 == nslab;
            Arow = last ? A0 : Arow + (long)GEMM_BK * lda;
            Brow = last ? B0 : Brow + (long)GEMM_BK * ldb;
            mc_slab<NJ, 1>(acc, f0, f1, Arow, Brow, lda, ldb, lane_off, lds_base, lds, 0, 0, wave, wm, wn, lane);
        }
    } else {
        int cur = 0;
        for (int s = 0; s < nslab; ++s) {
            const bool last = s + 1 == nslab;
            Arow = last ? A0 : Arow + (long)GEMM_BK * lda;
            Brow = last ? B0 : Brow + (long)GEMM_BK * ldb;
            const int nxt = cur ^ (2 * GEMM_LDS_TILE);
            mc_slab<NJ, -1>(acc, f0, f1, Arow, Brow, lda, ldb, lane_off, lds_base, lds, cur, nxt, wave, wm, wn, lane);
            cur = nxt;
        }
    }
    ring_wait_barrier<0>();
}

// Accumulate acc += opA[m0.., kb..ke) * opB[n0.., kb..ke)^T.
// A, B point at row m0 / n0, k = 0 of their panels.  kb, ke multiples of 16.  `lds` is the 73728-byte, 16-byte aligned
// dynamic LDS block.  Both operands M-contiguous: gemm_tile_mc above (NJ / nhalf: see there).  Otherwise (a K-contiguous
// operand cannot use LDS-direct loads: its slab rows are 16 doubles) the slab is staged global -> VGPR -> LDS, double
// buffered, one barrier per slab.
template <bool A_KC, bool B_KC, int NJ = 4, bool EVEN = false>
__device__ __forceinline__ void gemm_tile(Acc& acc, const double* __restrict__ A, long lda,
                                          const double* __restrict__ B, long ldb, int kb, int ke, double* lds, int nhalf = 0) {
    if constexpr (!A_KC && !B_KC) {
        gemm_tile_mc<NJ, EVEN>(acc, A, lda, B, ldb, kb, ke, lds, nhalf);
    } else {
        static_assert(NJ == 4, "half tiles exist for the M-contiguous form only");
        const int tid = threadIdx.x;
        const int lane = tid & 63;
        const int wave = tid >> 6;
        const int wm = (wave & 1) * 64;   // wave's m offset inside the tile
        const int wn = (wave >> 1) * 64;  // wave's n offset
        if (kb >= ke) return;
        // NOTE: LDS buffers are selected by integer offset from the one LDS base pointer.  Selecting between
        // pointers (double* buf[2]) makes hipcc lose the LDS address space and emit flat_load/flat_store, whose
        // s_waitcnt vmcnt(0) then drains the global prefetch before every MFMA group (measured: 72 % -> MFMA busy).
        // layout: [A0 | B0 | A1 | B1], each GEMM_LDS_TILE doubles
        Stage sa, sb;
        stage_load<A_KC>(sa, A, lda, kb, tid);
        stage_load<B_KC>(sb, B, ldb, kb, tid);
        stage_store<A_KC>(sa, lds, tid);
        stage_store<B_KC>(sb, lds + GEMM_LDS_TILE, tid);
        __syncthreads();
        int cur = 0;   // offset (doubles) of the buffer pair being consumed
        for (int k0 = kb; k0 < ke; k0 += GEMM_BK) {
            const bool more = (k0 + GEMM_BK) < ke;
            if (more) {
                stage_load<A_KC>(sa, A, lda, k0 + GEMM_BK, tid);
                stage_load<B_KC>(sb, B, ldb, k0 + GEMM_BK, tid);
            }
            const double* la = lds + cur;
            const double* lb = lds + cur + GEMM_LDS_TILE;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                double af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = frag_read<A_KC>(la, wm + 16 * i, kk, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = frag_read<B_KC>(lb, wn + 16 * j, kk, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc.v[i][j], 0, 0, 0);
            }
            const int nxt = cur ^ (2 * GEMM_LDS_TILE);
            if (more) {
                stage_store<A_KC>(sa, lds + nxt, tid);
                stage_store<B_KC>(sb, lds + nxt + GEMM_LDS_TILE, tid);
            }
            __syncthreads();
            cur = nxt;
        }
    }
}

// 128 x 64 output tile (A M- or K-contiguous, B K-contiguous), for products whose second dimension is small (the gradient
// contractions have n' = D <= 64 for the headline configuration; a 128-wide tile would waste half of its MFMAs).
// 4 waves stacked along m: wave w owns rows 32 w .. 32 w + 31 (2 A fragments) x all 64 columns (4 B fragments).
// LDS per buffer: A slab [16][144] + B slab [64][18] doubles; two buffers = 55296 B -> 2-3 workgroups per CU.
constexpr int GEMM_N64_LDS_B = 64 * GEMM_LDS_KC_LD;                       // 1152 doubles
constexpr int GEMM_N64_LDS_BUF = GEMM_LDS_TILE + GEMM_N64_LDS_B;          // 3456 doubles
constexpr int GEMM_N64_LDS_BYTES = 2 * GEMM_N64_LDS_BUF * 8;              // 55296 B

struct Acc64 {
    d4_t v[2][4];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
};

// A_LD (M-contiguous A only): doubles per k-row of the A slab image.  144 is conflict-free for the fragment reads; 136 (two-way
// conflicts on them) brings two buffers down to 53 248 B, so that THREE workgroups fit a CU's 160 KB: the HBM-bound gradient
// contraction wants bytes in flight, not LDS read bandwidth.
template <int A_LD>
constexpr int gemm_n64_lds_bytes() {
    return 2 * (16 * A_LD + GEMM_N64_LDS_B) * 8;
}
template <bool A_KC = false, int A_LD = GEMM_LDS_MC_LD>
__device__ __forceinline__ void gemm_tile_n64(Acc64& acc, const double* __restrict__ A, long lda,
                                              const double* __restrict__ B, long ldb, int kb, int ke, double* lds) {
    static_assert(!A_KC || A_LD == GEMM_LDS_MC_LD, "the K-contiguous image has its own padding");
    constexpr int A_SLAB = 16 * A_LD;                  // doubles (= GEMM_LDS_TILE for the default)
    constexpr int BUF = A_SLAB + GEMM_N64_LDS_B;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wm = (tid >> 6) * 32;
    if (kb >= ke) return;
    Stage sa;
    d2_t sbr[2];
    auto load_b = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;          // 512 x 16-byte pieces: n' = idx / 8, k pair = idx % 8
            const int n = idx >> 3, k2 = idx & 7;
            sbr[i] = *reinterpret_cast<const d2_t*>(B + (long)(k0 + 2 * k2) + (long)n * ldb);
        }
    };
    auto store_b = [&](double* l) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int n = idx >> 3, k2 = idx & 7;
            *reinterpret_cast<d2_t*>(l + n * GEMM_LDS_KC_LD + 2 * k2) = sbr[i];
        }
    };
    // M-contiguous A (the streamed operand of the gradient contractions): LDS-direct loads as in gemm_tile
    constexpr bool DIRECT_A = !A_KC;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto issue_a = [&](int k0, int bufoff) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * wave_u + r;
            slab_row_to_lds(A + 2 * lane + (long)(k0 + row) * lda, lds + bufoff + row * A_LD);
        }
    };
    if (DIRECT_A) issue_a(kb, 0);
    else stage_load<A_KC>(sa, A, lda, kb, tid);
    load_b(kb);
    if (DIRECT_A) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else stage_store<A_KC>(sa, lds, tid);
    store_b(lds + A_SLAB);
    __syncthreads();
    int cur = 0;
    for (int k0 = kb; k0 < ke; k0 += GEMM_BK) {
        const bool more = (k0 + GEMM_BK) < ke;
        if (more) {
            if (DIRECT_A) issue_a(k0 + GEMM_BK, cur ^ BUF);
            else stage_load<A_KC>(sa, A, lda, k0 + GEMM_BK, tid);
            load_b(k0 + GEMM_BK);
        }
        const double* la = lds + cur;
        const double* lb = lds + cur + A_SLAB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[2], bf[4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = A_KC ? frag_read<true>(la, wm + 16 * i, kk, lane)
                             : la[(4 * kk + (lane >> 4)) * A_LD + wm + 16 * i + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = frag_read<true>(lb, 16 * j, kk, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc.v[i][j], 0, 0, 0);
        }
        const int nxt = cur ^ BUF;
        if (more) {
            if (!DIRECT_A) stage_store<A_KC>(sa, lds + nxt, tid);
            store_b(lds + nxt + A_SLAB);
        }
        if (DIRECT_A) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur = nxt;
    }
}

// Coordinates of accumulator element (i, j, r) inside the 128x128 tile.
__device__ __forceinline__ int acc_m(int i) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave & 1) * 64 + 16 * i + (lane & 15);
}
__device__ __forceinline__ int acc_n(int j, int r) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave >> 1) * 64 + 16 * j + (lane >> 4) + 4 * r;
}

// Walk a 128 x 128 accumulator tile COLUMN BY COLUMN with whole-wave coalesced accesses.  In the accumulator layout a wave
// instruction touches four 128-byte runs 8 * ld bytes apart (16 lanes x 8 bytes each); epilogues that load or store global
// memory that way run at a fraction of the HBM rate (measured in the dataflow Cholesky: 16 us per 128 x 128 read-modify-write,
// 3.7 us once staged).  Here the tile goes through the GEMM's own LDS block (free after the k loop) in two passes of 64 columns
// ([64][144] doubles = 73 728 bytes), and f(col, row, v) is called with v = the tile's elements (row, col), (row + 1, col) for
// row = 2 * lane: a wave covers one 1 KB column per call, 16 bytes per lane.  Values are only moved, never re-associated.
// Every thread calls f 32 times (col = wave + 4 q within each half).
// HEAVY: f carries transcendental math (keep the column loop rolled: unrolled, its temporaries push the accumulators of the
// second half out of the register file).
template <bool HEAVY = false, class F>
__device__ __forceinline__ void acc_tile_by_columns(const Acc& acc, double* lds, F&& f) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if ((wave >> 1) == half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        lds[(16 * j + (lane >> 4) + 4 * r) * GEMM_LDS_MC_LD + (wave & 1) * 64 + 16 * i + (lane & 15)] = acc.v[i][j][r];
        }
        __syncthreads();
        if constexpr (HEAVY) {
#pragma unroll 2
            for (int q = 0; q < 16; ++q) {
                const int c = wave + 4 * q;
                const d2_t v = *reinterpret_cast<const d2_t*>(lds + c * GEMM_LDS_MC_LD + 2 * lane);
                f(64 * half + c, 2 * lane, v);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int c = wave + 4 * q;
                const d2_t v = *reinterpret_cast<const d2_t*>(lds + c * GEMM_LDS_MC_LD + 2 * lane);
                f(64 * half + c, 2 * lane, v);
            }
        }
    }
}

// XCD-aware tile order.  Workgroup b runs on XCD b % 8 (observed, speed only).
// Give each XCD a contiguous run of the linear tile order so that the 64 tiles
// resident on one XCD share operand panels in its private L2.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, within = b >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + within;
}

}  // namespace slsk
