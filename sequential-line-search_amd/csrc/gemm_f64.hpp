// fp64 MFMA GEMM building block for gfx950 (MI355X / CDNA4).
//
// One workgroup (256 threads = 4 waves, 2x2) computes a 128x128 tile of
//     C[m][n] (+)= sum_k opA[m][k] * opB[n][k]
// with v_mfma_f64_16x16x4_f64.  Each wave owns a 64x64 sub-tile = 4x4 MFMA
// tiles (128 accumulator VGPRs).  Operand tiles are staged global -> regs ->
// LDS in BK=16 slabs, double buffered, one barrier per slab (64 MFMAs per wave
// between barriers).
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

// Accumulate acc += opA[m0.., kb..ke) * opB[n0.., kb..ke)^T.
// A, B point at row m0 / n0, k = 0 of their panels.  kb, ke multiples of 16.
// The k loop starts at slab `kfirst` (kb <= kfirst < ke, multiple of 16) and wraps around at ke: tiles that share an
// operand panel are started a slab apart so that their global loads of one slab do not miss the L2 simultaneously.
// `lds` is the 73728-byte, 16-byte aligned dynamic LDS block.
// NJ = 4: the whole 128 x 128 tile (wave (wm, wn) owns a 64 x 64 quadrant).  NJ = 2: only the 64 columns [64 nhalf, 64 nhalf
// + 64) of it -- wave (wm, sub) owns 64 x 32 at column 64 nhalf + 32 sub, accumulators acc.v[i][0..1] -- with the same slabs,
// the same fragment layout and the same k order, so every element gets the bits the full tile would give it at half the
// MFMA work per workgroup (used to split the tiles of a partially filled last generation over twice as many workgroups).
template <bool A_KC, bool B_KC, int NJ = 4>
__device__ __forceinline__ void gemm_tile(Acc& acc, const double* __restrict__ A, long lda,
                                          const double* __restrict__ B, long ldb, int kb, int ke, double* lds,
                                          int kfirst = -1, int nhalf = 0) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (wave & 1) * 64;   // wave's m offset inside the tile
    const int wn = NJ == 4 ? (wave >> 1) * 64 : 64 * nhalf + (wave >> 1) * 32;  // wave's n offset
    if (kb >= ke) return;
    if (kfirst < kb || kfirst >= ke) kfirst = kb;

    // NOTE: LDS buffers are selected by integer offset from the one LDS base pointer.  Selecting between
    // pointers (double* buf[2]) makes hipcc lose the LDS address space and emit flat_load/flat_store, whose
    // s_waitcnt vmcnt(0) then drains the global prefetch before every MFMA group (measured: 72 % -> MFMA busy).
    // layout: [A0 | B0 | A1 | B1], each GEMM_LDS_TILE doubles
    // Both operands M-contiguous: a k-row of a slab is 128 contiguous doubles = 64 lanes x 16 B, exactly what one
    // global_load_lds_dwordx4 (LDS-direct load, gfx950) deposits at LDS base + 16 B * lane.  The slab then never passes
    // through VGPRs: no staging registers (-32 VGPRs), no ds_write, no vmcnt -> ds_write dependency in the MFMA stream
    // (gemm_probe_lds: 70.0 -> 71.2 TFLOP/s at 16384 x 8192 x 8192, 50 -> 56 TFLOP/s at 2048^3).  Wave w brings rows
    // 4w .. 4w+3 of both operands; the loads of slab s+1 are issued at the top of slab s and must have landed
    // (s_waitcnt vmcnt(0)) before the barrier that publishes the buffer.
    constexpr bool DIRECT = !A_KC && !B_KC;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto issue_direct = [&](int k0, int bufoff) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * wave_u + r;
            slab_row_to_lds(A + 2 * lane + (long)(k0 + row) * lda, lds + bufoff + row * GEMM_LDS_MC_LD);
            slab_row_to_lds(B + 2 * lane + (long)(k0 + row) * ldb, lds + bufoff + GEMM_LDS_TILE + row * GEMM_LDS_MC_LD);
        }
    };
    Stage sa, sb;
    if (DIRECT) {
        issue_direct(kfirst, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        stage_load<A_KC>(sa, A, lda, kfirst, tid);
        stage_load<B_KC>(sb, B, ldb, kfirst, tid);
        stage_store<A_KC>(sa, lds, tid);
        stage_store<B_KC>(sb, lds + GEMM_LDS_TILE, tid);
    }
    __syncthreads();

    int cur = 0;   // offset (doubles) of the buffer pair being consumed
    int knext = kfirst;
    const int nslab = (ke - kb) / GEMM_BK;
    for (int s = 0; s < nslab; ++s) {
        const bool more = (s + 1) < nslab;
        knext += GEMM_BK;
        if (knext >= ke) knext = kb;
        if (more) {
            if (DIRECT) {
                issue_direct(knext, cur ^ (2 * GEMM_LDS_TILE));
            } else {
                stage_load<A_KC>(sa, A, lda, knext, tid);
                stage_load<B_KC>(sb, B, ldb, knext, tid);
            }
        }
        const double* la = lds + cur;
        const double* lb = lds + cur + GEMM_LDS_TILE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[4], bf[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = frag_read<A_KC>(la, wm + 16 * i, kk, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = frag_read<B_KC>(lb, wn + 16 * j, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc.v[i][j], 0, 0, 0);
        }
        const int nxt = cur ^ (2 * GEMM_LDS_TILE);
        if (DIRECT) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (more) {
            stage_store<A_KC>(sa, lds + nxt, tid);
            stage_store<B_KC>(sb, lds + nxt + GEMM_LDS_TILE, tid);
        }
        __syncthreads();
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Ring form of the M-contiguous x M-contiguous tile: slabs of BK k-rows in a ring of STAGES LDS stages, the LDS-direct
// loads of slab s + STAGES - 1 issued as soon as the barrier of slab s has retired slab s - 1.  Same fragment layout and
// the same k order as gemm_tile, hence the same bits; what changes is how long a load may take before somebody waits for
// it: gemm_tile waits for slab s + 1 at the end of slab s (~48 MFMAs = 1.3 us after the issue), the ring with BK = 8 and
// four stages waits ~2.75 slabs = 88 MFMAs = 2.4 us after it, with the same 73 728 bytes of LDS (two workgroups per CU).
// The barrier is a bare s_barrier behind an explicit s_waitcnt: __syncthreads() carries a workgroup fence, for which the
// compiler emits s_waitcnt vmcnt(0) -- that drains the ring (it is why the first 4-stage attempt measured "no change").
// lgkmcnt(0): this wave's LDS reads of the retiring slab have returned before anybody's load may overwrite it.
template <int N>
__device__ __forceinline__ void ring_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

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
//     read F1 (slab s, k-group 1) | 16 MFMAs on F0 | wait + barrier (slab s + 1 landed, slab s read by everybody)
//     | issue slab s + 4 into slab s's stage | read F0 (slab s + 1, k-group 0) | 16 MFMAs on F1
// so every fragment read is issued 16 MFMAs (~1000 cycles) before its first use and the instructions behind a barrier are
// MFMAs whose operands are already in registers: the wave has no LDS-latency bubble per slab (gemm_tile and the plain ring
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
    auto mfma = [&](const double (&af)[4], const double (&bf)[NJ]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc.v[i][j], 0, 0, 0);
    };
    for (int s = 0; s < STAGES && s < nst; ++s) issue();
    wait_for(min(STAGES, nst) - 1);
    double a0[4], b0[NJ], a1[4], b1[NJ];
    read(a0, b0, 0, 0);
    int scur = 0;
    for (int s = 0; s < nst; ++s) {
        read(a1, b1, scur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nst) {
            wait_for(min(STAGES - 2, nst - 2 - s));
            if (s + STAGES < nst) issue();         // into slab s's stage
            read(a0, b0, (scur + 1) & (STAGES - 1), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        scur = (scur + 1) & (STAGES - 1);
    }
    ring_wait_barrier<0>();                        // the callers reuse the LDS block in their epilogues
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

template <bool A_KC = false>
__device__ __forceinline__ void gemm_tile_n64(Acc64& acc, const double* __restrict__ A, long lda,
                                              const double* __restrict__ B, long ldb, int kb, int ke, double* lds) {
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
            slab_row_to_lds(A + 2 * lane + (long)(k0 + row) * lda, lds + bufoff + row * GEMM_LDS_MC_LD);
        }
    };
    if (DIRECT_A) issue_a(kb, 0);
    else stage_load<A_KC>(sa, A, lda, kb, tid);
    load_b(kb);
    if (DIRECT_A) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else stage_store<A_KC>(sa, lds, tid);
    store_b(lds + GEMM_LDS_TILE);
    __syncthreads();
    int cur = 0;
    for (int k0 = kb; k0 < ke; k0 += GEMM_BK) {
        const bool more = (k0 + GEMM_BK) < ke;
        if (more) {
            if (DIRECT_A) issue_a(k0 + GEMM_BK, cur ^ GEMM_N64_LDS_BUF);
            else stage_load<A_KC>(sa, A, lda, k0 + GEMM_BK, tid);
            load_b(k0 + GEMM_BK);
        }
        const double* la = lds + cur;
        const double* lb = lds + cur + GEMM_LDS_TILE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[2], bf[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = frag_read<A_KC>(la, wm + 16 * i, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = frag_read<true>(lb, 16 * j, kk, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc.v[i][j], 0, 0, 0);
        }
        const int nxt = cur ^ GEMM_N64_LDS_BUF;
        if (more) {
            if (!DIRECT_A) stage_store<A_KC>(sa, lds + nxt, tid);
            store_b(lds + nxt + GEMM_LDS_TILE);
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
