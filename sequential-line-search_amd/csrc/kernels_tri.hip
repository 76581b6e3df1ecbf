// Tiled triangular products on the fp64 MFMA tile and what is built from them: the multi-launch Cholesky schedule (the fallback
// of the single-launch factorisation of kernels_chol.hip: diagonal block + panel + trailing update per 128-column step), the
// triangular inverse by recursive doubling (X = L^-1 and U = X^T), K^-1 = U U^T (replaces MatrixXd::inverse(),
// src/gaussian-process-regressor.cpp:159,211,231), the dispatch between the fused single launch and these separate launches, and
// the block triangular solves behind CholeskyFactor::solve (Eigen::LLT::solve, src/preference-regressor.cpp:293-330).
// Split off kernels_chol.hip in round 5.
#include <algorithm>
#include <cstdlib>

#include "gemm_f64.hpp"
#include "kernels.hpp"

namespace slsk {

// ---------------------------------------------------------------------------------------------------------
// generic tile GEMM with triangular k-ranges:  C = alpha * opA opB^T + beta * C
// ---------------------------------------------------------------------------------------------------------
struct GemmDesc {
    const double* A; long lda; long strideA;   // batch strides in elements
    const double* B; long ldb; long strideB;
    double* C; long ldc; long strideC;
    int mt, nt;        // tile grid
    int K;             // full k extent (multiple of 16)
    double alpha, beta;
    int tri;           // 1: only tiles tm >= tn + tri_off
    int tri_off;
    int order;         // tile issue order (longest k range first): 0 blockIdx = tm + tn*mt; 1 lower triangle row by row from
                       // tm = 0 (grid = mt (mt + 1) / 2, kmode 3); 2 rows from tm = mt - 1 down (kmode 2); 3 rows from tm = 0 (kmode 4)
    int kmode;         // 0: [0,K)  1: [128 tn, K)  2: [0, 128 (tm+1))  3: [128 max(tm,tn), K)  4: [128 tm, K)
    int vb_stride, vb_off, vb_limit;   // tile row (vb_on_n: tile column) valid iff batch*vb_stride + vb_off + tm (tn) < vb_limit
    int vb_on_n;
    int mirror;        // 1 (square lower-triangular outputs): tile (tm, tn), tm > tn, is also written transposed at (tn, tm)
};

// NJ = 4: one workgroup per 128 x 128 tile.  NJ = 2 (both operands M-contiguous only): two workgroups per tile, each the 64
// columns [64 h, 64 h + 64) -- same slabs, fragments and k order (gemm_tile_mc), so the same bits at twice the workgroup count,
// for launches whose 128 x 128 tiling leaves most of the chip's workgroup slots empty.
template <bool A_KC, bool B_KC, int NJ = 4>
__global__ __launch_bounds__(256, 2) void tri_gemm_kernel(GemmDesc g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int unit = NJ == 4 ? blockIdx.x : blockIdx.x >> 1;
    const int nhalf = NJ == 4 ? 0 : blockIdx.x & 1;
    int tm = unit % g.mt, tn = unit / g.mt;
    if (g.order == 1) {          // row-major enumeration of the lower triangle: all tiles of row tm share one k length
        int t = unit;
        tm = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while (tm * (tm + 1) / 2 > t) --tm;
        while ((tm + 1) * (tm + 2) / 2 <= t) ++tm;
        tn = t - tm * (tm + 1) / 2;
    } else if (g.order == 2) {
        tm = g.mt - 1 - unit / g.nt;
        tn = unit % g.nt;
    } else if (g.order == 3) {
        tm = unit / g.nt;
        tn = unit % g.nt;
    }
    const int batch = blockIdx.y;
    if (g.tri && tn + g.tri_off > tm) return;
    if (batch * g.vb_stride + g.vb_off + (g.vb_on_n ? tn : tm) >= g.vb_limit) return;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    int kb = 0, ke = g.K;
    if (g.kmode == 1) kb = NB * tn;
    else if (g.kmode == 2) ke = min(g.K, NB * (tm + 1));
    else if (g.kmode == 3) kb = NB * max(tm, tn);
    else if (g.kmode == 4) kb = NB * tm;
    const double* A = g.A + batch * g.strideA;
    const double* B = g.B + batch * g.strideB;
    double* C = g.C + batch * g.strideC;
    const double* Ap = A_KC ? A + (long)m0 * g.lda : A + m0;
    const double* Bp = B_KC ? B + (long)n0 * g.ldb : B + n0;
    Acc acc;
    acc.zero();
    gemm_tile<A_KC, B_KC, NJ, true>(acc, Ap, g.lda, Bp, g.ldb, kb, ke, lds, nhalf);   // kb, ke multiples of 128
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncol0 = NJ == 4 ? (wave >> 1) * 64 : 64 * nhalf + (wave >> 1) * 32;   // this wave's first column in the tile
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double* c = C + (long)(m0 + acc_m(i)) + (long)(n0 + ncol0 + 16 * j + (lane >> 4) + 4 * r) * g.ldc;
                double v = g.alpha * acc.v[i][j][r];
                if (g.beta != 0.0) v += g.beta * *c;
                *c = v;
                // the mirror tile from the same registers (lauum: K^-1 is used as a full matrix; a separate pass over the finished
                // matrix read and wrote N^2 / 2 elements again: 91 us at N = 8192).  Four consecutive columns per lane group and
                // store, a full 128-byte line per row over r = 0 .. 3.
                if (g.mirror && tm != tn) C[(long)(n0 + ncol0 + 16 * j + (lane >> 4) + 4 * r) + (long)(m0 + acc_m(i)) * g.ldc] = v;
            }
}

// LDS request of the tile kernels.  one_per_cu: 96 KB, so that only one workgroup fits a CU -- one wave per SIMD keeps the MFMA
// pipe to itself (see launch_acq_gemm); measured for trtri at N = 8192: 3.44 -> 3.19 ms, for lauum no difference.
// SLS_TRI_WG_PER_CU=1 / 2 forces either for every launch (A/B switch).
static int tri_lds_bytes(bool one_per_cu) {
    if (tune_set(TUNE_TRI_WG_PER_CU)) one_per_cu = tune(TUNE_TRI_WG_PER_CU, 0) == 1;
    return one_per_cu ? 96 * 1024 : GEMM_LDS_BYTES;
}
template <bool A_KC, bool B_KC>
static void launch_tri_gemm(hipStream_t s, const GemmDesc& g, int batches, bool one_per_cu = false) {
    const int lds_bytes = tri_lds_bytes(one_per_cu);
    ensure_dyn_lds((const void*)tri_gemm_kernel<A_KC, B_KC>, lds_bytes);
    if (g.mt <= 0 || g.nt <= 0 || batches <= 0) return;
    const int grid = g.order == 1 ? g.mt * (g.mt + 1) / 2 : g.mt * g.nt;
    hipLaunchKernelGGL((tri_gemm_kernel<A_KC, B_KC>), dim3(grid, batches), dim3(GEMM_THREADS), lds_bytes, s, g);
}
// both operands M-contiguous, half tiles (two workgroups per 128 x 128 tile)
static void launch_tri_gemm_mc_half(hipStream_t s, const GemmDesc& g, int batches, bool one_per_cu = false) {
    const int lds_bytes = tri_lds_bytes(one_per_cu);
    ensure_dyn_lds((const void*)tri_gemm_kernel<false, false, 2>, lds_bytes);
    if (g.mt <= 0 || g.nt <= 0 || batches <= 0) return;
    const int grid = g.order == 1 ? g.mt * (g.mt + 1) / 2 : g.mt * g.nt;
    hipLaunchKernelGGL((tri_gemm_kernel<false, false, 2>), dim3(2 * grid, batches), dim3(GEMM_THREADS), lds_bytes, s, g);
}

static GemmDesc mkdesc(const double* A, long lda, const double* B, long ldb, double* C, long ldc, int mt, int nt, int K,
                       double alpha, double beta) {
    GemmDesc g;
    g.A = A; g.lda = lda; g.strideA = 0;
    g.B = B; g.ldb = ldb; g.strideB = 0;
    g.C = C; g.ldc = ldc; g.strideC = 0;
    g.mt = mt; g.nt = nt; g.K = K; g.alpha = alpha; g.beta = beta;
    g.tri = 0; g.tri_off = 0; g.order = 0; g.kmode = 0; g.vb_stride = 0; g.vb_off = 0; g.vb_limit = 1 << 30; g.vb_on_n = 0; g.mirror = 0;
    return g;
}

void launch_gemm_plain(hipStream_t s, const double* A, long lda, bool a_kc, const double* B, long ldb, bool b_kc, double* C,
                       long ldc, int mt, int nt, int K, double alpha, double beta) {
    GemmDesc g = mkdesc(A, lda, B, ldb, C, ldc, mt, nt, K, alpha, beta);
    if (!a_kc && !b_kc) launch_tri_gemm<false, false>(s, g, 1);
    else if (!a_kc && b_kc) launch_tri_gemm<false, true>(s, g, 1);
    else if (a_kc && b_kc) launch_tri_gemm<true, true>(s, g, 1);
    else launch_tri_gemm<true, false>(s, g, 1);
}

// C_part[c] = A[:, K_c] * B[:, K_c]^T for `chunks` equal ranges K_c of the contraction (A M-contiguous, B K-contiguous): a tall
// product with few output tiles (the MAP gradient's Y = G X~: N x D x N, N / 128 tiles) spread over chunks x as many workgroups.
// The caller adds the partial results in chunk order (fixed order: deterministic).
void launch_gemm_splitk_nt(hipStream_t s, const double* A, long lda, const double* B, long ldb, double* Cpart, long ldc,
                           long part_stride, int mt, int nt, int K, int chunks) {
    const int Kc = K / chunks;                        // multiple of 128 (caller)
    GemmDesc g = mkdesc(A, lda, B, ldb, Cpart, ldc, mt, nt, Kc, 1.0, 0.0);
    g.strideA = (long)Kc * lda;
    g.strideB = Kc;
    g.strideC = part_stride;
    launch_tri_gemm<false, true>(s, g, chunks);
}

// Two-level right-looking factorisation.  Outer blocks of `nbo` 128-columns: inside an outer block every 128-step is
//   diag (one workgroup, LDS-resident)  ->  panel L_ij = A_ij T_jj^T for ALL rows below  ->  update of the REMAINING columns
//   of the outer block only (K = 128, at most nbo - 1 tile columns: a short launch),
// and the rest of the trailing matrix is updated once per outer block with K = 128 nbo (nbo x fewer read-modify-write
// sweeps over the trailing matrix and a k loop long enough to run at the GEMM rate; with nbo = 1 this is the plain
// one-level algorithm).
int potrf_default_nbo(int Np) {
    if (tune(TUNE_POTRF_NBO, 0) >= 1) return (int)tune(TUNE_POTRF_NBO, 0);
    return Np >= 8192 ? 4 : 1;   // measured (tools/probes/potrf_bench): two-level pays from N = 8192 (10.4 -> 9.5 ms), not below
}

void launch_potrf(hipStream_t s, double* A, int Np, double* Linv, int* info, int nbo, int* dataflow_sync) {
    const int nb = Np / NB;
    const long ld = Np;
    const int mode = potrf_default_mode(Np);
    if (dataflow_sync && nb >= 3 && mode == 3 && launch_potrf_dataflow(s, A, Np, Linv, info, dataflow_sync)) return;
    if (nbo < 1) nbo = potrf_default_nbo(Np);
    auto syrk = [&](hipStream_t st, int kcol0, int ktiles, int row0, int col0, int ncols) {
        // A[row0.., col0 .. col0+ncols) -= L[row0.., kcol0 .. kcol0+ktiles) L[col0.., same]^T on lower tiles (row >= col)
        const int mt = nb - row0;
        if (mt <= 0 || ncols <= 0) return;
        const double* Ap = A + (long)row0 * NB + (long)kcol0 * NB * ld;
        const double* Bp = A + (long)col0 * NB + (long)kcol0 * NB * ld;
        double* Cp = A + (long)row0 * NB + (long)col0 * NB * ld;
        GemmDesc u = mkdesc(Ap, ld, Bp, ld, Cp, ld, mt, ncols, ktiles * NB, -1.0, 1.0);
        u.tri = 1;
        u.tri_off = col0 - row0;     // tile (tm, tn) is on or below the diagonal iff row0 + tm >= col0 + tn
        launch_tri_gemm<false, false>(st, u, 1);
    };
    for (int J0 = 0; J0 < nb; J0 += nbo) {
        const int J1 = std::min(J0 + nbo, nb);      // outer block = tile columns [J0, J1)
        for (int j = J0; j < J1; ++j) {
            double* Ajj = A + (long)j * NB * (ld + 1);
            double* Tjj = Linv + (long)j * NB * (ld + 1);
            launch_chol_diag(s, Ajj, ld, Tjj, ld, info, j * NB);
            const int rem = nb - j - 1;
            if (rem <= 0) break;
            double* Apan = Ajj + NB;   // rows below the diagonal block, same columns
            // panel: L_ij = A_ij T_jj^T   (B operand elem(n,k) = T[n + k ld], M-contiguous)
            GemmDesc p = mkdesc(Apan, ld, Tjj, ld, Apan, ld, rem, 1, NB, 1.0, 0.0);
            launch_tri_gemm<false, false>(s, p, 1);
            // inner update: the remaining columns of this outer block
            syrk(s, j, 1, j + 1, j + 1, J1 - (j + 1));
        }
        if (J1 >= nb) break;
        const int kt = J1 - J0;
        syrk(s, J0, kt, J1, J1, nb - J1);   // nbo == 1: the inner update above had no columns, this is the whole update
    }
}

// dst block = (src block)^T for `batches` blocks of tr x tc 32 x 32 sub-tiles each (through a padded LDS tile: both sides
// coalesced).  Blocks whose first tile row lies beyond the matrix are skipped (vb_*, as in GemmDesc).
__global__ __launch_bounds__(256) void transpose_blocks_kernel(const double* __restrict__ src, double* __restrict__ dst, long ld,
                                                               long stride, int vb_stride, int vb_off, int vb_limit) {
    __shared__ double tile[32][33];
    const int batch = blockIdx.z;
    // rows of the source block beyond the matrix do not exist (last pair of a level): 128-row granularity
    if (batch * vb_stride + vb_off + (int)(blockIdx.x / 4) >= vb_limit) return;
    const double* S = src + batch * stride;
    double* Dp = dst + batch * stride;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) tile[r][tx] = S[(long)(32 * blockIdx.x + tx) + (long)(32 * blockIdx.y + r) * ld];   // tile[col][row]
    __syncthreads();
    for (int r = ty; r < 32; r += 8) Dp[(long)(32 * blockIdx.y + tx) + (long)(32 * blockIdx.x + r) * ld] = tile[tx][r];
}

// dst = src^T for a whole Np x Np matrix (handles with the solve-based sigma read EVERY block of U = (L^-1)^T; trtri leaves the
// strictly-lower blocks of U unwritten, and the rank-1 growth of sls_gp_append_point does not maintain U at all)
void launch_transpose_full(hipStream_t s, const double* src, double* dst, int Np) {
    hipLaunchKernelGGL(transpose_blocks_kernel, dim3(Np / 32, Np / 32, 1), dim3(256), 0, s, src, dst, (long)Np, 0L, 0, 0, 1 << 30);
}

// Linv (diagonal blocks already inverted) <- full lower-triangular inverse X = L^-1, and U <- X^T (upper triangular), level by
// level (recursive doubling).  Level with half-size h blocks, pair p = blocks F = [2hp, 2hp+h) | S = [2hp+h, min(2hp+2h, nb)):
//     W (F x S) = U_FF L_SF^T        (= (L_SF X_FF)^T)         k >= 128 tm
//     X_SF      = -X_SS W^T                                     k <  128 (tm + 1)
//     U_FS      = X_SF^T                                        (transpose_blocks_kernel)
// Keeping the transposed inverse next to the inverse makes BOTH products NT forms of column-major operands, i.e. both operands
// M-contiguous: they run on gemm_tile_mc (LDS-direct loads, 0.95 of the MFMA peak) instead of the staged K-contiguous path
// (0.6-0.7), and so does lauum (K^-1 = U U^T).  Every element is the same dot product in the same k order as in the NN form
// (tmp = L_SF X_FF, X_SF = -X_SS tmp), so the results are bit-identical to it.  W lives in `tmp` (the K^-1 buffer, free until
// lauum).  U's strictly-lower blocks are never read; its diagonal blocks are the transposed diagonal blocks of X.
void launch_trtri(hipStream_t s, const double* L, int Np, double* Linv, double* tmp, double* U) {
    const int nb = Np / NB;
    const long ld = Np;
    // diagonal blocks: U_jj = X_jj^T (zeros included)
    hipLaunchKernelGGL(transpose_blocks_kernel, dim3(4, 4, nb), dim3(256), 0, s, Linv, U, ld, (long)NB * (ld + 1), 0, 0, 1 << 30);
    for (int h = 1; h < nb; h *= 2) {
        const int pairs = (nb + 2 * h - 1) / (2 * h);
        const long pstride = (long)2 * h * NB * (ld + 1);
        const long off21 = (long)h * NB;          // block (S, F): rows of the second half, columns of the first
        const long off12 = (long)h * NB * ld;     // block (F, S)
        // fewer 128 x 128 tiles than two per CU: half tiles (same bits).  Measured per level at N = 8192 (rocprofv3 kernel trace, us per
        // product; half / whole tiles): 512 tiles (h = 16): 288 / 273; 256 tiles (h = 8): 110 / 117 (SLS_TRTRI_NARROW = the threshold)
        const bool narrow = (long)h * h * pairs <= tune(TUNE_TRTRI_NARROW, 511);
        // W = U_FF * L_SF^T : A = U_FF (M-contig, k >= 128 tm), B elem(n,k) = L_SF[n + k ld] (M-contig); columns n in S
        GemmDesc g1 = mkdesc(U, ld, L + off21, ld, tmp + off12, ld, h, h, h * NB, 1.0, 0.0);
        g1.strideA = g1.strideB = g1.strideC = pstride;
        g1.kmode = 4;
        g1.vb_stride = 2 * h; g1.vb_off = h; g1.vb_limit = nb; g1.vb_on_n = 1;
        g1.order = 3;       // k >= 128 tm: long rows first
        if (narrow) launch_tri_gemm_mc_half(s, g1, pairs, true);
        else launch_tri_gemm<false, false>(s, g1, pairs, true);
        // X_SF = -X_SS * W^T : A = X_SS (M-contig, k < 128 (tm+1)), B elem(n,k) = W[n + k ld] (M-contig); rows m in S
        GemmDesc g2 = mkdesc(Linv + (long)h * NB * (ld + 1), ld, tmp + off12, ld, Linv + off21, ld, h, h, h * NB, -1.0, 0.0);
        g2.strideA = g2.strideB = g2.strideC = pstride;
        g2.kmode = 2;
        g2.vb_stride = 2 * h; g2.vb_off = h; g2.vb_limit = nb;
        g2.order = 2;       // k < 128 (tm + 1): long rows first
        if (narrow) launch_tri_gemm_mc_half(s, g2, pairs, true);
        else launch_tri_gemm<false, false>(s, g2, pairs, true);
        // U_FS = X_SF^T
        hipLaunchKernelGGL(transpose_blocks_kernel, dim3(4 * h, 4 * h, pairs), dim3(256), 0, s, Linv + off21, U + off12,
                           ld, pstride, 2 * h, h, nb);
    }
}

// K^-1 = X^T X = U U^T with U = X^T (launch_trtri): lower tiles, A elem(m,k) = U[m + k ld], B elem(n,k) = U[n + k ld] (both
// M-contiguous), k >= 128 max(tm,tn) = 128 tm.
void launch_lauum(hipStream_t s, const double* U, int Np, double* Kinv) {
    const int nb = Np / NB;
    const long ld = Np;
    GemmDesc g = mkdesc(U, ld, U, ld, Kinv, ld, nb, nb, Np, 1.0, 0.0);
    g.tri = 1;
    g.kmode = 3;
    g.order = 1;            // rows from the top, longest k range first, no idle workgroups
    g.mirror = 1;           // both triangles in one pass
    // small matrices leave most workgroup slots empty: half tiles double the count (SLS_LAUUM_N64=0/1 overrides)
    const int n64_env = (int)tune(TUNE_LAUUM_N64, -1);
    // measured inside the C5 evaluation (N = 4096): 3.41 -> 3.24 ms per evaluation with half tiles (the longest tile's k loop,
    // 32 slabs of 13.6 us on a shared CU, bounds the launch); at N = 8192 whole tiles win (2.9 ms, 0.80 of peak)
    const bool narrow = n64_env >= 0 ? n64_env != 0 : nb <= 32;
    if (narrow) launch_tri_gemm_mc_half(s, g, 1);
    else launch_tri_gemm<false, false>(s, g, 1);
}

// A (SPD, lower triangle read) -> L in place, Linv = L^-1, U = Linv^T (blocks on and above the diagonal), Kinv = A^-1 (full).
// N <= 4096 with the single-launch schedule available: ONE launch (factorisation + fused inverse, potri_team); otherwise
// launch_potrf + launch_trtri + launch_lauum.  SLS_POTRI_FUSED=0 forces the separate launches (A/B, tests).  Returns true when
// the fused launch was used.  As with launch_potrf, info[1] != 0 afterwards means the single launch gave up: repeat on the
// multi-launch schedule (dataflow_sync = nullptr).
bool potri_fused_applies(int Np, bool have_sync) {
    const int nb = Np / NB;
    return tune_on(TUNE_POTRI_FUSED) && have_sync && nb >= 3 && nb <= 32 && potrf_default_mode(Np) == 3;
}
// linv_zeroed = false: the caller has NOT cleared Linv (the fused launch does not need it: it writes the diagonal tiles in full and
// the tiles below them, and leaves the tiles above the diagonal alone); the separate launches clear it here.
bool launch_potri(hipStream_t s, double* A, int Np, double* Linv, double* U, double* Kinv, int* info, int* dataflow_sync,
                  bool linv_zeroed) {
    if (potri_fused_applies(Np, dataflow_sync != nullptr) && launch_potri_dataflow(s, A, Np, Linv, U, Kinv, info, dataflow_sync))
        return true;
    if (!linv_zeroed) launch_fill(s, Linv, (long)Np * Np, 0.0);
    launch_potrf(s, A, Np, Linv, info, 0, dataflow_sync);
    launch_trtri(s, A, Np, Linv, Kinv, U);
    launch_lauum(s, U, Np, Kinv);
    return false;
}

// B <- (L L^T)^-1 B, B is Np x Rp (ld = Np).  Block forward / backward substitution, each step two tile GEMMs.
void launch_potrs(hipStream_t s, const double* L, const double* Linv, int Np, double* B, int Rp) {
    const int nb = Np / NB, rt = Rp / NB;
    const long ld = Np;
    for (int j = 0; j < nb; ++j) {   // forward: X_j = T_jj B_j ; B_i -= L_ij X_j (i > j)
        const double* Tjj = Linv + (long)j * NB * (ld + 1);
        double* Bj = B + (long)j * NB;
        GemmDesc a = mkdesc(Tjj, ld, Bj, ld, Bj, ld, 1, rt, NB, 1.0, 0.0);   // B operand elem(n,k) = Bj[k + n ld]: K-contig
        launch_tri_gemm<false, true>(s, a, 1);
        const int rem = nb - j - 1;
        if (rem > 0) {
            GemmDesc u = mkdesc(L + (long)j * NB * (ld + 1) + NB, ld, Bj, ld, Bj + NB, ld, rem, rt, NB, -1.0, 1.0);
            launch_tri_gemm<false, true>(s, u, 1);
        }
    }
    for (int j = nb - 1; j >= 0; --j) {   // backward: X_j = T_jj^T B_j ; B_i -= L_ji^T X_j (i < j)
        const double* Tjj = Linv + (long)j * NB * (ld + 1);
        double* Bj = B + (long)j * NB;
        GemmDesc a = mkdesc(Tjj, ld, Bj, ld, Bj, ld, 1, rt, NB, 1.0, 0.0);   // A elem(m,k) = T[k + m ld]: K-contig
        launch_tri_gemm<true, true>(s, a, 1);
        if (j > 0) {
            // rows i < j: A elem(m,k) = L[(j NB + k) + m ld] over block row j of L, columns 0.. j NB
            GemmDesc u = mkdesc(L + (long)j * NB, ld, Bj, ld, B, ld, j, rt, NB, -1.0, 1.0);
            launch_tri_gemm<true, true>(s, u, 1);
        }
    }
}

}  // namespace slsk
