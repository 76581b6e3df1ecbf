// Fused MAP-objective evaluation for the reference's own operating sizes (N <= 128 data points; the demos run N <= 100):
// ONE workgroup, ONE launch replaces the ~20 launches of the tiled pipeline (prep, Gram, potrf, trtri, lauum, GEMVs,
// weight/gradient contractions, reductions), which at these sizes is pure launch latency (340 us wall per evaluation of
// the GP marginal likelihood at N = 20, of which < 80 us are kernels).
//
// Reference arithmetic: src/gaussian-process-regressor.cpp:66-127,141-193 (log marginal likelihood and its gradient
// wrt a, b, l_1..l_D) and the GP term of src/preference-regressor.cpp:53-115; kernel scalars src/regressor.cpp:14-23.
//   K_y = K_f(a, l) + b I  ->  L, L^-1 (chol_diag_steps, LDS resident)  ->  logdet = 2 sum log L_ii,
//   K^-1 = L^-T L^-1 (MFMA tiles, overwriting L),  alpha = K^-1 y,  quad = y^T alpha,
//   W = 1/2 (alpha alpha^T - K^-1):  d/da = sum W.*K_f / a,  d/db = tr W,  d/dl_p = (1/l_p) sum_ij W_ij c_ij (x~_ip - x~_jp)^2
// Everything lives in the 160 KB LDS image of the diagonal-block kernel; the 16 padding doubles of every LDS column
// (rows 128..143 of the [128 x 144] matrix) serve as scratch for alpha, y, 1/l and the reduction slots.
#include "chol_diag.hpp"
#include "kernels.hpp"
#include "wave_reduce.hpp"
#include "../../include/sls_hip.h"

namespace slsk {

// scratch map (2048 doubles in the padding rows of the LDS matrix):
//   [0,128) alpha   [128,256) y   [256,384) 1/l   [384,1024) BTL contributions (map_opt) / the four waves' partial length-scale
//   gradients (small_grad)   [1024,1028) reduction slots   [1028,1032) a, b (map_opt)   [1040,1168) l (map_opt)
//   [1168,1296) length-scale gradient   [1296,1616) gradient wrt the optimiser's variables (map_opt)
//   [1616,1744) squared norms of the scaled points   [1744,1768) section timers (only when tracing)   [1768,1780) slots of the triple sum
//   [1872,2002) logarithms of a, b, l (map_opt)
constexpr int SC_ALPHA = 0, SC_Y = 128, SC_INVL = 256, SC_BTL = 384, SC_BTL_MAX = 640, SC_RED = 1024, SC_AB = 1028,
              SC_ELL = 1040, SC_GL = 1168, SC_GZ = 1296, SC_NX = 1616, SC_TRACE = 1744, SC_RED3 = 1768, SC_LZ = 1872;
static_assert(SC_TRACE + MAP_OPT_TRACE_SLOTS <= SC_LZ && SC_GZ + MAP_OPT_MAX_VARS <= SC_NX && SC_LZ + 2 + NLL_SMALL_MAX_D <= 2048 && 4 * NLL_SMALL_MAX_D <= SC_BTL_MAX, "LDS scratch map");

// (k >> 4) DL + 128 + (k & 15) with DL = 144 = 128 + 16: a shift and two adds instead of shift, mask, multiply, two adds -- these
// addresses are formed thousands of times per evaluation by waves that are alone on their SIMD (every instruction counts)
static_assert(DL == 144, "scratch / free-area addressing assumes DL = 128 + 16");
__device__ __forceinline__ double& small_scratch(double* As, int k) { return As[128 + k + ((k >> 4) << 7)]; }

__device__ __forceinline__ double small_block_sum(double v, double* As) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) small_scratch(As, SC_RED + (threadIdx.x >> 6)) = v;
    __syncthreads();
    return (small_scratch(As, SC_RED) + small_scratch(As, SC_RED + 1)) + (small_scratch(As, SC_RED + 2) + small_scratch(As, SC_RED + 3));
}

// three sums behind one pair of barriers (each in the order of small_block_sum: lanes, then (w0 + w1) + (w2 + w3))
__device__ __forceinline__ void small_block_sum3(double& v0, double& v1, double& v2, double* As) {
    v0 = wave_sum(v0);
    v1 = wave_sum(v1);
    v2 = wave_sum(v2);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        small_scratch(As, SC_RED3 + w) = v0;
        small_scratch(As, SC_RED3 + 4 + w) = v1;
        small_scratch(As, SC_RED3 + 8 + w) = v2;
    }
    __syncthreads();
    auto total = [&](int q) {
        return (small_scratch(As, SC_RED3 + q) + small_scratch(As, SC_RED3 + q + 1)) + (small_scratch(As, SC_RED3 + q + 2) + small_scratch(As, SC_RED3 + q + 3));
    };
    v0 = total(0);
    v1 = total(4);
    v2 = total(8);
}

// optional section timing (SLS_MAP_TRACE=1): ticks of the 100 MHz clock per section, kept by thread 0 in the LDS scratch (24 running
// totals in registers cost the untraced kernel ~50 scalar registers it does not have: spills)
struct SmallTrace {
    bool on = false;
    long long t_prev = 0;
    double* As = nullptr;
    __device__ __forceinline__ long long& slot_ref(int slot) { return reinterpret_cast<long long&>(small_scratch(As, SC_TRACE + slot)); }
    __device__ __forceinline__ void mark(int slot) {
        if (on) {
            const long long t_now = wall_clock64();
            if (threadIdx.x == 0) slot_ref(slot) += t_now - t_prev;
            t_prev = t_now;
        }
    }
    __device__ __forceinline__ void count(int slot) {
        if (on && threadIdx.x == 0) slot_ref(slot) += 1;
    }
};

// A wave's sequence of matrix-core steps (eight to twelve fragment reads, then four to eight products), software-pipelined: the
// fragments of step s + 1 are requested before the products of step s are issued, in the same basic block, into the other of two
// fragment sets (no copies: a copy would wait for the reads it copies).  One wave per SIMD has nothing else to hide the LDS latency
// behind, and a step of 8 reads + 4 dependent products took ~900 cycles back to back (measured, N = 58) against 256 of products.
//   It: done(), last() (the step completes a tile / unit), advance(); past the end an iterator keeps valid coordinates (the
//   prefetch of a step that does not exist reads addresses of the last one).   load(it, frag)  mma(it, frag)  fin(it)
struct SmallFrag {
    double a[4], b[4], c[4];
};
template <class It, class Load, class Mma, class Fin>
__device__ __forceinline__ void small_mfma_steps(It it, Load&& load, Mma&& mma, Fin&& fin) {
    if (it.done()) return;
    SmallFrag f0, f1;
    load(it, f0);
    for (;;) {
        It n1 = it;
        n1.advance();
        load(n1, f1);
        mma(it, f0);
        if (it.last()) fin(it);
        if (n1.done()) break;
        It n2 = n1;
        n2.advance();
        load(n2, f0);
        mma(n1, f1);
        if (n1.last()) fin(n1);
        if (n2.done()) break;
        it = n2;
    }
}

template <bool MATERN>
__device__ __forceinline__ void small_kern(double a, double q, double& k, double& c) {
    if (!MATERN) {
        k = a * exp(-0.5 * q);
        c = k;
    } else {
        const double s = sqrt(5.0 * q), e = exp(-s);
        k = a * (1.0 + s + (5.0 / 3.0) * q) * e;
        c = a * (5.0 / 3.0) * (1.0 + s) * e;
    }
}

// Where the design matrix is read from.  A single CU with one wave per SIMD cannot hide global-memory latency (measured: ~0.5 us
// per dependent round trip) and even an LDS-resident copy read once per pair and dimension is bound by the LDS bandwidth (three
// 512-byte reads per fused multiply-add), so whenever the N x N image leaves enough of the LDS block unused -- the columns right
// of the leading Nb = 16 ceil(N / 16) ones -- the CENTRED design matrix x - 0.5 is staged there once per launch and the Gram
// matrix and the length-scale gradient run on the matrix cores in the forms of the tiled pipeline (kernels_gram.hip,
// kernels_map.hip):
//     q_ij = |x~_i|^2 + |x~_j|^2 - 2 x~_i . x~_j,   x~ = (x - 0.5) / l      (one operand fragment scaled by 1 / l_d^2 on the fly)
//     dL/dl_p = (2 / l_p) sum_j [x~_jp^2 r_j + (Gl X~^2)_jp - 2 x~_jp (Gl X~)_jp],   Gl = strictly-lower 1/2 W o C,  r = Gl 1  (small_grad)
// Layout: point i, dimension d at offset d + i Dp of the free area, Dp = 16 ceil(D / 16) + 1, zero padded to Nb points and Dp - 1
// dimensions: both fragment shapes (16 lanes over points, stride Dp odd; 16 lanes over dimensions, contiguous) are conflict free.
// C3's shapes (N <= 96 at D = 32) always fit.  Otherwise the differences are formed directly: the Gram pass reads the transposed
// copy XTr[i + d * 128] (the 16 rows of a tile are 128 contiguous bytes per dimension) and the length-scale contraction the
// D x N original (a point's D values are contiguous).
struct SmallPts {
    const double* __restrict__ X;
    const double* __restrict__ XTr;
    int lds, base_col, Dp;
    // the pair scratch of the gradient beyond the slots a thread keeps in registers (see Fold / SmallPairs), behind the points when
    // that fits too: stash_off = its offset in the free area, or -1 (then global memory)
    int stash_off;
};

// The elements (i, j), j <= i < N, of the lower triangle dealt evenly over the 256 threads: columns j and N - 1 - j together hold
// N + 1 elements, so slot e = c + r (N + 1) of the folded rectangle [ceil(N / 2)] x [N + 1] is element (r + c, r) for c < N - r and
// (c - 1, N - 1 - r) beyond.  Thread t owns slots t, t + 256, ...: consecutive lanes walk down a column (conflict-free in the
// column-major image).  The kernel-function pass and the gradient-weight pass walk the same slots, so a pair's kernel value and
// derivative weight stay with their thread: the first SMALL_SR slots in registers, the rest in the pair scratch (slot order).
constexpr int SMALL_SR = 8;
struct SmallPairs {
    double k[SMALL_SR], c[SMALL_SR];
};
__device__ __forceinline__ int small_stash_len(int N) {
    const int total = ((N + 1) >> 1) * (N + 1);
    return total > SMALL_SR * 256 ? ((total - SMALL_SR * 256 + 127) & ~127) : 0;   // per array, whole LDS columns
}
struct Fold {
    int N, W, H, dr, dc, r, c;
    __device__ __forceinline__ explicit Fold(int N_) : N(N_), W(N_ + 1), H((N_ + 1) >> 1) {
        dr = 256 / W;
        dc = 256 - dr * W;
        r = (int)threadIdx.x / W;
        c = (int)threadIdx.x - r * W;
    }
    __device__ __forceinline__ int total() const { return H * W; }
    // the element of this thread's current slot (0, 0 and false beyond the triangle; odd N: the middle column pairs with itself and
    // its second copy is dropped), then on to the thread's next slot
    __device__ __forceinline__ bool next(int& i, int& j) {
        const bool first = c < N - r;
        const bool live = r < H && !(!first && 2 * r == N - 1);
        i = live ? (first ? r + c : c - 1) : 0;
        j = live ? (first ? r : N - 1 - r) : 0;
        c += dc;
        r += dr;
        if (c >= W) { c -= W; ++r; }
        return live;
    }
};
// element o of the free area (the columns right of the leading base_col ones, 128 rows each)
__device__ __forceinline__ double& small_free(double* As, int base_col, int o) { return As[base_col * DL + o + ((o >> 7) << 4)]; }   // (base_col + (o >> 7)) DL + (o & 127)
struct PtsLds {
    const double* As;
    int base_col, Dp;
    __device__ __forceinline__ double operator()(int i, int d) const {
        const int o = d + i * Dp;
        return As[base_col * DL + o + ((o >> 7) << 4)];
    }
};
struct PtsRows {   // XTr[i + d * 128]
    const double* __restrict__ XTr;
    __device__ __forceinline__ double operator()(int i, int d) const { return XTr[i + d * 128]; }
};
struct PtsCols {   // X[d + i * D]
    const double* __restrict__ X;
    int D;
    __device__ __forceinline__ double operator()(int i, int d) const { return X[d + (long)i * D]; }
};
// all threads; the caller's next barrier publishes the copy
__device__ __forceinline__ SmallPts small_pts_stage(double* As, const double* __restrict__ X, const double* __restrict__ XTr, int D, int N,
                                                    bool allow) {
    SmallPts p{X, XTr, 0, 16 * ((N + 15) >> 4), 16 * ((D + 15) >> 4) + 1, -1};
    const int Nb = p.base_col;
    // 1 / l of the padding dimensions D .. 127: zeros, written once (the fragment loads of the Gram pass read them unconditionally)
    for (int d = D + (int)threadIdx.x; d < 128; d += 256) small_scratch(As, SC_INVL + d) = 0.0;
    if (!allow || Nb * p.Dp > (128 - Nb) * 128) return p;
    p.lds = 1;
    if (Nb * p.Dp + 2 * small_stash_len(N) <= (128 - Nb) * 128) p.stash_off = Nb * p.Dp;
    for (int o = threadIdx.x; o < Nb * p.Dp; o += 256) {
        const int i = o / p.Dp, d = o - i * p.Dp;
        small_free(As, p.base_col, o) = (i < N && d < D) ? X[d + (long)i * D] - 0.5 : 0.0;
    }
    return p;
}

// May this evaluation use the norm expansion?  (uniform: every wave computes the same bound from the 1 / l in the scratch, zeros beyond D)
__device__ __forceinline__ bool small_mc_form_ok(double* As, int D) {
    const int l64 = threadIdx.x & 63;
    const double ilm = wave_max(fmax(small_scratch(As, SC_INVL + l64), small_scratch(As, SC_INVL + 64 + l64)));
    return ilm * ilm * (0.5 * D) * 2.3e-16 <= 1e-11;
}

// q_ij = sum_d ((x_id - x_jd) / l_d)^2 in dimension order, B dimensions' operands in flight at a time
template <int B, class Pts>
__device__ __forceinline__ double small_pair_q(const Pts& x, int D, int i, int j, double* As) {
    double q = 0.0;
    int d = 0;
    for (; d + B <= D; d += B) {
        double xi[B], xj[B], il[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            xi[u] = x(i, d + u);
            xj[u] = x(j, d + u);
            il[u] = small_scratch(As, SC_INVL + d + u);
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const double t = (xi[u] - xj[u]) * il[u];
            q = fma(t, t, q);
        }
    }
    for (; d < D; ++d) {
        const double t = (x(i, d) - x(j, d)) * small_scratch(As, SC_INVL + d);
        q = fma(t, t, q);
    }
    return q;
}

// One batch of up to four slots of the kernel-function pass: q_ij from qf, the kernel value into the image (a + b on the diagonal), the
// pair (k, c) to put(u, k, c).  No branches around the evaluations: sqrt and exp of the four elements interleave.
template <bool MATERN, class QF, class Put>
__device__ __forceinline__ void small_gram_batch(double* As, double a, double b, Fold& fold, QF&& qf, Put&& put) {
    int iq[4], jq[4];
    bool live[4];
    double dq[4], kq[4], cq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        live[u] = fold.next(iq[u], jq[u]);
        dq[u] = qf(iq[u], jq[u]);
    }
    __builtin_amdgcn_sched_barrier(0);   // every load of the batch before its first store (dead slots all read element (0, 0))
#pragma unroll
    for (int u = 0; u < 4; ++u) small_kern<MATERN>(a, dq[u] < 0.0 ? 0.0 : dq[u], kq[u], cq[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (live[u]) As[iq[u] + jq[u] * DL] = (iq[u] == jq[u]) ? a + b : kq[u];
        put(u, live[u], kq[u], cq[u]);
    }
}

// K_y = K_f(a, l) + b I into the LDS image (1/l in the scratch) and its Cholesky factorisation on the leading ceil(N/16) blocks: L in the
// lower triangle of As, L^-T in its strictly-upper tiles, the inverses of the diagonal tiles in Ts.  Returns log L_ii in thread i (0 beyond N).
// Only the lower triangle of K_y is formed (nothing reads the upper halves of the diagonal tiles: what the factorisation computes
// from them never reaches a result).
// pairs != nullptr: the kernel values k_ij and derivative weights c_ij stay with their threads for small_grad -- *pairs for a
// thread's first SMALL_SR slots, the pair scratch (LDS behind the points, or kc in global memory) beyond.
template <bool MATERN, class Idle = NoIdleWork>
__device__ __forceinline__ double small_build_factor(double* As, double* Ts, const SmallPts& pts, const Fold& fold0, int D, int N, double a,
                                                     double b, int* __restrict__ info, SmallPairs* pairs, double* __restrict__ kc,
                                                     SmallTrace& st, Idle&& idle0 = Idle{}) {
    const int tid = threadIdx.x;
    const int nb16 = (N + 15) >> 4;
    const int slen = small_stash_len(N);
    // the kernel function over this thread's slots: registers first, then the pair scratch
    auto elements = [&](auto&& qf) {
        Fold fold = fold0;
        const int total = fold.total();
#pragma unroll
        for (int s0 = 0; s0 < SMALL_SR; s0 += 4) {
            if (s0 * 256 >= total) break;
            small_gram_batch<MATERN>(As, a, b, fold, qf, [&](int u, bool, double k, double c) {
                if (pairs) {
                    pairs->k[s0 + u] = k;
                    pairs->c[s0 + u] = c;
                }
            });
        }
        for (int e0 = SMALL_SR * 256; e0 < total; e0 += 4 * 256)
            small_gram_batch<MATERN>(As, a, b, fold, qf, [&](int u, bool live, double k, double c) {
                const int q = e0 - SMALL_SR * 256 + 256 * u + tid;
                if (!pairs || !live) return;
                if (pts.stash_off >= 0) {
                    small_free(As, pts.base_col, pts.stash_off + q) = k;
                    if (MATERN) small_free(As, pts.base_col, pts.stash_off + slen + q) = c;
                } else {
                    kc[q] = k;
                    if (MATERN) kc[128 * 128 + q] = c;
                }
            });
    };
    // The matrix-core form takes q_ij from the norm expansion |x~_i|^2 + |x~_j|^2 - 2 x~_i . x~_j, whose cancellation error is about
    // eps (|x~_i|^2 + |x~_j|^2) <= eps D / (2 l_min^2) ABSOLUTE in q -- nothing at the usual length scales, but the DIRECT phase of the GP
    // MAP fit takes l down to 1e-8, where near-duplicate points (normal late in a line search) would get q wrong by O(1) and the
    // clamp at 0 would hide it.  Beyond 1e-11 the evaluation uses the direct differences (which are exact for near-duplicates):
    // every wave forms the bound itself (two reads, one wave-wide maximum), so the branch is uniform.
    const bool mc_form = pts.lds != 0 && small_mc_form_ok(As, D);
    if (mc_form) {
        // Matrix-core form.  (1) The dot products x~_i . x~_j / l^2 of the lower tiles, one 16 x 16 tile per wave and trip, straight
        // into the image; the diagonal of a diagonal tile is |x~_i|^2 and goes to the scratch (SC_NX).  (2) q_ij from the norm
        // expansion and the kernel function per element.
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fl = lane & 15, fk = lane >> 4;
        const PtsLds xc{As, pts.base_col, pts.Dp};
        const int Dk = pts.Dp - 1;
        // tile t of the lower tiles (column by column) belongs to wave t & 3; a step = 16 dimensions of a tile
        struct It {
            int ti, tj, d0, nb, Dk;
            bool fin;
            __device__ __forceinline__ bool done() const { return fin; }
            __device__ __forceinline__ bool last() const { return d0 + 16 >= Dk; }
            __device__ __forceinline__ void skip(int n) {   // n tiles on, n <= 4: at most four column changes
                int i = ti + n, j = tj;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (j < nb && i >= nb) { i = i - nb + j + 1; ++j; }
                if (j >= nb || i >= nb) { fin = true; return; }
                ti = i;
                tj = j;
                d0 = 0;
            }
            __device__ __forceinline__ void advance() {
                if (!last()) { d0 += 16; return; }
                skip(4);
            }
        };
        if (nb16 <= 4) {
            // N <= 64 (at most ten tiles, three per wave): the wave's tiles t = wave, wave + 4, wave + 8 advance TOGETHER through the
            // dimensions -- straight-line code per 16 dimensions (the 1 / l fragment once, eight reads and four products per tile on
            // independent accumulators), no step bookkeeping between the products.  Tile slots beyond the last tile repeat tile 0 and
            // are not stored.  (The generic stepper below spent ~1200 cycles per step of four products, 800 of them on its own
            // scalar and address arithmetic, each in full on the one wave of its SIMD.)
            const int ntile = nb16 * (nb16 + 1) / 2;
            int ti3[3], tj3[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                int rem = wave + 4 * m < ntile ? wave + 4 * m : 0, j = 0;
                while (rem >= nb16 - j) { rem -= nb16 - j; ++j; }
                tj3[m] = j;
                ti3[m] = j + rem;
            }
            d4_t acc3[3] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
            for (int d0 = 0; d0 < Dk; d0 += 16) {
                double il2[4], a3[3][4], b3[3][4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int d = d0 + 4 * kk + fk;
                    il2[kk] = small_scratch(As, SC_INVL + d);   // zeros beyond D
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        a3[m][kk] = xc(16 * ti3[m] + fl, d);
                        b3[m][kk] = xc(16 * tj3[m] + fl, d);
                    }
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) il2[kk] *= il2[kk];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int m = 0; m < 3; ++m) acc3[m] = mfma16(b3[m][kk], a3[m][kk] * il2[kk], acc3[m]);
                }
            }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (wave + 4 * m >= ntile) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    As[16 * ti3[m] + fl + (16 * tj3[m] + fk + 4 * r) * DL] = acc3[m][r];
                    if (ti3[m] == tj3[m] && fl == fk + 4 * r) small_scratch(As, SC_NX + 16 * ti3[m] + fl) = acc3[m][r];
                }
            }
        } else {
        It it0{0, 0, 0, nb16, Dk, false};
        if (wave) it0.skip(wave);
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        small_mfma_steps(
            it0,
            [&](const It& q, SmallFrag& f) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int d = q.d0 + 4 * kk + fk;
                    f.c[kk] = small_scratch(As, SC_INVL + d);   // zeros beyond D
                    f.a[kk] = xc(16 * q.ti + fl, d);
                    f.b[kk] = xc(16 * q.tj + fl, d);
                }
            },
            [&](const It&, const SmallFrag& f) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = mfma16(f.b[kk], f.a[kk] * (f.c[kk] * f.c[kk]), acc);
            },
            [&](const It& q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    As[16 * q.ti + fl + (16 * q.tj + fk + 4 * r) * DL] = acc[r];
                    if (q.ti == q.tj && fl == fk + 4 * r) small_scratch(As, SC_NX + 16 * q.ti + fl) = acc[r];
                    acc[r] = 0.0;
                }
            });
        }
        __syncthreads();
        if (st.on) {   // probe: pass (1) (slot 23), counted inside the Gram slot 8 as well
            const long long t_now = wall_clock64();
            if (tid == 0) st.slot_ref(23) += t_now - st.t_prev;
        }
        // identity padding of the last block: the padded points are zero rows, so rows N .. 16 nb16 - 1 of the lower triangle hold
        // zeros already; ones on their diagonal (pass (2) does not touch these rows)
        {
            const int i = N + (tid >> 4);   // at most 15 padding rows
            if (i < 16 * nb16 && (tid & 15) == 0) As[i + i * DL] = 1.0;
        }
        elements([&](int i, int j) { return small_scratch(As, SC_NX + i) + small_scratch(As, SC_NX + j) - 2.0 * As[i + j * DL]; });
    } else {
        // direct differences from the transposed copy in global memory; identity padding of the last block written out
        for (int idx = tid; idx < (16 * nb16 - N) * 16 * nb16; idx += 256) {
            const int i = N + idx / (16 * nb16), j = idx % (16 * nb16);
            if (j <= i) As[i + j * DL] = (i == j) ? 1.0 : 0.0;
        }
        elements([&](int i, int j) { return small_pair_q<8>(PtsRows{pts.XTr}, D, i, j, As); });
    }
    __syncthreads();
    st.mark(8);

    chol_diag_steps<true, false>(As, Ts, info, 0, nb16, idle0);
    __syncthreads();
    st.mark(9);

    // ---- log det: thread i's term log L_ii (the caller adds them up, small_alpha together with its own sums) ----
    return tid < N ? log(As[tid + tid * DL]) : 0.0;
}

// The strictly-lower tiles of the leading nb16 blocks copied into their mirror positions, one tile per wave and trip.  Each 16-lane
// group walks a wrapped diagonal of the tile (column fl, row fl + fk + 4 q), so both the row-major reads and the column-major writes
// touch 16 different banks.  The caller's barriers order the pass against its neighbours.
__device__ __forceinline__ void small_mirror_lower(double* As, int nb16) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fl = lane & 15, fk = lane >> 4;
    int t = 0;
    for (int i = 1; i < nb16; ++i)
        for (int j = 0; j < i; ++j, ++t) {
            if ((t & 3) != wave) continue;
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = As[(16 * j + fl) * DL + 16 * i + ((fl + fk + 4 * q) & 15)];   // (row 16i + r, col 16j + fl)
#pragma unroll
            for (int q = 0; q < 4; ++q) As[(16 * i + ((fl + fk + 4 * q) & 15)) * DL + 16 * j + fl] = v[q];   // (row 16j + fl, col 16i + r)
        }
}

// K^-1 = L^-T L^-1 as a full symmetric image over the dead factor (after small_build_factor: L in the lower triangle of As, L^-T in its
// strictly-upper tiles, the inverses of the diagonal tiles in Ts)
__device__ __forceinline__ void small_inverse_in_place(double* As, double* Ts, int N, SmallTrace* stp = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 15, fk = lane >> 4;
    const int nb16 = (N + 15) >> 4;
    // ---- K^-1 = L^-T L^-1, lower tiles (i >= j) into the lower triangle (L is no longer needed) ----
    // tile t (row-major over the lower tiles) belongs to wave t & 3; a step = one block k = i .. nb16 - 1 of the tile's sum (the sum
    // keeps its order).  Inputs live in the strictly-upper tiles and in Ts, results go to the lower tiles: a tile may be stored while
    // the next one's fragments are in flight.
    {
        struct It {
            int i, j, k, nb;
            bool fin;
            __device__ __forceinline__ bool done() const { return fin; }
            __device__ __forceinline__ bool last() const { return k == nb - 1; }
            __device__ __forceinline__ void skip(int n) {   // n tiles on, n <= 4: at most four row changes
                int r = i, c = j + n;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c > r) { c -= r + 1; ++r; }
                if (r >= nb) { fin = true; return; }
                i = r;
                j = c;
                k = r;
            }
            __device__ __forceinline__ void advance() {
                if (!last()) { ++k; return; }
                skip(4);
            }
        };
        constexpr int TS = 128 * DL;   // Ts = As + TS
        It it0{0, 0, 0, nb16, false};
        if (wave) it0.skip(wave);
        d4_t c = {0.0, 0.0, 0.0, 0.0};
        small_mfma_steps(
            it0,
            [&](const It& q, SmallFrag& f) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int kq = 4 * kk + fk;
                    const int row = (16 * q.k + kq) * DL + fl, tile = TS + kq + 16 * fl;
                    f.a[kk] = As[q.k == q.i ? tile + 256 * q.i : row + 16 * q.i];   // T[k][i] (kq, m)
                    f.b[kk] = As[q.k == q.j ? tile + 256 * q.j : row + 16 * q.j];   // T[k][j] (kq, n)
                }
            },
            [&](const It&, const SmallFrag& f) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) c = mfma16(f.b[kk], f.a[kk], c);
            },
            [&](const It& q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    As[(16 * q.j + fk + 4 * r) * DL + 16 * q.i + fl] = c[r];
                    c[r] = 0.0;
                }
            });
    }
    if (stp) stp->mark(16);
    __syncthreads();
    if (stp) stp->mark(17);
    // mirror the strictly-lower tiles into the upper triangle (L^-T is no longer needed): K^-1 becomes a full symmetric image
    small_mirror_lower(As, nb16);
    __syncthreads();
}


// K_y = K_f(a, l) + b I into the LDS image (1/l in the scratch), Cholesky + inverse on the leading ceil(N/16) blocks,
// K_y^-1 as a full symmetric image over the dead factor.  Returns log L_ii in thread i (their sum = logdet / 2: small_alpha).
template <bool MATERN, class Idle = NoIdleWork>
__device__ __forceinline__ double small_factor_inverse(double* As, double* Ts, const SmallPts& pts, const Fold& fold0, int D, int N, double a,
                                                       double b, int* __restrict__ info, SmallPairs* pairs, double* __restrict__ kc,
                                                       SmallTrace& st, Idle&& idle0 = Idle{}) {
    const double lg = small_build_factor<MATERN>(As, Ts, pts, fold0, D, N, a, b, info, pairs, kc, st, idle0);
    small_inverse_in_place(As, Ts, N, &st);
    st.mark(11);
    return lg;
}

// alpha = K^-1 y (y in the scratch) into the scratch; gb = 1/2 (alpha.alpha - tr K^-1), quad = y.alpha and ld = the sum of the
// threads' lg (log L_ii from small_factor_inverse: logdet / 2) in every thread -- one pair of barriers for the three sums
// side: work of the caller's for the upper two waves while the lower two form the product (it may write scratch that nobody reads
// before the barrier below)
template <class Side = NoIdleWork>
__device__ __forceinline__ void small_alpha(double* As, int N, double lg, double& ld, double& gb, double& quad, Side&& side = Side{}) {
    const int tid = threadIdx.x;
    // N <= 64: two threads per row (waves 0 and 1), each with one half of the columns -- one wave alone took ~2000 cycles for its
    // 8-column trips; the halves are added as p0 + p1
    const bool split = N <= 64;
    if (tid >= 128) side();
    else {
        // eight LDS operand pairs in flight, then their fused multiply-adds in column order (a loop of dependent load -> fma trips
        // ran at ~150 ns per column on the otherwise idle CU).  Columns N .. 8 ceil(N / 8) - 1 exist in the image (identity padding,
        // inside the 16-aligned block) and meet y = 0 there.
        const int i = split ? (tid & 63) : tid, part = split ? (tid >> 6) : 0;
        const int n8 = (N + 7) >> 3, h8 = split ? (n8 + 1) >> 1 : n8;
        double s = 0.0;
        if (i < N) {
            for (int j0 = 8 * part * h8; j0 < 8 * min(n8, (part + 1) * h8); j0 += 8) {
                double av[8], yv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    av[u] = As[i + (j0 + u) * DL];
                    yv[u] = small_scratch(As, SC_Y + j0 + u);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) s = fma(av[u], yv[u], s);
            }
        }
        small_scratch(As, (part ? SC_GZ : SC_ALPHA) + i) = s;   // the second halves: the optimiser-gradient slots (dead between evaluations)
    }
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    if (tid < 128) {
        double al = small_scratch(As, SC_ALPHA + tid);
        if (split && tid < 64) {
            al += small_scratch(As, SC_GZ + tid);
            small_scratch(As, SC_ALPHA + tid) = al;   // read by others behind the barriers of the sums below
        }
        if (tid < N) {
            s1 = al * al - As[tid + tid * DL];
            s2 = small_scratch(As, SC_Y + tid) * al;
        }
    }
    small_block_sum3(s1, s2, lg, As);
    gb = 0.5 * s1;
    quad = s2;
    ld = lg;
}

// Gradient contractions over the pairs i >= j with W = alpha alpha^T - K^-1 (src/gaussian-process-regressor.cpp:66-127 without
// the (D + 1) N x N tensor of src/regressor.cpp:110-134):
//   returns  sa = sum W.*K_f;  leaves  gl[d] = (1 / l_d) sum_{i>j} W_ij c_ij ((x_id - x_jd) / l_d)^2  in the scratch (SC_GL + d), d < D <= 128.
// (1) One pair per slot, the slots of the kernel-function pass (same thread, pair values from *pairs / the pair scratch): w, sa, and
// G_ij = 1/2 w c_ij over K^-1 in the lower triangle, zeros on its diagonal (nothing reads K^-1 after small_alpha; both contractions
// read the lower triangle only).  (2) The contraction: on the matrix cores (points staged in LDS) or with lanes over the
// DIMENSIONS: a wave takes whole rows i (dealt 0 1 2 3 3 2 1 0 over the waves), 64 / LP pairs of a row per step with
// LP = min(64, 2^ceil(log2 D)) lanes each (two dimensions per lane for D > 64), four steps' loads in flight; the point coordinates
// come from the D x N original (a pair's D values are contiguous), G from LDS.  The per-wave partial sums are added over the lanes of
// a dimension by xor butterflies and over the waves as (w0 + w1) + (w2 + w3): one fixed order.
// e1, e2: two more per-thread addends of the caller's, summed over the workgroup behind the same barriers as sa (in place).
template <bool MATERN>
__device__ __forceinline__ double small_grad(double* As, const SmallPts& pts, const Fold& fold0, const SmallPairs* pairs,
                                             const double* __restrict__ kc, int D, int N, double a, bool want_grad, SmallTrace& st,
                                             double& e1, double& e2) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb16 = (N + 15) >> 4;
    const bool mm = pts.lds != 0 && small_mc_form_ok(As, D);   // matrix-core contraction (the same expansion, the same guard as the Gram pass)
    double sa = 0.0;
    if (want_grad) {
        // one batch of four slots, without branches around the loads: alpha_i, alpha_j, K^-1_ij of the four first
        Fold fold = fold0;
        const int total = fold.total(), slen = small_stash_len(N);
        auto batch = [&](auto&& get) {
            int iq[4], jq[4];
            bool live[4];
            double ai[4], aj[4], kinv[4], kv[4], cv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                live[u] = fold.next(iq[u], jq[u]);
                ai[u] = small_scratch(As, SC_ALPHA + iq[u]);
                aj[u] = small_scratch(As, SC_ALPHA + jq[u]);
                kinv[u] = As[iq[u] + jq[u] * DL];
                get(u, kv[u], cv[u]);
            }
            __builtin_amdgcn_sched_barrier(0);   // dead slots read element (0, 0): before its owner's store
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool diag = iq[u] == jq[u];
                const double w = (diag ? 0.5 : 1.0) * (ai[u] * aj[u] - kinv[u]);
                sa = live[u] ? fma(w, diag ? a : kv[u], sa) : sa;
                if (live[u]) As[iq[u] + jq[u] * DL] = diag ? 0.0 : 0.5 * (w * (MATERN ? cv[u] : kv[u]));
            }
        };
#pragma unroll
        for (int s0 = 0; s0 < SMALL_SR; s0 += 4) {
            if (s0 * 256 >= total) break;
            batch([&](int u, double& k, double& c) {
                k = pairs->k[s0 + u];
                c = pairs->c[s0 + u];
            });
        }
        for (int e0 = SMALL_SR * 256; e0 < total; e0 += 4 * 256)
            batch([&](int u, double& k, double& c) {
                const int q = min(e0 - SMALL_SR * 256 + 256 * u + tid, slen - 1);   // any address inside the scratch may be read
                if (pts.stash_off >= 0) {
                    k = small_free(As, pts.base_col, pts.stash_off + q);
                    c = MATERN ? small_free(As, pts.base_col, pts.stash_off + slen + q) : 0.0;
                } else {
                    k = kc[q];
                    c = MATERN ? kc[128 * 128 + q] : 0.0;
                }
            });
        // padding rows of the last block: K^-1 is the identity there, G has no weight
        {
            const int i = N + (tid >> 4);
            if (i < 16 * nb16 && (tid & 15) == 0) As[i + i * DL] = 0.0;
        }
    }
    st.mark(18);
    small_block_sum3(sa, e1, e2, As);   // its barriers publish the lower triangle of G
    const double sa_t = sa;
    st.mark(12);
    if (!want_grad) return sa_t;

    if (mm) {
        // With G symmetric, sum_j x_jp (x_jp s_j - (G X~)_jp) needs the strictly-lower triangle Gl only:
        //     P_p = sum_j [ x_jp^2 r_j + (Gl X~^2)_jp - 2 x_jp (Gl X~)_jp ],   r = Gl 1 (row sums),  X~^2 elementwise
        // (the column sums of Gl weigh x_jp^2 exactly as Gl weighs the squares of the other factor) -- no mirror image of G, half
        // the block products.  (a) r: thread (j, part) adds its quarter (half for N > 64) of row j, parts combined in a fixed order.
        // (b) the products on the matrix cores: a unit = (block row tj, dimension tile tp), tj + 1 column blocks (steps) each; the
        // units, ordered by descending cost, are dealt forwards and backwards over the waves (u mod 8 = w or 7 - w: 1 + 4, 2 + 3
        // blocks ...).  Lane (fl, fk) holds (j = 16 tj + fl, p = 16 tp + fk + 4 q); a unit's sums over its 16 j go to the wave's
        // own partial array.
        const int fl = lane & 15, fk = lane >> 4;
        const PtsLds xc{As, pts.base_col, pts.Dp};
        const int Nb = 16 * nb16;
        const int shift = Nb <= 64 ? 6 : 7, parts = 256 >> shift, ck = Nb >> (8 - shift);   // ck = Nb / parts
        {
            const int j = tid & ((1 << shift) - 1), part = tid >> shift;
            double r = 0.0;
            for (int k0 = part * ck; k0 < (part + 1) * ck; k0 += 8) {
                double g[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) g[u] = As[min(j, Nb - 1) + min(k0 + u, Nb - 1) * DL];
#pragma unroll
                for (int u = 0; u < 8; ++u) r += (k0 + u < (part + 1) * ck && k0 + u < j) ? g[u] : 0.0;
            }
            small_scratch(As, SC_GZ + (part << shift) + j) = r;   // 256 partial sums: the optimiser-gradient slots, rewritten after this pass
        }
        small_scratch(As, SC_BTL + wave * NLL_SMALL_MAX_D + lane) = 0.0;
        small_scratch(As, SC_BTL + wave * NLL_SMALL_MAX_D + 64 + lane) = 0.0;
        __syncthreads();
        if (st.on) {   // probe: the row-sum step (slot 22), counted inside the pass's slot 19 as well
            const long long t_now = wall_clock64();
            if (tid == 0) st.slot_ref(22) += t_now - st.t_prev;
        }
        struct It {
            int tj, tp, kb, ntp, w;
            bool odd, fin;
            __device__ __forceinline__ bool done() const { return fin; }
            __device__ __forceinline__ bool last() const { return kb == tj; }
            __device__ __forceinline__ void skip(int n) {   // n units on
                int p = tp + n, j = tj;
                while (p >= ntp) { p -= ntp; --j; }
                if (j < 0) { fin = true; return; }
                tp = p;
                tj = j;
                kb = 0;
            }
            __device__ __forceinline__ void advance() {
                if (!last()) { ++kb; return; }
                skip(odd ? 1 + 2 * w : 7 - 2 * w);
                odd = !odd;
            }
        };
        // a finished unit: x^2 r - 2 x Y + Z summed over its 16 rows j, into the wave's partial array
        auto finish = [&](int tj, int tp, d4_t& ay, d4_t& az) {
            const int j = 16 * tj + fl;
            double rj = small_scratch(As, SC_GZ + j) + small_scratch(As, SC_GZ + (1 << shift) + j);
            if (parts == 4) rj += small_scratch(As, SC_GZ + 128 + j) + small_scratch(As, SC_GZ + 192 + j);
            double v[4];   // the four row sums first (independent: they interleave), then the four updates
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double xv = xc(j, 16 * tp + fk + 4 * r);
                v[r] = row_sum(fma(xv, fma(xv, rj, -2.0 * ay[r]), az[r]));
                ay[r] = 0.0;
                az[r] = 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = 16 * tp + fk + 4 * r;
                if (fl == 0 && p < D) small_scratch(As, SC_BTL + wave * NLL_SMALL_MAX_D + p) += v[r];
            }
        };
        It it0{nb16 - 1, 0, 0, (pts.Dp - 1) >> 4, wave, false, false};
        if (wave) it0.skip(wave);
        d4_t accy = {0.0, 0.0, 0.0, 0.0}, accz = {0.0, 0.0, 0.0, 0.0};
        small_mfma_steps(
            it0,
            [&](const It& q, SmallFrag& f) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    f.a[kk] = As[(16 * q.kb + 4 * kk + fk) * DL + 16 * q.tj + fl];
                    f.b[kk] = xc(16 * q.kb + 4 * kk + fk, 16 * q.tp + fl);
                }
            },
            [&](const It& q, const SmallFrag& f) {
                const bool diag = q.kb == q.tj;   // the diagonal block: its strictly-lower half
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double g = (!diag || 4 * kk + fk < fl) ? f.a[kk] : 0.0;
                    accy = mfma16(f.b[kk], g, accy);
                    accz = mfma16(f.b[kk] * f.b[kk], g, accz);
                }
            },
            [&](const It& q) { finish(q.tj, q.tp, accy, accz); });
        st.mark(19);
        __syncthreads();
        st.mark(20);
        if (tid < D) {
            const double p = (small_scratch(As, SC_BTL + tid) + small_scratch(As, SC_BTL + NLL_SMALL_MAX_D + tid)) +
                             (small_scratch(As, SC_BTL + 2 * NLL_SMALL_MAX_D + tid) + small_scratch(As, SC_BTL + 3 * NLL_SMALL_MAX_D + tid));
            const double il = small_scratch(As, SC_INVL + tid);
            small_scratch(As, SC_GL + tid) = 2.0 * il * (il * il) * p;
        }
        __syncthreads();
        st.mark(13);
        return sa_t;
    }

    int LP = 1;
    while (LP < D && LP < 64) LP <<= 1;
    const int PW = 64 / LP, dl = lane & (LP - 1), jj = lane / LP;
    const bool wide = D > 64;
    // lanes beyond D read dimension D - 1 and weigh it with 0
    const int d0 = min(dl, D - 1), d1 = min(dl + 64, D - 1);
    const bool v0 = dl < D, v1 = wide && dl + 64 < D;
    const double il0 = v0 ? small_scratch(As, SC_INVL + d0) : 0.0, il1 = v1 ? small_scratch(As, SC_INVL + d1) : 0.0;
    double acc0 = 0.0, acc1 = 0.0;
    {
        const PtsCols x{pts.X, D};
        constexpr int U = 8;
        for (int i = 1; i < N; ++i) {
            const int r8 = i & 7;
            if ((r8 < 4 ? r8 : 7 - r8) != wave) continue;
            const double xi0 = x(i, d0), xi1 = wide ? x(i, d1) : 0.0;
            for (int j0 = 0; j0 < i; j0 += U * PW) {
                double gv[U], x0[U], x1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = j0 + u * PW + jj, jc = min(j, i - 1);   // past the row's end: the last pair again, weight 0
                    const double g = As[i + jc * DL];   // the lower triangle (all lanes of a pair read one address)
                    gv[u] = j < i ? g : 0.0;
                    x0[u] = x(jc, d0);
                    x1[u] = wide ? x(jc, d1) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double t0 = (xi0 - x0[u]) * il0;
                    acc0 = fma(gv[u], t0 * t0, acc0);
                    if (wide) {
                        const double t1 = (xi1 - x1[u]) * il1;
                        acc1 = fma(gv[u], t1 * t1, acc1);
                    }
                }
            }
        }
    }
    for (int m = LP; m < 64; m <<= 1) {
        acc0 += __shfl_xor(acc0, m);
        if (wide) acc1 += __shfl_xor(acc1, m);
    }
    if (jj == 0) {
        if (v0) small_scratch(As, SC_BTL + wave * NLL_SMALL_MAX_D + dl) = acc0;
        if (v1) small_scratch(As, SC_BTL + wave * NLL_SMALL_MAX_D + dl + 64) = acc1;
    }
    __syncthreads();
    if (tid < D) {
        const double p = (small_scratch(As, SC_BTL + tid) + small_scratch(As, SC_BTL + NLL_SMALL_MAX_D + tid)) +
                         (small_scratch(As, SC_BTL + 2 * NLL_SMALL_MAX_D + tid) + small_scratch(As, SC_BTL + 3 * NLL_SMALL_MAX_D + tid));
        small_scratch(As, SC_GL + tid) = 2.0 * p * small_scratch(As, SC_INVL + tid);   // G = 1/2 w c
    }
    __syncthreads();
    st.mark(13);
    return sa_t;
}

template <bool MATERN>
__global__ __launch_bounds__(256) void nll_small_kernel(const NllSmallArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* As = reinterpret_cast<double*>(smem);
    double* Ts = As + 128 * DL;
    const int tid = threadIdx.x;
    const double* __restrict__ X = args.X;
    double* __restrict__ out = args.out + (long)blockIdx.x * args.out_stride;
    int* __restrict__ info = args.info + blockIdx.x;
    const double* __restrict__ in_dev = args.in_dev ? args.in_dev + (long)blockIdx.x * args.in_stride : nullptr;
    const int D = args.D, N = args.N, want_grad = args.want_grad;
    // hyper-parameters and targets travel in the kernel argument block (no upload); D > 32 (NLL_SMALL_MAX_ARG_D) falls back to a device buffer
    const double a = in_dev ? in_dev[0] : args.a, b = in_dev ? in_dev[1] : args.b;
    for (int d = tid; d < D; d += 256) small_scratch(As, SC_INVL + d) = 1.0 / (in_dev ? in_dev[2 + d] : args.ell[d]);
    for (int i = tid; i < 128; i += 256)
        small_scratch(As, SC_Y + i) = i < N ? (in_dev ? in_dev[2 + D + i] : args.y[i]) : 0.0;
    if (tid == 0) *info = 0;
    const SmallPts pts = small_pts_stage(As, X, args.XTr, D, N, args.x_lds != 0);
    __syncthreads();

    double* __restrict__ kc = want_grad ? args.kc : nullptr;
    SmallTrace st;
    const Fold fold0(N);
    SmallPairs pr;
    const double lg = small_factor_inverse<MATERN>(As, Ts, pts, fold0, D, N, a, b, info, want_grad ? &pr : nullptr, kc, st);
    double ld, gb, quad, e1 = 0.0, e2 = 0.0;
    small_alpha(As, N, lg, ld, gb, quad);
    if (tid < N && args.batch <= 1) out[NLL_SMALL_OUT_ALPHA + tid] = small_scratch(As, SC_ALPHA + tid);   // batch mode: 8 output words per parameter set
    const double sa_t = small_grad<MATERN>(As, pts, fold0, &pr, kc, D, N, a, want_grad != 0, st, e1, e2);
    if (want_grad && tid < D) out[NLL_SMALL_OUT_GL + tid] = small_scratch(As, SC_GL + tid);
    if (tid == 0) {
        out[0] = sa_t;
        out[1] = gb;
        out[2] = quad;
        out[3] = 2.0 * ld;
        out[4] = (double)__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------------
// map_opt_kernel: the whole MAP fit (or one objective evaluation) in ONE single-workgroup launch.
//
// Objective (maximised), with y = z[0..ny) or the fixed targets and (a, b, l) = the hyper variables or the fixed defaults:
//   f = sum_p log BTL_p(y)                                   src/preference-regressor.cpp:151-154, utils.hpp:25-29 (no max-subtraction)
//     - 1/2 y^T K^-1 y - 1/2 log|K| - N/2 log 2 pi           :165-170  /  src/gaussian-process-regressor.cpp:174-180
//     + log-normal priors of a, b, l_d (nh > 0)              :175-192  /  :181-192
//   df/dy = sum_p dBTL_p / BTL_p - K^-1 y                    :199-221, utils.hpp:31-52
//   df/d(a, b, l)                                            :53-115   /  :66-127   (small_grad, D <= 128)
// Optimiser: optim::MaximizeBounded of host/device.cpp statement by statement (projected gradient, two-loop recursion over
// the last 8 pairs, Armijo backtracking by halving, at most 31 trials per direction) as a flat state machine with one
// objective evaluation per turn.  The optimiser state (x, g, direction, history) is REPLICATED in the registers of each of
// the four waves -- lane l owns variables l, l + 64, l + 128 -- so its ~25 reductions per iteration are wave
// shuffles without a workgroup barrier; all waves execute the same instructions on the same values and therefore take the
// same branches.  The evaluation itself is the one-workgroup pipeline of nll_small_kernel; for fixed hyper-parameters
// (the reference's use_map_hyperparams = false, src/preference-regressor.cpp:161-162) K^-1 is built once and stays in LDS.
// `budget` < max_evals: the launch stops after `budget` evaluations and leaves the state in `state`; the next launch (fresh = 0)
// continues from it with the same machine code -- the one-launch-per-evaluation form the tests compare the single launch with.
// ---------------------------------------------------------------------------------------------------------
constexpr int MH = MAP_OPT_HIST;

template <int KV>
__device__ __forceinline__ double wave_dot(const double (&u)[KV], const double (&v)[KV]) {
    double p = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k) p += u[k] * v[k];
    return wave_sum(p);
}
// mathtoolbox::GetLogOfLogNormalDist / ...Derivative with lx = log x supplied by the caller (the optimiser's
// own variable when it runs in the logarithms) and hl = 1/2 log(2 pi s2) formed once per launch
__device__ __forceinline__ double dev_log_lognormal(double lx, double mu, double s2, double hl) {
    return -lx - hl - (lx - mu) * (lx - mu) / (2.0 * s2);
}
__device__ __forceinline__ double dev_log_lognormal_d(double x, double lx, double mu, double s2) { return (mu - s2 - lx) / (s2 * x); }

// KV: optimiser variables per lane (64 KV >= n)
template <bool MATERN, int KV>
__global__ __launch_bounds__(256) void map_opt_kernel(const MapOptArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* As = reinterpret_cast<double*>(smem);
    double* Ts = As + 128 * DL;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* __restrict__ X = args.X;
    const int D = args.D, N = args.N, ny = args.ny, nh = args.nh, n = ny + nh;
    const int P = args.n_prefs, F = args.flat_len;
    const bool btl_lds = F <= SC_BTL_MAX;
    int* __restrict__ info = args.info;
    double* __restrict__ state = args.state;
    double* __restrict__ out = args.out;
    SmallTrace st;
    st.on = args.trace != nullptr;
    st.As = As;
    if (st.on && tid == 0)
        for (int q = 0; q < MAP_OPT_TRACE_SLOTS; ++q) st.slot_ref(q) = 0;
    const long long tr_begin = st.on ? wall_clock64() : 0;
    st.t_prev = tr_begin;
#define MAP_T(slot) st.mark(slot)

    // ---- optimiser state: one replica per wave ----
    double x[KV], g[KV], xt[KV], d[KV], lo[KV], hi[KV], S[MH][KV], Y[MH][KV], rho[MH];
    double fx = 0.0, t = 1.0, sy_last = 0.0, yy_last = 1.0;
    int cnt = 0, bt = 0, evals = 0, phase = 0, done = 0;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int e = lane + 64 * k;
        lo[k] = e < n ? args.lower[e] : 0.0;
        hi[k] = e < n ? args.upper[e] : 0.0;
        x[k] = g[k] = d[k] = 0.0;
        xt[k] = e < n ? fmin(hi[k], fmax(lo[k], args.z0[e])) : 0.0;
#pragma unroll
        for (int h = 0; h < MH; ++h) S[h][k] = Y[h][k] = 0.0;
    }
#pragma unroll
    for (int h = 0; h < MH; ++h) rho[h] = 0.0;
    if (!args.fresh) {
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int e = lane + 64 * k;
            x[k] = state[0 * MAP_OPT_MAX_VARS + e];
            g[k] = state[1 * MAP_OPT_MAX_VARS + e];
            xt[k] = state[2 * MAP_OPT_MAX_VARS + e];
            d[k] = state[3 * MAP_OPT_MAX_VARS + e];
#pragma unroll
            for (int h = 0; h < MH; ++h) {
                S[h][k] = state[(4 + h) * MAP_OPT_MAX_VARS + e];
                Y[h][k] = state[(4 + MH + h) * MAP_OPT_MAX_VARS + e];
            }
        }
        const double* sc = state + (4 + 2 * MH) * MAP_OPT_MAX_VARS;
        fx = sc[0]; t = sc[1]; sy_last = sc[2]; yy_last = sc[3];
#pragma unroll
        for (int h = 0; h < MH; ++h) rho[h] = sc[4 + h];
        cnt = (int)sc[12]; bt = (int)sc[13]; evals = (int)sc[14]; phase = (int)sc[15]; done = (int)sc[16];
    }

    // ---- fixed inputs ----
    for (int i = tid; i < 128; i += 256) small_scratch(As, SC_Y + i) = (ny == 0 && i < N) ? args.y_fixed[i] : 0.0;
    if (nh == 0)
        for (int dd = tid; dd < D; dd += 256) small_scratch(As, SC_INVL + dd) = 1.0 / args.r0;
    if (tid == 0) *info = 0;
    const SmallPts pts = small_pts_stage(As, X, args.XTr, D, N, args.x_lds != 0);
    const Fold fold0(N);
    // the first preference tuple of this thread and the first tuple memberships of data point `tid`: indices in registers
    constexpr int RC = 4;
    int po = 0, pm = 0, pmem = 0, co = 0, cm = 0, cidx0 = 0, cidx1 = 0, cidx2 = 0, cidx3 = 0;
    // tuple tq (, tq + 32, ...) on the QUAD of threads 128 + 4 tq .. + 3 of the upper two waves, one member per lane (see btl_tuples):
    // offset, size and this lane's member of the quad's first tuple in registers
    const int tq = tid >= 128 ? (tid - 128) >> 2 : -1, ti = tid & 3;
    if (tq >= 0 && tq < P) {
        po = args.pref_off[tq];
        pm = args.pref_off[tq + 1] - po;
        pmem = args.pref_flat[po + min(ti, pm - 1)];
    }
    if (tid < ny && P > 0) {
        co = args.csc_off[tid];
        cm = args.csc_off[tid + 1] - co;
        if (cm > 0) cidx0 = args.csc_ent[co];
        if (cm > 1) cidx1 = args.csc_ent[co + 1];
        if (cm > 2) cidx2 = args.csc_ent[co + 2];
        if (cm > 3) cidx3 = args.csc_ent[co + 3];
    }
    __syncthreads();

    const double hl_a = 0.5 * log(2.0 * M_PI * args.s2_a), hl_b = 0.5 * log(2.0 * M_PI * args.s2_b), hl_r = 0.5 * log(2.0 * M_PI * args.s2_r);
    bool have_factor = false, bad = false;
    double ld = 0.0, a = args.a0, b = args.noiseless ? 0.0 : args.b0;
    double f_last = 0.0;
    int budget = args.budget;
    while (budget > 0 && !done) {
        --budget;
        // ---- publish the trial point: y into the scratch, hyper-parameters in linear space ----
        // (every wave holds the same replica: wave k & 3 publishes variables 64 k .. 64 k + 63)
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            if ((k & 3) == wave) {
                const int e = lane + 64 * k;
                if (e < ny) small_scratch(As, SC_Y + e) = xt[k];
                else if (e < n) {
                    const int hq = e - ny;
                    const double v = args.log_hyper ? exp(xt[k]) : xt[k];
                    small_scratch(As, SC_LZ + hq) = args.log_hyper ? xt[k] : log(xt[k]);
                    if (hq == 0) small_scratch(As, SC_AB) = v;
                    else if (hq == 1) small_scratch(As, SC_AB + 1) = args.noiseless ? 0.0 : v;
                    else {
                        small_scratch(As, SC_ELL + hq - 2) = v;
                        small_scratch(As, SC_INVL + hq - 2) = 1.0 / v;
                    }
                }
            }
        }
        __syncthreads();
        if (nh) {
            a = small_scratch(As, SC_AB);
            b = small_scratch(As, SC_AB + 1);
        }
        MAP_T(0);
        // ---- Bradley-Terry-Luce terms: tuple q on thread 128 + q (, + 128, ...).  They need the published goodness values only, so
        // the upper two waves form them where they would otherwise wait: during the pivot chain of the first diagonal tile when the
        // matrix is rebuilt (hyper-parameters among the variables), beside the product K^-1 y of the lower two waves otherwise.
        // contrib: the per-member terms d BTL_p / BTL_p, in the LDS scratch (flat_len <= 640) or in global memory -- two
        // instantiations of the same code, not a run-time pointer choice (a pointer that may be LDS or global is a generic
        // pointer: flat loads / stores)
        double lsum = 0.0, gsum = 0.0, gy = 0.0;
        auto btl_tuples_in = [&](auto&& contrib) {
            const double bs = args.btl_scale;
            // one tuple: o = its offset in the flat list, m = its size, (m0 .. m3) = its first RC members
            auto tuple_terms = [&](int o, int m, int m0, int m1, int m2, int m3) {
                // the first RC members from registers (a select chain: a dynamically indexed private array lives in scratch memory,
                // one flat load per member on the dependent path member -> y -> exp), the rest from global memory
                auto member = [&](int i) {
                    if (i >= RC) return args.pref_flat[o + i];
                    int r = m0;
                    r = (i == 1) ? m1 : r;
                    r = (i == 2) ? m2 : r;
                    r = (i == 3) ? m3 : r;
                    return r;
                };
                const double f0 = small_scratch(As, SC_Y + m0);
                double sum = 0.0;
                for (int i = 0; i < m; ++i) sum += exp(small_scratch(As, SC_Y + member(i)) / bs);
                const double v = exp(f0 / bs) / sum;                       // CalcBtl
                lsum += log(v);                                            // calc_log_likelihood
                const double tmp = -v * v / bs;                            // CalcBtlDerivative
                double sum2 = 0.0;
                for (int i = 1; i < m; ++i) {
                    const double r = exp((small_scratch(As, SC_Y + member(i)) - f0) / bs);   // used twice by the reference: once here
                    sum2 += r;
                    contrib(o + i) = r;
                }
                contrib(o) = (tmp * (-sum2)) / v;
                for (int i = 1; i < m; ++i) contrib(o + i) = (tmp * contrib(o + i)) / v;
            };
            // Up to four members: one per lane of the quad.  The exponentials of the members run side by side (an exp is ~120
            // cycles of issue on its wave, a log 480: a tuple's six exponentials, its logarithm and five divisions one after the other
            // were ~2000 cycles), the two sums are butterflies inside the quad -- (e0 + e1) + (e2 + e3), the reference's order
            // e0 + e1 + e2 for the three members of a line search's tuples -- and lane 0's values reach the others by quad
            // broadcast.  Larger tuples: lane 0 of the quad alone, member by member.
            auto quad_terms = [&](int o, int m, int mem) {
                const bool act = ti < m;
                const double yi = small_scratch(As, SC_Y + mem);
                const double f0 = dpp_f64<0x00>(yi);                                    // quad_perm [0,0,0,0]
                const double e = act ? exp(yi / bs) : 0.0;
                const double r = (act && ti >= 1) ? exp((yi - f0) / bs) : 0.0;          // used twice by the reference: once here
                double sum = e, sum2 = r;
                sum += dpp_f64<0xB1>(sum);
                sum2 += dpp_f64<0xB1>(sum2);
                sum += dpp_f64<0x4E>(sum);
                sum2 += dpp_f64<0x4E>(sum2);
                const double v = dpp_f64<0x00>(e) / sum;                                 // CalcBtl: exp(f0 / bs) / sum
                if (ti == 0) lsum += log(v);                                             // calc_log_likelihood
                const double tmp = -v * v / bs;                                          // CalcBtlDerivative
                if (act) contrib(o + ti) = ti == 0 ? (tmp * (-sum2)) / v : (tmp * r) / v;
            };
            if (tq >= 0) {
                for (int p = tq; p < P; p += 32) {          // (the quad's lanes run the same trips)
                    int o = po, m = pm, mem = pmem;
                    if (p != tq) {                          // more than 32 tuples: indices from global memory
                        o = args.pref_off[p];
                        m = args.pref_off[p + 1] - o;
                        mem = args.pref_flat[o + min(ti, m - 1)];
                    }
                    if (m <= RC) quad_terms(o, m, mem);
                    else if (ti == 0)
                        tuple_terms(o, m, args.pref_flat[o], args.pref_flat[o + 1], args.pref_flat[o + 2], args.pref_flat[o + 3]);
                }
            }
        };
        auto btl_tuples = [&]() {
            if (btl_lds) btl_tuples_in([&](int q) -> double& { return small_scratch(As, SC_BTL + q); });
            else btl_tuples_in([&](int q) -> double& { return args.btl_scratch[q]; });
        };
        auto btl_gather_in = [&](auto&& contrib) {
            if (tid < ny) {
                for (int i = 0; i < cm; ++i) {
                    int e;
                    if (i >= RC) e = args.csc_ent[co + i];
                    else {
                        e = cidx0;
                        e = (i == 1) ? cidx1 : e;
                        e = (i == 2) ? cidx2 : e;
                        e = (i == 3) ? cidx3 : e;
                    }
                    gsum += contrib(e);
                }
            }
        };
        const bool rebuild = nh || !have_factor;
        SmallPairs pr;   // kernel values and derivative weights of this thread's pairs, from the kernel-function pass to the gradient's
        double lg = 0.0, ld_now, gb, quad;
        if (rebuild) {
            lg = small_factor_inverse<MATERN>(As, Ts, pts, fold0, D, N, a, b, info, nh ? &pr : nullptr, args.kc, st, btl_tuples);
            have_factor = true;
            bad = __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            small_alpha(As, N, lg, ld_now, gb, quad);
            ld = ld_now;
        } else small_alpha(As, N, lg, ld_now, gb, quad, btl_tuples);
        MAP_T(1);
        if (bad && nh && tid == 0) *info = 0;   // every thread has read it (barriers of small_alpha); the next factorisation starts clean
        // the contributions are published (barriers of small_alpha): gathered now, small_grad reuses their scratch
        if (btl_lds) btl_gather_in([&](int q) -> double& { return small_scratch(As, SC_BTL + q); });
        else btl_gather_in([&](int q) -> double& { return args.btl_scratch[q]; });
        // the sum of the tuples' log-likelihoods and (hyper-parameters among the variables) of the D length-scale prior terms, one per
        // thread (:175-192), ride on the barriers of small_grad's own sum
        double sa_t = 0.0, btl_sum = lsum, reg_l = 0.0;
        if (nh) {
            reg_l = tid < D ? dev_log_lognormal(small_scratch(As, SC_LZ + 2 + tid), args.mu_r, args.s2_r, hl_r) : 0.0;
            sa_t = small_grad<MATERN>(As, pts, fold0, &pr, args.kc, D, N, a, true, st, btl_sum, reg_l);   // length-scale gradient -> scratch (SC_GL)
        } else btl_sum = small_block_sum(lsum, As);

        // ---- gradient of the BTL terms wrt the goodness values: the contributions gathered in tuple order (:202-216), minus alpha (:219) ----
        if (tid < ny) gy = gsum - small_scratch(As, SC_ALPHA + tid);
        MAP_T(3);

        // ---- value ----
        double f = btl_sum + (-0.5 * quad - 0.5 * (2.0 * ld) - 0.5 * N * log(2.0 * M_PI));
        if (nh) {
            // log-normal priors (:175-192)
            double reg = dev_log_lognormal(small_scratch(As, SC_LZ), args.mu_a, args.s2_a, hl_a);
            if (!args.noiseless) reg += dev_log_lognormal(small_scratch(As, SC_LZ + 1), args.mu_b, args.s2_b, hl_b);
            reg += reg_l;
            f += reg;
        }
        f_last = f;
        // ---- gradient wrt the optimiser's variables (minimisation: phi = -f) ----
        for (int e = tid; e < n; e += 256) {   // n <= 320; ny <= 128: the goodness values are on the first trip, thread e
            double gz;
            if (e < ny) gz = gy;
            else {
                const int hq = e - ny;
                double gx, xv;
                if (hq == 0) {
                    xv = a;
                    gx = sa_t / a + dev_log_lognormal_d(a, small_scratch(As, SC_LZ), args.mu_a, args.s2_a);
                } else if (hq == 1) {
                    xv = b;
                    gx = args.noiseless ? 0.0 : gb + dev_log_lognormal_d(b, small_scratch(As, SC_LZ + 1), args.mu_b, args.s2_b);
                } else {
                    xv = small_scratch(As, SC_ELL + hq - 2);
                    gx = small_scratch(As, SC_GL + hq - 2) + dev_log_lognormal_d(xv, small_scratch(As, SC_LZ + hq), args.mu_r, args.s2_r);
                }
                gz = args.log_hyper ? gx * xv : gx;
            }
            small_scratch(As, SC_GZ + e) = -gz;
            if (args.eval_only) out[MAP_OPT_OUT_G + e] = gz;
        }
        __syncthreads();
        MAP_T(4);
        ++evals;
        st.count(6);
        if (args.eval_only) {
            done = 1;
#pragma unroll
            for (int k = 0; k < KV; ++k) x[k] = xt[k];
            fx = -f;
            break;
        }
        double gt[KV];
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int e = lane + 64 * k;
            gt[k] = e < n ? small_scratch(As, SC_GZ + e) : 0.0;
        }
        const double ft = bad ? HUGE_VAL : -f;

        // ---- advance the optimiser by one evaluation (per wave, no barriers) ----
        bool need_dir = false;
        if (phase == 0) {
#pragma unroll
            for (int k = 0; k < KV; ++k) { x[k] = xt[k]; g[k] = gt[k]; }
            fx = ft;
            phase = 1;
            need_dir = true;
        } else {
            double sv[KV], yv[KV];
#pragma unroll
            for (int k = 0; k < KV; ++k) { sv[k] = xt[k] - x[k]; yv[k] = gt[k] - g[k]; }
            const double gs = wave_dot(g, sv);
            if (isfinite(ft) && ft <= fx + 1e-4 * gs) {
                const double sy = wave_dot(sv, yv), yy = wave_dot(yv, yv);
                if (sy > 1e-10 * yy && sy > 0.0) {
                    if (cnt == MH) {
#pragma unroll
                        for (int h = 0; h + 1 < MH; ++h) {
                            rho[h] = rho[h + 1];
#pragma unroll
                            for (int k = 0; k < KV; ++k) { S[h][k] = S[h + 1][k]; Y[h][k] = Y[h + 1][k]; }
                        }
                        cnt = MH - 1;
                    }
#pragma unroll
                    for (int h = 0; h < MH; ++h)
                        if (h == cnt) {
                            rho[h] = 1.0 / sy;
#pragma unroll
                            for (int k = 0; k < KV; ++k) { S[h][k] = sv[k]; Y[h][k] = yv[k]; }
                        }
                    ++cnt;
                    sy_last = sy;
                    yy_last = yy;
                }
                // NLopt's relative stopping tests on the accepted step (sls_nll_set_tolerances; 0 = off)
                if (lbfgs_f_stalled(fx, ft, args.ftol_rel)) done = 1;
                if (args.xtol_rel > 0.0) {
                    bool moved = false;
#pragma unroll
                    for (int k = 0; k < KV; ++k) moved = moved || (lane + 64 * k < n && lbfgs_x_moved(x[k], xt[k], args.xtol_rel) != 0.0);
                    if (!__any(moved)) done = 1;
                }
#pragma unroll
                for (int k = 0; k < KV; ++k) { x[k] = xt[k]; g[k] = gt[k]; }
                fx = ft;
                need_dir = true;
            } else {
                t *= 0.5;
                st.count(21);   // rejected trial points (probe)
                if (++bt > 30) done = 1;
            }
        }
        if (!done && need_dir) {
            if (evals >= args.max_evals) done = 1;
            else {
                double pg[KV], pm_ = 0.0, pn = 0.0;
#pragma unroll
                for (int k = 0; k < KV; ++k) {
                    double v = g[k];
                    if ((x[k] <= lo[k] && v > 0.0) || (x[k] >= hi[k] && v < 0.0)) v = 0.0;
                    pg[k] = v;
                    pm_ = fmax(pm_, fabs(v));
                    pn += v * v;
                }
                // max |projected gradient| > 0 somewhere?  A vote instead of the wave-wide maximum (only its sign is used; a dependent
                // reduction is ~180 cycles on this chain, tools/probes/lat_probe)
                if (!__any(pm_ > 0.0)) done = 1;
                else {
                    double al[MH];
#pragma unroll
                    for (int k = 0; k < KV; ++k) d[k] = pg[k];
#pragma unroll
                    for (int h = MH - 1; h >= 0; --h) {
                        al[h] = 0.0;
                        if (h < cnt) {
                            al[h] = rho[h] * wave_dot(S[h], d);
#pragma unroll
                            for (int k = 0; k < KV; ++k) d[k] -= al[h] * Y[h][k];
                        }
                    }
                    double gamma = cnt > 0 ? sy_last / yy_last : 1.0 / fmax(1.0, sqrt(wave_sum(pn)));   // |pg|^2 only where it is used
#pragma unroll
                    for (int k = 0; k < KV; ++k) d[k] *= gamma;
#pragma unroll
                    for (int h = 0; h < MH; ++h)
                        if (h < cnt) {
                            const double beta = rho[h] * wave_dot(Y[h], d);
#pragma unroll
                            for (int k = 0; k < KV; ++k) d[k] += S[h][k] * (al[h] - beta);
                        }
#pragma unroll
                    for (int k = 0; k < KV; ++k) d[k] = (pg[k] == 0.0) ? 0.0 : -d[k];
                    double gd = wave_dot(pg, d);
                    if (!(gd < 0.0)) {
                        cnt = 0;
                        gamma = 1.0 / fmax(1.0, sqrt(wave_sum(pn)));
#pragma unroll
                        for (int k = 0; k < KV; ++k) d[k] = -gamma * pg[k];
                        gd = wave_dot(pg, d);
                        if (!(gd < 0.0)) done = 1;
                    }
                    t = 1.0;
                    bt = 0;
                }
            }
        }
        if (!done) {
            if (evals >= args.max_evals) done = 1;
            else {
                double dv[KV];
#pragma unroll
                for (int k = 0; k < KV; ++k) {
                    xt[k] = fmin(hi[k], fmax(lo[k], x[k] + t * d[k]));
                    dv[k] = xt[k] - x[k];
                }
                double dd = 0.0;   // the lane's part of dv . dv: the sum of these non-negative terms is zero exactly when every one is
#pragma unroll
                for (int k = 0; k < KV; ++k) dd += dv[k] * dv[k];
                if (!__any(dd != 0.0)) done = 1;
            }
        }
        MAP_T(5);
    }
    if (st.on && tid == 0) {
        st.slot_ref(7) = wall_clock64() - tr_begin;
        for (int q = 0; q < MAP_OPT_TRACE_SLOTS; ++q) args.trace[q] += st.slot_ref(q);
    }

    // ---- results and the state for a continuation ----
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int e = lane + 64 * k;
            state[0 * MAP_OPT_MAX_VARS + e] = x[k];
            state[1 * MAP_OPT_MAX_VARS + e] = g[k];
            state[2 * MAP_OPT_MAX_VARS + e] = xt[k];
            state[3 * MAP_OPT_MAX_VARS + e] = d[k];
#pragma unroll
            for (int h = 0; h < MH; ++h) {
                state[(4 + h) * MAP_OPT_MAX_VARS + e] = S[h][k];
                state[(4 + MH + h) * MAP_OPT_MAX_VARS + e] = Y[h][k];
            }
            if (e < n) out[MAP_OPT_OUT_X + e] = x[k];
        }
        if (lane == 0) {
            double* sc = state + (4 + 2 * MH) * MAP_OPT_MAX_VARS;
            sc[0] = fx; sc[1] = t; sc[2] = sy_last; sc[3] = yy_last;
#pragma unroll
            for (int h = 0; h < MH; ++h) sc[4 + h] = rho[h];
            sc[12] = cnt; sc[13] = bt; sc[14] = evals; sc[15] = phase; sc[16] = done;
            out[0] = args.eval_only ? f_last : -fx;
            out[1] = evals;
            out[2] = done;
            out[3] = bad ? 1.0 : 0.0;
        }
    }
}

template <bool MATERN, int KV>
static void launch_map_opt_as(hipStream_t s, const MapOptArgs& args) {
    ensure_dyn_lds((const void*)map_opt_kernel<MATERN, KV>, DIAG_LDS_BYTES);
    hipLaunchKernelGGL((map_opt_kernel<MATERN, KV>), dim3(1), dim3(256), DIAG_LDS_BYTES, s, args);
}
void launch_map_opt(hipStream_t s, int kernel, const MapOptArgs& args) {
    const bool matern = kernel == SLS_KERNEL_ARD_MATERN52;
    // variables per lane: the smallest instantiation that holds n (padding lanes carry zeros: the sums have the same bits in every
    // instantiation; the optimiser's vector work, its history registers and the publishing step scale with KV)
    static_assert(MAP_OPT_MAX_VARS == 64 * 5, "map_opt_kernel is instantiated for 1, 2, 3 and 5 variables per lane");
    const int n = args.ny + args.nh;
    if (n <= 64) matern ? launch_map_opt_as<true, 1>(s, args) : launch_map_opt_as<false, 1>(s, args);
    else if (n <= 64 * 2) matern ? launch_map_opt_as<true, 2>(s, args) : launch_map_opt_as<false, 2>(s, args);
    else if (n <= 64 * 3) matern ? launch_map_opt_as<true, 3>(s, args) : launch_map_opt_as<false, 3>(s, args);
    else matern ? launch_map_opt_as<true, 5>(s, args) : launch_map_opt_as<false, 5>(s, args);
}

// ---------------------------------------------------------------------------------------------------------
// gp_fit_small_kernel: the whole fit of a GP handle for N <= 128 in ONE single-workgroup launch (GaussianProcessRegressor /
// PreferenceRegressor constructors, src/gaussian-process-regressor.cpp:198-232, src/preference-regressor.cpp:289-290, with the
// hoisted quantities of DESIGN.md 2): scaled design matrix + norms, K_y, L, L^-1, (L^-1)^T, K_y^-1, alpha, alpha o X~, the
// posterior mean at the data points with its first maximum, log|K_y|.  The tiled pipeline needs ~15 launches for the same at these
// sizes (175 us of device time + their launch overheads per fit; round 4, C3: one fit per SubmitFeedbackData).
// ---------------------------------------------------------------------------------------------------------
template <bool MATERN>
__global__ __launch_bounds__(256) void gp_fit_small_kernel(const GpFitSmallArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* As = reinterpret_cast<double*>(smem);
    double* Ts = As + 128 * DL;
    const int tid = threadIdx.x;
    const int D = p.D, N = p.N, Dcols = p.Dcols;
    constexpr int Np = 128;
    const int nb16 = (N + 15) >> 4, Nb = 16 * nb16;
    // ---- scaled, centred design matrix and its squared norms (prep_kernel's arithmetic) ----
    if (tid < Np) {
        double sq = 0.0;
        if (tid < N) {
            for (int d = 0; d < D; ++d) {
                const double xr = p.X[d + (long)tid * D];
                const double v = (xr - 0.5) * p.inv_ell[d];
                p.XT[tid + (long)d * Np] = v;
                p.XaT[tid + (long)d * Np] = xr;   // the raw transpose for the Gram pass (Np = 128); alpha o X~ replaces it at the end
                sq += v * v;
            }
            for (int d = D; d < Dcols; ++d) p.XT[tid + (long)d * Np] = 0.0;
        } else {
            for (int d = 0; d < Dcols; ++d) p.XT[tid + (long)d * Np] = 0.0;
        }
        p.nx[tid] = sq;
    }
    for (int d = tid; d < D; d += 256) small_scratch(As, SC_INVL + d) = p.inv_ell[d];
    for (int i = tid; i < 128; i += 256) small_scratch(As, SC_Y + i) = i < N ? p.y[i] : 0.0;
    if (tid == 0) *p.info = 0;
    const SmallPts pts = small_pts_stage(As, p.X, p.XaT, D, N, p.x_lds != 0);
    __syncthreads();

    SmallTrace st;
    const double lg = small_build_factor<MATERN>(As, Ts, pts, Fold(N), D, N, p.a, p.b, p.info, nullptr, nullptr, st);
    // ---- L, L^-1 and (L^-1)^T to global memory (identity padding outside the leading Nb x Nb block, zeros above / below) ----
    for (int idx = tid; idx < Np * Np; idx += 256) {
        const int i = idx & (Np - 1), j = idx >> 7;
        double l = 0.0, t = 0.0;
        if (i >= Nb || j >= Nb) {
            l = t = (i == j) ? 1.0 : 0.0;
        } else if (i >= j) {
            l = As[i + j * DL];
            // L^-1: same 16 x 16 tile -> the tile's inverse in Ts; other tiles -> the transposed slot in the upper triangle of As
            t = (i >> 4) == (j >> 4) ? Ts[256 * (i >> 4) + (i & 15) + 16 * (j & 15)] : As[j + i * DL];
        }
        p.L[idx] = l;
        p.Linv[idx] = t;
        p.U[j + (long)i * Np] = t;
    }
    __syncthreads();
    small_inverse_in_place(As, Ts, N);
    for (int idx = tid; idx < Np * Np; idx += 256) {
        const int i = idx & (Np - 1), j = idx >> 7;
        p.Kinv[idx] = (i < Nb && j < Nb) ? As[i + j * DL] : (i == j ? 1.0 : 0.0);
    }
    double ld, gb, quad;
    small_alpha(As, N, lg, ld, gb, quad);
    // ---- alpha, alpha o X~, the posterior mean at the data points (mu(x_i) = y_i - b alpha_i) and its first maximum ----
    double mv = -INFINITY;
    int mi = 0x7fffffff;
    if (tid < Np) {
        const double al = tid < N ? small_scratch(As, SC_ALPHA + tid) : 0.0;
        p.alpha[tid] = al;
        for (int d = 0; d < Dcols; ++d) p.XaT[tid + (long)d * Np] = al * p.XT[tid + (long)d * Np];
        if (tid < N) {
            const double m = small_scratch(As, SC_Y + tid) - p.b * al;
            p.mu_data[tid] = m;
            mv = m;
            mi = tid;
        }
    }
    // first maximum (Eigen maxCoeff): highest value, ties -> lowest index; all -inf / NaN -> index 0 (argmax_kernel's rule)
    __syncthreads();
    double* rv = &small_scratch(As, SC_BTL);
    if (tid < 128) {
        small_scratch(As, SC_BTL + tid) = mv;
        small_scratch(As, SC_BTL + 128 + tid) = (double)mi;
    }
    __syncthreads();
    if (tid == 0) {
        double bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = 0; i < N; ++i) {
            const double v = small_scratch(As, SC_BTL + i);
            if (v > bv) { bv = v; bi = i; }
        }
        p.scal[0] = bi == 0x7fffffff ? small_scratch(As, SC_BTL) : bv;
        p.d_idx[0] = bi == 0x7fffffff ? 0 : bi;
        p.scal[1] = 2.0 * ld;
        if (p.summary) {   // what the host reads after the fit, in one mapped block: no copies back
            p.summary[0] = bi == 0x7fffffff ? small_scratch(As, SC_BTL) : bv;
            p.summary[1] = 2.0 * ld;
            p.summary[2] = bi == 0x7fffffff ? 0.0 : (double)bi;
            p.summary[3] = (double)__hip_atomic_load(p.info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.summary[4] = 0.0;
        }
        (void)rv;
    }
}

void launch_gp_fit_small(hipStream_t s, int kernel, const GpFitSmallArgs& args) {
    ensure_dyn_lds((const void*)gp_fit_small_kernel<false>, DIAG_LDS_BYTES);
    ensure_dyn_lds((const void*)gp_fit_small_kernel<true>, DIAG_LDS_BYTES);
    if (kernel == SLS_KERNEL_ARD_MATERN52)
        hipLaunchKernelGGL(gp_fit_small_kernel<true>, dim3(1), dim3(256), DIAG_LDS_BYTES, s, args);
    else
        hipLaunchKernelGGL(gp_fit_small_kernel<false>, dim3(1), dim3(256), DIAG_LDS_BYTES, s, args);
}

void launch_nll_small(hipStream_t s, int kernel, const NllSmallArgs& args) {
    ensure_dyn_lds((const void*)nll_small_kernel<false>, DIAG_LDS_BYTES);
    ensure_dyn_lds((const void*)nll_small_kernel<true>, DIAG_LDS_BYTES);
    if (kernel == SLS_KERNEL_ARD_MATERN52)
        hipLaunchKernelGGL(nll_small_kernel<true>, dim3(args.batch > 1 ? args.batch : 1), dim3(256), DIAG_LDS_BYTES, s, args);
    else
        hipLaunchKernelGGL(nll_small_kernel<false>, dim3(args.batch > 1 ? args.batch : 1), dim3(256), DIAG_LDS_BYTES, s, args);
}

}  // namespace slsk
