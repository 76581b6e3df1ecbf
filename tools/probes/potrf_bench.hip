// potrf schedule experiments (debug tool, not shipped): one-level vs two-level blocking, look-ahead on a CU-masked side
// stream.  usage: potrf_bench [N ...]   env: none.  Prints ms per factorisation and the max deviation from the one-level L.
#include "../../sequential-line-search_amd/csrc/kernels_chol.hip"
#include "../../sequential-line-search_amd/csrc/kernels_tri.hip"
#include "../../sequential-line-search_amd/csrc/kernels_vec.hip"
#include <cmath>
#include <cstdio>
#include <set>
#include <string>
#include <vector>

__global__ void fill_spd(double* A, int Np) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= (long)Np * Np) return;
    const int i = idx % Np, j = idx / Np;
    const double d = (double)(i - j) / 40.0;
    A[idx] = exp(-0.5 * d * d) * (1.0 + 0.3 * cos(0.01 * (i + j))) * 0.5 + (i == j ? 0.05 : 0.0);
}
__global__ void maxdiff_lower(const double* a, const double* b, int Np, double* out) {
    __shared__ double red[256];
    double m = 0.0;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < (long)Np * Np; idx += gridDim.x * 256L) {
        const int i = idx % Np, j = idx / Np;
        if (i >= j) m = fmax(m, fabs(a[idx] - b[idx]));
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256, 2) void where_kernel(int* out) {
    if (threadIdx.x == 0) {
        unsigned x, h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        out[blockIdx.x] = ((x & 0xf) << 16) | (h & 0xff00);
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 5000) __builtin_amdgcn_s_sleep(32);
}

int main(int argc, char** argv) {
    using namespace slsk;
    setvbuf(stdout, nullptr, _IONBF, 0);
    std::vector<int> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = {2048, 4096, 8192};
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int* info; hipMalloc(&info, 8192);
    const bool quick = getenv("POTRF_BENCH_QUICK") != nullptr;
    for (int Np : sizes) {
        double *A0, *A, *Lref, *Linv, *red;
        const size_t bytes = (size_t)Np * Np * 8;
        hipMalloc(&A0, bytes); hipMalloc(&A, bytes); hipMalloc(&Lref, bytes); hipMalloc(&Linv, bytes); hipMalloc(&red, 1024 * 8);
        hipLaunchKernelGGL(fill_spd, dim3((unsigned)(((long)Np * Np + 255) / 256)), dim3(256), 0, s, A0, Np);
        hipMemsetAsync(Linv, 0, bytes, s);
        auto run = [&](int nbo, const char* label, bool is_ref) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemcpyAsync(A, A0, bytes, hipMemcpyDeviceToDevice, s);
                hipMemsetAsync(info, 0, 64, s);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, s);
                launch_potrf(s, A, Np, Linv, info, nbo, nullptr);
                hipEventRecord(e1, s); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
                hipEventDestroy(e0); hipEventDestroy(e1);
            }
            int inf2[2] = {0, 0}; hipMemcpy(inf2, info, 8, hipMemcpyDeviceToHost);
            const int inf = inf2[0] + 1000000 * inf2[1];
            double md = 0.0;
            if (is_ref) hipMemcpyAsync(Lref, A, bytes, hipMemcpyDeviceToDevice, s);
            else {
                hipLaunchKernelGGL(maxdiff_lower, dim3(1024), dim3(256), 0, s, A, Lref, Np, red);
                std::vector<double> h(1024); hipStreamSynchronize(s); hipMemcpy(h.data(), red, 1024 * 8, hipMemcpyDeviceToHost);
                for (double v : h) md = fmax(md, v);
            }
            hipStreamSynchronize(s);
            const double tf = (double)Np * Np * Np / 3.0 / (best * 1e-3) / 1e12;
            printf("N=%5d %-34s %8.3f ms  %6.2f TFLOP/s  info=%d  max|L - L_ref|=%.2e\n", Np, label, best, tf, inf, md);
        };
        run(1, "one-level (nbo=1)", true);
        {   // dataflow form (SLS_POTRF_DNBO / SLS_POTRF_DPR from the environment)
            int* dsync; hipMalloc(&dsync, potrf_dataflow_sync_ints(Np) * sizeof(int));
            float best = 1e30f; bool ok = true;
            for (int rep = 0; rep < 5 && ok; ++rep) {
                hipMemcpyAsync(A, A0, bytes, hipMemcpyDeviceToDevice, s);
                hipMemsetAsync(info, 0, 64, s);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, s);
                ok = launch_potrf_dataflow(s, A, Np, Linv, info, dsync);
                hipEventRecord(e1, s); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            int inf2[2] = {0, 0}; hipMemcpy(inf2, info, 8, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(maxdiff_lower, dim3(1024), dim3(256), 0, s, A, Lref, Np, red);
            std::vector<double> h(1024); hipStreamSynchronize(s); hipMemcpy(h.data(), red, 1024 * 8, hipMemcpyDeviceToHost);
            double md = 0.0; for (double v : h) md = fmax(md, v);
            printf("N=%5d %-34s %8.3f ms  %6.2f TFLOP/s  info=%d abort=%d applicable=%d  max|L - L_ref|=%.2e\n", Np, "dataflow single launch", best,
                   (double)Np * Np * Np / 3.0 / (best * 1e-3) / 1e12, inf2[0], inf2[1], (int)ok, md);
            if (getenv("POTRF_BENCH_TRACE") && ok) {
                const int nb = Np / 128;
                const size_t trn = (size_t)nb * 16 + 16 * 512;
                long long* tr; hipMalloc(&tr, trn * 8); hipMemset(tr, 0, trn * 8);
                hipMemcpyAsync(A, A0, bytes, hipMemcpyDeviceToDevice, s);
                hipMemsetAsync(info, 0, 64, s);
                launch_potrf_dataflow(s, A, Np, Linv, info, dsync, tr);
                hipStreamSynchronize(s);
                std::vector<long long> ht(trn); hipMemcpy(ht.data(), tr, ht.size() * 8, hipMemcpyDeviceToHost);
                {   // worker statistics
                    double task = 0, idle = 0, life = 0, tmax = 0, lmax = 0; long nu = 0, np_ = 0, rounds = 0; int nw = 0; double gsum = 0, rsum = 0; long lost = 0, cas = 0; double lost_t = 0;
                    for (int w = 0; w < 512; ++w) {
                        const long long* o = ht.data() + 16 * nb + 16 * w;
                        if (o[6] == 0) continue;
                        ++nw; task += o[0] / 100.0; idle += o[1] / 100.0; life += o[2] / 100.0; tmax = fmax(tmax, o[0] / 100.0); lmax = fmax(lmax, o[2] / 100.0);
                        nu += o[3]; np_ += o[4]; rounds += o[5]; gsum += o[7] / 100.0; rsum += o[8] / 100.0;
                        lost += o[11]; lost_t += o[12] / 100.0; cas += o[13];
                    }
                    printf("  workers with tiles: %d; per worker (us): in tasks mean %.0f max %.0f, idle rounds mean %.0f, lifetime mean %.0f max %.0f; tasks: %ld updates %ld panels, %ld scheduling rounds; mean task %.1f us; update tasks: k loop %.1f us, C read-modify-write + drain %.1f us (wave 0, mean)\n",
                           nw, task / nw, tmax, idle / nw, life / nw, lmax, nu, np_, rounds, task / (nu + np_), gsum / nu, rsum / nu);
                    if (cas) printf("  pools: %ld claim attempts for %ld tasks; %ld rounds saw ready items and lost every race (%.0f us per worker in them)\n", cas, nu + np_, lost, lost_t / nw);
                }
                printf("dataflow chain trace N=%d (us from the chain's step start): step | wait-for-tiles gemm1+signal gemm2 diag+signal | step start since step 0\n", Np);
                double waited = 0.0;
                for (int j = 0; j < nb - 1; ++j) {
                    const long long* g = ht.data() + 16 * j; const double t0 = (double)g[0];
                    auto us = [&](long long v) { return v ? ((double)v - t0) / 100.0 : -1.0; };
                    waited += us(g[1]);
                    if (j < 4 || j % 8 == 0 || j > nb - 3)
                        printf("  %3d | %7.1f %6.1f %6.1f %6.1f | %8.1f\n", j, us(g[1]), us(g[2]), us(g[3]), us(g[4]), ((double)g[0] - (double)ht[0]) / 100.0);
                    if (getenv("POTRF_BENCH_FINE") && j >= 1 && j < 4)
                        printf("        fine: solve loop done %.1f  image %.1f  stores issued %.1f  published %.1f | LL^T MFMAs done %.1f  image %.1f  tile subtracted %.1f\n",
                               us(g[5]), us(g[6]), us(g[7]), us(g[2]), us(g[8]), us(g[9]), us(g[3]));
                }
                printf("  chain waited %.1f us in total for its tiles\n", waited);
                if (getenv("POTRF_BENCH_FOLLOWER")) {
                    // streamed form: per step, relative to the END of the chain's diagonal block j (= start of its step j): when the
                    // follower's tiles (j+1, j), (j+1, j+1) had their updates, when its solve was done, when L_{j+1,j} was published
                    printf("  follower (us after diagonal block j finished): step: tiles ready / solve done / published | chain's wait | last update of "
                           "tile (j+2, j+1), first half: begin / end\n");
                    for (int j = 0; j < nb - 1; ++j) {
                        const long long* g = ht.data() + 16 * j; const double t0 = (double)g[0];
                        auto us = [&](long long v) { return v ? ((double)v - t0) / 100.0 : -1.0; };
                        printf("   %2d: %6.1f %6.1f %6.1f | %5.1f | %6.1f %6.1f | panel tile (j+2, j): task start %6.1f solve done %6.1f stores issued %6.1f\n", j, us(g[11]), us(g[12]), us(g[13]), us(g[1]), us(g[14]), us(g[15]), us(g[5]), us(g[6]), us(g[7]));
                    }
                }
                hipFree(tr);
            }
            hipFree(dsync);
        }
        if (getenv("POTRF_BENCH_POTRI")) {
            // factor + inverse: separate launches (potrf dataflow + diag inverses + trtri + lauum) against the fused single launch
            // (launch_potri_dataflow); max deviations of L^-1 (lower), U = (L^-1)^T (blocks on / above the diagonal) and K^-1 (full)
            double *X1, *U1, *K1, *X2, *U2, *K2;
            hipMalloc(&X1, bytes); hipMalloc(&U1, bytes); hipMalloc(&K1, bytes); hipMalloc(&X2, bytes); hipMalloc(&U2, bytes); hipMalloc(&K2, bytes);
            int* dsync; hipMalloc(&dsync, potrf_dataflow_sync_ints(Np) * sizeof(int));
            auto timeit = [&](bool fused, double* X, double* U, double* K) {
                float best = 1e30f; bool ok = true;
                for (int rep = 0; rep < 5 && ok; ++rep) {
                    hipMemcpyAsync(A, A0, bytes, hipMemcpyDeviceToDevice, s);
                    hipMemsetAsync(info, 0, 64, s);
                    hipMemsetAsync(X, 0, bytes, s);
                    hipMemsetAsync(U, fused ? 0xff : 0, bytes, s);      // the fused form must not depend on the contents of U / K
                    hipMemsetAsync(K, fused ? 0xff : 0, bytes, s);
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    hipEventRecord(e0, s);
                    if (fused) ok = launch_potri_dataflow(s, A, Np, X, U, K, info, dsync);
                    else {
                        ok = launch_potrf_dataflow(s, A, Np, X, info, dsync);
                        launch_trtri(s, A, Np, X, K, U);
                        launch_lauum(s, U, Np, K);
                    }
                    hipEventRecord(e1, s); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < best) best = ms;
                }
                int inf2[2] = {0, 0}; hipMemcpy(inf2, info, 8, hipMemcpyDeviceToHost);
                if (!ok) printf("N=%5d %-34s not applicable at this size (the fused form serves N <= 4096)\n", Np, fused ? "potri fused single launch" : "potrf + trtri + lauum");
                else printf("N=%5d %-34s %8.3f ms  info=%d abort=%d applicable=%d\n", Np, fused ? "potri fused single launch" : "potrf + trtri + lauum", best,
                            inf2[0], inf2[1], (int)ok);
                return ok;
            };
            timeit(false, X1, U1, K1);
            if (getenv("POTRF_BENCH_TRACE")) {
                // where the fused launch spends its time: the chain's end, the factorisation's workers, the inverse's team
                const int nb = Np / 128;
                const size_t trn = (size_t)nb * 16 + 16 * 512;
                long long* tr; hipMalloc(&tr, trn * 8); hipMemset(tr, 0, trn * 8);
                hipMemcpyAsync(A, A0, bytes, hipMemcpyDeviceToDevice, s);
                hipMemsetAsync(info, 0, 64, s); hipMemsetAsync(X2, 0, bytes, s);
                launch_potri_dataflow(s, A, Np, X2, U2, K2, info, dsync, tr);
                hipStreamSynchronize(s);
                std::vector<long long> ht(trn); hipMemcpy(ht.data(), tr, trn * 8, hipMemcpyDeviceToHost);
                const double t_start = (double)ht[0], t_chain_end = (double)ht[16 * (nb - 2) + 4];
                int n1 = 0, n2 = 0; double busy1 = 0, busy2 = 0, life2 = 0, end2_max = 0, end2_mean = 0; long tasks2 = 0;
                for (int w = 0; w < 512; ++w) {
                    const long long* o = ht.data() + 16 * nb + 16 * w;
                    if (o[6] > 0) { ++n1; busy1 += o[0] / 100.0; }
                    else if (o[6] < 0) {
                        ++n2; busy2 += o[0] / 100.0; life2 += o[2] / 100.0; tasks2 += o[3];
                        const double e = ((double)o[10] - t_start) / 100.0; end2_max = fmax(end2_max, e); end2_mean += e;
                    }
                }
                printf("  fused trace N=%d: chain ends at %.0f us; factorisation's workers: %d, mean busy %.0f us; inverse team: %d workgroups, %ld tasks, "
                       "mean busy %.0f us (mean task %.1f us), mean life %.0f us, last task ends at mean %.0f / max %.0f us\n",
                       Np, (t_chain_end - t_start) / 100.0, n1, n1 ? busy1 / n1 : 0.0, n2, tasks2, n2 ? busy2 / n2 : 0.0, tasks2 ? busy2 / tasks2 : 0.0,
                       n2 ? life2 / n2 : 0.0, n2 ? end2_mean / n2 : 0.0, end2_max);
                hipFree(tr);
            }
            if (timeit(true, X2, U2, K2)) {
                std::vector<double> hx1((size_t)Np * Np), hx2((size_t)Np * Np);
                auto cmp = [&](const double* d1, const double* d2, int mode, const char* name) {   // mode 0 lower, 1 upper blocks, 2 full
                    hipMemcpy(hx1.data(), d1, bytes, hipMemcpyDeviceToHost); hipMemcpy(hx2.data(), d2, bytes, hipMemcpyDeviceToHost);
                    double md = 0.0, mx = 0.0; long bad = 0;
                    for (long j = 0; j < Np; ++j)
                        for (long i = 0; i < Np; ++i) {
                            if (mode == 0 && i < j) continue;
                            if (mode == 1 && (i / 128) > (j / 128)) continue;
                            const double a = hx1[i + j * Np], b = hx2[i + j * Np];
                            if (!(fabs(a - b) <= 1e300)) { ++bad; continue; }
                            md = fmax(md, fabs(a - b)); mx = fmax(mx, fabs(a));
                        }
                    printf("    %-6s max|fused - separate| = %.3e (max |value| %.3e, non-finite %ld)\n", name, md, mx, bad);
                };
                cmp(X1, X2, 0, "L^-1"); cmp(U1, U2, 1, "U"); cmp(K1, K2, 2, "K^-1");
                // the fused form once more with another split of the chip between the two teams: must give the same bits
                const char* w1 = getenv("SLS_POTRI_W1");
                const std::string keep = w1 ? w1 : "";
                setenv("SLS_POTRI_W1", Np <= 1024 ? "5" : "71", 1);
                if (timeit(true, X1, U1, K1)) {
                    printf("  same bits with another team split?\n");
                    cmp(X1, X2, 0, "L^-1"); cmp(U1, U2, 1, "U"); cmp(K1, K2, 2, "K^-1");
                }
                if (w1) setenv("SLS_POTRI_W1", keep.c_str(), 1); else unsetenv("SLS_POTRI_W1");
            }
            if (const char* st = getenv("POTRF_BENCH_STRESS")) {
                // the fused launch over and over (with a changing split of the chip): every repetition must reproduce the first one's
                // factor, inverse factor and inverse bit for bit -- a tile read before it was complete shows up here
                const int reps = atoi(st);
                double* red2; hipMalloc(&red2, 1024 * 8);
                auto maxdiff = [&](const double* p1, const double* p2) {
                    hipLaunchKernelGGL(maxdiff_lower, dim3(1024), dim3(256), 0, s, p1, p2, Np, red2);
                    std::vector<double> h(1024); hipStreamSynchronize(s); hipMemcpy(h.data(), red2, 1024 * 8, hipMemcpyDeviceToHost);
                    double md = 0.0; for (double v : h) md = fmax(md, v != v ? 1e300 : v);
                    return md;
                };
                double *L0; hipMalloc(&L0, bytes);
                int bad = 0, aborted = 0;
                for (int rep = 0; rep <= reps; ++rep) {
                    char w1[16]; snprintf(w1, sizeof w1, "%d", 40 + 13 * (rep % 9));
                    if (rep % 3 == 1) setenv("SLS_POTRI_W1", w1, 1); else unsetenv("SLS_POTRI_W1");
                    hipMemcpyAsync(A, A0, bytes, hipMemcpyDeviceToDevice, s);
                    hipMemsetAsync(info, 0, 64, s);
                    hipMemsetAsync(X1, 0, bytes, s); hipMemsetAsync(U1, 0x7f, bytes, s); hipMemsetAsync(K1, 0x7f, bytes, s);
                    if (!launch_potri_dataflow(s, A, Np, X1, U1, K1, info, dsync)) { printf("  stress: not applicable\n"); break; }
                    hipStreamSynchronize(s);
                    int inf2[2]; hipMemcpy(inf2, info, 8, hipMemcpyDeviceToHost);
                    if (inf2[0] || inf2[1]) { ++aborted; continue; }
                    if (rep == 0) {   // on the SAME stream: a device-to-device hipMemcpy may return before it is done, and the next repetition's memsets run on s
                        hipMemcpyAsync(L0, A, bytes, hipMemcpyDeviceToDevice, s); hipMemcpyAsync(X2, X1, bytes, hipMemcpyDeviceToDevice, s);
                        hipMemcpyAsync(K2, K1, bytes, hipMemcpyDeviceToDevice, s); hipStreamSynchronize(s);
                        continue;
                    }
                    // K^-1 is full and symmetric: its lower triangle is the whole information
                    const double dl = maxdiff(A, L0), dx = maxdiff(X1, X2), dk = maxdiff(K1, K2);
                    if (dl != 0.0 || dx != 0.0 || dk != 0.0) {
                        if (++bad <= 5) printf("  stress rep %d (W1 %s): max|dL| %.3e  max|dL^-1| %.3e  max|dK^-1| %.3e\n", rep, rep % 3 == 1 ? w1 : "default", dl, dx, dk);
                    }
                }
                unsetenv("SLS_POTRI_W1");
                printf("N=%5d stress: %d fused launches, %d differ from the first one, %d gave up\n", Np, reps, bad, aborted);
                hipFree(L0); hipFree(red2);
            }
            hipFree(X1); hipFree(U1); hipFree(K1); hipFree(X2); hipFree(U2); hipFree(K2); hipFree(dsync);
        }
        if (getenv("POTRF_BENCH_BATCH")) {
            // P independent factorisations of the same matrix in ONE launch, each on 1/P of the chip's workgroups, against P launches
            // in sequence: time per problem, and every problem's factor against the single-problem factor (must be identical bits)
            for (int P : {2, 3, 4, 6, 8}) {
                if ((size_t)P * bytes * 2 > (size_t)160 << 30) break;
                double *Ab, *Lb; int* sb;
                const size_t sync_ints = (potrf_dataflow_sync_ints(Np) + 63) / 64 * 64;
                hipMalloc(&Ab, bytes * P); hipMalloc(&Lb, bytes * P); hipMalloc(&sb, sync_ints * sizeof(int) * P);
                float best = 1e30f; bool ok = true;
                for (int rep = 0; rep < 4 && ok; ++rep) {
                    for (int q = 0; q < P; ++q) hipMemcpyAsync(Ab + (size_t)q * Np * Np, A0, bytes, hipMemcpyDeviceToDevice, s);
                    hipMemsetAsync(info, 0, 64, s);
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    hipEventRecord(e0, s);
                    ok = launch_potrf_dataflow_batch(s, Ab, Np, Lb, info, sb, P, (long)Np * Np, (long)sync_ints, false);
                    hipEventRecord(e1, s); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < best) best = ms;
                }
                int inf[16] = {0}; hipMemcpy(inf, info, 64, hipMemcpyDeviceToHost);
                int bad = 0; for (int q = 0; q < 2 * P; ++q) bad |= inf[q];
                double md = 0.0;
                for (int q = 0; q < P && ok; ++q) {
                    hipLaunchKernelGGL(maxdiff_lower, dim3(1024), dim3(256), 0, s, Ab + (size_t)q * Np * Np, A, Np, red);
                    std::vector<double> h(1024); hipStreamSynchronize(s); hipMemcpy(h.data(), red, 1024 * 8, hipMemcpyDeviceToHost);
                    for (double v : h) md = fmax(md, v);
                }
                printf("N=%5d batch of %d in one launch: %8.3f ms = %7.3f ms per problem (applicable=%d, info|abort=%d, max|L_q - L_single|=%.2e; "
                       "suggested P = %d)\n", Np, P, best, best / P, (int)ok, bad, md, potrf_dataflow_max_problems(Np));
                hipFree(Ab); hipFree(Lb); hipFree(sb);
            }
        }
        if (quick) { hipFree(A0); hipFree(A); hipFree(Lref); hipFree(Linv); hipFree(red); continue; }
        for (int nbo : {2, 4, 8}) {
            char lab[96];
            snprintf(lab, sizeof lab, "two-level nbo=%d", nbo);
            run(nbo, lab, false);
        }
        hipFree(A0); hipFree(A); hipFree(Lref); hipFree(Linv); hipFree(red);
    }
    return 0;
}
