// C-ABI of libsls_hip (include/sls_hip.h): host orchestration of the gfx950 kernels.  No CPU fallback.
#include <cstddef>
#include <cstring>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>

#include "common.hpp"
#include "kernels.hpp"

using namespace slsk;


namespace slsk {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace slsk

// ---- device block cache (common.hpp) -------------------------------------------------------------------------
namespace slsk {
namespace {
struct Pool {
    std::mutex mtx;
    std::multimap<size_t, void*> free_blocks;
    size_t cached = 0;
    long synced_epoch = -1;
};
// bumped by every C-ABI entry point: "device work may have been queued since the last device-wide synchronisation"
std::atomic<long> g_work_epoch{0};
Pool& pool_of(int dev) {
    static std::mutex m;
    static std::map<int, std::unique_ptr<Pool>> pools;
    std::lock_guard<std::mutex> lock(m);
    auto& p = pools[dev];
    if (!p) p.reset(new Pool());
    return *p;
}
size_t pool_limit() {
    return (size_t)tune(TUNE_POOL_MB, 16384) << 20;
}
}  // namespace

void note_entry() { g_work_epoch.fetch_add(1); }


void* pool_alloc(size_t bytes) {
    if (bytes == 0) bytes = 8;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Pool& P = pool_of(dev);
    {
        std::lock_guard<std::mutex> lock(P.mtx);
        auto it = P.free_blocks.find(bytes);
        if (it != P.free_blocks.end()) {
            void* p = it->second;
            P.free_blocks.erase(it);
            P.cached -= bytes;
            return p;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {          // out of memory: give the cache back and try once more
        pool_trim(dev);
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        throw HipFail{SLS_ERR_HIP};
    }
    return p;
}

void pool_free(void* p, size_t bytes, bool in_flight) {
    if (!p) return;
    if (bytes == 0) bytes = 8;
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess) dev = attr.device;   // freed from another thread / current device
    Pool& P = pool_of(dev);
    {
        std::lock_guard<std::mutex> lock(P.mtx);
        if (P.cached + bytes <= pool_limit()) {
            // A cached block may be handed out again at once (to any stream): everything queued on it must have finished.
            // hipFree would have synchronised implicitly; here ONE device-wide synchronisation covers all the blocks
            // released after the same entry point (a handle's ~30 buffers cost one ~5 us call, normally on an idle device
            // because handles and temporaries are released after their stream has been synchronised).
            const long epoch = g_work_epoch.load();
            // (in_flight: the block is released by the entry point that queued work on it -- a buffer regrown between two
            // launches -- so the epoch's earlier synchronisation does not cover it: synchronise again before it can be reused
            // by another context's stream)
            if (P.synced_epoch != epoch || in_flight) {
                int cur = 0;
                (void)hipGetDevice(&cur);
                if (cur != dev) (void)hipSetDevice(dev);
                (void)hipDeviceSynchronize();
                if (cur != dev) (void)hipSetDevice(cur);
                P.synced_epoch = epoch;
            }
            P.free_blocks.emplace(bytes, p);
            P.cached += bytes;
            return;
        }
    }
    (void)hipFree(p);
}

void pool_trim(int device) {
    Pool& P = pool_of(device);
    std::vector<void*> blocks;
    {
        std::lock_guard<std::mutex> lock(P.mtx);
        for (auto& kv : P.free_blocks) blocks.push_back(kv.second);
        P.free_blocks.clear();
        P.cached = 0;
    }
    for (void* b : blocks) (void)hipFree(b);
}
}  // namespace slsk

// ---- context --------------------------------------------------------------------------------------------
hipEvent_t sls_ctx::get_event() {
    if (!event_pool.empty()) {
        hipEvent_t e = event_pool.back();
        event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    SLS_HIP(hipEventCreate(&e));
    return e;
}
void sls_ctx::prof_begin(const char*, hipEvent_t& e0) {
    e0 = get_event();
    SLS_HIP(hipEventRecord(e0, stream));
}
void sls_ctx::prof_end(const char* name, hipEvent_t e0) {
    hipEvent_t e1 = get_event();
    (void)hipEventRecord(e1, stream);
    ProfEntry& pe = prof[name];
    pe.pending.emplace_back(e0, e1);
    pe.launches += 1;
}
void sls_ctx::prof_collect() {
    for (auto& kv : prof) {
        for (auto& pr : kv.second.pending) {
            (void)hipEventSynchronize(pr.second);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, pr.first, pr.second);
            kv.second.ms += ms;
            event_pool.push_back(pr.first);
            event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

void sls_ctx::potrf_tick_rearm() {
    if (!potrf_persistent_ok && potrf_rearm > 0 && --potrf_rearm == 0) potrf_persistent_ok = true;
}

bool sls_ctx::potrf_df_available(int Np) const { return potrf_persistent_ok && slsk::potrf_default_mode(Np) == 3; }
int* sls_ctx::potrf_df_sync(int Np) {
    if (!potrf_df_available(Np)) return nullptr;
    potrf_df.ensure((slsk::potrf_dataflow_sync_ints(Np) + 1) / 2);
    potrf_single_pending = Np >= 384;   // smaller matrices never take the single-launch form (launch_potrf)
    return reinterpret_cast<int*>(potrf_df.p);
}

namespace slsk {
bool potrf_gave_up(sls_ctx* c, int abort_flag, int attempt) {
    if (abort_flag == 0) {
        // a single-launch factorisation ran to the end: the back-off of earlier give-ups starts over (a context that saw a few
        // transient residency failures would otherwise sit on a 4096-fit back-off for the rest of its life)
        if (c->potrf_single_pending && c->potrf_persistent_ok) c->potrf_rearm_next = 16;
        c->potrf_single_pending = false;
        return false;
    }
    c->potrf_single_pending = false;
    if (attempt > 0 || !c->potrf_persistent_ok) {
        set_error("Cholesky factorisation aborted: a device-side wait expired");
        throw HipFail{SLS_ERR_HIP};
    }
    // The single-launch form needs every workgroup resident at once; another kernel on the device (another process, a
    // profiler, RCCL) can prevent that.  This context uses the multi-launch schedule for the next fits and then tries again,
    // backing off (16, 64, 256, ... 4096 fits) while the device stays shared; the count is visible through
    // sls_prof_get("potrf_fallbacks") and in bench.py's line.  The two schedules agree to rounding, not bit for bit.
    c->potrf_persistent_ok = false;
    c->potrf_rearm = c->potrf_rearm_next;
    c->potrf_rearm_next = std::min(4096, c->potrf_rearm_next * 4);
    c->potrf_fallbacks += 1;
    return true;
}
}  // namespace slsk

#define SLS_TRY slsk::note_entry(); try {
#define SLS_CATCH                                   \
    }                                               \
    catch (const slsk::HipFail& f) { return f.code; } \
    catch (const std::exception& e) {               \
        slsk::set_error("exception: %s", e.what()); \
        return SLS_ERR_INVALID;                     \
    }                                               \
    return SLS_OK;

extern "C" const char* sls_last_error(void) { return slsk::g_err; }
extern "C" int sls_version(void) { return 100; }

extern "C" int sls_ctx_create(int device, sls_ctx** out) {
    SLS_TRY
    SLS_REQUIRE(out != nullptr, "sls_ctx_create: out is NULL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        set_error("sls_ctx_create: no HIP device available (%s); this library has no CPU fallback",
                  e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return SLS_ERR_NO_DEVICE;
    }
    SLS_REQUIRE(device >= 0 && device < ndev, "sls_ctx_create: device %d out of range (%d devices)", device, ndev);
    SLS_HIP(hipSetDevice(device));
    std::unique_ptr<sls_ctx> c(new sls_ctx());
    c->device = device;
    SLS_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    SLS_HIP(hipMalloc((void**)&c->d_info, 4096));   // ints 0..15: potrf info ([0] pivot, [1] abort); 32..47: acq_gemm generation gates; 64..1023: sync words of the persistent potrf
    *out = c.release();
    SLS_CATCH
}

void* sls_ctx::host_take(size_t bytes, bool mapped, size_t* got) {
    int best = -1;
    for (int i = 0; i < (int)host_free.size(); ++i)
        if (host_free[i].mapped == mapped && host_free[i].bytes >= bytes && (best < 0 || host_free[i].bytes < host_free[best].bytes)) best = i;
    if (best >= 0) {
        HostBlock b = host_free[best];
        host_free.erase(host_free.begin() + best);
        *got = b.bytes;
        return b.p;
    }
    void* p = nullptr;
    const size_t sz = std::max<size_t>(bytes, 4096);
    if (hipHostMalloc(&p, sz, mapped ? hipHostMallocMapped : hipHostMallocDefault) != hipSuccess) {
        slsk::set_error("hipHostMalloc(%zu) failed", sz);
        throw slsk::HipFail{SLS_ERR_HIP};
    }
    *got = sz;
    return p;
}
void sls_ctx::host_give(void* p, size_t bytes, bool mapped) {
    if (!p) return;
    if (host_free.size() >= 16) {
        (void)hipHostFree(p);
        return;
    }
    host_free.push_back(HostBlock{p, bytes, mapped});
}

static void ctx_destroy_now(sls_ctx* ctx);
namespace slsk {
void ctx_retain(sls_ctx* c) { ++c->live_handles; }
void ctx_release(sls_ctx* c) {
    bool last;
    {
        std::unique_lock<std::recursive_mutex> lock(c->mtx);
        last = --c->live_handles == 0 && c->destroy_requested;
    }
    if (last) ctx_destroy_now(c);
}
}  // namespace slsk

extern "C" int sls_ctx_destroy(sls_ctx* ctx) {
    if (!ctx) return SLS_OK;
    {
        std::unique_lock<std::recursive_mutex> lock(ctx->mtx);
        if (ctx->live_handles > 0) {       // handles outlive the context: the last one frees it (slsk::ctx_release)
            ctx->destroy_requested = true;
            return SLS_OK;
        }
    }
    ctx_destroy_now(ctx);
    return SLS_OK;
}
static void ctx_destroy_now(sls_ctx* ctx) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& s : ctx->slots) {   // the evaluation slots: their streams, their mapped blocks back to the pool that is freed next
        if (s->stream) {
            (void)hipStreamSynchronize(s->stream);
            (void)hipStreamDestroy(s->stream);
        }
        if (s->host) ctx->host_give(s->host, s->bytes, true);
    }
    ctx->slots.clear();
    for (auto& b : ctx->host_free) (void)hipHostFree(b.p);
    ctx->host_free.clear();
    ctx->prof_collect();
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->d_info) (void)hipFree(ctx->d_info);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int sls_tuning_reload(void) {
    slsk::tuning_reload();
    return SLS_OK;
}

extern "C" int sls_device_trim_cache(int device) {
    slsk::note_entry();
    (void)hipDeviceSynchronize();
    slsk::pool_trim(device);
    return SLS_OK;
}

extern "C" int sls_ctx_set_stream(sls_ctx* ctx, void* hip_stream) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (ctx) lock_ = std::unique_lock<std::recursive_mutex>(ctx->mtx);
    SLS_REQUIRE(ctx, "ctx is NULL");
    SLS_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    SLS_CATCH
}
extern "C" int sls_ctx_synchronize(sls_ctx* ctx) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (ctx) lock_ = std::unique_lock<std::recursive_mutex>(ctx->mtx);
    SLS_REQUIRE(ctx, "ctx is NULL");
    SLS_HIP(hipStreamSynchronize(ctx->stream));
    SLS_CATCH
}
extern "C" int sls_ctx_set_candidate_chunk(sls_ctx* ctx, int chunk) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (ctx) lock_ = std::unique_lock<std::recursive_mutex>(ctx->mtx);
    SLS_REQUIRE(ctx && chunk >= 128, "candidate chunk must be >= 128");
    ctx->cand_chunk = round_up(chunk, 128);
    SLS_CATCH
}
extern "C" int sls_prof_enable(sls_ctx* ctx, int on) {
    if (!ctx) return SLS_ERR_INVALID;
    std::unique_lock<std::recursive_mutex> lock_(ctx->mtx);
    ctx->prof_on = on != 0;
    return SLS_OK;
}
extern "C" int sls_prof_reset(sls_ctx* ctx) {
    if (!ctx) return SLS_ERR_INVALID;
    std::unique_lock<std::recursive_mutex> lock_(ctx->mtx);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->prof_collect();
    ctx->prof.clear();
    return SLS_OK;
}
extern "C" int sls_prof_get(sls_ctx* ctx, const char* name, double* total_ms, long* launches) {
    if (!ctx || !name) return SLS_ERR_INVALID;
    std::unique_lock<std::recursive_mutex> lock_(ctx->mtx);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->prof_collect();
    if (std::string(name) == "potrf_fallbacks") {          // not a timing: how often a single-launch Cholesky gave up
        if (total_ms) *total_ms = 0.0;
        if (launches) *launches = ctx->potrf_fallbacks;
        return SLS_OK;
    }
    auto it = ctx->prof.find(name);
    if (total_ms) *total_ms = it == ctx->prof.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == ctx->prof.end() ? 0 : it->second.launches;
    return SLS_OK;
}

// ---- helpers --------------------------------------------------------------------------------------------
static void h2d(sls_ctx* c, double* dst, const double* src, size_t n) {
    SLS_HIP(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
}
static void d2h(sls_ctx* c, double* dst, const double* src, size_t n) {
    SLS_HIP(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
}
static void sync(sls_ctx* c) { SLS_HIP(hipStreamSynchronize(c->stream)); }
// Up to this many doubles of query points / results / matrices go through a handle's page-locked staging block (32 MB); larger
// transfers take the plain (pageable, blocking) copies.
static const size_t IO_STAGE_MAX = (size_t)4 << 20;

static void check_theta(const double* theta, int D) {
    SLS_REQUIRE(theta != nullptr, "theta is NULL");
    SLS_REQUIRE(theta[0] > 0.0, "theta[0] (signal variance) must be positive");
    for (int d = 0; d < D; ++d) SLS_REQUIRE(theta[1 + d] > 0.0, "length scale %d must be positive", d);
}

// copies a padded (ld = Np) device matrix region [N x N] into a dense host N x N
static void d2h_matrix(sls_ctx* c, double* dst, const double* src, int N, int Np) {
    SLS_HIP(hipMemcpy2DAsync(dst, (size_t)N * 8, src, (size_t)Np * 8, (size_t)N * 8, N, hipMemcpyDeviceToHost, c->stream));
}
static void h2d_matrix(sls_ctx* c, double* dst, const double* src, int N, int Np) {
    SLS_HIP(hipMemcpy2DAsync(dst, (size_t)Np * 8, src, (size_t)N * 8, (size_t)N * 8, N, hipMemcpyHostToDevice, c->stream));
}

// ---- GP handle --------------------------------------------------------------------------------------------
struct sls_gp {
    sls_ctx* ctx = nullptr;
    int D = 0, N = 0, Np = 0, Dp = 0, Dcols = 0, kernel = 0;
    double a = 0, b = 0;
    std::vector<double> theta, Xh, yh;
    std::vector<double> il_h, ypad_h;   // upload staging that lives with the handle: no synchronisation between the uploads and the fit
    bool host_stale = false;   // sls_gp_refit_dev replaced the device X / y: Xh / yh are refreshed before their next use
    DBuf X, y, inv_ell, XT, XaT, nx, L, Linv, Kinv, U, alpha, tvec, mu_data, scal, gemv_part, idx_buf;   // U = (L^-1)^T, needed by the fit only
    const double* linv_clean_p = nullptr;   // the Linv buffer / leading dimension whose blocks above the diagonal are known to be zero
    int linv_clean_np = 0;
    long* d_idx = nullptr;             // idx_buf's block (pooled: a hipMalloc / hipFree pair per handle synchronised the device)
    // what the host reads after a fit -- max mu, log|K_y|, arg max, the factorisation's two info words -- in a page-locked block the
    // last kernel of the fit writes directly (mapped): one synchronisation, no copies back (they were three blocking pageable copies)
    double* sum_host = nullptr;
    double* sum_dev = nullptr;
    size_t sum_bytes = 0;
    int best_index = 0;
    double mu_best = 0, logdet = 0;
    // evaluation workspace (grown on demand)
    DBuf Ks, Cs, P, Vs, parts, Gs, Gm, Gpart, XsT, ns, raw, outv, outg, outm, outs;
    int ws_chunk = 0;
    // L-BFGS state
    DBuf pair_mu, pair_sg, pair_dmu, pair_dsg;
    DBuf lb_x, lb_g, lb_dir, lb_xt, lb_scr, lb_S, lb_Y, lb_rho, lb_f, lb_t, lb_val, lb_grad, lb_xc;
    DBuf lb_int_buf;          // pooled (a hipMalloc / hipFree pair per handle cost ~0.2 ms per submit of the reference's demo and synchronised the device)
    int* lb_int = nullptr;    // lb_int_buf's block: hlen | hpos | nbt | done | live list A | live list B | live count (+ padding) | block counts
    int lb_Sp = 0, lb_m = 0;
    // 0: sigma^2 = a - k^T K^-1 k with the explicit inverse (GaussianProcessRegressor); 1: a - |L^-1 k|^2, the Cholesky solve of
    // PreferenceRegressor (sls_gp_set_sigma_mode)
    int sigma_mode = 0;
    // changes whenever the predictor this handle stands for changes (fit, refit, appended point, sigma mode); process-wide unique, so
    // a new handle at a recycled address never repeats a number (sls_gp_generation: the host layer keys its replicas on it)
    long generation = 0;
    // statistics of the last sls_acq_maximize* call on this handle (sls_acq_last_stats)
    long stat_issued = 0, stat_cap = 0;
    int stat_rounds = 0, stat_live_end = 0;
    // page-locked, device-mapped block for small value-only evaluations (query points in, values out: no copy calls)
    double* zc_host = nullptr;
    double* zc_dev = nullptr;
    size_t zc_bytes = 0;
    // page-locked staging of host-supplied query points and their results (predict / acquisition entry points on the tiled path):
    // uploads and downloads are truly asynchronous, one synchronisation per call (pageable buffers: blocking staged copies, a
    // synchronisation behind each, and a fresh 0.5 MB temporary per transfer)
    // Concurrent const evaluations (src/acquisition-function.cpp:125-144: Predict* of ONE regressor from hardware_concurrency worker
    // threads).  A small evaluation (a few query points on a wave-path handle) does not take the context's lock: it borrows a SLOT of
    // the handle -- a stream of its own and a page-locked, device-mapped block for the query points and the results -- under a
    // SHARED lock on the fitted state; everything that changes the state (refit, appended point, sigma mode) holds it exclusively.
    // The slots belong to the CONTEXT (sls_ctx::slots): a stream costs ~1 ms to create and the facade builds a new handle per submit.
    std::shared_mutex state_mtx;
    double* io_host = nullptr;
    size_t io_bytes = 0;
    double* io_stage(size_t doubles) {
        if (doubles * 8 > io_bytes) {
            if (io_host) {
                (void)hipStreamSynchronize(ctx->stream);   // an upload from the old block may still be in flight (growth is rare)
                ctx->host_give(io_host, io_bytes, false);
            }
            io_host = nullptr;
            io_host = static_cast<double*>(ctx->host_take(doubles * 8, false, &io_bytes));
        }
        return io_host;
    }
    ~sls_gp() {   // sls_gp_destroy holds the context's lock
        if (sum_host) ctx->host_give(sum_host, sum_bytes, true);
        if (zc_host) ctx->host_give(zc_host, zc_bytes, true);
        if (io_host) ctx->host_give(io_host, io_bytes, false);
    }
};

// N <= 128: the whole fit is one single-workgroup launch (kernels_small.hip); SLS_FIT_SMALL=0 forces the tiled pipeline (A/B, tests)
static bool gp_fit_small_ok(const sls_gp* g) {
    if (g->N > NLL_SMALL_MAX_N || g->Np != 128 || g->D > 128) return false;
    return tune_on(TUNE_FIT_SMALL);
}

static std::atomic<long> g_gp_generation{0};
static void gp_fit_device(sls_gp* g) {
    g->generation = ++g_gp_generation;
    sls_ctx* c = g->ctx;
    const int N = g->N, Np = g->Np, D = g->D;
    if (gp_fit_small_ok(g)) {
        GpFitSmallArgs f;
        f.X = g->X.p; f.y = g->y.p; f.inv_ell = g->inv_ell.p;
        f.D = D; f.N = N; f.Dcols = g->Dcols; f.a = g->a; f.b = g->b;
        f.XT = g->XT.p; f.nx = g->nx.p; f.XaT = g->XaT.p; f.L = g->L.p; f.Linv = g->Linv.p; f.U = g->U.p; f.Kinv = g->Kinv.p;
        f.alpha = g->alpha.p; f.mu_data = g->mu_data.p; f.scal = g->scal.p; f.d_idx = g->d_idx; f.info = c->d_info;
        f.summary = g->sum_dev;
        f.x_lds = tune_on(TUNE_SMALL_XLDS) ? 1 : 0;
        SLS_HIP(hipMemsetAsync(c->d_info, 0, 64, c->stream));
        ProfScope ps(c, "fit_small");
        launch_gp_fit_small(c->stream, g->kernel, f);
        return;
    }
    KernelSpec ks{g->kernel, g->a};
    {
        ProfScope ps(c, "gram");
        launch_prep_points(c->stream, g->X.p, D, N, g->inv_ell.p, g->XT.p, Np, Np, g->Dcols, g->nx.p);
        launch_gram_sym(c->stream, g->XT.p, Np, g->Dp, g->nx.p, Np, N, ks, g->b, g->L.p, true);
    }
    SLS_HIP(hipMemsetAsync(c->d_info, 0, 64, c->stream));
    // L^-1 is read as a full matrix (gemv, the plain-product predict): its blocks above the diagonal must be zero.  Nothing ever
    // writes them -- the factorisation, the inverse and the rank-1 growth touch blocks on and below the diagonal only -- so they are
    // cleared when the buffer is new or its leading dimension changed, not on every refit (a 512 MB sweep, 91 us, at N = 8192).
    if (g->linv_clean_p != g->Linv.p || g->linv_clean_np != Np) {
        launch_fill(c->stream, g->Linv.p, (long)Np * Np, 0.0);
        g->linv_clean_p = g->Linv.p;
        g->linv_clean_np = Np;
    }
    c->potrf_tick_rearm();
    int* df_sync = c->potrf_df_sync(Np);
    if (potri_fused_applies(Np, df_sync != nullptr)) {
        // N <= 4096: factorisation, L^-1, its transpose and K_y^-1 in ONE launch (the inverse is built behind the chain by the CUs the
        // factorisation leaves idle)
        ProfScope ps(c, "potri");
        if (!launch_potri(c->stream, g->L.p, Np, g->Linv.p, g->U.p, g->Kinv.p, c->d_info, df_sync))
            ps.rename("potrf+trtri+lauum");   // the fused launch declined (too few CUs resident): the three separate launches ran
    } else {
        {
            ProfScope ps(c, "potrf");
            launch_potrf(c->stream, g->L.p, Np, g->Linv.p, c->d_info, 0, df_sync);
        }
        {
            ProfScope ps(c, "trtri");
            launch_trtri(c->stream, g->L.p, Np, g->Linv.p, g->Kinv.p, g->U.p);
        }
        {
            ProfScope ps(c, "lauum");
            launch_lauum(c->stream, g->U.p, Np, g->Kinv.p);
        }
    }
    // alpha = Linv^T (Linv y);  mu at the data points = y - b alpha;  x_best = first argmax  (regressor.cpp:29-43 hoisted)
    launch_gemv_n(c->stream, g->Linv.p, Np, g->y.p, g->tvec.p, g->gemv_part.p, true);
    launch_gemv_t(c->stream, g->Linv.p, Np, g->tvec.p, g->alpha.p, true);
    launch_scale_rows(c->stream, g->XT.p, g->alpha.p, g->XaT.p, Np, Np, g->Dcols);
    if (g->sigma_mode == 1) launch_transpose_full(c->stream, g->Linv.p, g->U.p, Np);   // every block of U = (L^-1)^T
    // mu at the data points, its first maximum, log|K_y| and the info words: one launch, results straight into the mapped block
    launch_fit_summary(c->stream, g->y.p, g->alpha.p, g->b, N, g->mu_data.p, g->L.p, Np, c->d_info, g->scal.p, g->d_idx, g->sum_dev);
}

static void gp_fetch_summary(sls_gp* g, int attempt = 0) {
    sls_ctx* c = g->ctx;
    sync(c);                         // the fit's last kernel has written the mapped block
    const double* sm = g->sum_host;
    const int info[2] = {(int)sm[3], (int)sm[4]};
    if (potrf_gave_up(c, info[1], attempt)) {
        gp_fit_device(g);            // once more, on the multi-launch schedule (the fit rebuilds K_y from X)
        gp_fetch_summary(g, 1);
        return;
    }
    if (info[0] != 0) {
        set_error("Cholesky failed: K_y is not positive definite (pivot %d)", info[0] - 1);
        throw HipFail{SLS_ERR_NOT_SPD};
    }
    g->best_index = (int)sm[2];
    g->mu_best = sm[0];
    g->logdet = sm[1];
}
// (re)size every per-handle buffer for the host copies g->Xh / g->yh, upload and fit
static void gp_setup(sls_gp* g) {
    sls_ctx* ctx = g->ctx;
    const int D = g->D, N = (int)g->yh.size();
    g->N = N;
    g->Np = round_up(N, 128); g->Dp = round_up(D, 16); g->Dcols = round_up(D, 128);
    const size_t Np = g->Np;
    g->X.ensure((size_t)D * N); g->y.ensure(Np); g->inv_ell.ensure(g->Dcols);
    g->XT.ensure(Np * g->Dcols); g->XaT.ensure(Np * g->Dcols); g->nx.ensure(Np);
    g->L.ensure(Np * Np); g->Linv.ensure(Np * Np); g->Kinv.ensure(Np * Np); g->U.ensure(Np * Np);
    g->alpha.ensure(Np); g->tvec.ensure(Np); g->mu_data.ensure(Np); g->scal.ensure(8);
    g->gemv_part.ensure((Np / 128) * Np);
    g->ws_chunk = 0;   // the evaluation workspace depends on Np
    g->idx_buf.ensure(8);
    g->d_idx = reinterpret_cast<long*>(g->idx_buf.p);
    if (!g->sum_host) {
        // [0..5) the fit's summary, [8..16 + D) the maximiser's best start
        g->sum_host = static_cast<double*>(ctx->host_take((size_t)(16 + D) * 8, true, &g->sum_bytes));
        SLS_HIP(hipHostGetDevicePointer((void**)&g->sum_dev, g->sum_host, 0));
    }
    g->il_h.assign(g->Dcols, 0.0);
    g->ypad_h.assign(Np, 0.0);
    for (int d = 0; d < D; ++d) g->il_h[d] = 1.0 / g->theta[1 + d];
    std::copy(g->yh.begin(), g->yh.end(), g->ypad_h.begin());
    h2d(ctx, g->X.p, g->Xh.data(), (size_t)D * N);
    h2d(ctx, g->y.p, g->ypad_h.data(), Np);
    h2d(ctx, g->inv_ell.p, g->il_h.data(), g->Dcols);
    gp_fit_device(g);            // enqueued behind the uploads; the staging vectors are members
    gp_fetch_summary(g);         // the one synchronisation of the fit
}

extern "C" int sls_gp_create(sls_ctx* ctx, const double* X, int D, int N, const double* y, const double* theta, double b,
                             int kernel, sls_gp** out) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (ctx) lock_ = std::unique_lock<std::recursive_mutex>(ctx->mtx);
    SLS_REQUIRE(ctx && out, "sls_gp_create: NULL argument");
    SLS_REQUIRE(D >= 1 && N >= 1, "sls_gp_create: need D >= 1 and N >= 1 (got D=%d N=%d)", D, N);
    SLS_REQUIRE(X && y, "sls_gp_create: X / y is NULL");
    SLS_REQUIRE(kernel == SLS_KERNEL_ARD_SQUARED_EXPONENTIAL || kernel == SLS_KERNEL_ARD_MATERN52, "unknown kernel %d", kernel);
    SLS_REQUIRE(b >= 0.0, "noise level must be >= 0");
    check_theta(theta, D);
    SLS_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<sls_gp> g(new sls_gp());
    g->ctx = ctx; g->D = D; g->kernel = kernel; g->a = theta[0]; g->b = b;
    g->theta.assign(theta, theta + D + 1);
    g->Xh.assign(X, X + (size_t)D * N);
    g->yh.assign(y, y + N);
    gp_setup(g.get());
    slsk::ctx_retain(ctx);
    *out = g.release();
    SLS_CATCH
}

extern "C" int sls_gp_refit_dev(sls_gp* g, const double* X_dev, const double* y_dev) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (g) {
        lock_ = std::unique_lock<std::recursive_mutex>(g->ctx->mtx);
        (void)hipSetDevice(g->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    SLS_REQUIRE(g && X_dev && y_dev, "sls_gp_refit_dev: NULL argument");
    std::unique_lock<std::shared_mutex> state_(g->state_mtx);
    sls_ctx* c = g->ctx;
    SLS_HIP(hipMemcpyAsync(g->X.p, X_dev, (size_t)g->D * g->N * 8, hipMemcpyDeviceToDevice, c->stream));
    SLS_HIP(hipMemcpyAsync(g->y.p, y_dev, (size_t)g->N * 8, hipMemcpyDeviceToDevice, c->stream));
    g->host_stale = true;
    gp_fit_device(g);
    gp_fetch_summary(g);
    SLS_CATCH
}

extern "C" int sls_gp_destroy(sls_gp* gp) {
    if (!gp) return SLS_OK;
    slsk::note_entry();
    sls_ctx* c = gp->ctx;
    {
        std::unique_lock<std::recursive_mutex> lock_(c->mtx);
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        // evaluations that run on the context's slots without its lock (eval_in_slot) hold the state lock shared until their slot's
        // stream has drained: taking it exclusively waits for them.  Released before the delete (it is a member).
        { std::unique_lock<std::shared_mutex> drain_(gp->state_mtx); }
        delete gp;
    }
    slsk::ctx_release(c);
    return SLS_OK;
}

extern "C" int sls_gp_generation(sls_gp* g, long* generation) {
    if (!g || !generation) return SLS_ERR_INVALID;
    std::unique_lock<std::recursive_mutex> lock_(g->ctx->mtx);
    *generation = g->generation;
    return SLS_OK;
}

extern "C" int sls_gp_get_summary(sls_gp* g, int* best_index, double* mu_best, double* logdet) {
    if (!g) return SLS_ERR_INVALID;
    std::unique_lock<std::recursive_mutex> lock_(g->ctx->mtx);
    if (best_index) *best_index = g->best_index;
    if (mu_best) *mu_best = g->mu_best;
    if (logdet) *logdet = g->logdet;
    return SLS_OK;
}

namespace slsk {
int gp_export_inputs(sls_gp* g, int* D, int* N, int* kernel, int* sigma_mode, int* device, double* b, std::vector<double>* X,
                     std::vector<double>* y, std::vector<double>* theta) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_(g->ctx->mtx);
    (void)hipSetDevice(g->ctx->device);
    *D = g->D; *N = g->N; *kernel = g->kernel; *sigma_mode = g->sigma_mode; *device = g->ctx->device; *b = g->b;
    *theta = g->theta;
    X->resize((size_t)g->D * g->N);
    y->resize(g->N);
    d2h(g->ctx, X->data(), g->X.p, X->size());
    d2h(g->ctx, y->data(), g->y.p, y->size());
    sync(g->ctx);
    SLS_CATCH
}
}  // namespace slsk

extern "C" int sls_gp_set_sigma_mode(sls_gp* g, int mode) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (g) {
        lock_ = std::unique_lock<std::recursive_mutex>(g->ctx->mtx);
        (void)hipSetDevice(g->ctx->device);
    }
    SLS_REQUIRE(g && (mode == SLS_SIGMA_EXPLICIT_INVERSE || mode == SLS_SIGMA_CHOLESKY_SOLVE), "sls_gp_set_sigma_mode: bad argument");
    std::unique_lock<std::shared_mutex> state_(g->state_mtx);
    if (mode == SLS_SIGMA_CHOLESKY_SOLVE && g->sigma_mode != mode)
        launch_transpose_full(g->ctx->stream, g->Linv.p, g->U.p, g->Np);   // every block of U = (L^-1)^T (trtri writes the upper ones only)
    if (g->sigma_mode != mode) g->generation = ++g_gp_generation;
    g->sigma_mode = mode;
    sync(g->ctx);   // slot streams are not ordered behind the context's stream: the state is complete before the lock goes
    SLS_CATCH
}

extern "C" int sls_gp_get_matrix(sls_gp* g, int what, double* out) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (g) {
        lock_ = std::unique_lock<std::recursive_mutex>(g->ctx->mtx);
        (void)hipSetDevice(g->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    SLS_REQUIRE(g && out, "sls_gp_get_matrix: NULL argument");
    sls_ctx* c = g->ctx;
    const int N = g->N, Np = g->Np;
    // Up to Np = 2048 the padded matrix comes back as ONE contiguous copy into the handle's page-locked staging block and the N x N
    // part is taken out on the host (a strided copy into pageable memory is executed row by row, blocking; the factor's upper
    // triangle is cleared on the way out instead of by a device copy + launch).  PreferenceRegressor fetches K and L after every fit.
    const bool staged = (size_t)Np * Np <= IO_STAGE_MAX;
    auto fetch_staged = [&](const double* dev, bool lower_only) {
        double* st = g->io_stage((size_t)Np * Np);
        d2h(c, st, dev, (size_t)Np * Np);
        sync(c);
        for (int j = 0; j < N; ++j) {
            const double* col = st + (size_t)j * Np;
            double* o = out + (size_t)j * N;
            if (lower_only) {
                std::fill(o, o + std::min(j, N), 0.0);
                std::copy(col + j, col + N, o + j);
            } else {
                std::copy(col, col + N, o);
            }
        }
    };
    switch (what) {
        case SLS_GP_K_Y: {
            // m_K_y is not kept resident (the Cholesky factor overwrites it): rebuild the full symmetric matrix
            DBuf K;
            K.ensure((size_t)Np * Np);
            launch_gram_sym(c->stream, g->XT.p, Np, g->Dp, g->nx.p, Np, N, KernelSpec{g->kernel, g->a}, g->b, K.p, false);
            if (staged) fetch_staged(K.p, false);
            else {
                d2h_matrix(c, out, K.p, N, Np);
                sync(c);
            }
            break;
        }
        case SLS_GP_K_Y_INV:
            if (staged) fetch_staged(g->Kinv.p, false);
            else {
                d2h_matrix(c, out, g->Kinv.p, N, Np);
                sync(c);
            }
            break;
        case SLS_GP_CHOL_L: {
            if (staged) {
                fetch_staged(g->L.p, true);
                break;
            }
            DBuf T;
            T.ensure((size_t)Np * Np);
            SLS_HIP(hipMemcpyAsync(T.p, g->L.p, (size_t)Np * Np * 8, hipMemcpyDeviceToDevice, c->stream));
            launch_zero_upper(c->stream, T.p, Np);
            d2h_matrix(c, out, T.p, N, Np);
            sync(c);
            break;
        }
        case SLS_GP_ALPHA: d2h(c, out, g->alpha.p, N); sync(c); break;
        case SLS_GP_MU_DATA: d2h(c, out, g->mu_data.p, N); sync(c); break;
        default: SLS_REQUIRE(false, "sls_gp_get_matrix: unknown selector %d", what);
    }
    SLS_CATCH
}

// ---- batched evaluation -------------------------------------------------------------------------------------
static void ensure_eval_ws(sls_gp* g, int chunk) {
    if (chunk <= g->ws_chunk) return;
    const size_t Np = g->Np, C = chunk;
    const size_t nbt = Np / 128;
    g->Ks.ensure(C * Np);
    if (g->kernel == SLS_KERNEL_ARD_MATERN52) g->Cs.ensure(C * Np);
    g->P.ensure(C * Np);
    g->parts.ensure(10 * nbt * C);  // mu, ca: one per row tile; kw, cw: one per half row tile; + kw, cw of the solve-based sigma
    g->Gs.ensure(C * g->Dcols);
    g->Gm.ensure(C * g->Dcols);
    g->XsT.ensure(C * g->Dcols);
    g->ns.ensure(C);
    g->ws_chunk = chunk;
}

struct EvalOut {
    // candidate-major device outputs with leading dimension ldo (full candidate count, padded); any may be NULL
    long ldo = 0;
    double *mu = nullptr, *sigma = nullptr, *dmu = nullptr, *dsigma = nullptr, *val = nullptr, *grad = nullptr;
    int acq = SLS_ACQ_EXPECTED_IMPROVEMENT;
    double ucb_h = 1.0;
};

// value-only evaluations take the triangular L^-1 contraction (half the flops); SLS_TRI_PREDICT=0 forces the K^-1 form
static bool tri_predict() {
    return tune_on(TUNE_TRI_PREDICT);
}

// xr: raw candidate coordinates, S candidates: candidate-major xr[n + d*ldr], or (point_major) xr[d + n*D] as the host hands them
// over (the device transposes: prep_kernel<false>, same arithmetic and same bits as the candidate-major form).
static void eval_candidates(sls_gp* g, const double* xr, long ldr, int S, const EvalOut& o, bool point_major = false) {
    sls_ctx* c = g->ctx;
    const int Np = g->Np, N = g->N, D = g->D;
    const bool want_grad = o.dmu || o.dsigma || o.grad;
    const int chunk_max = std::min(round_up(S, 128), c->cand_chunk);
    ensure_eval_ws(g, chunk_max);
    const int nbt = Np / 128;
    KernelSpec ks{g->kernel, g->a};
    for (int s0 = 0; s0 < S; s0 += chunk_max) {
        const int sc = std::min(chunk_max, S - s0);
        const int Sp = round_up(sc, 128);
        const long ldk = Sp;
        double* Cs = (g->kernel == SLS_KERNEL_ARD_MATERN52) ? g->Cs.p : g->Ks.p;
        double* mu_part = g->parts.p;
        double* ca_part = mu_part + (size_t)nbt * ldk;
        double* kw_part = ca_part + (size_t)nbt * ldk;
        double* cw_part = kw_part + (size_t)2 * nbt * ldk;
        {
            ProfScope ps(c, "cross_gram");
            if (point_major) launch_prep_points(c->stream, xr + (size_t)s0 * D, D, sc, g->inv_ell.p, g->XsT.p, ldk, Sp, g->Dcols, g->ns.p);
            else launch_prep_cands(c->stream, xr + s0, ldr, D, sc, g->inv_ell.p, g->XsT.p, ldk, Sp, g->Dcols, g->ns.p);
            launch_cross_gram(c->stream, g->XsT.p, ldk, g->ns.p, Sp, g->XT.p, Np, g->nx.p, Np, N, g->Dp, ks, g->alpha.p, g->Ks.p, Cs,
                              ldk, mu_part, ca_part);
        }
        int split_first = 0x7fffffff;
        const bool solve_grad = g->sigma_mode == 1 && (want_grad || !tri_predict());
        {
            ProfScope ps(c, want_grad ? "acq_gemm" : "var_gemm");
            if (solve_grad) {
                // handles of a PreferenceRegressor: w = K^-1 k as L^-T (L^-1 k) (LLT.solve, src/preference-regressor.cpp:299-330) --
                // V = K* L^-T as a plain product, then the acq_gemm tile kernel with (L^-1)^T in the place of K^-1 and V in the
                // place of K*: P = C* o W and the c.w partial sums come out as usual (its k.w sums are not used: sigma below)
                g->Vs.ensure((size_t)chunk_max * Np);
                launch_gemm_plain(c->stream, g->Ks.p, ldk, false, g->Linv.p, Np, false, g->Vs.p, ldk, Sp / 128, Np / 128, Np, 1.0, 0.0);
                split_first = launch_acq_gemm(c->stream, g->Vs.p, Cs, ldk, Sp, g->U.p, Np, g->P.p, kw_part, cw_part, c->d_info + 32);
            } else if (want_grad || !tri_predict())
                split_first = launch_acq_gemm(c->stream, g->Ks.p, Cs, ldk, Sp, g->Kinv.p, Np, g->P.p, kw_part, cw_part, c->d_info + 32);
            else
                launch_var_gemm(c->stream, g->Ks.p, ldk, Sp, g->Linv.p, Np, kw_part, cw_part);
        }
        // ... and sigma from the triangular form |L^-1 k|^2
        double* kw_solve = nullptr;
        if (solve_grad) {
            kw_solve = cw_part + (size_t)2 * nbt * ldk;
            ProfScope ps(c, "var_gemm");
            launch_var_gemm(c->stream, g->Ks.p, ldk, Sp, g->Linv.p, Np, kw_solve, kw_solve + (size_t)2 * nbt * ldk);
        }
        if (want_grad) {
            ProfScope ps(c, "grad_gemm");
            double* part = nullptr;
            if (D <= 64 && grad_gemm_wants_split(Sp)) {
                g->Gpart.ensure((size_t)8 * Sp * 64);
                part = g->Gpart.p;
            }
            launch_grad_gemm(c->stream, g->P.p, Cs, ldk, Sp, g->XT.p, g->XaT.p, Np, Np, D <= 64 ? -g->Dcols : g->Dcols, g->Gs.p, g->Gm.p,
                             part);
        }
        {
            ProfScope ps(c, "finalize");
            FinalizeArgs f;
            f.S = sc; f.D = D; f.nbt = nbt; f.ldk = ldk; f.ntm = Sp / 128; f.split_first = split_first;
            f.mu_part = mu_part; f.ca_part = ca_part; f.kw_part = kw_part; f.cw_part = cw_part;
            f.kw_solve_part = kw_solve;
            f.Gs = g->Gs.p; f.Gm = g->Gm.p; f.XsT = g->XsT.p; f.inv_ell = g->inv_ell.p;
            f.a = g->a; f.mu_best = g->mu_best; f.ucb_h = o.ucb_h; f.acq = o.acq;
            f.ldo = o.ldo;
            f.mu = o.mu ? o.mu + s0 : nullptr;
            f.sigma = o.sigma ? o.sigma + s0 : nullptr;
            f.dmu = o.dmu ? o.dmu + s0 : nullptr;
            f.dsigma = o.dsigma ? o.dsigma + s0 : nullptr;
            f.val = o.val ? o.val + s0 : nullptr;
            f.grad = o.grad ? o.grad + s0 : nullptr;
            launch_finalize(c->stream, f);
        }
    }
}

// Small problems: one wavefront per query point (kernels_wave.hip, evaluation-only mode) instead of the tiled pipeline.
// Xs_dev: D x M column-major device copy of the query points; outputs as in EvalOut.
static bool eval_small(sls_gp* g, const double* Xs_dev, int M, const EvalOut& o) {
    const bool allow = tune_on(TUNE_WAVE_PATH);
    if (!allow || g->Np > WAVE_PATH_MAX_NP || g->D > WAVE_PATH_MAX_D || M > 4096) return false;
    sls_ctx* c = g->ctx;
    WaveArgs w;
    w.S = M; w.D = g->D; w.N = g->N; w.Np = g->Np; w.m = 1; w.n_local = 0; w.acq = o.acq;
    w.matern = g->kernel == SLS_KERNEL_ARD_MATERN52;
    w.a = g->a; w.mu_best = g->mu_best; w.ucb_h = o.ucb_h; w.c1 = 0; w.shrink = 0; w.gtol = 0; w.max_backtracks = 0;
    w.XT = g->XT.p; w.inv_ell = g->inv_ell.p; w.Kinv = g->Kinv.p; w.alpha = g->alpha.p; w.starts = Xs_dev;
    w.solve_sigma = g->sigma_mode == 1; w.Linv = g->Linv.p; w.U = g->U.p;
    w.x_out = nullptr; w.f_out = nullptr; w.ld = o.ldo;
    w.ev_mu = o.mu; w.ev_sigma = o.sigma; w.ev_dmu = o.dmu; w.ev_dsigma = o.dsigma; w.ev_val = o.val; w.ev_grad = o.grad;
    ProfScope ps(c, "acq_wave");
    launch_maximize_wave(c->stream, w);
    return true;
}

// host D x M column-major -> device candidate-major raw coordinates (clamping is NOT applied here)
static void upload_candidates(sls_gp* g, const double* Xs, int M, DBuf& raw, int Mp) {
    sls_ctx* c = g->ctx;
    const int D = g->D;
    std::vector<double> t((size_t)Mp * D, 0.5);
    for (int m = 0; m < M; ++m)
        for (int d = 0; d < D; ++d) t[(size_t)m + (size_t)d * Mp] = Xs[d + (size_t)m * D];
    raw.ensure((size_t)Mp * D);
    h2d(c, raw.p, t.data(), (size_t)Mp * D);
    sync(c);
}

// evaluate M host-supplied query points (D x M column-major) into the candidate-major device outputs of `o`
static void eval_host_points(sls_gp* g, const double* Xs, int M, int Mp, const EvalOut& o) {
    sls_ctx* c = g->ctx;
    const bool allow = tune_on(TUNE_WAVE_PATH);
    const size_t n = (size_t)g->D * M;
    if (allow && g->Np <= WAVE_PATH_MAX_NP && g->D <= WAVE_PATH_MAX_D && M <= 4096) {
        g->raw.ensure(n);
        if (n <= IO_STAGE_MAX && tune_on(TUNE_IO_STAGE)) {   // through the page-locked block: a copy from pageable memory blocks the host
            double* st = g->io_stage(n);
            std::memcpy(st, Xs, n * sizeof(double));
            h2d(c, g->raw.p, st, n);
        } else
            h2d(c, g->raw.p, Xs, n);
        if (eval_small(g, g->raw.p, M, o)) return;
    }
    if (n <= IO_STAGE_MAX && tune_on(TUNE_IO_STAGE)) {   // SLS_IO_STAGE=0: the transposing pageable upload of rounds 1-3 (A/B, tests)
        // as handed over (point-major), through page-locked memory: no transposition on the host, no synchronisation behind the upload
        double* st = g->io_stage(n);
        std::memcpy(st, Xs, n * sizeof(double));
        g->raw.ensure(n);
        h2d(c, g->raw.p, st, n);
        eval_candidates(g, g->raw.p, 0, M, o, true);
        return;
    }
    upload_candidates(g, Xs, M, g->raw, Mp);
    eval_candidates(g, g->raw.p, Mp, M, o);
}

// Results back to the host: up to two candidate-major device arrays (rows x Mp each; rows = 1: a vector, rows = D: transposed into
// D x M column-major) with ONE synchronisation, through the page-locked staging block when they fit.
static void download_cm2(sls_gp* g, const double* devA, double* hostA, const double* devB, double* hostB, int M, int Mp, int rows) {
    sls_ctx* c = g->ctx;
    const size_t each = (size_t)Mp * rows;
    const int cnt = (hostA ? 1 : 0) + (hostB ? 1 : 0);
    if (cnt == 0) {
        sync(c);   // nothing to fetch, but the evaluation in flight reads the staging block the next call writes into
        return;
    }
    std::vector<double> pageable;
    double* t;
    if (each * cnt <= IO_STAGE_MAX) t = g->io_stage(each * cnt);
    else {
        pageable.resize(each * cnt);
        t = pageable.data();
    }
    double* tA = t;
    double* tB = hostA ? t + each : t;
    if (hostA) d2h(c, tA, devA, each);
    if (hostB) d2h(c, tB, devB, each);
    sync(c);
    auto out = [&](const double* src, double* host) {
        if (rows == 1) std::copy(src, src + M, host);
        else
            for (int m = 0; m < M; ++m)
                for (int d = 0; d < rows; ++d) host[d + (size_t)m * rows] = src[(size_t)m + (size_t)d * Mp];
    };
    if (hostA) out(tA, hostA);
    if (hostB) out(tB, hostB);
}
static void download_cm(sls_gp* g, const double* dev, int M, int Mp, int rows, double* host /* rows x M col-major or M */) {
    download_cm2(g, dev, host, nullptr, nullptr, M, Mp, rows);
}

// ---- concurrent small evaluations (sls_gp::EvalSlot) ---------------------------------------------------------
// What one call wants; any pointer may be NULL.  Row outputs (dmu, dsigma, grad) are D x M column-major on the host.
struct SmallEvalOut {
    double *mu = nullptr, *sigma = nullptr, *dmu = nullptr, *dsigma = nullptr, *val = nullptr, *grad = nullptr;
    int acq = SLS_ACQ_EXPECTED_IMPROVEMENT;
    double ucb_h = 1.0;
};
constexpr int SLOT_MAX_POINTS = 64;
// true: evaluated (results are in the caller's arrays).  false: not applicable -- the caller takes the context's lock and the
// general path.  Holds no lock of the context; the caller must NOT hold it either (a mutator waiting for the state lock would
// then wait for this call, not the other way round -- there is no inversion, but the point of the path is to stay off that lock).
static bool eval_in_slot(sls_gp* g, const double* Xs, int M, const SmallEvalOut& o) {
    sls_ctx* c = g->ctx;
    if (M < 1 || M > SLOT_MAX_POINTS || !tune_on(TUNE_EVAL_SLOTS) || !tune_on(TUNE_WAVE_PATH) || c->prof_on) return false;
    if (g->D > WAVE_PATH_MAX_D) return false;          // D never changes for a handle
    (void)hipSetDevice(c->device);
    const int D = g->D;
    // Lock order: the slot (and, for the rare growth of its block, the context's lock, released again) FIRST, the shared lock on the
    // fitted state LAST and on its own.  Mutators hold the context's lock and then the state lock exclusively: taking the context's
    // lock under the shared state lock here would deadlock against them.
    const size_t n_in = (size_t)D * M, n_out = (size_t)(3 + 3 * D) * M;
    // ---- borrow a slot ----
    sls_ctx::EvalSlot* slot = nullptr;
    {
        std::unique_lock<std::mutex> lk(c->slot_mtx);
        for (;;) {
            for (auto& s : c->slots)
                if (!s->busy) { slot = s.get(); break; }
            if (slot) break;
            if ((int)c->slots.size() < sls_ctx::MAX_SLOTS) {
                c->slots.emplace_back(new sls_ctx::EvalSlot());
                slot = c->slots.back().get();
                break;
            }
            c->slot_cv.wait(lk);
        }
        slot->busy = true;
    }
    struct Release {
        sls_ctx* c;
        sls_ctx::EvalSlot* s;
        ~Release() {
            {
                std::lock_guard<std::mutex> lk(c->slot_mtx);
                s->busy = false;
            }
            c->slot_cv.notify_one();
        }
    } release{c, slot};
    if (!slot->stream) SLS_HIP(hipStreamCreateWithFlags(&slot->stream, hipStreamNonBlocking));
    const size_t need = (n_in + n_out) * sizeof(double);
    if (need > slot->bytes) {
        std::lock_guard<std::recursive_mutex> ctx_lock(c->mtx);   // the context's page-locked pool is not thread safe; growth is rare
        if (slot->host) c->host_give(slot->host, slot->bytes, true);
        slot->host = nullptr;
        slot->bytes = 0;
        slot->host = static_cast<double*>(c->host_take(std::max(need, (size_t)(4 + 4 * D) * SLOT_MAX_POINTS * sizeof(double)), true, &slot->bytes));
        SLS_HIP(hipHostGetDevicePointer((void**)&slot->dev, slot->host, 0));
    }
    std::shared_lock<std::shared_mutex> state_(g->state_mtx);
    if (g->Np > WAVE_PATH_MAX_NP) return false;        // grown past the wave path by appended points: the general path
    std::memcpy(slot->host, Xs, n_in * sizeof(double));
    double* od = slot->dev + n_in;       // device view of the outputs: mu | sigma | val | dmu | dsigma | grad, candidate-major, ld = M
    double* oh = slot->host + n_in;
    WaveArgs w;
    w.S = M; w.D = D; w.N = g->N; w.Np = g->Np; w.m = 1; w.n_local = 0; w.acq = o.acq;
    w.matern = g->kernel == SLS_KERNEL_ARD_MATERN52;
    w.a = g->a; w.mu_best = g->mu_best; w.ucb_h = o.ucb_h; w.c1 = 0; w.shrink = 0; w.gtol = 0; w.max_backtracks = 0;
    w.XT = g->XT.p; w.inv_ell = g->inv_ell.p; w.Kinv = g->Kinv.p; w.alpha = g->alpha.p; w.starts = slot->dev;
    w.solve_sigma = g->sigma_mode == 1; w.Linv = g->Linv.p; w.U = g->U.p;
    w.x_out = nullptr; w.f_out = nullptr; w.ld = M;
    w.ev_mu = o.mu ? od : nullptr;
    w.ev_sigma = o.sigma ? od + M : nullptr;
    w.ev_val = o.val ? od + 2 * (size_t)M : nullptr;
    w.ev_dmu = o.dmu ? od + 3 * (size_t)M : nullptr;
    w.ev_dsigma = o.dsigma ? od + (size_t)(3 + D) * M : nullptr;
    w.ev_grad = o.grad ? od + (size_t)(3 + 2 * D) * M : nullptr;
    launch_maximize_wave(slot->stream, w);
    SLS_HIP(hipStreamSynchronize(slot->stream));
    auto vec = [&](double* host, size_t off) {
        if (host) std::memcpy(host, oh + off, sizeof(double) * M);
    };
    auto rows = [&](double* host, size_t off) {
        if (!host) return;
        for (int m = 0; m < M; ++m)
            for (int d = 0; d < D; ++d) host[d + (size_t)m * D] = oh[off + (size_t)m + (size_t)d * M];
    };
    vec(o.mu, 0); vec(o.sigma, M); vec(o.val, 2 * (size_t)M);
    rows(o.dmu, 3 * (size_t)M); rows(o.dsigma, (size_t)(3 + D) * M); rows(o.grad, (size_t)(3 + 2 * D) * M);
    return true;
}

extern "C" int sls_gp_predict(sls_gp* g, const double* Xs, int M, double* mu, double* sigma) {
    SLS_TRY
    if (g && Xs && M >= 1 && M <= SLOT_MAX_POINTS) {
        SmallEvalOut so;
        so.mu = mu; so.sigma = sigma;
        if (eval_in_slot(g, Xs, M, so)) return SLS_OK;
    }
    std::unique_lock<std::recursive_mutex> lock_;
    if (g) {
        lock_ = std::unique_lock<std::recursive_mutex>(g->ctx->mtx);
        (void)hipSetDevice(g->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    SLS_REQUIRE(g && Xs && M >= 0, "sls_gp_predict: bad argument");
    if (M == 0) return SLS_OK;
    const int Mp = round_up(M, 128);
    g->outm.ensure(Mp); g->outs.ensure(Mp);
    EvalOut o;
    o.ldo = Mp; o.mu = g->outm.p; o.sigma = g->outs.p;
    eval_host_points(g, Xs, M, Mp, o);
    download_cm2(g, g->outm.p, mu, g->outs.p, sigma, M, Mp, 1);
    SLS_CATCH
}

extern "C" int sls_gp_predict_grad(sls_gp* g, const double* Xs, int M, double* dmu, double* dsigma) {
    SLS_TRY
    if (g && Xs && M >= 1 && M <= SLOT_MAX_POINTS) {
        SmallEvalOut so;
        so.dmu = dmu; so.dsigma = dsigma;
        if (eval_in_slot(g, Xs, M, so)) return SLS_OK;
    }
    std::unique_lock<std::recursive_mutex> lock_;
    if (g) {
        lock_ = std::unique_lock<std::recursive_mutex>(g->ctx->mtx);
        (void)hipSetDevice(g->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    SLS_REQUIRE(g && Xs && M >= 0, "sls_gp_predict_grad: bad argument");
    if (M == 0) return SLS_OK;
    const int Mp = round_up(M, 128), D = g->D;
    g->outv.ensure((size_t)Mp * D); g->outg.ensure((size_t)Mp * D);
    EvalOut o;
    o.ldo = Mp; o.dmu = g->outv.p; o.dsigma = g->outg.p;
    eval_host_points(g, Xs, M, Mp, o);
    download_cm2(g, g->outv.p, dmu, g->outg.p, dsigma, M, Mp, D);
    SLS_CATCH
}

extern "C" int sls_acq_eval(sls_gp* g, int acq_type, double ucb_h, const double* Xs, int M, double* val, double* grad) {
    SLS_TRY
    if (g && Xs && M >= 1 && M <= SLOT_MAX_POINTS && (acq_type == SLS_ACQ_EXPECTED_IMPROVEMENT || acq_type == SLS_ACQ_GP_UCB)) {
        SmallEvalOut so;
        so.val = val; so.grad = grad; so.acq = acq_type; so.ucb_h = ucb_h;
        if (eval_in_slot(g, Xs, M, so)) return SLS_OK;
    }
    std::unique_lock<std::recursive_mutex> lock_;
    if (g) {
        lock_ = std::unique_lock<std::recursive_mutex>(g->ctx->mtx);
        (void)hipSetDevice(g->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    SLS_REQUIRE(g && Xs && M >= 0, "sls_acq_eval: bad argument");
    SLS_REQUIRE(acq_type == SLS_ACQ_EXPECTED_IMPROVEMENT || acq_type == SLS_ACQ_GP_UCB, "unknown acquisition type %d", acq_type);
    if (M == 0) return SLS_OK;
    const int Mp = round_up(M, 128), D = g->D;
    {
        // Value-only evaluation of a small batch on a small problem -- one iteration of DIRECT (host/direct.cpp; the reference's
        // default global phase, src/acquisition-function.cpp:155-165) -- : the query points are read from, and the values written to,
        // a page-locked block the device maps: ONE launch + one synchronisation per call instead of upload + launch + download
        // (~60 -> ~30 us per call; ten calls per SubmitFeedbackData).  SLS_EVAL_ZEROCOPY=0: the copying path.
        if (!grad && val && tune_on(TUNE_WAVE_PATH) && tune_on(TUNE_EVAL_ZEROCOPY) && g->Np <= WAVE_PATH_MAX_NP &&
            D <= WAVE_PATH_MAX_D && M <= 4096) {
            sls_ctx* c = g->ctx;
            const size_t need = ((size_t)D * M + M) * sizeof(double);
            if (need > g->zc_bytes) {
                if (g->zc_host) c->host_give(g->zc_host, g->zc_bytes, true);
                g->zc_host = nullptr;
                g->zc_bytes = 0;
                g->zc_host = static_cast<double*>(c->host_take(need * 2, true, &g->zc_bytes));
                SLS_HIP(hipHostGetDevicePointer((void**)&g->zc_dev, g->zc_host, 0));
            }
            std::memcpy(g->zc_host, Xs, sizeof(double) * (size_t)D * M);
            EvalOut oz;
            oz.ldo = Mp; oz.val = g->zc_dev + (size_t)D * M; oz.acq = acq_type; oz.ucb_h = ucb_h;
            if (eval_small(g, g->zc_dev, M, oz)) {
                sync(c);
                std::memcpy(val, g->zc_host + (size_t)D * M, sizeof(double) * M);
                return SLS_OK;
            }
        }
    }
    g->outm.ensure(Mp);
    if (grad) g->outg.ensure((size_t)Mp * D);
    EvalOut o;
    o.ldo = Mp; o.val = g->outm.p; o.grad = grad ? g->outg.p : nullptr; o.acq = acq_type; o.ucb_h = ucb_h;
    eval_host_points(g, Xs, M, Mp, o);
    if (val) download_cm(g, g->outm.p, M, Mp, 1, val);
    if (grad) download_cm(g, g->outg.p, M, Mp, D, grad);
    SLS_CATCH
}

// ---- multi-start maximiser -------------------------------------------------------------------------------------
extern "C" void sls_lbfgs_default_opts(sls_lbfgs_opts* o) {
    o->struct_size = (int)sizeof(sls_lbfgs_opts);
    o->history = 6; o->c1 = 1e-4; o->shrink = 0.5; o->gtol = 0.0; o->max_backtracks = 20; o->ftol_rel = 0.0; o->xtol_rel = 0.0;
}
// the caller's struct may be shorter than this library's (an older header): its struct_size bytes over the defaults
static sls_lbfgs_opts read_lbfgs_opts(const sls_lbfgs_opts* in) {
    sls_lbfgs_opts o;
    sls_lbfgs_default_opts(&o);
    if (!in) return o;
    const int min_size = (int)(offsetof(sls_lbfgs_opts, max_backtracks) + sizeof(int));   // the members of the first version
    SLS_REQUIRE(in->struct_size >= min_size && in->struct_size <= (int)sizeof(sls_lbfgs_opts),
                "sls_lbfgs_opts.struct_size = %d: expected %d .. %d (call sls_lbfgs_default_opts first)", in->struct_size, min_size,
                (int)sizeof(sls_lbfgs_opts));
    memcpy(&o, in, (size_t)in->struct_size);
    o.struct_size = (int)sizeof(sls_lbfgs_opts);
    return o;
}

static void ensure_lbfgs(sls_gp* g, int Sp, int m) {
    if (Sp <= g->lb_Sp && m <= g->lb_m) return;
    const size_t D = g->D, S = Sp;
    g->lb_x.ensure(S * D); g->lb_g.ensure(S * D); g->lb_dir.ensure(S * D); g->lb_xt.ensure(S * D); g->lb_scr.ensure(S * D);
    const size_t Dh = D <= 16 ? 16 : (D <= 64 ? 64 : D);   // lbfgs_step_reg_kernel keeps rows of 4 DPL doubles per (start, pair)
    g->lb_S.ensure(S * Dh * m); g->lb_Y.ensure(S * Dh * m); g->lb_rho.ensure(S * m);
    g->lb_f.ensure(S); g->lb_t.ensure(S); g->lb_val.ensure(S); g->lb_grad.ensure(S * D); g->lb_xc.ensure(S * D);
    g->lb_int_buf.ensure(((S * 6 + 128 + S / 1024 + 8) * sizeof(int) + 7) / 8);   // ... | count (64) | block counts
    g->lb_int = reinterpret_cast<int*>(g->lb_int_buf.p);
    g->lb_Sp = Sp; g->lb_m = m;
}

// value (+ gradient) of the acquisition at candidate-major points xr; gs != nullptr: sigma / dsigma come from gs
// (objective_for_multiple_points, src/acquisition-function.cpp:63-110)
static void eval_acq(sls_gp* g, sls_gp* gs, const double* xr, long ldr, int S, int acq_type, double ucb_h, double* val,
                     double* grad, long ldo) {
    if (!gs) {
        EvalOut eo;
        eo.ldo = ldo; eo.val = val; eo.grad = grad; eo.acq = acq_type; eo.ucb_h = ucb_h;
        eval_candidates(g, xr, ldr, S, eo);
        return;
    }
    const size_t D = g->D;
    g->pair_mu.ensure(ldo); g->pair_sg.ensure(ldo);
    if (grad) { g->pair_dmu.ensure(ldo * D); g->pair_dsg.ensure(ldo * D); }
    EvalOut a, b;
    a.ldo = ldo; a.mu = g->pair_mu.p; a.dmu = grad ? g->pair_dmu.p : nullptr;
    b.ldo = ldo; b.sigma = g->pair_sg.p; b.dsigma = grad ? g->pair_dsg.p : nullptr;
    eval_candidates(g, xr, ldr, S, a);
    eval_candidates(gs, xr, ldr, S, b);
    launch_combine(g->ctx->stream, S, (int)D, ldo, g->pair_mu.p, g->pair_sg.p, g->pair_dmu.p, g->pair_dsg.p, acq_type, g->mu_best,
                   ucb_h, val, grad);
}

static void maximize_impl(sls_gp* g, sls_gp* gs, int acq_type, double ucb_h, const double* starts_dev, int S, int n_local,
                          const sls_lbfgs_opts* opts_in, long off, double* x_out, double* val_out, long* idx_out,
                          double* x_stars, double* y_stars) {
    sls_ctx* c = g->ctx;
    SLS_REQUIRE(S >= 1 && n_local >= 1, "sls_acq_maximize: need S >= 1 and n_local >= 1");
    SLS_REQUIRE(acq_type == SLS_ACQ_EXPECTED_IMPROVEMENT || acq_type == SLS_ACQ_GP_UCB, "unknown acquisition type %d", acq_type);
    const sls_lbfgs_opts o = read_lbfgs_opts(opts_in);
    SLS_REQUIRE(o.history >= 1 && o.history <= 8, "L-BFGS history must be in 1..8");
    const int Sp = round_up(S, 128), D = g->D;
    ensure_lbfgs(g, Sp, o.history);
    g->stat_issued = 0; g->stat_cap = (long)S * n_local; g->stat_rounds = 0; g->stat_live_end = 0;
    LbfgsState st;
    st.live = nullptr; st.nlive = S; st.ldv = Sp;
    st.S = S; st.D = D; st.m = o.history; st.ld = Sp;
    st.x = g->lb_x.p; st.g = g->lb_g.p; st.dir = g->lb_dir.p; st.xt = g->lb_xt.p; st.scr = g->lb_scr.p;
    st.Sh = g->lb_S.p; st.Yh = g->lb_Y.p; st.rho = g->lb_rho.p; st.f = g->lb_f.p; st.t = g->lb_t.p;
    st.hlen = g->lb_int; st.hpos = g->lb_int + Sp; st.nbt = g->lb_int + 2 * Sp; st.done = g->lb_int + 3 * Sp;
    st.c1 = o.c1; st.shrink = o.shrink; st.gtol = o.gtol; st.max_backtracks = o.max_backtracks;
    st.ftol_rel = o.ftol_rel; st.xtol_rel = o.xtol_rel;
    // Small problems: one wavefront per start runs the whole search in a single launch (kernels_wave.hip).  The choice
    // depends only on the fitted state and the start count, so repeated / sharded calls take the same path.
    bool used_wave = false;
    const unsigned long long* wave_useful = nullptr;   // the one-wavefront-per-start run's count of useful evaluations (device)
    {
        const bool allow = tune_on(TUNE_WAVE_PATH);
        if (allow && !gs && g->Np <= WAVE_PATH_MAX_NP && D <= WAVE_PATH_MAX_D && S <= 4096) {
            WaveArgs w;
            w.S = S; w.D = D; w.N = g->N; w.Np = g->Np; w.m = o.history; w.n_local = n_local; w.acq = acq_type;
            w.matern = g->kernel == SLS_KERNEL_ARD_MATERN52;
            w.a = g->a; w.mu_best = g->mu_best; w.ucb_h = ucb_h; w.c1 = o.c1; w.shrink = o.shrink; w.gtol = o.gtol;
            w.max_backtracks = o.max_backtracks;
            w.ftol_rel = o.ftol_rel; w.xtol_rel = o.xtol_rel;
            w.XT = g->XT.p; w.inv_ell = g->inv_ell.p; w.Kinv = g->Kinv.p; w.alpha = g->alpha.p; w.starts = starts_dev;
            w.solve_sigma = g->sigma_mode == 1; w.Linv = g->Linv.p; w.U = g->U.p;
            w.x_out = st.x; w.f_out = st.f; w.ld = Sp;
            w.ev_mu = w.ev_sigma = w.ev_dmu = w.ev_dsigma = w.ev_val = w.ev_grad = nullptr;
            // evaluations of starts that were still moving (finished starts run idle to keep the barriers uniform): counted
            // by the kernel into the live-count words of the integer scratch
            unsigned long long* d_useful = reinterpret_cast<unsigned long long*>(g->lb_int + 6 * (size_t)Sp + 32);
            SLS_HIP(hipMemsetAsync(d_useful, 0, sizeof(unsigned long long), c->stream));
            w.useful = d_useful;
            long long* d_trace = nullptr;
            if (tune_set(TUNE_WAVE_TRACE)) {
                d_trace = reinterpret_cast<long long*>(g->lb_int + 6 * (size_t)Sp + 64);
                SLS_HIP(hipMemsetAsync(d_trace, 0, 9 * sizeof(long long), c->stream));
                w.trace = d_trace;
            }
            {
                ProfScope ps(c, "acq_wave");
                launch_maximize_wave(c->stream, w);
            }
            if (d_trace) {
                long long tr[9] = {};
                SLS_HIP(hipMemcpyAsync(tr, d_trace, sizeof(tr), hipMemcpyDeviceToHost, c->stream));
                sync(c);
                fprintf(stderr, "wave trace (us): S %d N %d D %d evals %lld | kvec %.1f  Kinv.k %.1f  sums %.1f  grad+acq %.1f  direction %.1f  "
                        "bookkeeping %.1f  total %.1f  (shader clock %.0f MHz)\n", S, g->N, D, tr[6], tr[0] * 0.01, tr[1] * 0.01, tr[2] * 0.01, tr[3] * 0.01, tr[4] * 0.01,
                        tr[5] * 0.01, tr[7] * 0.01, tr[7] > 0 ? (double)tr[8] / (tr[7] * 0.01) : 0.0);
            }
            wave_useful = d_useful;   // read back with the best start, behind the one synchronisation of the run
            g->stat_rounds = n_local;
            used_wave = true;
        }
    }
    if (!used_wave) {
        // Lock-step rounds over the ACTIVE SET.  NLopt's max_evals is a cap per start, not a quota
        // (src/acquisition-function.cpp:128-129): a start that can no longer move (stationary projected gradient, null
        // step, exhausted backtracking) is finished.  After every round the starts still moving are compacted, in
        // increasing order, into dense 128-wide tiles, so cross_gram / acq_gemm / grad_gemm only see live columns.  A
        // candidate's arithmetic does not depend on the column it occupies, so every start ends with the same bits as
        // in the uncompacted schedule (SLS_COMPACT=0: every start is re-evaluated every round; tests compare the two).
        const bool compact = tune_on(TUNE_COMPACT);
        int* live_a = g->lb_int + 4 * (size_t)Sp;
        int* live_b = live_a + Sp;
        int* d_count = live_b + Sp;
        int* d_blocks = d_count + 64;     // per-block counts of the compaction
        launch_clamp_starts(c->stream, starts_dev, D, S, st.xt, Sp, Sp);
        const double* trial = st.xt;      // candidate-major trial points of this round, leading dimension Sp
        const int* live = nullptr;        // identity
        int nlive = S, moving = S;
        st.ldv = Sp;
        for (int ev = 0; ev < n_local && nlive > 0; ++ev) {
            eval_acq(g, gs, trial, Sp, nlive, acq_type, ucb_h, g->lb_val.p, g->lb_grad.p, Sp);
            g->stat_issued += compact ? nlive : moving;       // SLS_COMPACT=0 evaluates finished starts too: they do not count
            g->stat_rounds += 1;
            {
                ProfScope ps(c, "lbfgs");
                st.live = live; st.nlive = nlive;
                launch_lbfgs_step(c->stream, st, g->lb_val.p, g->lb_grad.p, ev == 0);
                if (ev + 1 < n_local) {
                    int* live_next = (live == live_a) ? live_b : live_a;
                    launch_compact_live(c->stream, live, nlive, st.done, live_next, d_count, d_blocks);
                    if (compact) {
                        launch_gather_trials(c->stream, st.xt, Sp, D, live_next, d_count, nlive, g->lb_xc.p, Sp);
                        live = live_next;
                        trial = g->lb_xc.p;
                    }
                }
            }
            if (ev + 1 < n_local) {
                int cnt = 0;
                SLS_HIP(hipMemcpyAsync(&cnt, d_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                sync(c);
                if (compact) nlive = cnt;
                else moving = cnt;                            // statistics only: the launch shapes stay at S
            }
        }
        g->stat_live_end = compact ? nlive : moving;
    }   // !used_wave
    // best start and its coordinates: one launch into the handle's mapped block, one synchronisation, no copies
    launch_argmax_neg_gather(c->stream, st.f, S, st.x, Sp, D, g->sum_dev + 8, wave_useful);
    sync(c);
    const double* best = g->sum_host + 8;
    const long bi = (long)best[1];
    if (wave_useful) g->stat_issued = (long)best[2];
    if (val_out) *val_out = best[0];
    if (idx_out) *idx_out = bi + off;
    if (x_out) std::copy(best + 8, best + 8 + D, x_out);
    if (x_stars) download_cm(g, st.x, S, Sp, D, x_stars);
    if (y_stars) {
        download_cm(g, st.f, S, Sp, 1, y_stars);
        for (int i = 0; i < S; ++i) y_stars[i] = -y_stars[i];
    }
}

extern "C" int sls_acq_last_stats(sls_gp* g, long* evals_issued, long* evals_cap, int* rounds, int* live_at_end) {
    if (!g) return SLS_ERR_INVALID;
    std::unique_lock<std::recursive_mutex> lock_(g->ctx->mtx);
    if (evals_issued) *evals_issued = g->stat_issued;
    if (evals_cap) *evals_cap = g->stat_cap;
    if (rounds) *rounds = g->stat_rounds;
    if (live_at_end) *live_at_end = g->stat_live_end;
    return SLS_OK;
}

extern "C" int sls_acq_maximize(sls_gp* g, int acq_type, double ucb_h, const double* starts, int S, int n_local,
                                const sls_lbfgs_opts* opts, long start_index_offset, double* x_out, double* val_out,
                                long* idx_out, double* x_stars, double* y_stars) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (g) {
        lock_ = std::unique_lock<std::recursive_mutex>(g->ctx->mtx);
        (void)hipSetDevice(g->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    SLS_REQUIRE(g && starts, "sls_acq_maximize: NULL argument");
    SLS_REQUIRE(S >= 1, "sls_acq_maximize: need S >= 1");
    sls_ctx* c = g->ctx;
    DBuf sd;
    sd.ensure((size_t)g->D * S);
    h2d(c, sd.p, starts, (size_t)g->D * S);
    maximize_impl(g, nullptr, acq_type, ucb_h, sd.p, S, n_local, opts, start_index_offset, x_out, val_out, idx_out, x_stars,
                  y_stars);
    SLS_CATCH
}

static void check_pair(sls_gp* g, sls_gp* gs) {
    SLS_REQUIRE(g && gs, "NULL handle");
    SLS_REQUIRE(g->ctx == gs->ctx && g->D == gs->D, "the two regressors must share the context and the dimensionality");
}

extern "C" int sls_acq_maximize_pair(sls_gp* g, sls_gp* gs, int acq_type, double ucb_h, const double* starts, int S, int n_local,
                                     const sls_lbfgs_opts* opts, double* x_out, double* val_out, long* idx_out) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (g) {
        lock_ = std::unique_lock<std::recursive_mutex>(g->ctx->mtx);
        (void)hipSetDevice(g->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    check_pair(g, gs);
    SLS_REQUIRE(starts && S >= 1, "sls_acq_maximize_pair: bad argument");
    sls_ctx* c = g->ctx;
    DBuf sd;
    sd.ensure((size_t)g->D * S);
    h2d(c, sd.p, starts, (size_t)g->D * S);
    maximize_impl(g, gs, acq_type, ucb_h, sd.p, S, n_local, opts, 0, x_out, val_out, idx_out, nullptr, nullptr);
    SLS_CATCH
}

extern "C" int sls_acq_eval_pair(sls_gp* g, sls_gp* gs, int acq_type, double ucb_h, const double* Xs, int M, double* val,
                                 double* grad) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (g) {
        lock_ = std::unique_lock<std::recursive_mutex>(g->ctx->mtx);
        (void)hipSetDevice(g->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    check_pair(g, gs);
    SLS_REQUIRE(Xs && M >= 0, "sls_acq_eval_pair: bad argument");
    SLS_REQUIRE(acq_type == SLS_ACQ_EXPECTED_IMPROVEMENT || acq_type == SLS_ACQ_GP_UCB, "unknown acquisition type %d", acq_type);
    if (M == 0) return SLS_OK;
    const int Mp = round_up(M, 128), D = g->D;
    upload_candidates(g, Xs, M, g->raw, Mp);
    g->outm.ensure(Mp);
    if (grad) g->outg.ensure((size_t)Mp * D);
    eval_acq(g, gs, g->raw.p, Mp, M, acq_type, ucb_h, g->outm.p, grad ? g->outg.p : nullptr, Mp);
    if (val) download_cm(g, g->outm.p, M, Mp, 1, val);
    if (grad) download_cm(g, g->outg.p, M, Mp, D, grad);
    SLS_CATCH
}

extern "C" int sls_acq_maximize_dev(sls_gp* g, int acq_type, double ucb_h, const double* starts_dev, int S, int n_local,
                                    const sls_lbfgs_opts* opts, long start_index_offset, double* x_out, double* val_out,
                                    long* idx_out) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (g) {
        lock_ = std::unique_lock<std::recursive_mutex>(g->ctx->mtx);
        (void)hipSetDevice(g->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    SLS_REQUIRE(g && starts_dev, "sls_acq_maximize_dev: NULL argument");
    maximize_impl(g, nullptr, acq_type, ucb_h, starts_dev, S, n_local, opts, start_index_offset, x_out, val_out, idx_out, nullptr,
                  nullptr);
    SLS_CATCH
}

// ---- free functions ----------------------------------------------------------------------------------------------
extern "C" int sls_gram(sls_ctx* c, const double* X, int D, int N, const double* theta, double b, int kernel, double* K_out) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (c) lock_ = std::unique_lock<std::recursive_mutex>(c->mtx);
    SLS_REQUIRE(c && X && K_out && D >= 1 && N >= 1, "sls_gram: bad argument");
    check_theta(theta, D);
    SLS_HIP(hipSetDevice(c->device));
    const int Np = round_up(N, 128), Dp = round_up(D, 16);
    DBuf Xd, il, XT, nx, K;
    Xd.ensure((size_t)D * N); il.ensure(Dp); XT.ensure((size_t)Np * Dp); nx.ensure(Np); K.ensure((size_t)Np * Np);
    std::vector<double> ilh(Dp, 0.0);
    for (int d = 0; d < D; ++d) ilh[d] = 1.0 / theta[1 + d];
    h2d(c, Xd.p, X, (size_t)D * N);
    h2d(c, il.p, ilh.data(), Dp);
    launch_prep_points(c->stream, Xd.p, D, N, il.p, XT.p, Np, Np, Dp, nx.p);
    launch_gram_sym(c->stream, XT.p, Np, Dp, nx.p, Np, N, KernelSpec{kernel, theta[0]}, b, K.p, false);
    d2h_matrix(c, K_out, K.p, N, Np);
    sync(c);
    SLS_CATCH
}

extern "C" int sls_gram_cross(sls_ctx* c, const double* X, int D, int N, const double* Xs, int M, const double* theta, int kernel,
                              double* Ks_out) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (c) lock_ = std::unique_lock<std::recursive_mutex>(c->mtx);
    SLS_REQUIRE(c && X && Xs && Ks_out && D >= 1 && N >= 1 && M >= 1, "sls_gram_cross: bad argument");
    check_theta(theta, D);
    SLS_HIP(hipSetDevice(c->device));
    const int Np = round_up(N, 128), Mp = round_up(M, 128), Dp = round_up(D, 16);
    DBuf Xd, Xsd, il, XT, XsT, nx, ns, Ks, Cs;
    Xd.ensure((size_t)D * N); Xsd.ensure((size_t)D * M); il.ensure(Dp);
    XT.ensure((size_t)Np * Dp); XsT.ensure((size_t)Mp * Dp); nx.ensure(Np); ns.ensure(Mp);
    Ks.ensure((size_t)Mp * Np);
    const bool matern = kernel == SLS_KERNEL_ARD_MATERN52;
    if (matern) Cs.ensure((size_t)Mp * Np);
    std::vector<double> ilh(Dp, 0.0);
    for (int d = 0; d < D; ++d) ilh[d] = 1.0 / theta[1 + d];
    h2d(c, Xd.p, X, (size_t)D * N);
    h2d(c, Xsd.p, Xs, (size_t)D * M);
    h2d(c, il.p, ilh.data(), Dp);
    launch_prep_points(c->stream, Xd.p, D, N, il.p, XT.p, Np, Np, Dp, nx.p);
    launch_prep_points(c->stream, Xsd.p, D, M, il.p, XsT.p, Mp, Mp, Dp, ns.p);
    launch_cross_gram(c->stream, XsT.p, Mp, ns.p, Mp, XT.p, Np, nx.p, Np, N, Dp, KernelSpec{kernel, theta[0]}, nullptr, Ks.p,
                      matern ? Cs.p : Ks.p, Mp, nullptr, nullptr);
    // device image is candidate-major [m + i*Mp]; the API returns N x M column-major (column m = k(xs_m, X))
    std::vector<double> t((size_t)Mp * Np);
    d2h(c, t.data(), Ks.p, (size_t)Mp * Np);
    sync(c);
    for (int m = 0; m < M; ++m)
        for (int i = 0; i < N; ++i) Ks_out[i + (size_t)m * N] = t[(size_t)m + (size_t)i * Mp];
    SLS_CATCH
}

static void upload_padded_spd(sls_ctx* c, DBuf& A, const double* src, int N, int Np) {
    A.ensure((size_t)Np * Np);
    launch_fill(c->stream, A.p, (long)Np * Np, 0.0);
    h2d_matrix(c, A.p, src, N, Np);
    if (Np > N) {
        std::vector<double> ones(Np - N, 1.0);
        SLS_HIP(hipMemcpy2DAsync(A.p + (size_t)N * (Np + 1), (size_t)(Np + 1) * 8, ones.data(), 8, 8, Np - N,
                                 hipMemcpyHostToDevice, c->stream));
        sync(c);
    }
}

extern "C" int sls_potrf(sls_ctx* c, double* A, int N) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (c) lock_ = std::unique_lock<std::recursive_mutex>(c->mtx);
    SLS_REQUIRE(c && A && N >= 1, "sls_potrf: bad argument");
    SLS_HIP(hipSetDevice(c->device));
    const int Np = round_up(N, 128);
    DBuf Ad, Li;
    Li.ensure((size_t)Np * Np);
    int info2[2] = {0, 0};
    std::vector<double> out((size_t)N * N);
    for (int attempt = 0;; ++attempt) {
        upload_padded_spd(c, Ad, A, N, Np);
        SLS_HIP(hipMemsetAsync(c->d_info, 0, 64, c->stream));
        c->potrf_tick_rearm();
        launch_potrf(c->stream, Ad.p, Np, Li.p, c->d_info, 0, c->potrf_df_sync(Np));
        launch_zero_upper(c->stream, Ad.p, Np);
        SLS_HIP(hipMemcpyAsync(info2, c->d_info, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        d2h_matrix(c, out.data(), Ad.p, N, Np);
        sync(c);
        if (!potrf_gave_up(c, info2[1], attempt)) break;
    }
    std::copy(out.begin(), out.end(), A);
    const int info = info2[0];
    if (info != 0) {
        set_error("sls_potrf: matrix is not positive definite (pivot %d)", info - 1);
        return SLS_ERR_NOT_SPD;
    }
    SLS_CATCH
}

extern "C" int sls_potrs(sls_ctx* c, const double* L, int N, double* B, int nrhs) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (c) lock_ = std::unique_lock<std::recursive_mutex>(c->mtx);
    SLS_REQUIRE(c && L && B && N >= 1 && nrhs >= 1, "sls_potrs: bad argument");
    SLS_HIP(hipSetDevice(c->device));
    const int Np = round_up(N, 128), Rp = round_up(nrhs, 128);
    DBuf Ld, Li, Bd;
    upload_padded_spd(c, Ld, L, N, Np);
    launch_zero_upper(c->stream, Ld.p, Np);
    Li.ensure((size_t)Np * Np);
    launch_diag_inverse(c->stream, Ld.p, Np, Li.p);
    Bd.ensure((size_t)Np * Rp);
    launch_fill(c->stream, Bd.p, (long)Np * Rp, 0.0);
    SLS_HIP(hipMemcpy2DAsync(Bd.p, (size_t)Np * 8, B, (size_t)N * 8, (size_t)N * 8, nrhs, hipMemcpyHostToDevice, c->stream));
    launch_potrs(c->stream, Ld.p, Li.p, Np, Bd.p, Rp);
    SLS_HIP(hipMemcpy2DAsync(B, (size_t)N * 8, Bd.p, (size_t)Np * 8, (size_t)N * 8, nrhs, hipMemcpyDeviceToHost, c->stream));
    sync(c);
    SLS_CATCH
}

extern "C" int sls_potri(sls_ctx* c, const double* L, int N, double* Ainv) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (c) lock_ = std::unique_lock<std::recursive_mutex>(c->mtx);
    SLS_REQUIRE(c && L && Ainv && N >= 1, "sls_potri: bad argument");
    SLS_HIP(hipSetDevice(c->device));
    const int Np = round_up(N, 128);
    DBuf Ld, Li, Ki, Ui;
    upload_padded_spd(c, Ld, L, N, Np);
    launch_zero_upper(c->stream, Ld.p, Np);
    Li.ensure((size_t)Np * Np);
    Ki.ensure((size_t)Np * Np);
    Ui.ensure((size_t)Np * Np);
    launch_fill(c->stream, Li.p, (long)Np * Np, 0.0);
    launch_diag_inverse(c->stream, Ld.p, Np, Li.p);
    launch_trtri(c->stream, Ld.p, Np, Li.p, Ki.p, Ui.p);
    launch_lauum(c->stream, Ui.p, Np, Ki.p);
    d2h_matrix(c, Ainv, Ki.p, N, Np);
    sync(c);
    SLS_CATCH
}

// Grow the fitted state by one observation in O(N^2): Schur-complement update of K^-1, one new row of L and L^-1,
// alpha, mu at the data points, arg max and log-determinant.  Replaces the full refit of the dummy regressor in
// FindNextPoints (src/acquisition-function.cpp:280-293).  When N is a multiple of 128 the padded buffers are full and
// the handle is rebuilt from scratch on the device instead (same results).
extern "C" int sls_gp_append_point(sls_gp* g, const double* x, double y_new) {
    SLS_TRY
    std::unique_lock<std::recursive_mutex> lock_;
    if (g) {
        lock_ = std::unique_lock<std::recursive_mutex>(g->ctx->mtx);
        (void)hipSetDevice(g->ctx->device);   // the handle's device, whatever the caller's current device is
    }
    SLS_REQUIRE(g && x, "sls_gp_append_point: NULL argument");
    std::unique_lock<std::shared_mutex> state_(g->state_mtx);   // ends with gp_fetch_summary's synchronisation: the state is complete
    sls_ctx* c = g->ctx;
    const int D = g->D, N = g->N, Np = g->Np;
    if (g->host_stale) {   // the device copies are authoritative after sls_gp_refit_dev
        d2h(c, g->Xh.data(), g->X.p, (size_t)D * N);
        d2h(c, g->yh.data(), g->y.p, (size_t)N);
        sync(c);
        g->host_stale = false;
    }
    g->Xh.insert(g->Xh.end(), x, x + D);
    g->yh.push_back(y_new);
    if (N % 128 == 0) {
        gp_setup(g);   // buffers full: re-size with one more 128-block and refit on the device
        return SLS_OK;
    }
    // candidate-major upload of the single point, cross covariances k = k(X, x) through the regular evaluation kernels
    upload_candidates(g, x, 1, g->raw, 128);
    ensure_eval_ws(g, 128);
    g->outv.ensure(4 * (size_t)Np + 8);
    double* kvec = g->outv.p;            // k (Np, strided copy of Ks row 0)
    double* uvec = kvec + Np;
    double* lvec = uvec + Np;
    double* tmp = lvec + Np;
    double* scal = tmp + Np;
    KernelSpec ks{g->kernel, g->a};
    launch_prep_cands(c->stream, g->raw.p, 128, D, 1, g->inv_ell.p, g->XsT.p, 128, 128, g->Dcols, g->ns.p);
    double* Cs = (g->kernel == SLS_KERNEL_ARD_MATERN52) ? g->Cs.p : g->Ks.p;
    launch_cross_gram(c->stream, g->XsT.p, 128, g->ns.p, 128, g->XT.p, Np, g->nx.p, Np, N, g->Dp, ks, nullptr, g->Ks.p, Cs, 128,
                      nullptr, nullptr);
    // Ks[n + i*128], n = 0 -> stride-128 gather into a dense vector (padding rows i >= N are 0)
    SLS_HIP(hipMemcpy2DAsync(kvec, 8, g->Ks.p, 128 * 8, 8, Np, hipMemcpyDeviceToDevice, c->stream));
    launch_gemv_n(c->stream, g->Kinv.p, Np, kvec, uvec, g->gemv_part.p);     // u = K^-1 k   (padding: identity block x 0 = 0)
    launch_gemv_n(c->stream, g->Linv.p, Np, kvec, lvec, g->gemv_part.p, true);     // l = L^-1 k
    launch_append_dots(c->stream, kvec, uvec, lvec, g->y.p, N, scal);
    const double kappa = g->a + g->b;
    launch_append_update(c->stream, g->Kinv.p, g->L.p, g->Linv.p, g->alpha.p, Np, N, uvec, lvec, scal, kappa, y_new);
    // new training point joins X, y, the scaled copies and the hoisted quantities
    SLS_HIP(hipMemcpyAsync(g->y.p + N, &g->yh[N], 8, hipMemcpyHostToDevice, c->stream));
    g->X.ensure((size_t)D * (N + 1));   // may reallocate: refill from the host copy
    SLS_HIP(hipMemcpyAsync(g->X.p, g->Xh.data(), (size_t)D * (N + 1) * 8, hipMemcpyHostToDevice, c->stream));
    g->N = N + 1;
    g->generation = ++g_gp_generation;
    launch_prep_points(c->stream, g->X.p, D, N + 1, g->inv_ell.p, g->XT.p, Np, Np, g->Dcols, g->nx.p);
    launch_scale_rows(c->stream, g->XT.p, g->alpha.p, g->XaT.p, Np, Np, g->Dcols);
    if (g->sigma_mode == 1) launch_transpose_full(c->stream, g->Linv.p, g->U.p, Np);   // the rank-1 growth does not maintain U
    SLS_HIP(hipMemsetAsync(c->d_info, 0, 64, c->stream));
    launch_fit_summary(c->stream, g->y.p, g->alpha.p, g->b, N + 1, g->mu_data.p, g->L.p, Np, c->d_info, g->scal.p, g->d_idx, g->sum_dev);
    gp_fetch_summary(g);
    SLS_REQUIRE(std::isfinite(g->logdet), "sls_gp_append_point: the extended K_y is not positive definite");
    SLS_CATCH
}

