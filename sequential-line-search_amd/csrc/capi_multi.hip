// Multi-GPU maximisation behind the C ABI (include/sls_hip.h "multi-GPU" section).
//
// The multi-start search shards over its starts (src/acquisition-function.cpp:125-141: the iterations of the reference's
// parallel loop share only the const regressor).  Each GPU holds a replica of the fitted state (the fit is ~N^3 flops from a
// few MB of X, cheaper to repeat than to broadcast the 512 MB inverse), runs its contiguous slice of the starts, and
// contributes (value, global start index, x[D]) to ONE ncclAllGather; every rank then takes the first maximum (highest
// value, ties -> lowest global index = Eigen maxCoeff, :146-153).  No other collective.
//
// Two deployment shapes share the exchange code:
//   sls_multi_*   one process drives n devices, one host thread per device, communicators from ncclCommInitAll;
//   sls_comm_*    one process per GPU (torch.distributed.run / mpirun), communicator from ncclCommInitRank with a
//                 unique id the caller distributes (bench.py broadcasts it over its rendezvous store).
// RCCL is loaded lazily with dlopen: a single-GPU user never touches it.  This file is a pure client of the C ABI of
// capi.hip (sls_gp_create / sls_acq_maximize / ...) plus RCCL and std::thread.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

using namespace slsk;

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl& rccl_state() {
    static Rccl r;
    return r;
}

// nullptr (and rccl_state().why filled) when RCCL cannot be loaded
Rccl* rccl() {
    Rccl& r = rccl_state();
    static std::once_flag once;
    std::call_once(once, [&r] {
        // a copy already mapped into the process (e.g. the one PyTorch ships) wins over a second load
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!r.lib)
            for (const char* n : names)
                if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.lib) {
            const char* e = dlerror();   // one call: dlerror() clears the state it returns
            r.why = std::string("cannot load librccl: ") + (e ? e : "?");
            return;
        }
#define SLS_SYM(field, name)                                                     \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));           \
    if (!r.field) { r.why = std::string("librccl lacks ") + name; r.lib = nullptr; return; }
        SLS_SYM(GetUniqueId, "ncclGetUniqueId")
        SLS_SYM(CommInitRank, "ncclCommInitRank")
        SLS_SYM(CommInitAll, "ncclCommInitAll")
        SLS_SYM(AllGather, "ncclAllGather")
        SLS_SYM(GroupStart, "ncclGroupStart")
        SLS_SYM(GroupEnd, "ncclGroupEnd")
        SLS_SYM(CommDestroy, "ncclCommDestroy")
        SLS_SYM(GetErrorString, "ncclGetErrorString")
#undef SLS_SYM
    });
    return r.lib ? &r : nullptr;
}

#define SLS_NCCL(x)                                                                                      \
    do {                                                                                                 \
        ncclResult_t r_ = (x);                                                                           \
        if (r_ != ncclSuccess) {                                                                         \
            slsk::set_error("%s failed: %s (%s:%d)", #x, rccl()->GetErrorString(r_), __FILE__, __LINE__); \
            throw slsk::HipFail{SLS_ERR_HIP};                                                            \
        }                                                                                                \
    } while (0)

// first maximum over `world` records of (value, index, x[D]) laid out back to back
void merge_first_max(const double* rec, int world, int D, double* val, long* idx, double* x) {
    int best = -1;
    for (int r = 0; r < world; ++r) {
        const double* p = rec + (size_t)r * (2 + D);
        if (p[1] < 0.0) continue;                       // rank held no starts
        if (best < 0) { best = r; continue; }
        const double* q = rec + (size_t)best * (2 + D);
        if (p[0] > q[0] || (p[0] == q[0] && p[1] < q[1])) best = r;
    }
    if (best < 0) best = 0;
    const double* q = rec + (size_t)best * (2 + D);
    if (val) *val = q[0];
    if (idx) *idx = (long)q[1];
    if (x) std::memcpy(x, q + 2, sizeof(double) * (size_t)D);
}

// contiguous slice [lo, hi) of S starts owned by shard r of n (the first S % n shards get one more)
void shard_range(int S, int r, int n, int* lo, int* hi) {
    const int base = S / n, rem = S % n;
    *lo = r * base + std::min(r, rem);
    *hi = *lo + base + (r < rem ? 1 : 0);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// one process per GPU
// ---------------------------------------------------------------------------------------------------------
struct sls_comm {
    sls_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double* d_buf = nullptr;   // send record followed by the world gathered records
    size_t cap = 0;            // doubles
};

extern "C" int sls_comm_unique_id(char* out128) {
    try {
        SLS_REQUIRE(out128, "sls_comm_unique_id: out is NULL");
        Rccl* r = rccl();
        SLS_REQUIRE(r, "sls_comm_unique_id: %s", rccl_state().why.c_str());
        ncclUniqueId id;
        SLS_NCCL(r->GetUniqueId(&id));
        static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
        std::memcpy(out128, &id, 128);
    } catch (const HipFail& f) { return f.code; }
    return SLS_OK;
}

extern "C" int sls_comm_create(sls_ctx* ctx, const char* id128, int rank, int world, sls_comm** out) {
    try {
        SLS_REQUIRE(ctx && id128 && out && world >= 1 && rank >= 0 && rank < world, "sls_comm_create: bad argument");
        Rccl* r = rccl();
        if (!r) {
            set_error("sls_comm_create: RCCL is not available in this process: %s", rccl_state().why.c_str());
            return SLS_ERR_HIP;
        }
        std::unique_lock<std::recursive_mutex> lock(ctx->mtx);
        SLS_HIP(hipSetDevice(ctx->device));
        std::unique_ptr<sls_comm> c(new sls_comm());
        c->ctx = ctx; c->rank = rank; c->world = world;
        ncclUniqueId id;
        std::memcpy(&id, id128, 128);
        SLS_NCCL(r->CommInitRank(&c->comm, world, id, rank));
        slsk::ctx_retain(ctx);
        *out = c.release();
    } catch (const HipFail& f) { return f.code; }
    return SLS_OK;
}

extern "C" int sls_comm_destroy(sls_comm* c) {
    if (!c) return SLS_OK;
    sls_ctx* ctx = c->ctx;
    (void)hipSetDevice(ctx->device);
    if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
    if (c->d_buf) (void)hipFree(c->d_buf);
    delete c;
    slsk::ctx_release(ctx);
    return SLS_OK;
}

extern "C" int sls_comm_allgather_best(sls_comm* c, double value, long index, const double* x, int D, double* value_out,
                                       long* index_out, double* x_out) {
    try {
        SLS_REQUIRE(c && x && D >= 1, "sls_comm_allgather_best: bad argument");
        sls_ctx* ctx = c->ctx;
        std::unique_lock<std::recursive_mutex> lock(ctx->mtx);
        SLS_HIP(hipSetDevice(ctx->device));
        const size_t rec = 2 + (size_t)D, need = rec * (1 + (size_t)c->world);
        if (need > c->cap) {
            if (c->d_buf) (void)hipFree(c->d_buf);
            c->d_buf = nullptr;
            SLS_HIP(hipMalloc((void**)&c->d_buf, need * sizeof(double)));
            c->cap = need;
        }
        std::vector<double> h(need);
        h[0] = value; h[1] = (double)index;            // exact below 2^53
        std::memcpy(h.data() + 2, x, sizeof(double) * (size_t)D);
        SLS_HIP(hipMemcpyAsync(c->d_buf, h.data(), rec * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        SLS_NCCL(rccl()->AllGather(c->d_buf, c->d_buf + rec, rec, ncclDouble, c->comm, ctx->stream));
        SLS_HIP(hipMemcpyAsync(h.data() + rec, c->d_buf + rec, rec * c->world * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SLS_HIP(hipStreamSynchronize(ctx->stream));
        merge_first_max(h.data() + rec, c->world, D, value_out, index_out, x_out);
    } catch (const HipFail& f) { return f.code; }
    return SLS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// one process, n devices
// ---------------------------------------------------------------------------------------------------------
struct sls_multi {
    std::vector<int> devices;
    std::vector<sls_ctx*> ctxs;
    std::vector<ncclComm_t> comms;      // empty: host merge (a device listed twice -- RCCL refuses two ranks on one GPU)
    std::vector<double*> d_buf;         // per shard: send record + n gathered records
    size_t cap = 0;
    std::string rccl_note;
};

struct sls_multi_gp {
    sls_multi* m = nullptr;
    int D = 0;
    std::vector<sls_gp*> gps;
    std::vector<char> borrowed;   // gps[r] is the caller's own handle (sls_multi_gp_create_from): not destroyed with the replicas
};

namespace {
// run f(r) for r = 0..n-1 on n host threads; first failure's code + message are re-published on the calling thread
template <class F>
int for_each_shard(int n, F f) {
    std::vector<int> rc(n, SLS_OK);
    std::vector<std::string> msg(n);
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r)
        th.emplace_back([&, r] {
            rc[r] = f(r);
            if (rc[r] != SLS_OK) msg[r] = sls_last_error();   // thread-local on the worker
        });
    for (auto& t : th) t.join();
    for (int r = 0; r < n; ++r)
        if (rc[r] != SLS_OK) {
            set_error("shard %d: %s", r, msg[r].c_str());
            return rc[r];
        }
    return SLS_OK;
}
}  // namespace

extern "C" int sls_multi_create(const int* devices, int n, sls_multi** out) {
    std::unique_ptr<sls_multi> m(new sls_multi());
    try {
        SLS_REQUIRE(devices && out && n >= 1 && n <= 64, "sls_multi_create: need 1..64 devices");
        m->devices.assign(devices, devices + n);
        m->ctxs.assign(n, nullptr);
        for (int r = 0; r < n; ++r) {
            const int rc = sls_ctx_create(devices[r], &m->ctxs[r]);
            if (rc != SLS_OK) {
                for (sls_ctx* c : m->ctxs) sls_ctx_destroy(c);
                return rc;
            }
        }
        const bool distinct = std::set<int>(m->devices.begin(), m->devices.end()).size() == (size_t)n;
        const bool want = tune_on(TUNE_MULTI_RCCL);
        if (!distinct) m->rccl_note = "host merge: a device is listed more than once (RCCL allows one rank per GPU)";
        else if (!want) m->rccl_note = "host merge: SLS_MULTI_RCCL=0";
        else if (!rccl()) m->rccl_note = "host merge: " + rccl_state().why;
        else {
            m->comms.assign(n, nullptr);
            const ncclResult_t r_ = rccl()->CommInitAll(m->comms.data(), n, m->devices.data());
            if (r_ != ncclSuccess) {
                m->rccl_note = std::string("host merge: ncclCommInitAll failed: ") + rccl()->GetErrorString(r_);
                m->comms.clear();
            } else {
                m->rccl_note = "ncclAllGather";
            }
        }
        m->d_buf.assign(n, nullptr);
        *out = m.release();
    } catch (const HipFail& f) { return f.code; }
    return SLS_OK;
}

extern "C" int sls_multi_destroy(sls_multi* m) {
    if (!m) return SLS_OK;
    for (size_t r = 0; r < m->ctxs.size(); ++r) {
        (void)hipSetDevice(m->devices[r]);
        if (r < m->comms.size() && m->comms[r]) (void)rccl()->CommDestroy(m->comms[r]);
        if (m->d_buf[r]) (void)hipFree(m->d_buf[r]);
        sls_ctx_destroy(m->ctxs[r]);
    }
    delete m;
    return SLS_OK;
}

extern "C" int sls_multi_size(const sls_multi* m) { return m ? (int)m->devices.size() : 0; }
extern "C" const char* sls_multi_exchange(const sls_multi* m) { return m ? m->rccl_note.c_str() : ""; }
extern "C" sls_ctx* sls_multi_ctx(sls_multi* m, int shard) {
    return (m && shard >= 0 && shard < (int)m->ctxs.size()) ? m->ctxs[shard] : nullptr;
}

extern "C" int sls_multi_gp_create(sls_multi* m, const double* X, int D, int N, const double* y, const double* theta, double b,
                                   int kernel, sls_multi_gp** out) {
    if (!m || !out) { set_error("sls_multi_gp_create: NULL argument"); return SLS_ERR_INVALID; }
    std::unique_ptr<sls_multi_gp> g(new sls_multi_gp());
    g->m = m; g->D = D;
    const int n = (int)m->devices.size();
    g->gps.assign(n, nullptr);
    const int rc = for_each_shard(n, [&](int r) { return sls_gp_create(m->ctxs[r], X, D, N, y, theta, b, kernel, &g->gps[r]); });
    if (rc != SLS_OK) {
        for (sls_gp* h : g->gps) sls_gp_destroy(h);
        return rc;
    }
    *out = g.release();
    return SLS_OK;
}

// Replicas of an EXISTING fitted handle: the shard on the primary's own device (the first one, if the device is listed several
// times) IS the primary -- its fit is not repeated --, the other shards are fitted from the primary's data (X, y, theta, b, kernel,
// sigma mode), read back from its device once.  The primary must outlive the replicas and must not be refitted / grown while
// they are in use (a grown primary needs new replicas).
extern "C" int sls_multi_gp_create_from(sls_multi* m, sls_gp* primary, sls_multi_gp** out) {
    if (!m || !primary || !out) { set_error("sls_multi_gp_create_from: NULL argument"); return SLS_ERR_INVALID; }
    int D = 0, N = 0, kernel = 0, sigma_mode = 0, pdev = 0;
    double b = 0.0;
    std::vector<double> X, y, theta;
    int rc = slsk::gp_export_inputs(primary, &D, &N, &kernel, &sigma_mode, &pdev, &b, &X, &y, &theta);
    if (rc != SLS_OK) return rc;
    std::unique_ptr<sls_multi_gp> g(new sls_multi_gp());
    g->m = m; g->D = D;
    const int n = (int)m->devices.size();
    g->gps.assign(n, nullptr);
    g->borrowed.assign(n, 0);
    for (int r = 0; r < n; ++r)
        if (m->devices[r] == pdev) {
            g->gps[r] = primary;
            g->borrowed[r] = 1;
            break;
        }
    rc = for_each_shard(n, [&](int r) {
        if (g->borrowed[r]) return (int)SLS_OK;
        const int rr = sls_gp_create(m->ctxs[r], X.data(), D, N, y.data(), theta.data(), b, kernel, &g->gps[r]);
        if (rr != SLS_OK) return rr;
        return sls_gp_set_sigma_mode(g->gps[r], sigma_mode);
    });
    if (rc != SLS_OK) {
        for (int r = 0; r < n; ++r)
            if (!g->borrowed[r]) sls_gp_destroy(g->gps[r]);
        return rc;
    }
    *out = g.release();
    return SLS_OK;
}

extern "C" int sls_multi_gp_destroy(sls_multi_gp* g) {
    if (!g) return SLS_OK;
    for (size_t r = 0; r < g->gps.size(); ++r) {
        if (!g->borrowed.empty() && g->borrowed[r]) continue;
        (void)hipSetDevice(g->m->devices[r]);
        sls_gp_destroy(g->gps[r]);
    }
    delete g;
    return SLS_OK;
}

extern "C" sls_gp* sls_multi_gp_shard(sls_multi_gp* g, int shard) {
    return (g && shard >= 0 && shard < (int)g->gps.size()) ? g->gps[shard] : nullptr;
}

extern "C" int sls_multi_acq_maximize(sls_multi_gp* g, int acq_type, double ucb_h, const double* starts, int S, int n_local,
                                      const sls_lbfgs_opts* opts, double* x_out, double* val_out, long* idx_out,
                                      long* evals_issued) {
    try {
        SLS_REQUIRE(g && starts && S >= 1, "sls_multi_acq_maximize: bad argument");
        sls_multi* m = g->m;
        const int n = (int)m->devices.size(), D = g->D;
        const size_t rec = 2 + (size_t)D;
        std::vector<double> local(rec * n, 0.0);
        std::vector<long> issued(n, 0);
        // every shard: its contiguous slice of the starts with the global index offset, then its record on its device
        int rc = for_each_shard(n, [&](int r) {
            int lo, hi;
            shard_range(S, r, n, &lo, &hi);
            double* p = local.data() + rec * r;
            p[0] = 0.0; p[1] = -1.0;                      // "no starts" marker
            if (hi <= lo) return (int)SLS_OK;
            long idx = 0;
            const int rcc = sls_acq_maximize(g->gps[r], acq_type, ucb_h, starts + (size_t)lo * D, hi - lo, n_local, opts, lo, p + 2,
                                             p, &idx, nullptr, nullptr);
            if (rcc != SLS_OK) return rcc;
            p[1] = (double)idx;
            (void)sls_acq_last_stats(g->gps[r], &issued[r], nullptr, nullptr, nullptr);
            return (int)SLS_OK;
        });
        if (rc != SLS_OK) return rc;
        if (evals_issued) {
            *evals_issued = 0;
            for (long v : issued) *evals_issued += v;
        }
        std::vector<double> gathered(rec * n);
        if (!m->comms.empty()) {
            // the single exchange of the step: one grouped ncclAllGather of (value, index, x[D]) over the n devices
            if (rec * (1 + n) > m->cap) {
                for (int r = 0; r < n; ++r) {
                    SLS_HIP(hipSetDevice(m->devices[r]));
                    if (m->d_buf[r]) (void)hipFree(m->d_buf[r]);
                    m->d_buf[r] = nullptr;
                    SLS_HIP(hipMalloc((void**)&m->d_buf[r], rec * (1 + n) * sizeof(double)));
                }
                m->cap = rec * (1 + n);
            }
            for (int r = 0; r < n; ++r) {
                SLS_HIP(hipSetDevice(m->devices[r]));
                SLS_HIP(hipMemcpyAsync(m->d_buf[r], local.data() + rec * r, rec * sizeof(double), hipMemcpyHostToDevice,
                                       m->ctxs[r]->stream));
            }
            SLS_NCCL(rccl()->GroupStart());
            for (int r = 0; r < n; ++r)
                SLS_NCCL(rccl()->AllGather(m->d_buf[r], m->d_buf[r] + rec, rec, ncclDouble, m->comms[r], m->ctxs[r]->stream));
            SLS_NCCL(rccl()->GroupEnd());
            SLS_HIP(hipSetDevice(m->devices[0]));
            SLS_HIP(hipMemcpyAsync(gathered.data(), m->d_buf[0] + rec, rec * n * sizeof(double), hipMemcpyDeviceToHost,
                                   m->ctxs[0]->stream));
            for (int r = 0; r < n; ++r) {
                SLS_HIP(hipSetDevice(m->devices[r]));
                SLS_HIP(hipStreamSynchronize(m->ctxs[r]->stream));
            }
        } else {
            gathered = local;
        }
        merge_first_max(gathered.data(), n, D, val_out, idx_out, x_out);
    } catch (const HipFail& f) { return f.code; }
    return SLS_OK;
}

extern "C" int sls_multi_gp_predict(sls_multi_gp* g, const double* Xs, int M, double* mu, double* sigma) {
    try {
        SLS_REQUIRE(g && Xs && M >= 0, "sls_multi_gp_predict: bad argument");
        const int n = (int)g->m->devices.size(), D = g->D;
        return for_each_shard(n, [&](int r) {
            int lo, hi;
            shard_range(M, r, n, &lo, &hi);
            if (hi <= lo) return (int)SLS_OK;
            return sls_gp_predict(g->gps[r], Xs + (size_t)lo * D, hi - lo, mu ? mu + lo : nullptr, sigma ? sigma + lo : nullptr);
        });
    } catch (const HipFail& f) { return f.code; }
}

// ---- MAP objective over several devices: the B independent points of one DIRECT iteration dealt round-robin ----------------
// The N^3 factorisation of one evaluation does not shard; the evaluations of a batch do: point k runs on shard
// k mod n, each shard one sls_gp_nll_batch over its points, the values come back through host memory (no collective: B doubles).
// Every point is evaluated by the same kernels as on one device: the values are bit-identical to sls_gp_nll_batch.
struct sls_multi_nll {
    sls_multi* m = nullptr;
    int D = 0;
    std::vector<sls_nll*> hs;
};

extern "C" int sls_multi_nll_create(sls_multi* m, const double* X, int D, int N, int kernel, sls_multi_nll** out) {
    if (!m || !out) { set_error("sls_multi_nll_create: NULL argument"); return SLS_ERR_INVALID; }
    std::unique_ptr<sls_multi_nll> g(new sls_multi_nll());
    g->m = m; g->D = D;
    const int n = (int)m->devices.size();
    g->hs.assign(n, nullptr);
    const int rc = for_each_shard(n, [&](int r) { return sls_nll_create(m->ctxs[r], X, D, N, kernel, &g->hs[r]); });
    if (rc != SLS_OK) {
        for (sls_nll* h : g->hs) sls_nll_destroy(h);
        return rc;
    }
    *out = g.release();
    return SLS_OK;
}

extern "C" int sls_multi_nll_destroy(sls_multi_nll* g) {
    if (!g) return SLS_OK;
    for (size_t r = 0; r < g->hs.size(); ++r) {
        (void)hipSetDevice(g->m->devices[r]);
        sls_nll_destroy(g->hs[r]);
    }
    delete g;
    return SLS_OK;
}

extern "C" int sls_multi_gp_nll_batch(sls_multi_nll* g, const double* y, const double* xs, int B, double* values) {
    try {
        SLS_REQUIRE(g && y && xs && values && B >= 0, "sls_multi_gp_nll_batch: bad argument");
        const int n = (int)g->hs.size(), W = g->D + 2;
        return for_each_shard(n, [&](int r) {
            std::vector<double> mine, vals;
            for (int k = r; k < B; k += n) mine.insert(mine.end(), xs + (size_t)k * W, xs + (size_t)(k + 1) * W);
            const int nb = (int)(mine.size() / W);
            if (nb == 0) return (int)SLS_OK;
            vals.resize(nb);
            const int rc = sls_gp_nll_batch(g->hs[r], y, mine.data(), nb, vals.data());
            if (rc != SLS_OK) return rc;
            for (int q = 0; q < nb; ++q) values[r + (size_t)q * n] = vals[q];
            return (int)SLS_OK;
        });
    } catch (const HipFail& f) { return f.code; }
}
