// Shared host-side plumbing of libsls_hip: context, error handling, device buffers, per-kernel event timing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sls_hip.h"
#include "kernels.hpp"

namespace slsk {

void set_error(const char* fmt, ...);
// true: the persistent factorisation gave up (a bounded device-side wait expired, e.g. another stream kept CUs busy so
// that its workgroups were not all resident).  The context falls back to the multi-launch schedule for good and the
// caller repeats the factorisation once; a second failure throws HipFail{SLS_ERR_HIP}.
bool potrf_gave_up(sls_ctx* c, int abort_flag, int attempt);
// what a fitted handle was created from (capi_multi.hip: replicas of an existing handle)
int gp_export_inputs(sls_gp* g, int* D, int* N, int* kernel, int* sigma_mode, int* device, double* b, std::vector<double>* X,
                     std::vector<double>* y, std::vector<double>* theta);

struct HipFail {
    int code;
};

#define SLS_HIP(x)                                                                                 \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            slsk::set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            throw slsk::HipFail{SLS_ERR_HIP};                                                      \
        }                                                                                          \
    } while (0)

#define SLS_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            slsk::set_error(__VA_ARGS__);      \
            throw slsk::HipFail{SLS_ERR_INVALID}; \
        }                                      \
    } while (0)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Per-device cache of freed device blocks, keyed by exact size: hipMalloc / hipFree cost 50-300 us each and a GP handle owns
// ~30 buffers, which made CREATING a handle at N = 2048 take longer than FITTING it (BASELINE config C2).  A regressor that
// is rebuilt with the same shapes (every step of the reference's optimisers) gets its blocks back without touching the
// driver.  At most SLS_POOL_MB (default 16384) MB stay cached per device; beyond that blocks go back to the driver.
void note_entry();                           // every C-ABI entry point: device work may be queued from here on
void* pool_alloc(size_t bytes);              // throws HipFail on failure
void pool_free(void* p, size_t bytes, bool in_flight = false);   // in_flight: work queued in THIS entry may still use the block
void pool_trim(int device);                  // return every cached block of `device` to the driver

// Owning device buffer of doubles (or raw bytes).
struct DBuf {
    double* p = nullptr;
    size_t n = 0;  // doubles
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf() { release(); }
    void release(bool in_flight = false) {
        if (p) pool_free(p, n * sizeof(double), in_flight);
        p = nullptr;
        n = 0;
    }
    void ensure(size_t doubles) {
        if (doubles <= n && p) return;
        release(true);   // a regrow in the middle of an entry point: kernels queued a moment ago may still read the old block
        p = static_cast<double*>(pool_alloc(doubles * sizeof(double)));
        n = doubles;
    }
};

struct ProfEntry {
    double ms = 0;
    long launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace slsk

struct sls_ctx {
    // One stream, one info word, one profiling table per context: every entry point that touches the device takes this
    // lock, so a context (and all handles created from it) may be used from several host threads, one call at a time.
    std::recursive_mutex mtx;
    // handles (sls_gp, sls_nll, sls_comm) created from this context and still alive.  sls_ctx_destroy with live handles only marks
    // the context: the last handle to go frees it (garbage-collected bindings destroy context and handles in no particular order,
    // and a handle's destructor takes this lock).
    int live_handles = 0;
    bool destroy_requested = false;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int cand_chunk = 16384;
    bool prof_on = false;
    std::map<std::string, slsk::ProfEntry> prof;
    std::vector<hipEvent_t> event_pool;
    // scratch
    slsk::DBuf scratch;   // generic host<->device staging
    int* d_info = nullptr;  // device int[4]: potrf info etc.
    bool potrf_persistent_ok = true;     // cleared when a single-launch factorisation gave up (bounded wait expired)
    int potrf_rearm = 0;                 // fits left on the multi-launch schedule before the single-launch form is tried again
    int potrf_rearm_next = 16;           // ... and how many it will be after the next give-up (16, 64, 256, ... 4096)
    void potrf_tick_rearm();             // once per fit, BEFORE the factorisation is enqueued
    long potrf_fallbacks = 0;            // how often a single-launch factorisation gave up (sls_prof_get("potrf_fallbacks"))
    slsk::DBuf potrf_df;                 // flag tables of the dataflow form (grown on demand)
    int* potrf_df_sync(int Np);          // nullptr: single-launch form switched off for this context (no side effects)
    bool potrf_df_available(int Np) const;   // the same question without allocating the flag tables
    bool potrf_single_pending = false;   // a single-launch factorisation was handed its flag tables since the last verdict
    // streams + device-mapped page-locked blocks lent to concurrent small evaluations (capi.hip: eval_in_slot), created on demand,
    // kept for the life of the context
    struct EvalSlot {
        hipStream_t stream = nullptr;
        double* host = nullptr;
        double* dev = nullptr;
        size_t bytes = 0;
        bool busy = false;
    };
    std::mutex slot_mtx;
    std::condition_variable slot_cv;
    std::vector<std::unique_ptr<EvalSlot>> slots;
    static constexpr int MAX_SLOTS = 16;   // concurrent small evaluations per context; further callers wait for a free slot

    // page-locked host blocks (result blocks the small-problem kernels write directly, staging for uploads) handed out to the
    // handles of this context and taken back when a handle dies: hipHostMalloc + hipHostFree cost ~260 us per block (round 4, C3:
    // two blocks per SubmitFeedbackData were 0.5 ms of its 4 ms)
    struct HostBlock {
        void* p;
        size_t bytes;
        bool mapped;
    };
    std::vector<HostBlock> host_free;
    void* host_take(size_t bytes, bool mapped, size_t* got);
    void host_give(void* p, size_t bytes, bool mapped);

    hipEvent_t get_event();
    void prof_begin(const char* name, hipEvent_t& e0);
    void prof_end(const char* name, hipEvent_t e0);
    void prof_collect();
};

namespace slsk {
void ctx_retain(sls_ctx* c);     // a handle was created (caller holds c->mtx)
void ctx_release(sls_ctx* c);    // a handle is gone (caller does NOT hold c->mtx: this may free the context)
// RAII kernel-timing scope: records HIP events on the context's stream when profiling is enabled.
struct ProfScope {
    sls_ctx* c;
    const char* name;
    hipEvent_t e0 = nullptr;
    ProfScope(sls_ctx* c_, const char* n) : c(c_), name(n) {
        if (c->prof_on) c->prof_begin(name, e0);
    }
    void rename(const char* n) { name = n; }   // the launch behind the scope turned out to be something else (fused form declined)
    ~ProfScope() {
        if (c->prof_on) c->prof_end(name, e0);
    }
};
}  // namespace slsk
