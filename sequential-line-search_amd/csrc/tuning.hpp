// Run-time switches of the device library (A/B runs and test hooks; every default is the measured best).
#pragma once
#include <atomic>
#include <cstdlib>
#include <mutex>

namespace slsk {
// ONE table for every SLS_* environment variable the device library understands, parsed once (first use) instead of a getenv per
// call site and call.  sls_tuning_reload() (C ABI; the Python binding's tuning_reload()) re-reads the environment: tests switch
// paths inside one process.  tune(K, dflt): the variable's integer value, dflt when it is not set.  DESIGN.md lists the meanings.
#define SLS_TUNING_KEYS(X)                                                                                                      \
    X(POOL_MB) X(FIT_SMALL) X(TRI_PREDICT) X(WAVE_PATH) X(IO_STAGE) X(EVAL_ZEROCOPY) X(WAVE_TRACE) X(COMPACT) \
    X(NLL_SMALL) X(SMALL_ZEROCOPY) X(NLL_BATCH) X(MAP_DEVICE) X(MAP_TRACE) X(SMALL_XLDS) X(MULTI_RCCL) X(PERSIST)                \
    X(ACQ_WG_PER_CU) X(GATE_PHASE) X(TAIL_SPLIT) X(LBFGS_REG) X(TRI_WG_PER_CU) X(POTRF_MODE) X(POTRF_DNBO) X(POTRF_NBO)          \
    X(LAUUM_N64) X(TRTRI_NARROW) X(POTRI_FUSED) X(POTRF_STREAM) X(POTRF_SPLIT) X(POTRI_W1) X(POTRF_TIMEOUT_TICKS) X(POTRF_DNEAR) X(POTRI_PLAST)  \
    X(POTRI_CX) X(POTRI_CK) X(WAVE_STAGE) X(WAVE_COOP) X(GRAD_SPLIT_TILES) X(EVAL_SLOTS) X(GATE_EVERY) X(POTRF_FUSE_SYRK) X(POTRI_POOL) X(POTRF_POOL) X(POTRI_POOL_KEEP) X(POTRI_POOL_NEAR) X(POTRI_POOL_NEAR_W)
enum TuneKey {
#define SLS_TK(name) TUNE_##name,
    SLS_TUNING_KEYS(SLS_TK)
#undef SLS_TK
    TUNE_COUNT
};
// One word per key: bit 0 = "set", the value above it.  Atomic (relaxed): sls_tuning_reload() is a public entry point and may run
// while other host threads -- the lock-free evaluation slots, the multi-GPU workers -- read the table; a reader sees the old or the
// new setting of a key, never a torn one.
struct SlsTuning {
    std::atomic<long long> word[TUNE_COUNT];
};
namespace tuning_detail {
inline SlsTuning& table() {
    static SlsTuning t;
    return t;
}
inline void parse() {
    static const char* const names[TUNE_COUNT] = {
#define SLS_TK(name) "SLS_" #name,
        SLS_TUNING_KEYS(SLS_TK)
#undef SLS_TK
    };
    SlsTuning& t = table();
    for (int k = 0; k < TUNE_COUNT; ++k) {
        const char* v = getenv(names[k]);
        t.word[k].store(v ? (((long long)atol(v)) << 1) | 1 : 0, std::memory_order_relaxed);
    }
}
}  // namespace tuning_detail
inline const SlsTuning& tuning() {
    static std::once_flag once;
    std::call_once(once, tuning_detail::parse);
    return tuning_detail::table();
}
inline void tuning_reload() {
    (void)tuning();
    tuning_detail::parse();
}
inline long tune(TuneKey k, long dflt) {
    const long long w = tuning().word[k].load(std::memory_order_relaxed);
    return (w & 1) ? (long)(w >> 1) : dflt;
}
inline bool tune_on(TuneKey k, bool dflt = true) { return tune(k, dflt ? 1 : 0) != 0; }
inline bool tune_set(TuneKey k) { return (tuning().word[k].load(std::memory_order_relaxed) & 1) != 0; }

}  // namespace slsk
