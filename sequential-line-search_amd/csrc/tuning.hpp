// Run-time switches of the device library (A/B runs and test hooks; every default is the measured best).
#pragma once

namespace slsk {
// ONE table for every SLS_* environment variable the device library understands, parsed once (first use) instead of a getenv per
// call site and call.  sls_tuning_reload() (C ABI; the Python binding's tuning_reload()) re-reads the environment: tests switch
// paths inside one process.  tune(K, dflt): the variable's integer value, dflt when it is not set.  DESIGN.md lists the meanings.
#define SLS_TUNING_KEYS(X)                                                                                                      \
    X(POOL_MB) X(POTRF_LOOKAHEAD) X(FIT_SMALL) X(TRI_PREDICT) X(WAVE_PATH) X(IO_STAGE) X(EVAL_ZEROCOPY) X(WAVE_TRACE) X(COMPACT) \
    X(NLL_SMALL) X(SMALL_ZEROCOPY) X(NLL_BATCH) X(MAP_DEVICE) X(MAP_TRACE) X(SMALL_XLDS) X(MULTI_RCCL) X(PERSIST)                \
    X(ACQ_WG_PER_CU) X(GATE_PHASE) X(TAIL_SPLIT) X(LBFGS_REG) X(TRI_WG_PER_CU) X(POTRF_MODE) X(POTRF_DNBO) X(POTRF_NBO)          \
    X(LAUUM_N64) X(POTRI_FUSED) X(POTRF_STREAM) X(POTRF_SPLIT) X(POTRI_W1) X(POTRF_TIMEOUT_TICKS) X(POTRF_DNEAR) X(POTRI_PLAST)  \
    X(POTRI_CX) X(POTRI_CK) X(WAVE_STAGE) X(WAVE_COOP) X(GRAD_SPLIT_TILES) X(EVAL_SLOTS) X(GATE_EVERY)
enum TuneKey {
#define SLS_TK(name) TUNE_##name,
    SLS_TUNING_KEYS(SLS_TK)
#undef SLS_TK
    TUNE_COUNT
};
struct SlsTuning {
    bool has[TUNE_COUNT];
    long val[TUNE_COUNT];
};
const SlsTuning& tuning();   // capi.hip
void tuning_reload();
inline long tune(TuneKey k, long dflt) {
    const SlsTuning& t = tuning();
    return t.has[k] ? t.val[k] : dflt;
}
inline bool tune_on(TuneKey k, bool dflt = true) { return tune(k, dflt ? 1 : 0) != 0; }
inline bool tune_set(TuneKey k) { return tuning().has[k]; }

}  // namespace slsk
