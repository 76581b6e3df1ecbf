// Wave-wide sum / maximum with every lane receiving the result (shared by the one-wavefront-per-start maximiser and the
// one-workgroup MAP kernels): four DPP steps inside each 16-lane row (quad swaps, half-row and row mirrors: both partners of a
// step combine the same two values, so all lanes of a row hold the same bits), then two steps across the rows on gfx950's
// v_permlane16_swap / v_permlane32_swap (rows 0|1 and 2|3, then the two halves of the wave) -- all VALU, no LDS crossbar.
// The plain butterfly is six 64-bit __shfl_xor = twelve ds_bpermute_b32; the optimisers run 15-25 DEPENDENT reductions per
// iteration on ONE wave (measured, C3 local phase: 4.0 us of a 17.5 us evaluation round).
// A variant that fetched the four row totals with v_readlane (results in SGPRs) faulted on the device in one instantiation
// (round 4) and was dropped; -DSLS_SHFL_REDUCE builds the plain six-step butterfly.
#pragma once
#include <hip/hip_runtime.h>

namespace slsk {

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    const long v = __builtin_bit_cast(long, x);
    const unsigned lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, (long)(((unsigned long)hi << 32) | lo));
}
// W32 = false: a = rows (0, 0, 2, 2), b = rows (1, 1, 3, 3) of v;  W32 = true: a = lower half twice, b = upper half twice
template <bool W32>
__device__ __forceinline__ void row_swap(double v, double& a, double& b) {
    const long x = __builtin_bit_cast(long, v);
    const unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
    unsigned alo, blo, ahi, bhi;
    if (W32) {
        const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        alo = rl[0]; blo = rl[1]; ahi = rh[0]; bhi = rh[1];
    } else {
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        alo = rl[0]; blo = rl[1]; ahi = rh[0]; bhi = rh[1];
    }
    a = __builtin_bit_cast(double, (long)(((unsigned long)ahi << 32) | alo));
    b = __builtin_bit_cast(double, (long)(((unsigned long)bhi << 32) | blo));
}
// sum over the 16 lanes of each row, every lane of the row receiving it
__device__ __forceinline__ double row_sum(double v) {
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#ifdef SLS_SHFL_REDUCE
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
#else
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    double a, b;
    row_swap<false>(v, a, b);
    v = a + b;
    row_swap<true>(v, a, b);
    return a + b;
#endif
}
__device__ __forceinline__ double wave_max(double v) {
#ifdef SLS_SHFL_REDUCE
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
#else
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    double a, b;
    row_swap<false>(v, a, b);
    v = fmax(a, b);
    row_swap<true>(v, a, b);
    return fmax(a, b);
#endif
}

}  // namespace slsk
