// common.h -- shared host/device helpers for the gfx950 kernels. Internal (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/innr_hip.h"

namespace innr {

// ---- error plumbing --------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define INNR_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::innr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                              __LINE__);                                                      \
            return (_e == hipErrorOutOfMemory) ? INNR_E_OOM : INNR_E_HIP;                     \
        }                                                                                     \
    } while (0)

#define INNR_TRY(expr)                  \
    do {                                \
        innr_status _s = (expr);        \
        if (_s != INNR_OK) return _s;   \
    } while (0)

// ---- total order on f32 (core::f32::total_cmp) --------------------------------------------------
// ord(x): uint32, monotone increasing in total_cmp order (-NaN < -inf < ... < -0 < +0 < ... < +inf < +NaN)
__host__ __device__ __forceinline__ uint32_t f32_ord(float x) {
    uint32_t b = __builtin_bit_cast(uint32_t, x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord_f32(uint32_t o) {
    uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __builtin_bit_cast(float, b);
}

// Candidate = 64-bit composite, larger = better:  [ pref(32) | ~idx(32) ]
//   pref = ord(score) for similarities (DOT/COSINE), ~ord(dist) for distances (L2SQ);
//   ~idx makes the lower index win among equal scores (the reference's stable-sort tie rule).
// Composite 0 is reserved for "empty slot" (needs idx == 0xFFFFFFFF, which N < 2^32-1 excludes).
__host__ __device__ __forceinline__ uint64_t cand_make(uint32_t pref, uint32_t idx) {
    return ((uint64_t)pref << 32) | (uint64_t)(~idx);
}
__host__ __device__ __forceinline__ uint32_t cand_pref(uint64_t c) { return (uint32_t)(c >> 32); }
__host__ __device__ __forceinline__ uint32_t cand_idx(uint64_t c) { return ~(uint32_t)c; }

template <bool SMALLER_IS_BETTER>
__host__ __device__ __forceinline__ uint32_t score_pref(float s) {
    uint32_t o = f32_ord(s);
    return SMALLER_IS_BETTER ? ~o : o;
}
__host__ __device__ __forceinline__ float pref_score(uint32_t pref, bool smaller_is_better) {
    return ord_f32(smaller_is_better ? ~pref : pref);
}

// lib.rs:178
#define INNR_NORM_EPSILON 1e-9f

// ---- candidate-list geometry (shared by the scan and GEMM kernels and the select kernel) -------
// A producer (one wave) owns, per query, a list of CAP composites in global scratch. It appends every
// score that passes the list's threshold and compacts the list to its best KP when fewer than BURST free
// slots remain (BURST = most appends one producer step can make for one query).
constexpr int kBurst = 256;
__host__ __device__ constexpr int cand_cap(int KP) { return 4 * KP + kBurst; }

static inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

}  // namespace innr

// ---- exact (reference-order) f32 arithmetic ------------------------------------------------------
// The reference's portable loops are fl(acc + fl(a*b)): two roundings, never an FMA. HIP's __fmul_rn /
// __fadd_rn are plain operators (contractable) and __fsqrt_rn is the NATIVE (approximate) sqrt, so none
// of them is used. These helpers pin the semantics; the library is also compiled with -ffp-contract=off.
// sqrtf and '/' lower to the correctly rounded expansions (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt), verified in the ISA: v_div_scale/fmas/fixup, v_sqrt + fixup.
namespace innr {
namespace ex {
__device__ __forceinline__ float mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
// a - b where a NaN in b keeps its sign. gfx950's v_sub_f32 is a + (-b): it flips the sign of a propagated NaN,
// while the host ISAs the reference runs on (x86 subss, aarch64 fsub) return the NaN operand unchanged -- and
// total_cmp orders -NaN first / +NaN last, so the sign decides where a NaN distance lands in the result.
__device__ __forceinline__ float sub_keepnan(float a, float b) {
#pragma clang fp contract(off)
    const float d = a - b;
    return (b != b) ? b : d;
}
// acc + a*b with two roundings
__device__ __forceinline__ float mad2(float acc, float a, float b) {
#pragma clang fp contract(off)
    float p = a * b;
    return acc + p;
}
__device__ __forceinline__ float div(float a, float b) { return a / b; }
__device__ __forceinline__ float sqrt(float a) { return __builtin_sqrtf(a); }
}  // namespace ex
}  // namespace innr
