// Common macros for the host/device arithmetic headers of the EL PASSO HIP path.
//
// The headers under csrc/elp/ are written once and compiled twice:
//   * by hipcc for gfx950 (the product: kernels in ../kernels.hip behind the C-ABI of include/elpasso.h)
//   * by g++ for the host, ONLY by tests/host_twin (unit-testing the same formulas in a container
//     that has no GPU).  The host build is test infrastructure and is never linked into the product.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ELP_HD __host__ __device__
#else
#define ELP_HD
#endif

#define ELP_INL ELP_HD inline __attribute__((always_inline))

// "Heavy" routines are real (non-inlined) functions on the device: a pairing inlined into one kernel would be
// several MB of straight-line code; a call hierarchy (fp_mul <- fp2_mul <- fp6_mul <- fp12_mul ...) keeps the
// hot loop inside the instruction cache.  On the host the attribute does not matter.
// ELP_FPMUL: linkage of fp_mul/fp_sqr themselves.  -DELP_FPMUL_INLINE=1 inlines the limb code into the Fp2-level leaf
// functions (no callee-saved register traffic around every product); 0 keeps them as calls.
#ifndef ELP_FPMUL_INLINE
#define ELP_FPMUL_INLINE 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ELP_HEAVY static ELP_HD __attribute__((noinline))
#define ELP_UNROLL _Pragma("unroll")
#define ELP_NOUNROLL _Pragma("nounroll")
#else
#define ELP_HEAVY ELP_HD inline
#define ELP_UNROLL
#define ELP_NOUNROLL
#endif

#if defined(__HIP_DEVICE_COMPILE__) && !ELP_FPMUL_INLINE
#define ELP_FPMUL ELP_HEAVY
#else
#define ELP_FPMUL ELP_INL
#endif
// ELP_FP2: linkage of fp2_mul / fp2_sqr.  -DELP_FP2_INLINE=1 inlines them into the Fp6-level and point-formula functions,
// which then keep their Fp2 temporaries in registers instead of passing them through private memory.
#ifndef ELP_FP2_INLINE
#define ELP_FP2_INLINE 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !ELP_FP2_INLINE
#define ELP_FP2 ELP_HEAVY
#else
#define ELP_FP2 ELP_INL
#endif

// ELP_FP6: linkage of the Fp6-level routines (fp6_mul, fp6_sqr, fp6_mul_by_01, fp6_mul_by_fp2).  -DELP_FP6_INLINE=1 makes the
// Fp12-level routines (fp12_mul, fp12_sqr, sparse line product) single leaf functions.
#ifndef ELP_FP6_INLINE
#define ELP_FP6_INLINE 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !ELP_FP6_INLINE
#define ELP_FP6 ELP_HEAVY
#else
#define ELP_FP6 ELP_INL
#endif

namespace elp {
typedef uint32_t u32;
typedef uint64_t u64;
// Per-lane "hot slot": ELP_HOT_WORDS 32-bit words of LDS that a kernel hands to the device routines (KeyCtx::hot) for the one
// accumulator that is read and written by every step of a long loop (Miller value, exponentiation accumulator, point accumulator).
// Routines take it through generic references, so the same code runs on private memory when the slot is absent (host twin, hot == 0)
// or too small for the type.  4 resident waves x 64 lanes x 432 B = 108 KB of the CU's 160 KB.
constexpr int ELP_HOT_WORDS = 108;
template <class T>
ELP_INL T* hot_as(u32* hot) {
  return (hot != nullptr && sizeof(T) <= (size_t)ELP_HOT_WORDS * 4) ? reinterpret_cast<T*>(hot) : nullptr;
}
// scalar-field (Fr) Montgomery parameters of a curve; specialised in params_<curve>.h
template <class C>
struct FrOf;
}  // namespace elp
