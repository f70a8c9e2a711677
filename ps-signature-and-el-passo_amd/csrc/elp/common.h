// Common macros for the host/device arithmetic headers of the EL PASSO HIP path.
//
// The headers under csrc/elp/ are written once and compiled twice:
//   * by hipcc for gfx950 (the product: kernels in ../elpasso_impl.h, instantiated per curve and layout by ../elpasso_*.hip, behind the
//     C-ABI of include/elpasso.h)
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

// -DELP_NO_FORCE_INLINE (host builds of the test twins under sanitizers only): the routines stay ordinary inline functions, so that the instrumented build is made of
// many small functions instead of a few with 10^5 instructions (which neither g++ nor clang finish instrumenting in half an hour)
#if defined(ELP_NO_FORCE_INLINE) && !defined(__HIPCC__)
#define ELP_INL inline
#else
#define ELP_INL ELP_HD inline __attribute__((always_inline))
#endif

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

// A device function that is not a kernel and makes no call is a LEAF: it never saves its return address s[30:31], and the branch-relaxation pass of ROCm 7.2's LLVM expands
// the far branches of a leaf longer than 128 KB through exactly that register pair (profiles/r05_bls_fault.md).  ELP_NONLEAF() at the top of a routine that may grow past
// 128 KB in some build (-DELP_NONLEAF_GUARD=1) makes it a caller -- one empty call -- so that the prologue saves the pair.
#ifndef ELP_NONLEAF_GUARD
#define ELP_NONLEAF_GUARD 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && ELP_NONLEAF_GUARD
namespace elp {
static __device__ __attribute__((noinline)) void elp_nonleaf_anchor() { asm volatile("" ::: "memory"); }
}
#define ELP_NONLEAF() ::elp::elp_nonleaf_anchor()
#else
#define ELP_NONLEAF() ((void)0)
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

// -DELP_BISECT=k (experiments only: tools/probes/bls_fault_repro.sh): the PS-verification path of a translation unit returns early after step k, so that a run-time
// fault of a non-default build can be pinned to the first step whose code it needs.  Undefined in every product build.
#if defined(ELP_BISECT) && defined(__HIP_DEVICE_COMPILE__)
#define ELP_BISECT_AT(k) (ELP_BISECT == (k))
#else
#define ELP_BISECT_AT(k) false
#endif

namespace elp {
typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------------------------------------------
// Lane pairs.  A curve-traits class wrapped as Paired<B> selects the TWO-LANES-PER-ITEM layout of the Fp2 tower: lanes 2i and
// 2i+1 of a wave work on the same item, the even lane holds the real component of every Fp2 value and the odd lane the imaginary one
// (struct Fp2<C> then has ONE Fp member).  Everything built from Fp2 -- Fp6, Fp12, G2, Miller loop, final exponentiation -- keeps
// half of its state per lane (an Fp12 value is 54 registers instead of 108 for BN254), so the verification kernels fit 256 registers
// and run two waves per SIMD; an Fp2 product is one fp_mul_pair per lane after a DPP exchange of the operands with the neighbouring
// lane (v_mov_b32 quad_perm:[1,0,3,2], no LDS).  Base-field (G1) work does not split this way; the two lanes take different G1 jobs
// of the item instead (pipeline.h).  Rules for paired code: every branch whose body exchanges data must be taken by both lanes of a
// pair (conditions computed from Fp2 values go through fp2_is_zero_exact & co., which agree on both lanes by construction).
template <class B>
struct Paired : B {};
template <class C>
struct PairInfo {
  static constexpr bool paired = false;
  typedef C Base;
};
template <class B>
struct PairInfo<Paired<B>> {
  static constexpr bool paired = true;
  typedef B Base;
};
template <class C>
ELP_HD constexpr bool is_paired() { return PairInfo<C>::paired; }

#if defined(__HIP_DEVICE_COMPILE__)
ELP_INL bool pair_odd() { return (threadIdx.x & 1u) != 0; }             // workgroups are one-dimensional with an even size
// bound_ctrl = true: every source lane of quad_perm is inside the row, and no "old" value has to be preloaded (one instruction instead of two)
ELP_INL int32_t pair_swap_i32(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true); }
#else
// Host twin (tests only): the two lanes of a pair are two threads; the twin installs the hook that swaps a buffer with the partner
// thread's (a rendezvous, so a divergence between the lanes of a pair shows up as a hang / mismatch in the CPU tests).
inline thread_local int elp_pair_parity = 0;
inline void (*elp_pair_exchange_hook)(void* buf, size_t bytes) = nullptr;
inline bool pair_odd() { return elp_pair_parity != 0; }
inline int32_t pair_swap_i32(int32_t v) {
  elp_pair_exchange_hook(&v, sizeof v);
  return v;
}
#endif
// both lanes always execute the exchange (a lane that skipped it would leave its partner reading an inactive lane on the device and
// desynchronise the rendezvous of the host twin)
ELP_INL bool pair_and(bool b) {   // true iff true on both lanes of the pair
  const int o = pair_swap_i32(b ? 1 : 0);
  return b & (o != 0);
}
ELP_INL bool pair_or(bool b) {
  const int o = pair_swap_i32(b ? 1 : 0);
  return b | (o != 0);
}
// Per-lane "hot slot": ELP_HOT_WORDS 32-bit words of LDS that a kernel hands to the device routines (KeyCtx::hot) for the one
// accumulator that is read and written by every step of a long loop (Miller value, exponentiation accumulator, point accumulator).
// Routines take it through generic references, so the same code runs on private memory when the slot is absent (host twin, hot == 0)
// or too small for the type.  4 resident waves x 64 lanes x 432 B = 108 KB of the CU's 160 KB.
// Paired kernels run 8 waves per CU with half-size values: 54 words (216 B) per lane, 110 KB per CU.
constexpr int ELP_HOT_WORDS = 108;
#ifndef ELP_HOT_WORDS_PAIRED_N
#define ELP_HOT_WORDS_PAIRED_N 54      /* experiments: a translation unit may shrink its paired hot slot (hot_as<> falls back to private memory for values that no longer fit) */
#endif
constexpr int ELP_HOT_WORDS_PAIRED = ELP_HOT_WORDS_PAIRED_N;
template <class C>
ELP_HD constexpr int hot_words() { return is_paired<C>() ? ELP_HOT_WORDS_PAIRED : ELP_HOT_WORDS; }
template <class T, class C>
ELP_INL T* hot_as(u32* hot) {   // the BN254 Fp12 fills the slot exactly in either layout; smaller accumulators (Jacobian points) fit as well
  return (hot != nullptr && sizeof(T) <= (size_t)hot_words<C>() * 4) ? reinterpret_cast<T*>(hot) : nullptr;
}
// scalar-field (Fr) Montgomery parameters of a curve; specialised in params_<curve>.h
template <class C>
struct FrOf;
template <class B>
struct FrOf<Paired<B>> : FrOf<B> {};
}  // namespace elp
