// Base-field arithmetic Fp for gfx950: Montgomery form over UNSATURATED SIGNED limbs.
//
// Replaces mcl::FpT (third-parties/mcl, reached through G1/G2/pairing at e.g. src/ps-verifier.cc:73-137).
//
// Why not 32-bit saturated limbs: gfx950 has v_mad_u64_u32 / v_mad_i64_i32 (64-bit accumulate) but no multiply-add with
// carry-in, and every carry instruction (v_add_co/v_addc_co, v_lshl_add_u64) issues at the same half rate as the
// multiply itself (profiles/r01_ubench_valu.log); hipcc's saturated CIOS needs ~640 instructions for 120 MACs.
// Representation used instead (NL limbs, NL = 9 for BN254, 14 for BLS12-381):
//     value = sum_i v[i] * 2^(30 i),   v[i] signed 32-bit,   "carried" means |v[i]| <= 2^29 + small for i < NL-1
//     Montgomery radix R = 2^(30 NL) (>= 16 bits above p).
//   * a product column  sum_{i+j=k} a_i b_j + m_i p_j  of 2 NL terms of magnitude < 2^58 fits a signed 64-bit accumulator,
//     so a Montgomery product is 2 NL^2 + NL multiply-accumulates and ~4 cheap instructions per column: no carry chains;
//   * add / sub / neg are limb-wise (sub needs no "+ k p" correction: limbs are signed) followed by one PARALLEL carry pass;
//   * values are kept modulo p only loosely: |value| may grow to ~2^8 p between multiplications and a product always
//     returns |value| < ~1.1 p (R/p = 2^16.8), so no conditional subtractions anywhere on the hot path;
//   * canonical form (for I/O, hashing, equality) = one Montgomery product by 1 + a sequential carry (fp_to_std).
// Invariants used by the group law: (1) the point at infinity and literal zeros are stored with all limbs exactly 0
// (fp_is_zero_exact); (2) the Montgomery product of two multiples of p is literally zero, so "x == 0 (mod p)" can be read
// off x^2 without any reduction (jac_madd); a general modular zero test (fp_is_zero) costs one product and is kept off the
// hot loops.
#pragma once
#include "common.h"

namespace elp {

typedef int32_t i32;
typedef int64_t i64;

template <class C>
struct Fp {  // Montgomery form, signed radix-2^LB limbs (C::LB = 29 or 30)
  i32 v[C::NL];
};
template <class C>
struct StdFp {  // canonical integer in [0, p) (or any N-word integer before range checking), little-endian 32-bit words
  u32 w[C::N];
};

// -DELP_BOUND_CHECK (host twin only): every limb-wise operation is redone in 64 bits and asserted to fit int32, and every
// product asserts that its column sums cannot overflow the signed 64-bit accumulator.
#if defined(ELP_BOUND_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
#include <assert.h>
#include <stdlib.h>
#define ELP_ASSERT_I32(x) assert((x) >= -2147483647LL - 1 && (x) <= 2147483647LL)
#else
#define ELP_ASSERT_I32(x) ((void)0)
#endif

// -DELP_COUNT_OPS (host twin only, tools/count_ops.py): counts Montgomery products / squares so the VALU roofline of a kernel
// can be stated in modular multiplications per item.
#if defined(ELP_COUNT_OPS) && !defined(__HIP_DEVICE_COMPILE__)
inline unsigned long long elp_op_counts[4] = {0, 0, 0, 0};   // fp_mul, fp_sqr, fp_mul_pair, fp_mul_quad
#define ELP_COUNT_OP(i) (elp_op_counts[i]++)
#else
#define ELP_COUNT_OP(i) ((void)0)
#endif

// Keeps a compile-time constant out of the optimiser's sight (in a scalar register on the device): a multiply-add by a modulus limb
// that happens to be a power of two would otherwise be rewritten into a slower 64-bit shift-and-subtract sequence.
ELP_INL i32 elp_opaque(i32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+s"(x));
#else
  asm("" : "+r"(x));
#endif
  return x;
}
// limb geometry of a field: C::LB bits per limb (29 for the 9-limb BN254 field, 30 for the 14-limb BLS12-381 field)
#define ELP_LIMB_BITS (C::LB)
#define ELP_LIMB_HALF ((i32)1 << (C::LB - 1))
#define ELP_LIMB_MASK ((((u32)1) << C::LB) - 1)

// load a generated constant: ELP_LOAD_FP(x, C::curve_b(i_))
#define ELP_LOAD_FP(dst, expr_i)                               \
  do {                                                         \
    ELP_UNROLL                                                 \
    for (int i_ = 0; i_ < C::NL; i_++) (dst).v[i_] = (expr_i); \
  } while (0)

template <class C>
ELP_INL Fp<C> fp_zero() {
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) r.v[i] = 0;
  return r;
}
template <class C>
ELP_INL Fp<C> fp_one() {
  Fp<C> r;
  ELP_LOAD_FP(r, C::one(i_));
  return r;
}
// all limbs literally zero (see the invariant in the header comment)
template <class C>
ELP_INL bool fp_is_zero_exact(const Fp<C>& a) {
  i32 t = 0;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) t |= a.v[i];
  return t == 0;
}
template <class C>
ELP_INL Fp<C> fp_select(bool c, const Fp<C>& a, const Fp<C>& b) {  // c ? a : b
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}

template <int LB>
ELP_INL i32 elp_balanced(u32 x) {  // low LB bits of x as a signed value in [-2^(LB-1), 2^(LB-1))
  return (i32)(x << (32 - LB)) >> (32 - LB);
}
#define elp_balanced30 elp_balanced<C::LB>

// One parallel carry pass: every limb below the top keeps its balanced low 30 bits and receives the carry of its lower
// neighbour; no dependency chain.  Any int32 input; for |limb| <= 2^31 - 1 the output is |limb| <= 2^29 + 2.
template <class C>
ELP_INL void fp_carry(Fp<C>& a) {
  i32 c[C::NL];
  ELP_UNROLL
  for (int i = 0; i < C::NL - 1; i++) {
    c[i] = ((a.v[i] >> (ELP_LIMB_BITS - 1)) + 1) >> 1;          // round(a / 2^30) without overflow for any int32 input
    a.v[i] -= (i32)((u32)c[i] << ELP_LIMB_BITS);
  }
  ELP_UNROLL
  for (int i = 1; i < C::NL; i++) a.v[i] += c[i - 1];
}

// Same pass for inputs known to satisfy |limb| < 2^31 - 2^29 (sums or differences of two carried values, doubled carried
// values): round(a / 2^30) = (a + 2^29) >> 30 cannot overflow, the balanced low part is one signed bit-field extract.
template <class C>
ELP_INL void fp_carry_fast(Fp<C>& a) {
  i32 c[C::NL];
  ELP_UNROLL
  for (int i = 0; i < C::NL - 1; i++) {
    ELP_ASSERT_I32((i64)a.v[i] + (i64)ELP_LIMB_HALF);
    c[i] = (a.v[i] + ELP_LIMB_HALF) >> ELP_LIMB_BITS;
    a.v[i] = elp_balanced30((u32)a.v[i]);
  }
  ELP_UNROLL
  for (int i = 1; i < C::NL; i++) a.v[i] += c[i - 1];
}

template <class C>
ELP_INL Fp<C> fp_add(const Fp<C>& a, const Fp<C>& b) {
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) {
    ELP_ASSERT_I32((i64)a.v[i] + (i64)b.v[i]);
    r.v[i] = a.v[i] + b.v[i];
  }
  fp_carry_fast(r);
  return r;
}
template <class C>
ELP_INL Fp<C> fp_sub(const Fp<C>& a, const Fp<C>& b) {
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) {
    ELP_ASSERT_I32((i64)a.v[i] - (i64)b.v[i]);
    r.v[i] = a.v[i] - b.v[i];
  }
  fp_carry_fast(r);
  return r;
}
template <class C>
ELP_INL Fp<C> fp_neg(const Fp<C>& a) {
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) r.v[i] = -a.v[i];
  return r;
}
template <class C>
ELP_INL Fp<C> fp_dbl(const Fp<C>& a) {
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) {
    ELP_ASSERT_I32((i64)a.v[i] * 2);
    r.v[i] = a.v[i] * 2;
  }
  fp_carry_fast(r);
  return r;
}
// lazy variants: no carry pass (caller guarantees the result feeds at most one side of a product, or carries later)
template <class C>
ELP_INL Fp<C> fp_add_lazy(const Fp<C>& a, const Fp<C>& b) {
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) {
    ELP_ASSERT_I32((i64)a.v[i] + (i64)b.v[i]);
    r.v[i] = a.v[i] + b.v[i];
  }
  return r;
}
template <class C>
ELP_INL Fp<C> fp_sub_lazy(const Fp<C>& a, const Fp<C>& b) {
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) {
    ELP_ASSERT_I32((i64)a.v[i] - (i64)b.v[i]);
    r.v[i] = a.v[i] - b.v[i];
  }
  return r;
}


// Weak modular reduction: subtracts round(value / p) * p, estimated from the top limbs (error < 2^-9), leaving
// |value| <= 0.51 p.  Additions never need it (products bound the magnitude); it exists for the few formulas in which a
// value is carried forward LINEARLY across many steps without passing through a product (Granger-Scott squaring), where
// the magnitude would otherwise double per step.  ~6 instructions per limb.
template <class C>
ELP_INL void fp_reduce_weak(Fp<C>& a) {
  constexpr int NL = C::NL;
  i64 top = a.v[C::QI];
  if (C::QI + 1 < NL) top += (i64)a.v[C::QI + 1] << ELP_LIMB_BITS;
  const i32 q = (i32)((top * C::QK + ((i64)1 << (C::QS - 1))) >> C::QS);
  i64 t = 0;
  ELP_UNROLL
  for (int i = 0; i < NL - 1; i++) {
    t += (i64)a.v[i] - (i64)q * C::modl(i);
    i32 lo = elp_balanced30((u32)t);
    a.v[i] = lo;
    t = (t - lo) >> ELP_LIMB_BITS;
  }
  t += (i64)a.v[NL - 1] - (i64)q * C::modl(NL - 1);
  a.v[NL - 1] = (i32)t;
}

// The multiply-add of the column sums.  Written in C the compiler software-pipelines the columns: it starts the next column in a fresh accumulator
// while the current one waits for its Montgomery digit and merges the two with a 64-bit add per column -- 17 extra instructions per product, which a
// kernel at two waves per SIMD (multiply-add pipe bound) pays in full.  -DELP_ASM_MAC=1 (device builds) issues the multiply-adds as
// opaque v_mad_i64_i32 statements chained through ONE accumulator register pair, which pins the serial order the source states.  Measured (round 2,
// 65 536 proofs, W = 20): 4-5 % fewer vector instructions per product, and the BN254 paired kernel at TWO waves per SIMD gains (19.9 -> 19.0 ms), but
// every kernel that runs ONE wave per SIMD loses, because a lone wave cannot hide the latency of back-to-back dependent multiply-adds: plain kernel
// 18.6 -> 22.6 ms, a lone paired wave 9.9 -> 11.6 ms (and the paired layout is only chosen for batches that give one wave per SIMD); BLS12-381 paired
// 56.9 -> 58.5 ms.  The compiler's pipelining is what these kernels need, so the switch stays off; it is kept for the record and for experiments
// (profiles/r02_asm_mac_experiment.log).
#ifndef ELP_ASM_MAC
#define ELP_ASM_MAC 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && ELP_ASM_MAC
#define ELP_MAC(acc, x, y)                                                                                    \
  do {                                                                                                        \
    unsigned long long cy_;                                                                                   \
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy_) : "v"((i32)(x)), "v"((i32)(y)));            \
  } while (0)
#define ELP_MAC_S(acc, x, ys) /* second factor in a scalar register (modulus limbs) */                        \
  do {                                                                                                        \
    unsigned long long cy_;                                                                                   \
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy_) : "v"((i32)(x)), "s"((i32)(ys)));           \
  } while (0)
#else
#define ELP_MAC(acc, x, y) ((acc) += (i64)(x) * (y))
#define ELP_MAC_S(acc, x, ys) ((acc) += (i64)(x) * (ys))
#endif

// Montgomery product a*b*R^-1 (mod p), column-wise with a single signed 64-bit accumulator.
// Requirements: |a limb| * |b limb| summed over a column stays below 2^62 (true for carried inputs; one of the two may
// be a lazy sum of two carried values).  Output is carried and |value| < |a||b|/R + 0.51 p.
template <class C>
ELP_FPMUL Fp<C> fp_mul(Fp<C> a, Fp<C> b) {
  constexpr int NL = C::NL;
  ELP_COUNT_OP(0);
#if defined(ELP_BOUND_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
  {
    long double ma = 0, mb = 0;
    for (int i = 0; i < NL - 1; i++) {
      if (llabs((long long)a.v[i]) > ma) ma = llabs((long long)a.v[i]);
      if (llabs((long long)b.v[i]) > mb) mb = llabs((long long)b.v[i]);
    }
    // worst column: NL products of the operands + NL products m_i p_j (|m_i| <= 2^29, |p_j| <= 2^29) + carry-in
    assert(ma * mb * NL + (long double)NL * (long double)((i64)1 << (2 * C::LB - 2)) + 1.0e18L < 9223372036854775807.0L);
    assert(llabs((long long)a.v[NL - 1]) < (1LL << (C::LB - 1)) && llabs((long long)b.v[NL - 1]) < (1LL << (C::LB - 1)));
  }
#endif
  i32 m[NL], pl[NL];
  ELP_UNROLL
  for (int i = 0; i < NL; i++) pl[i] = elp_opaque(C::modl(i));
  Fp<C> r;
  i64 acc = 0;
  ELP_UNROLL
  for (int k = 0; k < NL; k++) {
    ELP_UNROLL
    for (int i = 0; i <= k; i++) ELP_MAC(acc, a.v[i], b.v[k - i]);
    ELP_UNROLL
    for (int i = 0; i < k; i++) ELP_MAC_S(acc, m[i], pl[k - i]);
    m[k] = elp_balanced30((u32)acc * C::INVL);
    ELP_MAC_S(acc, m[k], pl[0]);
    acc >>= ELP_LIMB_BITS;                                      // exact: the low 30 bits are zero now
  }
  ELP_UNROLL
  for (int k = NL; k < 2 * NL - 1; k++) {
    ELP_UNROLL
    for (int i = k - NL + 1; i < NL; i++) ELP_MAC(acc, a.v[i], b.v[k - i]);
    ELP_UNROLL
    for (int i = k - NL + 1; i < NL; i++) ELP_MAC_S(acc, m[i], pl[k - i]);
    r.v[k - NL] = elp_balanced30((u32)acc);
    acc = (acc + ELP_LIMB_HALF) >> ELP_LIMB_BITS;               // = (acc - balanced low part) >> 30
  }
  r.v[NL - 1] = (i32)acc;
  return r;
}

// Montgomery reduction of a two-term inner product: (a*b + c*d) * R^-1 (mod p) with ONE reduction (243 multiply-adds instead of
// 324 for two products), the shape of both components of an Fp2 product.  All four operands must be carried (|limb| <= 2^29 + 2):
// a column then holds 2 NL operand products + NL products m_i p_j, which fits the signed 64-bit accumulator only for NL <= 9.
template <class C>
ELP_FPMUL Fp<C> fp_mul_pair(Fp<C> a, Fp<C> b, Fp<C> c, Fp<C> d) {
  constexpr int NL = C::NL;
  static_assert(C::HEADROOM >= 3, "two-product columns overflow the 64-bit accumulator of this field");
  ELP_COUNT_OP(2);
#if defined(ELP_BOUND_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
  {
    long double mab = 0, mcd = 0, ma = 0, mb = 0, mc = 0, md = 0;
    for (int i = 0; i < NL - 1; i++) {
      if (llabs((long long)a.v[i]) > ma) ma = llabs((long long)a.v[i]);
      if (llabs((long long)b.v[i]) > mb) mb = llabs((long long)b.v[i]);
      if (llabs((long long)c.v[i]) > mc) mc = llabs((long long)c.v[i]);
      if (llabs((long long)d.v[i]) > md) md = llabs((long long)d.v[i]);
    }
    mab = ma * mb;
    mcd = mc * md;
    assert((mab + mcd) * NL + (long double)NL * (long double)((i64)1 << (2 * C::LB - 2)) + 1.0e18L < 9223372036854775807.0L);
    assert(llabs((long long)a.v[NL - 1]) < (1LL << (C::LB - 1)) && llabs((long long)b.v[NL - 1]) < (1LL << (C::LB - 1)));
    assert(llabs((long long)c.v[NL - 1]) < (1LL << (C::LB - 1)) && llabs((long long)d.v[NL - 1]) < (1LL << (C::LB - 1)));
  }
#endif
  i32 m[NL], pl[NL];
  ELP_UNROLL
  for (int i = 0; i < NL; i++) pl[i] = elp_opaque(C::modl(i));
  Fp<C> r;
  i64 acc = 0;
  ELP_UNROLL
  for (int k = 0; k < NL; k++) {
    ELP_UNROLL
    for (int i = 0; i <= k; i++) ELP_MAC(acc, a.v[i], b.v[k - i]);
    ELP_UNROLL
    for (int i = 0; i <= k; i++) ELP_MAC(acc, c.v[i], d.v[k - i]);
    ELP_UNROLL
    for (int i = 0; i < k; i++) ELP_MAC_S(acc, m[i], pl[k - i]);
    m[k] = elp_balanced30((u32)acc * C::INVL);
    ELP_MAC_S(acc, m[k], pl[0]);
    acc >>= ELP_LIMB_BITS;
  }
  ELP_UNROLL
  for (int k = NL; k < 2 * NL - 1; k++) {
    ELP_UNROLL
    for (int i = k - NL + 1; i < NL; i++) ELP_MAC(acc, a.v[i], b.v[k - i]);
    ELP_UNROLL
    for (int i = k - NL + 1; i < NL; i++) ELP_MAC(acc, c.v[i], d.v[k - i]);
    ELP_UNROLL
    for (int i = k - NL + 1; i < NL; i++) ELP_MAC_S(acc, m[i], pl[k - i]);
    r.v[k - NL] = elp_balanced30((u32)acc);
    acc = (acc + ELP_LIMB_HALF) >> ELP_LIMB_BITS;
  }
  r.v[NL - 1] = (i32)acc;
  return r;
}

// Four-term inner product with one reduction: (a*b + c*d + e*f + g*h) * R^-1 (mod p), 405 multiply-adds.  Both components of an
// Fp2 inner product x*y + z*w have this shape.  Needs the headroom of the 29-bit field: sum of the four operand-magnitude products
// <= 13 (asserted under ELP_BOUND_CHECK).
template <class C>
ELP_FPMUL Fp<C> fp_mul_quad(Fp<C> a, Fp<C> b, Fp<C> c, Fp<C> d, Fp<C> e, Fp<C> f, Fp<C> g, Fp<C> h) {
  constexpr int NL = C::NL;
  static_assert(C::HEADROOM >= 14, "four-product columns need the 29-bit field");
  ELP_COUNT_OP(3);
#if defined(ELP_BOUND_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
  {
    const Fp<C>* ops[8] = {&a, &b, &c, &d, &e, &f, &g, &h};
    long double mx[8];
    for (int q = 0; q < 8; q++) {
      mx[q] = 0;
      for (int i = 0; i < NL - 1; i++)
        if (llabs((long long)ops[q]->v[i]) > mx[q]) mx[q] = llabs((long long)ops[q]->v[i]);
      assert(llabs((long long)ops[q]->v[NL - 1]) < (1LL << (C::LB - 1)));
    }
    assert((mx[0] * mx[1] + mx[2] * mx[3] + mx[4] * mx[5] + mx[6] * mx[7]) * NL + (long double)NL * (long double)((i64)1 << (2 * C::LB - 2)) + 1.0e18L <
           9223372036854775807.0L);
  }
#endif
  i32 m[NL], pl[NL];
  ELP_UNROLL
  for (int i = 0; i < NL; i++) pl[i] = elp_opaque(C::modl(i));
  Fp<C> r;
  i64 acc = 0;
  ELP_UNROLL
  for (int k = 0; k < NL; k++) {
    ELP_UNROLL
    for (int i = 0; i <= k; i++) ELP_MAC(acc, a.v[i], b.v[k - i]);
    ELP_UNROLL
    for (int i = 0; i <= k; i++) ELP_MAC(acc, c.v[i], d.v[k - i]);
    ELP_UNROLL
    for (int i = 0; i <= k; i++) ELP_MAC(acc, e.v[i], f.v[k - i]);
    ELP_UNROLL
    for (int i = 0; i <= k; i++) ELP_MAC(acc, g.v[i], h.v[k - i]);
    ELP_UNROLL
    for (int i = 0; i < k; i++) ELP_MAC_S(acc, m[i], pl[k - i]);
    m[k] = elp_balanced30((u32)acc * C::INVL);
    ELP_MAC_S(acc, m[k], pl[0]);
    acc >>= ELP_LIMB_BITS;
  }
  ELP_UNROLL
  for (int k = NL; k < 2 * NL - 1; k++) {
    ELP_UNROLL
    for (int i = k - NL + 1; i < NL; i++) ELP_MAC(acc, a.v[i], b.v[k - i]);
    ELP_UNROLL
    for (int i = k - NL + 1; i < NL; i++) ELP_MAC(acc, c.v[i], d.v[k - i]);
    ELP_UNROLL
    for (int i = k - NL + 1; i < NL; i++) ELP_MAC(acc, e.v[i], f.v[k - i]);
    ELP_UNROLL
    for (int i = k - NL + 1; i < NL; i++) ELP_MAC(acc, g.v[i], h.v[k - i]);
    ELP_UNROLL
    for (int i = k - NL + 1; i < NL; i++) ELP_MAC_S(acc, m[i], pl[k - i]);
    r.v[k - NL] = elp_balanced30((u32)acc);
    acc = (acc + ELP_LIMB_HALF) >> ELP_LIMB_BITS;
  }
  r.v[NL - 1] = (i32)acc;
  return r;
}

// Montgomery square: the symmetric half of the a*a columns is computed once and doubled.
template <class C>
ELP_FPMUL Fp<C> fp_sqr(Fp<C> a) {
  constexpr int NL = C::NL;
  ELP_COUNT_OP(1);
  i32 m[NL], pl[NL];
  ELP_UNROLL
  for (int i = 0; i < NL; i++) pl[i] = elp_opaque(C::modl(i));
  Fp<C> r;
  i64 acc = 0;
  ELP_UNROLL
  for (int k = 0; k < 2 * NL - 1; k++) {
    i64 s = 0;
    ELP_UNROLL
    for (int i = (k < NL ? 0 : k - NL + 1); 2 * i < k; i++) s += (i64)a.v[i] * a.v[k - i];
    acc += 2 * s;
    if ((k & 1) == 0) acc += (i64)a.v[k / 2] * a.v[k / 2];
    if (k < NL) {
      ELP_UNROLL
      for (int i = 0; i < k; i++) acc += (i64)m[i] * pl[k - i];
      m[k] = elp_balanced30((u32)acc * C::INVL);
      acc += (i64)m[k] * pl[0];
      acc >>= ELP_LIMB_BITS;
    } else {
      ELP_UNROLL
      for (int i = k - NL + 1; i < NL; i++) acc += (i64)m[i] * pl[k - i];
      r.v[k - NL] = elp_balanced30((u32)acc);
      acc = (acc + ELP_LIMB_HALF) >> ELP_LIMB_BITS;
    }
  }
  r.v[NL - 1] = (i32)acc;
  return r;
}

// a^e for a public exponent given as N 32-bit words through a constexpr accessor: fixed 4-bit windows, MSB first
// (32 N squarings, at most 8 N + 14 products; 0^e = 0).
template <class C, class E>
ELP_HEAVY Fp<C> fp_pow_const(const Fp<C>& a, E expo) {
  Fp<C> t[16];
  t[1] = a;
  ELP_NOUNROLL
  for (int i = 2; i < 16; i++) t[i] = fp_mul<C>(t[i - 1], a);
  Fp<C> r = fp_one<C>();
  bool started = false;
  ELP_NOUNROLL
  for (int i = C::N * 8 - 1; i >= 0; i--) {
    const int nib = (int)((expo(i >> 3) >> (4 * (i & 7))) & 15);
    if (started) {
      ELP_NOUNROLL
      for (int s = 0; s < 4; s++) r = fp_sqr<C>(r);
    }
    if (nib != 0) {
      r = started ? fp_mul<C>(r, t[nib]) : t[nib];
      started = true;
    }
  }
  return r;
}
template <class C>
struct ExpPm2 {
  ELP_HD u32 operator()(int i) const { return C::pm2(i); }
};
template <class C>
struct ExpPp1d4 {
  ELP_HD u32 operator()(int i) const { return C::pp1d4(i); }
};
template <class C>
struct ExpPm1d2 {
  ELP_HD u32 operator()(int i) const { return C::pm1d2(i); }
};

// ---- canonical form.  fp_to_std: Montgomery product by 1 gives a representative in (-0.6 p, 0.6 p); lift to [0, p) and
// emit 32-bit words.
template <class C>
ELP_HEAVY StdFp<C> fp_to_std(const Fp<C>& a) {
  constexpr int NL = C::NL;
  Fp<C> one = fp_zero<C>();
  one.v[0] = 1;
  Fp<C> t = fp_mul<C>(a, one);
  // sequential carry to digits in [0, 2^30); the top limb keeps the sign of the value
  i32 c = 0;
  for (int i = 0; i < NL - 1; i++) {
    i32 x = t.v[i] + c;
    c = x >> ELP_LIMB_BITS;
    t.v[i] = x & (i32)ELP_LIMB_MASK;
  }
  t.v[NL - 1] += c;
  if (t.v[NL - 1] < 0) {  // negative representative: add p once
    c = 0;
    for (int i = 0; i < NL - 1; i++) {
      i32 x = t.v[i] + C::modl(i) + c;
      c = x >> ELP_LIMB_BITS;
      t.v[i] = x & (i32)ELP_LIMB_MASK;
    }
    t.v[NL - 1] += C::modl(NL - 1) + c;
  }
  StdFp<C> s;
  for (int j = 0; j < C::N; j++) {  // pack LB-bit digits into 32-bit words
    int bit = 32 * j, li = bit / C::LB, sh = bit % C::LB;
    u64 x = (u64)(u32)t.v[li] >> sh;
    int have = C::LB - sh;
    for (int k = li + 1; have < 32 && k < NL; k++, have += C::LB) x |= (u64)(u32)t.v[k] << have;
    s.w[j] = (u32)x;
  }
  return s;
}
// canonical words (value < 2^(32 N), normally < p) -> Montgomery form
template <class C>
ELP_HEAVY Fp<C> fp_from_std(const StdFp<C>& s) {
  constexpr int NL = C::NL;
  Fp<C> t;
  for (int i = 0; i < NL; i++) {  // unpack into unsigned LB-bit digits
    int bit = C::LB * i, wi = bit >> 5, sh = bit & 31;
    u64 x = 0;
    if (wi < C::N) x = (u64)s.w[wi] >> sh;
    if (wi + 1 < C::N && sh > 32 - C::LB) x |= (u64)s.w[wi + 1] << (32 - sh);
    t.v[i] = (i32)((u32)x & ELP_LIMB_MASK);
  }
  fp_carry(t);  // balance the digits
  Fp<C> r2;
  ELP_LOAD_FP(r2, C::r2(i_));
  return fp_mul<C>(t, r2);
}
// modular zero / equality tests (one product each; keep out of inner loops)
template <class C>
ELP_HEAVY bool fp_is_zero(const Fp<C>& a) {
  StdFp<C> s = fp_to_std<C>(a);
  u32 t = 0;
  for (int i = 0; i < C::N; i++) t |= s.w[i];
  return t == 0;
}
template <class C>
ELP_INL bool fp_eq(const Fp<C>& a, const Fp<C>& b) {
  return fp_is_zero<C>(fp_sub_lazy(a, b));
}

template <class C>
ELP_HEAVY Fp<C> fp_inv_pow(const Fp<C>& a) {  // a^(p-2); inv(0) = 0 (the Fermat inversion, kept as a cross-check)
  return fp_pow_const<C>(a, ExpPm2<C>());
}

// Inversion by Bernstein-Yang division steps ("safegcd", the half-delta variant also used by libsecp256k1's modinv32), in batches of LB
// steps on the low limbs with a 2x2 transition matrix applied to the full-length (f, g) and (d, e).  Branch-free, so the lanes of a
// wave stay converged; about a sixth of the cost of the Fermat power.  inv(0) = 0.
template <class C>
ELP_HEAVY Fp<C> fp_inv(const Fp<C>& a) {
  constexpr int NL = C::NL;
  constexpr int LB = C::LB;
  const i32 MLB = (i32)ELP_LIMB_MASK;
  // x = a / R as a plain integer in [0, p), digits in [0, 2^LB), signed top limb
  Fp<C> one = fp_zero<C>();
  one.v[0] = 1;
  Fp<C> gq = fp_mul<C>(a, one);
  {
    i32 c = 0;
    for (int i = 0; i < NL - 1; i++) {
      i32 x = gq.v[i] + c;
      c = x >> LB;
      gq.v[i] = x & MLB;
    }
    gq.v[NL - 1] += c;
    const i32 neg = gq.v[NL - 1] >> 31;                   // all ones when the representative is negative: add p
    c = 0;
    for (int i = 0; i < NL - 1; i++) {
      i32 x = gq.v[i] + (C::modl(i) & neg) + c;
      c = x >> LB;
      gq.v[i] = x & MLB;
    }
    gq.v[NL - 1] += (C::modl(NL - 1) & neg) + c;
  }
  Fp<C> fq, dq = fp_zero<C>(), eq = fp_zero<C>();
  ELP_LOAD_FP(fq, C::modl(i_));
  eq.v[0] = 1;
  i32 zeta = -1;
  ELP_NOUNROLL
  for (int it = 0; it < C::INV_ITERS; it++) {
    // LB division steps on the low limbs
    u32 u = 1, v = 0, q = 0, r = 1, f = (u32)fq.v[0], g = (u32)gq.v[0];
    ELP_NOUNROLL
    for (int s = 0; s < LB; s++) {
      u32 m1 = (u32)(zeta >> 31);
      const u32 m2 = 0u - (g & 1u);
      const u32 x = (f ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1;
      g += x & m2;
      q += y & m2;
      r += z & m2;
      m1 &= m2;
      zeta = (i32)((u32)zeta ^ m1) - 1;
      f += g & m1;
      u += q & m1;
      v += r & m1;
      g >>= 1;
      u <<= 1;
      v <<= 1;
    }
    const i32 tu = (i32)u, tv = (i32)v, tq = (i32)q, tr = (i32)r;
    // (d, e) <- t (d, e) / 2^LB mod p
    {
      const i32 sd = dq.v[NL - 1] >> 31, se = eq.v[NL - 1] >> 31;
      i32 md = (tu & sd) + (tv & se), me = (tq & sd) + (tr & se);
      i64 cd = (i64)tu * dq.v[0] + (i64)tv * eq.v[0];
      i64 ce = (i64)tq * dq.v[0] + (i64)tr * eq.v[0];
      md -= (i32)((C::PINVL * (u32)cd + (u32)md) & (u32)MLB);
      me -= (i32)((C::PINVL * (u32)ce + (u32)me) & (u32)MLB);
      cd += (i64)C::modl(0) * md;
      ce += (i64)C::modl(0) * me;
      cd >>= LB;
      ce >>= LB;
      ELP_UNROLL
      for (int i = 1; i < NL; i++) {
        cd += (i64)tu * dq.v[i] + (i64)tv * eq.v[i] + (i64)C::modl(i) * md;
        ce += (i64)tq * dq.v[i] + (i64)tr * eq.v[i] + (i64)C::modl(i) * me;
        dq.v[i - 1] = (i32)cd & MLB;
        eq.v[i - 1] = (i32)ce & MLB;
        cd >>= LB;
        ce >>= LB;
      }
      dq.v[NL - 1] = (i32)cd;
      eq.v[NL - 1] = (i32)ce;
    }
    // (f, g) <- t (f, g) / 2^LB
    {
      i64 cf = (i64)tu * fq.v[0] + (i64)tv * gq.v[0];
      i64 cg = (i64)tq * fq.v[0] + (i64)tr * gq.v[0];
      cf >>= LB;
      cg >>= LB;
      ELP_UNROLL
      for (int i = 1; i < NL; i++) {
        cf += (i64)tu * fq.v[i] + (i64)tv * gq.v[i];
        cg += (i64)tq * fq.v[i] + (i64)tr * gq.v[i];
        fq.v[i - 1] = (i32)cf & MLB;
        gq.v[i - 1] = (i32)cg & MLB;
        cf >>= LB;
        cg >>= LB;
      }
      fq.v[NL - 1] = (i32)cf;
      gq.v[NL - 1] = (i32)cg;
    }
  }
  // f = +-1 now (or +-p when x = 0, d = 0): d = +-x^-1 in (-2p, p); back to Montgomery form with one product by R^2
  const i32 sf = fq.v[NL - 1] >> 31;
  ELP_UNROLL
  for (int i = 0; i < NL; i++) dq.v[i] = (dq.v[i] ^ sf) - sf;
  fp_carry(dq);                                  // digits in [0, 2^LB) -> balanced limbs, as fp_mul expects of its operands
  Fp<C> r2;
  ELP_LOAD_FP(r2, C::r2(i_));
  return fp_mul<C>(dq, r2);
}
// square root for p = 3 (mod 4): returns false if a is not a square
template <class C>
ELP_HEAVY bool fp_sqrt(Fp<C>& r, const Fp<C>& a) {
  r = fp_pow_const<C>(a, ExpPp1d4<C>());
  return fp_eq(fp_sqr<C>(r), a);
}
// Legendre symbol: +1, 0, -1
template <class C>
ELP_HEAVY int fp_legendre(const Fp<C>& a) {
  if (fp_is_zero<C>(a)) return 0;
  Fp<C> t = fp_pow_const<C>(a, ExpPm1d2<C>());
  return fp_eq(t, fp_one<C>()) ? 1 : -1;
}

// ---- std-form helpers (plain integers, not field elements)
template <class C>
ELP_INL bool std_is_zero(const StdFp<C>& a) {
  u32 t = 0;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) t |= a.w[i];
  return t == 0;
}
template <class C>
ELP_INL bool std_in_range(const StdFp<C>& a) {  // a < p ?
  u64 br = 0;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) {
    u64 t = (u64)a.w[i] - C::mod(i) - br;
    br = (t >> 32) & 1;
  }
  return br != 0;
}
template <class C>
ELP_INL StdFp<C> std_load_le(const uint8_t* b) {  // FBYTES little-endian bytes
  StdFp<C> r;
  for (int i = 0; i < C::N; i++)
    r.w[i] = (u32)b[4 * i] | ((u32)b[4 * i + 1] << 8) | ((u32)b[4 * i + 2] << 16) | ((u32)b[4 * i + 3] << 24);
  return r;
}
template <class C>
ELP_INL void std_store_le(uint8_t* b, const StdFp<C>& a) {
  for (int i = 0; i < C::N; i++) {
    b[4 * i] = (uint8_t)a.w[i];
    b[4 * i + 1] = (uint8_t)(a.w[i] >> 8);
    b[4 * i + 2] = (uint8_t)(a.w[i] >> 16);
    b[4 * i + 3] = (uint8_t)(a.w[i] >> 24);
  }
}

}  // namespace elp
