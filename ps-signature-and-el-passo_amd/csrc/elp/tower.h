// Extension tower Fp2 = Fp[i]/(i^2+1), Fp6 = Fp2[v]/(v^3 - xi), Fp12 = Fp6[w]/(w^2 - v), xi = 1 + i.
// Replaces mcl::Fp2T / Fp6T / Fp12T (third-parties/mcl) as used under pairing() at src/ps-verifier.cc:31-34,134-137.
#pragma once
#include "fp.h"

namespace elp {

// ------------------------------------------------------------------ Fp2
// Two layouts (common.h, "Lane pairs"): the plain one keeps both components in the lane; for C = Paired<B> the lane keeps ONE component
// (even lane: real part, odd lane: imaginary part) and the routines below exchange operands with the partner lane where the algebra
// couples the components.  Everything above Fp2 is written against these routines only and therefore serves both layouts.
template <class C, bool P = PairInfo<C>::paired>
struct Fp2;
template <class C>
struct Fp2<C, false> {
  Fp<C> c0, c1;
};
template <class C>
struct Fp2<C, true> {
  Fp<C> c;   // this lane's component
};

// ---- pair primitives on field elements
template <class C>
ELP_INL Fp<C> fp_pair_swap(const Fp<C>& a) {   // the partner lane's value
  Fp<C> r;
#if defined(__HIP_DEVICE_COMPILE__)
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) r.v[i] = pair_swap_i32(a.v[i]);
#else
  r = a;
  elp_pair_exchange_hook(&r, sizeof r);
#endif
  return r;
}
template <class C>
ELP_INL Fp<C> fp_cneg(bool c, const Fp<C>& a) {   // c ? -a : a
  const i32 m = c ? -1 : 0;
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) r.v[i] = (a.v[i] ^ m) - m;
  return r;
}
// a generated Fp2 constant: this lane's component (paired) or both
#define ELP_LOAD_FP2(dst, expr_c)                                        \
  do {                                                                   \
    if constexpr (is_paired<C>()) {                                      \
      const bool odd_ = pair_odd();                                      \
      ELP_UNROLL                                                         \
      for (int i_ = 0; i_ < C::NL; i_++) {                               \
        const i32 e0_ = [&](int c_) { return (expr_c); }(0);             \
        const i32 e1_ = [&](int c_) { return (expr_c); }(1);             \
        (dst).c.v[i_] = odd_ ? e1_ : e0_;                                \
      }                                                                  \
    } else {                                                             \
      ELP_UNROLL                                                         \
      for (int i_ = 0; i_ < C::NL; i_++) {                               \
        (dst).c0.v[i_] = [&](int c_) { return (expr_c); }(0);            \
        (dst).c1.v[i_] = [&](int c_) { return (expr_c); }(1);            \
      }                                                                  \
    }                                                                    \
  } while (0)

template <class C>
ELP_INL Fp2<C> fp2_zero() {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_zero<C>();
  } else {
    r.c0 = fp_zero<C>();
    r.c1 = fp_zero<C>();
  }
  return r;
}
template <class C>
ELP_INL Fp2<C> fp2_one() {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_select(pair_odd(), fp_zero<C>(), fp_one<C>());
  } else {
    r.c0 = fp_one<C>();
    r.c1 = fp_zero<C>();
  }
  return r;
}
template <class C>
ELP_INL bool fp2_is_zero(const Fp2<C>& a) {          // modular test (two products)
  if constexpr (is_paired<C>()) return pair_and(fp_is_zero<C>(a.c));
  else return fp_is_zero<C>(a.c0) && fp_is_zero<C>(a.c1);
}
template <class C>
ELP_INL bool fp2_is_zero_exact(const Fp2<C>& a) {    // literal zero limbs
  if constexpr (is_paired<C>()) {
    i32 t = 0;
    ELP_UNROLL
    for (int i = 0; i < C::NL; i++) t |= a.c.v[i];
    t |= pair_swap_i32(t);
    return t == 0;
  } else {
    return fp_is_zero_exact(a.c0) && fp_is_zero_exact(a.c1);
  }
}
template <class C>
ELP_INL Fp2<C> fp2_sub_lazy(const Fp2<C>& a, const Fp2<C>& b);
template <class C>
ELP_INL bool fp2_eq(const Fp2<C>& a, const Fp2<C>& b) {
  if constexpr (is_paired<C>()) return fp2_is_zero<C>(fp2_sub_lazy(a, b));
  else return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1);
}
// component-wise operations: ELP_FP2_CW(r, expression in terms of X(a) ...) would hide too much; spelled out instead
template <class C>
ELP_INL Fp2<C> fp2_add(const Fp2<C>& a, const Fp2<C>& b) {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_add(a.c, b.c);
  } else {
    r.c0 = fp_add(a.c0, b.c0);
    r.c1 = fp_add(a.c1, b.c1);
  }
  return r;
}
template <class C>
ELP_INL Fp2<C> fp2_sub(const Fp2<C>& a, const Fp2<C>& b) {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_sub(a.c, b.c);
  } else {
    r.c0 = fp_sub(a.c0, b.c0);
    r.c1 = fp_sub(a.c1, b.c1);
  }
  return r;
}
template <class C>
ELP_INL Fp2<C> fp2_neg(const Fp2<C>& a) {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_neg(a.c);
  } else {
    r.c0 = fp_neg(a.c0);
    r.c1 = fp_neg(a.c1);
  }
  return r;
}
template <class C>
ELP_INL Fp2<C> fp2_dbl(const Fp2<C>& a) {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_dbl(a.c);
  } else {
    r.c0 = fp_dbl(a.c0);
    r.c1 = fp_dbl(a.c1);
  }
  return r;
}
template <class C>
ELP_INL Fp2<C> fp2_conj(const Fp2<C>& a) {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_cneg(pair_odd(), a.c);
  } else {
    r.c0 = a.c0;
    r.c1 = fp_neg(a.c1);
  }
  return r;
}
template <class C>
ELP_INL Fp2<C> fp2_select(bool c, const Fp2<C>& a, const Fp2<C>& b) {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_select(c, a.c, b.c);
  } else {
    r.c0 = fp_select(c, a.c0, b.c0);
    r.c1 = fp_select(c, a.c1, b.c1);
  }
  return r;
}
// lazy (carry-free) variants; results must be carried (fp2_carry) before they are stored or multiplied on both sides
template <class C>
ELP_INL Fp2<C> fp2_add_lazy(const Fp2<C>& a, const Fp2<C>& b) {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_add_lazy(a.c, b.c);
  } else {
    r.c0 = fp_add_lazy(a.c0, b.c0);
    r.c1 = fp_add_lazy(a.c1, b.c1);
  }
  return r;
}
template <class C>
ELP_INL Fp2<C> fp2_sub_lazy(const Fp2<C>& a, const Fp2<C>& b) {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_sub_lazy(a.c, b.c);
  } else {
    r.c0 = fp_sub_lazy(a.c0, b.c0);
    r.c1 = fp_sub_lazy(a.c1, b.c1);
  }
  return r;
}
template <class C>
ELP_INL Fp2<C> fp2_carry(Fp2<C> a) {
  if constexpr (is_paired<C>()) {
    fp_carry(a.c);
  } else {
    fp_carry(a.c0);
    fp_carry(a.c1);
  }
  return a;
}
template <class C>
ELP_INL Fp2<C> fp2_carry_fast(Fp2<C> a) {   // |limb| < 2^31 - 2^(LB-1): lazy sums of up to six carried values at LB = 29
  if constexpr (is_paired<C>()) {
    fp_carry_fast(a.c);
  } else {
    fp_carry_fast(a.c0);
    fp_carry_fast(a.c1);
  }
  return a;
}
template <class C>
ELP_INL void fp2_reduce_weak(Fp2<C>& a) {
  if constexpr (is_paired<C>()) {
    fp_reduce_weak(a.c);
  } else {
    fp_reduce_weak(a.c0);
    fp_reduce_weak(a.c1);
  }
}
// Fields with C::HEADROOM >= 14 (BN254 at 29-bit limbs): a product accepts operands whose limb magnitudes, in units of a carried
// limb, satisfy A*B <= 13 (fp_mul) or A*B + C*D <= 13 (fp_mul_pair), so the formulas below add and subtract without carrying and
// carry each result once.  int32 limbs hold lazy sums of up to seven carried values.
template <class C>
ELP_HD constexpr bool fp_roomy() { return C::HEADROOM >= 14; }
// (a0 + a1 i)(1 + i) = (a0 - a1) + (a0 + a1) i.   Paired: own -+ partner (even lane: own - partner, odd lane: own + partner)
template <class C>
ELP_INL Fp2<C> fp2_mul_xi_lazy(const Fp2<C>& a) {   // input carried, output limbs <= 2^30
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_add_lazy(a.c, fp_cneg(!pair_odd(), fp_pair_swap(a.c)));
  } else {
    r.c0 = fp_sub_lazy(a.c0, a.c1);
    r.c1 = fp_add_lazy(a.c0, a.c1);
  }
  return r;
}
template <class C>
ELP_INL Fp2<C> fp2_mul_xi(const Fp2<C>& a) {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_add(a.c, fp_cneg(!pair_odd(), fp_pair_swap(a.c)));
  } else {
    r.c0 = fp_sub(a.c0, a.c1);
    r.c1 = fp_add(a.c0, a.c1);
  }
  return r;
}
// Paired product: with (p, q) the partner's copies of (a, b),
//    even lane:  a0 b0 - a1 b1 = own_a * own_b + p_a * (-p_b)          odd lane:  a1 b0 + a0 b1 = own_a * p_b + p_a * own_b
// i.e. ONE two-term inner product per lane, the same instruction stream on both lanes, operands chosen by lane parity.
template <class C>
ELP_FP2 void fp2_mul(Fp2<C>& r, const Fp2<C>& a, const Fp2<C>& b) {  // operands carried
  if constexpr (is_paired<C>()) {
    const bool odd = pair_odd();
    const Fp<C> pa = fp_pair_swap(a.c), pb = fp_pair_swap(b.c);
    const Fp<C> y = fp_select(odd, pb, b.c);
    const Fp<C> w = fp_select(odd, b.c, fp_neg(pb));
    if constexpr (C::HEADROOM >= 3) {
      r.c = fp_mul_pair<C>(a.c, y, pa, w);
    } else {
      Fp<C> t = fp_sub_lazy(fp_mul<C>(a.c, y), fp_neg(fp_mul<C>(pa, w)));
      fp_carry(t);
      r.c = t;
    }
    return;
  } else {
  if constexpr (C::HEADROOM >= 3) {
    // schoolbook with one reduction per component: the same 486 multiply-adds as Karatsuba with three reductions, but no operand
    // sums, no output differences and no carry passes
    Fp<C> c0 = fp_mul_pair<C>(a.c0, b.c0, fp_neg(a.c1), b.c1);
    Fp<C> c1 = fp_mul_pair<C>(a.c0, b.c1, a.c1, b.c0);
    r.c0 = c0;
    r.c1 = c1;
    return;
  }
  // Karatsuba, 3 Fp mul
  Fp<C> t0 = fp_mul<C>(a.c0, b.c0);
  Fp<C> t1 = fp_mul<C>(a.c1, b.c1);
  // one operand of a product may be a lazy two-term sum when 9 limbs are summed per column (BN254); with 14 limbs
  // (BLS12-381) the accumulator has no room for it
  Fp<C> sa = (C::HEADROOM >= 3) ? fp_add_lazy(a.c0, a.c1) : fp_add(a.c0, a.c1);
  Fp<C> s = fp_mul<C>(sa, fp_add(b.c0, b.c1));
  r.c0 = fp_sub(t0, t1);
  Fp<C> u = fp_sub_lazy(fp_sub_lazy(s, t0), t1);   // three carried terms: |limb| < 1.5 * 2^30
  fp_carry(u);
  r.c1 = u;
  }
}
// r = a*b + c*d in Fp2: one reduction per component where the field has the headroom (operand magnitudes as for fp_mul_quad)
template <class C>
ELP_FP2 void fp2_mul_pair(Fp2<C>& r, const Fp2<C>& a, const Fp2<C>& b, const Fp2<C>& c, const Fp2<C>& d) {
  if constexpr (is_paired<C>() && fp_roomy<C>()) {
    const bool odd = pair_odd();
    const Fp<C> pa = fp_pair_swap(a.c), pb = fp_pair_swap(b.c), pc = fp_pair_swap(c.c), pd = fp_pair_swap(d.c);
    r.c = fp_mul_quad<C>(a.c, fp_select(odd, pb, b.c), pa, fp_select(odd, b.c, fp_neg(pb)), c.c, fp_select(odd, pd, d.c), pc,
                         fp_select(odd, d.c, fp_neg(pd)));
  } else if constexpr (!is_paired<C>() && fp_roomy<C>()) {
    Fp<C> c0 = fp_mul_quad<C>(a.c0, b.c0, fp_neg(a.c1), b.c1, c.c0, d.c0, fp_neg(c.c1), d.c1);
    Fp<C> c1 = fp_mul_quad<C>(a.c0, b.c1, a.c1, b.c0, c.c0, d.c1, c.c1, d.c0);
    r.c0 = c0;
    r.c1 = c1;
  } else {
    Fp2<C> t, u;
    fp2_mul<C>(t, a, b);
    fp2_mul<C>(u, c, d);
    r = fp2_add(t, u);
  }
}
// Paired square: even lane (a0 + a1)(a0 - a1), odd lane (2 a1) a0 -- one product per lane.
template <class C>
ELP_FP2 void fp2_sqr(Fp2<C>& r, const Fp2<C>& a) {  // 2 Fp mul; operand carried
  if constexpr (is_paired<C>()) {
    const bool odd = pair_odd();
    const Fp<C> pa = fp_pair_swap(a.c);
    Fp<C> x = fp_add_lazy(a.c, fp_select(odd, a.c, pa));          // a0 + a1 | 2 a1
    Fp<C> y = fp_select(odd, pa, fp_sub_lazy(a.c, pa));            // a0 - a1 | a0
    if constexpr (!fp_roomy<C>()) {
      fp_carry(x);
      fp_carry(y);
    }
    r.c = fp_mul<C>(x, y);
    return;
  } else {
  if constexpr (fp_roomy<C>()) {
    Fp<C> c0 = fp_mul<C>(fp_add_lazy(a.c0, a.c1), fp_sub_lazy(a.c0, a.c1));   // 2 x 2
    Fp<C> c1 = fp_mul<C>(fp_add_lazy(a.c0, a.c0), a.c1);                       // 2 x 1
    r.c0 = c0;
    r.c1 = c1;
    return;
  }
  Fp<C> t = fp_mul<C>(a.c0, a.c1);
  Fp<C> sa = (C::HEADROOM >= 3) ? fp_add_lazy(a.c0, a.c1) : fp_add(a.c0, a.c1);
  Fp<C> u = fp_mul<C>(sa, fp_sub(a.c0, a.c1));
  r.c0 = u;
  r.c1 = fp_dbl(t);
  }
}
template <class C>
ELP_INL Fp2<C> fp2_mulv(const Fp2<C>& a, const Fp2<C>& b) {
  Fp2<C> r;
  fp2_mul<C>(r, a, b);
  return r;
}
template <class C>
ELP_INL Fp2<C> fp2_sqrv(const Fp2<C>& a) {
  Fp2<C> r;
  fp2_sqr<C>(r, a);
  return r;
}
template <class C>
ELP_INL Fp2<C> fp2_mul_fp(const Fp2<C>& a, const Fp<C>& s) {   // s: the same base-field value on both lanes of a pair
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_mul<C>(a.c, s);
  } else {
    r.c0 = fp_mul<C>(a.c0, s);
    r.c1 = fp_mul<C>(a.c1, s);
  }
  return r;
}
// norm a0^2 + a1^2 (a base-field value; paired: the same on both lanes).  Exactly zero limbs for the literal zero.
template <class C>
ELP_INL Fp<C> fp2_norm(const Fp2<C>& a) {
  if constexpr (is_paired<C>()) {
    const Fp<C> s = fp_sqr<C>(a.c);
    return fp_add(s, fp_pair_swap(s));
  } else {
    return fp_add(fp_sqr<C>(a.c0), fp_sqr<C>(a.c1));
  }
}
// conj(a) * s for a base-field s (with s = 1 / norm(a): the inverse of a)
template <class C>
ELP_INL Fp2<C> fp2_conj_mul_fp(const Fp2<C>& a, const Fp<C>& s) {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_cneg(pair_odd(), fp_mul<C>(a.c, s));
  } else {
    r.c0 = fp_mul<C>(a.c0, s);
    r.c1 = fp_neg(fp_mul<C>(a.c1, s));
  }
  return r;
}
template <class C>
ELP_HEAVY void fp2_inv(Fp2<C>& r, const Fp2<C>& a) {
  r = fp2_conj_mul_fp<C>(a, fp_inv<C>(fp2_norm<C>(a)));
}
// ---- moving between the layouts (paired <-> both components in one lane), for the few routines that are not worth pairing
template <class C>
struct Fp2Full {   // both components, plain layout over the SAME field traits
  Fp<C> c0, c1;
};
template <class C>
ELP_INL Fp2Full<C> fp2_gather(const Fp2<C>& a) {
  Fp2Full<C> r;
  if constexpr (is_paired<C>()) {
    const Fp<C> p = fp_pair_swap(a.c);
    const bool odd = pair_odd();
    r.c0 = fp_select(odd, p, a.c);
    r.c1 = fp_select(odd, a.c, p);
  } else {
    r.c0 = a.c0;
    r.c1 = a.c1;
  }
  return r;
}
template <class C>
ELP_INL Fp2<C> fp2_scatter(const Fp<C>& c0, const Fp<C>& c1) {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_select(pair_odd(), c1, c0);
  } else {
    r.c0 = c0;
    r.c1 = c1;
  }
  return r;
}
// Square root in Fp2 for p = 3 (mod 4) ("complex method"), two Fp exponentiations and no inversion (round 5; three to four before).  With n = sqrt(a0^2 + a1^2),
// alpha = (a0 + n) / 2, t = alpha^((p - 3) / 4) and x = t alpha:  x t = alpha^((p - 1) / 2) is the Legendre symbol of alpha.  +1: x^2 = alpha, 1 / x = t and the
// root is x + (a1 t / 2) i.  -1: x^2 = -alpha, 1 / x = -t and the root is (-a1 t / 2) + x i (then u^2 - v^2 = (a0 - n) / 2 + (a0 + n) / 2 = a0, 2 u v = a1).
// Returns false when a is not a square (its norm is not one in Fp).  Which of the two roots comes out is not specified: the one caller (g2_deserialize) picks by the
// encoding's parity flag.  Paired layout: both lanes run the whole computation on gathered components (decompression only; not on the verification path proper).
template <class C>
ELP_HEAVY bool fp2_sqrt(Fp2<C>& r, const Fp2<C>& a_in) {
  const Fp2Full<C> a = fp2_gather<C>(a_in);
  if (fp_is_zero<C>(a.c1)) {           // a in Fp: sqrt(a0), or i sqrt(-a0) (one of a0, -a0 is a square)
    const Fp<C> s = fp_pow_const<C>(a.c0, ExpPp1d4<C>());
    const bool real = fp_eq(fp_sqr<C>(s), a.c0);
    r = real ? fp2_scatter<C>(s, fp_zero<C>()) : fp2_scatter<C>(fp_zero<C>(), s);
    return true;
  }
  Fp<C> n;
  if (!fp_sqrt<C>(n, fp_add(fp_sqr<C>(a.c0), fp_sqr<C>(a.c1)))) return false;
  Fp<C> inv2;
  ELP_LOAD_FP(inv2, C::inv2(i_));
  const Fp<C> alpha = fp_mul<C>(fp_add(a.c0, n), inv2);        // != 0: alpha = 0 would mean n = -a0, a1 = 0
  const Fp<C> t = fp_pow_const<C>(alpha, ExpPm3d4<C>());
  const Fp<C> x = fp_mul<C>(t, alpha);
  const bool qr = fp_eq(fp_mul<C>(x, t), fp_one<C>());
  const Fp<C> y = fp_mul<C>(fp_mul<C>(a.c1, t), inv2);
  r = qr ? fp2_scatter<C>(x, y) : fp2_scatter<C>(fp_neg(y), x);
  return true;
}
template <class C>
ELP_INL Fp2<C> fp2_frob_coeff(int n, int k) {  // gamma_{n,k} = xi^(k (p^n - 1)/6), k = 1..5
  Fp2<C> g;
  if (n == 1) {
    ELP_LOAD_FP2(g, C::frob1(k, c_, i_));
  } else if (n == 2) {
    ELP_LOAD_FP2(g, C::frob2(k, c_, i_));
  } else {
    ELP_LOAD_FP2(g, C::frob3(k, c_, i_));
  }
  return g;
}

// ------------------------------------------------------------------ Fp6
template <class C>
struct Fp6 {
  Fp2<C> c0, c1, c2;
};
template <class C>
ELP_INL void fp6_add(Fp6<C>& r, const Fp6<C>& a, const Fp6<C>& b) {
  r.c0 = fp2_add(a.c0, b.c0);
  r.c1 = fp2_add(a.c1, b.c1);
  r.c2 = fp2_add(a.c2, b.c2);
}
template <class C>
ELP_INL void fp6_sub(Fp6<C>& r, const Fp6<C>& a, const Fp6<C>& b) {
  r.c0 = fp2_sub(a.c0, b.c0);
  r.c1 = fp2_sub(a.c1, b.c1);
  r.c2 = fp2_sub(a.c2, b.c2);
}
template <class C>
ELP_INL void fp6_neg(Fp6<C>& r, const Fp6<C>& a) {
  r.c0 = fp2_neg(a.c0);
  r.c1 = fp2_neg(a.c1);
  r.c2 = fp2_neg(a.c2);
}
template <class C>
ELP_INL void fp6_mul_by_v(Fp6<C>& r, const Fp6<C>& a) {  // (c0,c1,c2) v = (xi c2, c0, c1)
  Fp2<C> t = fp2_mul_xi(a.c2);
  r.c2 = a.c1;
  r.c1 = a.c0;
  r.c0 = t;
}
template <class C>
ELP_FP6 void fp6_mul(Fp6<C>& r, const Fp6<C>& a, const Fp6<C>& b) {  // Karatsuba, 6 Fp2 mul; operands carried
  if constexpr (fp_roomy<C>()) {
    Fp2<C> t0, t1, t2, s, r0, r1, r2;
    fp2_mul<C>(t0, a.c0, b.c0);
    fp2_mul<C>(t1, a.c1, b.c1);
    fp2_mul<C>(t2, a.c2, b.c2);
    fp2_mul<C>(s, fp2_add_lazy(a.c1, a.c2), fp2_add_lazy(b.c1, b.c2));                                    // (2 x 2) + (2 x 2)
    r0 = fp2_carry(fp2_add_lazy(t0, fp2_mul_xi_lazy(fp2_sub_lazy(fp2_sub_lazy(s, t1), t2))));             // 1 + 2 * 3 = 7
    fp2_mul<C>(s, fp2_add_lazy(a.c0, a.c1), fp2_add_lazy(b.c0, b.c1));
    r1 = fp2_carry_fast(fp2_add_lazy(fp2_sub_lazy(fp2_sub_lazy(s, t0), t1), fp2_mul_xi_lazy(t2)));        // 3 + 2
    fp2_mul<C>(s, fp2_add_lazy(a.c0, a.c2), fp2_add_lazy(b.c0, b.c2));
    r2 = fp2_carry_fast(fp2_add_lazy(fp2_sub_lazy(fp2_sub_lazy(s, t0), t2), t1));                         // 4
    r.c0 = r0;
    r.c1 = r1;
    r.c2 = r2;
    return;
  }
  Fp2<C> t0, t1, t2, s;
  fp2_mul<C>(t0, a.c0, b.c0);
  fp2_mul<C>(t1, a.c1, b.c1);
  fp2_mul<C>(t2, a.c2, b.c2);
  Fp2<C> r0, r1, r2;
  // post-additions: chains of up to three carried terms are summed lazily and carried once
  fp2_mul<C>(s, fp2_add(a.c1, a.c2), fp2_add(b.c1, b.c2));
  r0 = fp2_carry(fp2_add_lazy(t0, fp2_mul_xi_lazy(fp2_carry(fp2_sub_lazy(fp2_sub_lazy(s, t1), t2)))));
  fp2_mul<C>(s, fp2_add(a.c0, a.c1), fp2_add(b.c0, b.c1));
  r1 = fp2_carry(fp2_add_lazy(fp2_carry(fp2_sub_lazy(fp2_sub_lazy(s, t0), t1)), fp2_mul_xi_lazy(t2)));
  fp2_mul<C>(s, fp2_add(a.c0, a.c2), fp2_add(b.c0, b.c2));
  r2 = fp2_carry(fp2_add_lazy(fp2_carry(fp2_sub_lazy(fp2_sub_lazy(s, t0), t2)), t1));
  r.c0 = r0;
  r.c1 = r1;
  r.c2 = r2;
}
template <class C>
ELP_FP6 void fp6_sqr(Fp6<C>& r, const Fp6<C>& a) {  // CH-SQR2: 2 mul + 3 sqr in Fp2
  Fp2<C> s0, s1, s2, s3, s4;
  fp2_sqr<C>(s0, a.c0);
  fp2_mul<C>(s1, a.c0, a.c1);
  s1 = fp2_dbl(s1);
  fp2_sqr<C>(s2, fp2_add(fp2_sub(a.c0, a.c1), a.c2));
  fp2_mul<C>(s3, a.c1, a.c2);
  s3 = fp2_dbl(s3);
  fp2_sqr<C>(s4, a.c2);
  r.c0 = fp2_add(s0, fp2_mul_xi(s3));
  r.c1 = fp2_add(s1, fp2_mul_xi(s4));
  r.c2 = fp2_sub(fp2_add(fp2_add(s1, s2), s3), fp2_add(s0, s4));
}
// a * (b0 + b1 v)
template <class C>
ELP_FP6 void fp6_mul_by_01(Fp6<C>& r, const Fp6<C>& a, const Fp2<C>& b0, const Fp2<C>& b1) {  // 5 Fp2 mul; operands carried
  if constexpr (fp_roomy<C>()) {
    Fp2<C> t0, t1, s, r0, r1, r2;
    fp2_mul<C>(t0, a.c0, b0);
    fp2_mul<C>(t1, a.c1, b1);
    fp2_mul<C>(s, fp2_add_lazy(a.c1, a.c2), b1);                                                  // 2 x 1
    r0 = fp2_carry_fast(fp2_add_lazy(t0, fp2_mul_xi_lazy(fp2_sub_lazy(s, t1))));                  // 1 + 2 * 2
    fp2_mul<C>(s, fp2_add_lazy(a.c0, a.c1), fp2_add_lazy(b0, b1));                                // 2 x 2
    r1 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(s, t0), t1));                                   // 3
    fp2_mul<C>(s, a.c2, b0);
    r2 = fp2_carry_fast(fp2_add_lazy(s, t1));                                                     // 2
    r.c0 = r0;
    r.c1 = r1;
    r.c2 = r2;
    return;
  }
  Fp2<C> t0, t1, s, r0, r1, r2;
  fp2_mul<C>(t0, a.c0, b0);
  fp2_mul<C>(t1, a.c1, b1);
  fp2_mul<C>(s, fp2_add(a.c1, a.c2), b1);  // a1 b1 + a2 b1
  r0 = fp2_add(t0, fp2_mul_xi(fp2_sub(s, t1)));
  fp2_mul<C>(s, fp2_add(a.c0, a.c1), fp2_add(b0, b1));
  r1 = fp2_sub(fp2_sub(s, t0), t1);
  fp2_mul<C>(s, a.c2, b0);
  r2 = fp2_add(s, t1);
  r.c0 = r0;
  r.c1 = r1;
  r.c2 = r2;
}
template <class C>
ELP_FP6 void fp6_mul_by_fp2(Fp6<C>& r, const Fp6<C>& a, const Fp2<C>& b) {  // 3 Fp2 mul
  fp2_mul<C>(r.c0, a.c0, b);
  fp2_mul<C>(r.c1, a.c1, b);
  fp2_mul<C>(r.c2, a.c2, b);
}
template <class C>
ELP_HEAVY void fp6_inv(Fp6<C>& r, const Fp6<C>& x) {
  Fp2<C> A, B, Cc, t, F;
  fp2_sqr<C>(A, x.c0);
  fp2_mul<C>(t, x.c1, x.c2);
  A = fp2_sub(A, fp2_mul_xi(t));
  fp2_sqr<C>(B, x.c2);
  B = fp2_mul_xi(B);
  fp2_mul<C>(t, x.c0, x.c1);
  B = fp2_sub(B, t);
  fp2_sqr<C>(Cc, x.c1);
  fp2_mul<C>(t, x.c0, x.c2);
  Cc = fp2_sub(Cc, t);
  Fp2<C> u, v;
  fp2_mul<C>(u, x.c2, B);
  fp2_mul<C>(v, x.c1, Cc);
  fp2_mul<C>(F, x.c0, A);
  F = fp2_add(F, fp2_mul_xi(fp2_add(u, v)));
  Fp2<C> Fi;
  fp2_inv<C>(Fi, F);
  fp2_mul<C>(r.c0, A, Fi);
  fp2_mul<C>(r.c1, B, Fi);
  fp2_mul<C>(r.c2, Cc, Fi);
}

// ------------------------------------------------------------------ Fp12
template <class C>
struct Fp12 {
  Fp6<C> c0, c1;
};
template <class C>
ELP_INL void fp12_set_one(Fp12<C>& r) {
  r.c0.c0 = fp2_one<C>();
  r.c0.c1 = fp2_zero<C>();
  r.c0.c2 = fp2_zero<C>();
  r.c1.c0 = fp2_zero<C>();
  r.c1.c1 = fp2_zero<C>();
  r.c1.c2 = fp2_zero<C>();
}
template <class C>
ELP_INL bool fp12_is_one(const Fp12<C>& a) {
  return fp2_eq(a.c0.c0, fp2_one<C>()) && fp2_is_zero(a.c0.c1) && fp2_is_zero(a.c0.c2) && fp2_is_zero(a.c1.c0) &&
         fp2_is_zero(a.c1.c1) && fp2_is_zero(a.c1.c2);
}
template <class C>
ELP_INL bool fp12_eq(const Fp12<C>& a, const Fp12<C>& b) {
  return fp2_eq(a.c0.c0, b.c0.c0) && fp2_eq(a.c0.c1, b.c0.c1) && fp2_eq(a.c0.c2, b.c0.c2) && fp2_eq(a.c1.c0, b.c1.c0) &&
         fp2_eq(a.c1.c1, b.c1.c1) && fp2_eq(a.c1.c2, b.c1.c2);
}
template <class C>
ELP_HEAVY void fp12_mul(Fp12<C>& r, const Fp12<C>& a, const Fp12<C>& b) {  // 3 Fp6 mul
  if constexpr (fp_roomy<C>()) {
    Fp6<C> t0, t1, s, u;
    fp6_mul<C>(t0, a.c0, b.c0);
    fp6_mul<C>(t1, a.c1, b.c1);
    fp6_add(s, a.c0, a.c1);               // carried: fp6_mul sums its operands lazily once more
    fp6_add(u, b.c0, b.c1);
    fp6_mul<C>(s, s, u);
    r.c1.c0 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(s.c0, t0.c0), t1.c0));
    r.c1.c1 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(s.c1, t0.c1), t1.c1));
    r.c1.c2 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(s.c2, t0.c2), t1.c2));
    r.c0.c0 = fp2_carry_fast(fp2_add_lazy(t0.c0, fp2_mul_xi_lazy(t1.c2)));      // t0 + v t1
    r.c0.c1 = fp2_carry_fast(fp2_add_lazy(t0.c1, t1.c0));
    r.c0.c2 = fp2_carry_fast(fp2_add_lazy(t0.c2, t1.c1));
    return;
  }
  Fp6<C> t0, t1, s, u;
  fp6_mul<C>(t0, a.c0, b.c0);
  fp6_mul<C>(t1, a.c1, b.c1);
  fp6_add(s, a.c0, a.c1);
  fp6_add(u, b.c0, b.c1);
  fp6_mul<C>(s, s, u);
  fp6_sub(s, s, t0);
  fp6_sub(r.c1, s, t1);
  fp6_mul_by_v(t1, t1);
  fp6_add(r.c0, t0, t1);
}
template <class C>
ELP_INL void fp12_sqr_inl(Fp12<C>& r, const Fp12<C>& a) {  // complex squaring, 2 Fp6 mul
  if constexpr (fp_roomy<C>()) {
    Fp6<C> t, s, u;
    fp6_mul<C>(t, a.c0, a.c1);                                                   // a0 a1
    fp6_add(s, a.c0, a.c1);                                                       // a0 + a1, carried
    u.c0 = fp2_carry_fast(fp2_add_lazy(a.c0.c0, fp2_mul_xi_lazy(a.c1.c2)));       // a0 + v a1, carried
    u.c1 = fp2_add(a.c0.c1, a.c1.c0);
    u.c2 = fp2_add(a.c0.c2, a.c1.c1);
    fp6_mul<C>(s, s, u);                                                          // a0^2 + v a1^2 + (1+v) a0 a1
    Fp2<C> r00 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(s.c0, t.c0), fp2_mul_xi_lazy(t.c2)));   // s - t - v t: 1 + 1 + 2
    Fp2<C> r01 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(s.c1, t.c1), t.c0));
    Fp2<C> r02 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(s.c2, t.c2), t.c1));
    r.c0.c0 = r00;
    r.c0.c1 = r01;
    r.c0.c2 = r02;
    r.c1.c0 = fp2_dbl(t.c0);
    r.c1.c1 = fp2_dbl(t.c1);
    r.c1.c2 = fp2_dbl(t.c2);
    return;
  }
  Fp6<C> t, s, u;
  fp6_mul<C>(t, a.c0, a.c1);           // a0 a1
  fp6_add(s, a.c0, a.c1);               // a0 + a1
  fp6_mul_by_v(u, a.c1);
  fp6_add(u, u, a.c0);                  // a0 + v a1
  fp6_mul<C>(s, s, u);                  // (a0+a1)(a0+v a1) = a0^2 + v a1^2 + (1+v) a0 a1
  fp6_sub(s, s, t);
  fp6_mul_by_v(u, t);
  fp6_sub(r.c0, s, u);
  fp6_add(r.c1, t, t);
}
template <class C>
ELP_HEAVY void fp12_sqr(Fp12<C>& r, const Fp12<C>& a) {
  fp12_sqr_inl<C>(r, a);
}
template <class C>
ELP_INL void fp12_conj(Fp12<C>& r, const Fp12<C>& a) {  // a^(p^6)
  r.c0 = a.c0;
  fp6_neg(r.c1, a.c1);
}
template <class C>
ELP_HEAVY void fp12_inv(Fp12<C>& r, const Fp12<C>& a) {
  Fp6<C> t0, t1;
  fp6_sqr<C>(t0, a.c0);
  fp6_sqr<C>(t1, a.c1);
  fp6_mul_by_v(t1, t1);
  fp6_sub(t0, t0, t1);
  fp6_inv<C>(t1, t0);
  fp6_mul<C>(r.c0, a.c0, t1);
  fp6_mul<C>(t0, a.c1, t1);
  fp6_neg(r.c1, t0);
}
// a^(p^n), n = 1, 2, 3.  Coefficient of v^i w^j (= w^(2i+j)) is conjugated n times and scaled by gamma_{n,2i+j}.
template <class C>
ELP_HEAVY void fp12_frob(Fp12<C>& r, const Fp12<C>& a, int n) {
  const bool cj = (n & 1) != 0;
  Fp2<C> t;
  t = a.c0.c0;
  r.c0.c0 = cj ? fp2_conj(t) : t;
  t = a.c1.c0;
  fp2_mul<C>(r.c1.c0, cj ? fp2_conj(t) : t, fp2_frob_coeff<C>(n, 1));
  t = a.c0.c1;
  fp2_mul<C>(r.c0.c1, cj ? fp2_conj(t) : t, fp2_frob_coeff<C>(n, 2));
  t = a.c1.c1;
  fp2_mul<C>(r.c1.c1, cj ? fp2_conj(t) : t, fp2_frob_coeff<C>(n, 3));
  t = a.c0.c2;
  fp2_mul<C>(r.c0.c2, cj ? fp2_conj(t) : t, fp2_frob_coeff<C>(n, 4));
  t = a.c1.c2;
  fp2_mul<C>(r.c1.c2, cj ? fp2_conj(t) : t, fp2_frob_coeff<C>(n, 5));
}

// Sparse product used by the Miller loop.
//  D-type twist: line = a + b w + c w^3   -> c0 = (a,0,0), c1 = (b,c,0)      ("034")
//  M-type twist: line = a + b w^2 + c w^3 -> c0 = (a,b,0), c1 = (0,c,0)      ("014")
template <class C>
ELP_INL void fp12_mul_by_line_inl(Fp12<C>& f, const Fp2<C>& a, const Fp2<C>& b, const Fp2<C>& c) {
  if constexpr (fp_roomy<C>() && C::TWIST_D) {
    Fp6<C> t0, t1, t2, s;
    fp6_mul_by_fp2<C>(t0, f.c0, a);          // f0 * a
    fp6_mul_by_01<C>(t1, f.c1, b, c);        // f1 * (b + c v)
    fp6_add(s, f.c0, f.c1);                  // carried
    fp6_mul_by_01<C>(t2, s, fp2_add(a, b), c);
    f.c1.c0 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(t2.c0, t0.c0), t1.c0));
    f.c1.c1 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(t2.c1, t0.c1), t1.c1));
    f.c1.c2 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(t2.c2, t0.c2), t1.c2));
    f.c0.c0 = fp2_carry_fast(fp2_add_lazy(t0.c0, fp2_mul_xi_lazy(t1.c2)));      // t0 + v t1
    f.c0.c1 = fp2_carry_fast(fp2_add_lazy(t0.c1, t1.c0));
    f.c0.c2 = fp2_carry_fast(fp2_add_lazy(t0.c2, t1.c1));
    return;
  }
  if (C::TWIST_D) {
    Fp6<C> t0, t1, t2, s;
    fp6_mul_by_fp2<C>(t0, f.c0, a);          // f0 * a
    fp6_mul_by_01<C>(t1, f.c1, b, c);        // f1 * (b + c v)
    fp6_add(s, f.c0, f.c1);
    fp6_mul_by_01<C>(t2, s, fp2_add(a, b), c);  // (f0+f1)(a + b + c v)
    fp6_sub(t2, t2, t0);
    fp6_sub(f.c1, t2, t1);
    fp6_mul_by_v(t1, t1);
    fp6_add(f.c0, t0, t1);
  } else {
    Fp6<C> t0, t1, t2, s;
    fp6_mul_by_01<C>(t0, f.c0, a, b);        // f0 * (a + b v)
    // f1 * (c v): (x0,x1,x2) * c v = (xi x2 c, x0 c, x1 c)
    fp6_mul_by_fp2<C>(t1, f.c1, c);
    fp6_mul_by_v(t1, t1);
    fp6_add(s, f.c0, f.c1);
    fp6_mul_by_01<C>(t2, s, a, fp2_add(b, c));  // (f0+f1)(a + (b+c) v)
    fp6_sub(t2, t2, t0);
    fp6_sub(f.c1, t2, t1);
    fp6_mul_by_v(t1, t1);
    fp6_add(f.c0, t0, t1);
  }
}

template <class C>
ELP_HEAVY void fp12_mul_by_line(Fp12<C>& f, const Fp2<C>& a, const Fp2<C>& b, const Fp2<C>& c) {
  fp12_mul_by_line_inl<C>(f, a, b, c);
}

// f <- f * l1 * l2 for two D-type lines l_i = a_i + b_i w + c_i w^3 (a_i, b_i carried; c_i of magnitude <= 2): the two sparse elements
// are multiplied first (6 Fp2 products), their product has five non-zero Fp2 coefficients
//     L0 = (a1 a2 + xi c1 c2,  b1 b2,  b1 c2 + c1 b2),   L1 = (a1 b2 + a2 b1,  a1 c2 + a2 c1,  0)
// and f * (L0 + L1 w) costs 6 + 5 + 6 Fp2 products: 23 instead of 26 for two sparse products, and one pass over f instead of two.
// M-type lines l_i = a_i + b_i v + c_i v w (a_i of magnitude <= 2; b_i, c_i carried): the same six products give
//     L0 = (a1 a2 + xi c1 c2,  a1 b2 + a2 b1,  b1 b2),   L1 = (0,  a1 c2 + a2 c1,  b1 c2 + b2 c1) = v (y_ac + y_bc v)
// and f1 * L1 = v * (f1 * (y_ac + y_bc v)), again 6 + 5 + 6 products.
template <class C>
ELP_INL void fp12_mul_by_two_lines_m_inl(Fp12<C>& f, const Fp2<C>& a1_in, const Fp2<C>& b1, const Fp2<C>& c1, const Fp2<C>& a2_in,
                                         const Fp2<C>& b2, const Fp2<C>& c2) {
  static_assert(!C::TWIST_D && C::HEADROOM >= 14, "M-type twist over a field with lazy-sum headroom");
  const Fp2<C> a1 = fp2_carry_fast(a1_in), a2 = fp2_carry_fast(a2_in);
  Fp2<C> taa, tbb, tcc, tbc, tab, tac;
  fp2_mul<C>(taa, a1, a2);
  fp2_mul<C>(tbb, b1, b2);
  fp2_mul<C>(tcc, c1, c2);
  fp2_mul<C>(tbc, fp2_add_lazy(b1, c1), fp2_add_lazy(b2, c2));                   // 2 x 2
  fp2_mul<C>(tab, fp2_add_lazy(a1, b1), fp2_add_lazy(a2, b2));
  fp2_mul<C>(tac, fp2_add_lazy(a1, c1), fp2_add_lazy(a2, c2));
  const Fp2<C> yab = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(tab, taa), tbb));
  const Fp2<C> yac = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(tac, taa), tcc));
  const Fp2<C> ybc = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(tbc, tbb), tcc));
  Fp6<C> L0, L1s;                                                                // L1s = L0 + L1 (for the Karatsuba cross term)
  L0.c0 = fp2_carry_fast(fp2_add_lazy(taa, fp2_mul_xi_lazy(tcc)));               // 1 + 2
  L0.c1 = yab;
  L0.c2 = tbb;
  L1s.c0 = L0.c0;
  L1s.c1 = fp2_add(L0.c1, yac);
  L1s.c2 = fp2_add(L0.c2, ybc);
  Fp6<C> t0, u, t2, s;
  fp6_mul<C>(t0, f.c0, L0);
  fp6_mul_by_01<C>(u, f.c1, yac, ybc);                                           // t1 = f1 * L1 = v u = (xi u2, u0, u1)
  fp6_add(s, f.c0, f.c1);
  fp6_mul<C>(t2, s, L1s);
  const Fp2<C> xu2 = fp2_carry_fast(fp2_mul_xi_lazy(u.c2));
  f.c1.c0 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(t2.c0, t0.c0), xu2));
  f.c1.c1 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(t2.c1, t0.c1), u.c0));
  f.c1.c2 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(t2.c2, t0.c2), u.c1));
  f.c0.c0 = fp2_carry_fast(fp2_add_lazy(t0.c0, fp2_mul_xi_lazy(u.c1)));          // t0 + v t1 = t0 + (xi u1, xi u2, u0)
  f.c0.c1 = fp2_carry_fast(fp2_add_lazy(t0.c1, xu2));
  f.c0.c2 = fp2_carry_fast(fp2_add_lazy(t0.c2, u.c0));
}

template <class C>
ELP_INL void fp12_mul_by_two_lines_inl(Fp12<C>& f, const Fp2<C>& a1, const Fp2<C>& b1, const Fp2<C>& c1_in, const Fp2<C>& a2,
                                       const Fp2<C>& b2, const Fp2<C>& c2_in) {
  static_assert(C::TWIST_D && C::HEADROOM >= 14, "written for the D-type twist over the 29-bit field");
  const Fp2<C> c1 = fp2_carry_fast(c1_in), c2 = fp2_carry_fast(c2_in);
  Fp2<C> taa, tbb, tcc, tbc, tab, tac;
  fp2_mul<C>(taa, a1, a2);
  fp2_mul<C>(tbb, b1, b2);
  fp2_mul<C>(tcc, c1, c2);
  fp2_mul<C>(tbc, fp2_add_lazy(b1, c1), fp2_add_lazy(b2, c2));                   // 2 x 2
  fp2_mul<C>(tab, fp2_add_lazy(a1, b1), fp2_add_lazy(a2, b2));
  fp2_mul<C>(tac, fp2_add_lazy(a1, c1), fp2_add_lazy(a2, c2));
  Fp6<C> L0, L1s;                                                                // L1s = L0 + L1 (for the Karatsuba cross term)
  L0.c0 = fp2_carry_fast(fp2_add_lazy(taa, fp2_mul_xi_lazy(tcc)));               // 1 + 2
  L0.c1 = tbb;
  L0.c2 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(tbc, tbb), tcc));             // 3
  const Fp2<C> y0 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(tab, taa), tbb));
  const Fp2<C> y1 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(tac, taa), tcc));
  L1s.c0 = fp2_add(L0.c0, y0);
  L1s.c1 = fp2_add(L0.c1, y1);
  L1s.c2 = L0.c2;
  Fp6<C> t0, t1, t2, s;
  fp6_mul<C>(t0, f.c0, L0);
  fp6_mul_by_01<C>(t1, f.c1, y0, y1);
  fp6_add(s, f.c0, f.c1);
  fp6_mul<C>(t2, s, L1s);
  f.c1.c0 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(t2.c0, t0.c0), t1.c0));
  f.c1.c1 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(t2.c1, t0.c1), t1.c1));
  f.c1.c2 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(t2.c2, t0.c2), t1.c2));
  f.c0.c0 = fp2_carry_fast(fp2_add_lazy(t0.c0, fp2_mul_xi_lazy(t1.c2)));         // t0 + v t1
  f.c0.c1 = fp2_carry_fast(fp2_add_lazy(t0.c1, t1.c0));
  f.c0.c2 = fp2_carry_fast(fp2_add_lazy(t0.c2, t1.c1));
}

// Granger-Scott squaring for elements of the cyclotomic subgroup (after the easy part of the final exponentiation).
template <class C>
ELP_INL void fp12_cyc_sqr_inl(Fp12<C>& r, const Fp12<C>& a) {
  // view Fp12 as three Fp4 = Fp2[s]/(s^2 - xi):  (g0 + g1 s) with pairs (c0.c0,c1.c1), (c1.c0,c0.c2), (c0.c1,c1.c2)
  const Fp2<C>&z0 = a.c0.c0, &z4 = a.c0.c1, &z3 = a.c0.c2, &z2 = a.c1.c0, &z1 = a.c1.c1, &z5 = a.c1.c2;
  Fp2<C> t0, t1, t2, t3, t4, t5, tmp;
  if constexpr (fp_roomy<C>()) {
    // per Fp4 block: A0 = z0^2 + xi z1^2 and A1 = 2 z0 z1 carried once each (3 -> 1), then 3 A -+ 2 z summed lazily (5) straight
    // into the weak reduction, which accepts any int32 limbs and returns carried ones
    Fp2<C> n[6];
    const Fp2<C>* za[3] = {&z0, &z2, &z4};
    const Fp2<C>* zb[3] = {&z1, &z3, &z5};
    ELP_UNROLL
    for (int k = 0; k < 3; k++) {
      fp2_sqr<C>(t0, *za[k]);
      fp2_sqr<C>(t1, *zb[k]);
      fp2_sqr<C>(tmp, fp2_add(*za[k], *zb[k]));
      Fp2<C> A1 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(tmp, t0), t1));        // 2 za zb
      Fp2<C> A0 = fp2_carry_fast(fp2_add_lazy(t0, fp2_mul_xi_lazy(t1)));          // za^2 + xi zb^2
      n[2 * k] = A0;
      n[2 * k + 1] = A1;
    }
    // A = (n0, n1) from (z0, z1); B = (n2, n3) from (z2, z3); Cq = (n4, n5) from (z4, z5)
    auto three_minus = [](const Fp2<C>& A, const Fp2<C>& z) {   // 3 A - 2 z
      return fp2_sub_lazy(fp2_add_lazy(fp2_add_lazy(A, A), A), fp2_add_lazy(z, z));
    };
    auto three_plus = [](const Fp2<C>& A, const Fp2<C>& z) {    // 3 A + 2 z
      return fp2_add_lazy(fp2_add_lazy(fp2_add_lazy(A, A), A), fp2_add_lazy(z, z));
    };
    Fp2<C> o0 = three_minus(n[0], z0);                 // z0' = 3 A0 - 2 z0
    Fp2<C> o1 = three_plus(n[1], z1);                  // z1' = 3 A1 + 2 z1
    Fp2<C> xc1 = fp2_carry_fast(fp2_mul_xi_lazy(n[5]));
    Fp2<C> o2 = three_plus(xc1, z2);                   // z2' = 3 xi C1 + 2 z2
    Fp2<C> o3 = three_minus(n[4], z3);                 // z3' = 3 C0 - 2 z3
    Fp2<C> o4 = three_minus(n[2], z4);                 // z4' = 3 B0 - 2 z4
    Fp2<C> o5 = three_plus(n[3], z5);                  // z5' = 3 B1 + 2 z5
    fp2_reduce_weak(o0); fp2_reduce_weak(o1); fp2_reduce_weak(o2); fp2_reduce_weak(o3); fp2_reduce_weak(o4); fp2_reduce_weak(o5);
    r.c0.c0 = o0;
    r.c0.c1 = o4;
    r.c0.c2 = o3;
    r.c1.c0 = o2;
    r.c1.c1 = o1;
    r.c1.c2 = o5;
    return;
  }
  // (z0 + z1 s)^2 = (z0^2 + xi z1^2) + 2 z0 z1 s
  fp2_sqr<C>(t0, z0);
  fp2_sqr<C>(t1, z1);
  fp2_sqr<C>(tmp, fp2_add(z0, z1));
  t5 = fp2_sub(fp2_sub(tmp, t0), t1);          // 2 z0 z1
  t0 = fp2_add(t0, fp2_mul_xi(t1));            // z0^2 + xi z1^2
  Fp2<C> A0 = t0, A1 = t5;
  fp2_sqr<C>(t0, z2);
  fp2_sqr<C>(t1, z3);
  fp2_sqr<C>(tmp, fp2_add(z2, z3));
  t5 = fp2_sub(fp2_sub(tmp, t0), t1);
  t0 = fp2_add(t0, fp2_mul_xi(t1));
  Fp2<C> B0 = t0, B1 = t5;
  fp2_sqr<C>(t0, z4);
  fp2_sqr<C>(t1, z5);
  fp2_sqr<C>(tmp, fp2_add(z4, z5));
  t5 = fp2_sub(fp2_sub(tmp, t0), t1);
  t0 = fp2_add(t0, fp2_mul_xi(t1));
  Fp2<C> C0 = t0, C1 = t5;
  // z0' = 3 A0 - 2 z0 ; z1' = 3 A1 + 2 z1
  Fp2<C> n0 = fp2_add(fp2_dbl(fp2_sub(A0, z0)), A0);
  Fp2<C> n1 = fp2_add(fp2_dbl(fp2_add(A1, z1)), A1);
  // z2' = 3 xi C1 + 2 z2 ; z3' = 3 C0 - 2 z3
  Fp2<C> xc1 = fp2_mul_xi(C1);
  Fp2<C> n2 = fp2_add(fp2_dbl(fp2_add(xc1, z2)), xc1);
  Fp2<C> n3 = fp2_add(fp2_dbl(fp2_sub(C0, z3)), C0);
  // z4' = 3 B0 - 2 z4 ; z5' = 3 B1 + 2 z5
  Fp2<C> n4 = fp2_add(fp2_dbl(fp2_sub(B0, z4)), B0);
  Fp2<C> n5 = fp2_add(fp2_dbl(fp2_add(B1, z5)), B1);
  (void)t2; (void)t3; (void)t4;
  // every output depends linearly on the matching input coefficient (3 A - 2 z): without a reduction the magnitude
  // would double per squaring (62 consecutive squarings in fp12_exp_absz)
  fp2_reduce_weak(n0); fp2_reduce_weak(n1); fp2_reduce_weak(n2); fp2_reduce_weak(n3); fp2_reduce_weak(n4); fp2_reduce_weak(n5);
  r.c0.c0 = n0;
  r.c0.c1 = n4;
  r.c0.c2 = n3;
  r.c1.c0 = n2;
  r.c1.c1 = n1;
  r.c1.c2 = n5;
}

template <class C>
ELP_HEAVY void fp12_cyc_sqr(Fp12<C>& r, const Fp12<C>& a) {
  fp12_cyc_sqr_inl<C>(r, a);
}

// ---- Compressed squaring in the cyclotomic subgroup (Karabina).  With t = w, s = w^3 (s^2 = xi) an element is alpha = a + b t + c t^2 over
// Fp4 = Fp2[s]: a = z0 + z1 s, b = z2 + z3 s, c = z4 + z5 s (the pairs of fp12_cyc_sqr_inl).  In the Granger-Scott formulas above the new (b, c) depend on the
// old (b, c) only, so a run of squarings can carry FOUR Fp2 coefficients instead of six: 6 Fp2 squarings per step instead of 9.  The dropped block follows
// from the subgroup relation  a c = b^2 - conj(c)  (equate the true square with the Granger-Scott form):  a = (b^2 - conj(c)) conj(c) / N(c),
// N(c) = z4^2 - xi z5^2 in Fp2 -- one Fp2 inversion, shared by all the decompressions of an exponentiation (fp12_exp_u64 in pairing.h).
template <class C>
struct CycComp {
  Fp2<C> z2, z3, z4, z5;
};
template <class C>
ELP_INL void fp12_to_comp(CycComp<C>& r, const Fp12<C>& a) {
  r.z2 = a.c1.c0;
  r.z3 = a.c0.c2;
  r.z4 = a.c0.c1;
  r.z5 = a.c1.c2;
}
// (x0 + x1 s)^2 = (x0^2 + xi x1^2) + 2 x0 x1 s, both parts carried
template <class C>
ELP_INL void fp4_sqr(Fp2<C>& r0, Fp2<C>& r1, const Fp2<C>& x0, const Fp2<C>& x1) {
  Fp2<C> t0, t1, tmp;
  fp2_sqr<C>(t0, x0);
  fp2_sqr<C>(t1, x1);
  fp2_sqr<C>(tmp, fp2_add(x0, x1));
  if constexpr (fp_roomy<C>()) {
    r1 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(tmp, t0), t1));
    r0 = fp2_carry_fast(fp2_add_lazy(t0, fp2_mul_xi_lazy(t1)));
  } else {
    r1 = fp2_sub(fp2_sub(tmp, t0), t1);
    r0 = fp2_add(t0, fp2_mul_xi(t1));
  }
}
template <class C>
ELP_INL void cyc_comp_sqr_inl(CycComp<C>& r, const CycComp<C>& a) {
  Fp2<C> B0, B1, C0, C1;
  fp4_sqr<C>(B0, B1, a.z2, a.z3);
  fp4_sqr<C>(C0, C1, a.z4, a.z5);
  Fp2<C> o2, o3, o4, o5;
  if constexpr (fp_roomy<C>()) {
    auto three_minus = [](const Fp2<C>& A, const Fp2<C>& z) { return fp2_sub_lazy(fp2_add_lazy(fp2_add_lazy(A, A), A), fp2_add_lazy(z, z)); };
    auto three_plus = [](const Fp2<C>& A, const Fp2<C>& z) { return fp2_add_lazy(fp2_add_lazy(fp2_add_lazy(A, A), A), fp2_add_lazy(z, z)); };
    const Fp2<C> xc1 = fp2_carry_fast(fp2_mul_xi_lazy(C1));
    o2 = three_plus(xc1, a.z2);    // z2' = 3 xi C1 + 2 z2
    o3 = three_minus(C0, a.z3);    // z3' = 3 C0 - 2 z3
    o4 = three_minus(B0, a.z4);    // z4' = 3 B0 - 2 z4
    o5 = three_plus(B1, a.z5);     // z5' = 3 B1 + 2 z5
  } else {
    const Fp2<C> xc1 = fp2_mul_xi(C1);
    o2 = fp2_add(fp2_dbl(fp2_add(xc1, a.z2)), xc1);
    o3 = fp2_add(fp2_dbl(fp2_sub(C0, a.z3)), C0);
    o4 = fp2_add(fp2_dbl(fp2_sub(B0, a.z4)), B0);
    o5 = fp2_add(fp2_dbl(fp2_add(B1, a.z5)), B1);
  }
  fp2_reduce_weak(o2);   // every output depends linearly on the matching input (3 A -+ 2 z): keep the magnitude bounded, as in fp12_cyc_sqr_inl
  fp2_reduce_weak(o3);
  fp2_reduce_weak(o4);
  fp2_reduce_weak(o5);
  r.z2 = o2;
  r.z3 = o3;
  r.z4 = o4;
  r.z5 = o5;
}
template <class C>
ELP_HEAVY void cyc_comp_sqr(CycComp<C>& r, const CycComp<C>& a) {
  cyc_comp_sqr_inl<C>(r, a);
}
// n squarings in a row with the four coefficients held in registers (as single calls every step moves them through memory twice)
template <class C>
ELP_HEAVY void cyc_comp_sqr_n(CycComp<C>& a, int n) {
  CycComp<C> c = a;
  ELP_NOUNROLL
  for (int i = 0; i < n; i++) cyc_comp_sqr_inl<C>(c, c);
  a = c;
}
// N(c) = z4^2 - xi z5^2 (the Fp4/Fp2 norm of c): zero iff c = 0
template <class C>
ELP_INL Fp2<C> cyc_comp_norm(const CycComp<C>& a) {
  Fp2<C> s4, s5;
  fp2_sqr<C>(s4, a.z4);
  fp2_sqr<C>(s5, a.z5);
  return fp2_sub(s4, fp2_mul_xi(s5));
}
// the full element from its compressed form and ninv = 1 / N(c)
template <class C>
ELP_HEAVY void cyc_decompress(Fp12<C>& r, const CycComp<C>& a, const Fp2<C>& ninv) {
  Fp2<C> B0, B1;
  fp4_sqr<C>(B0, B1, a.z2, a.z3);
  const Fp2<C> n0 = fp2_sub(B0, a.z4), n1 = fp2_add(B1, a.z5);                       // b^2 - conj(c)
  // (n0 + n1 s)(z4 - z5 s) = (n0 z4 - xi n1 z5) + (n1 z4 - n0 z5) s
  Fp2<C> m0, m1, t;
  fp2_mul<C>(m0, n0, a.z4);
  fp2_mul<C>(t, n1, a.z5);
  m0 = fp2_sub(m0, fp2_mul_xi(t));
  fp2_mul<C>(m1, n1, a.z4);
  fp2_mul<C>(t, n0, a.z5);
  m1 = fp2_sub(m1, t);
  fp2_mul<C>(r.c0.c0, m0, ninv);     // z0
  fp2_mul<C>(r.c1.c1, m1, ninv);     // z1
  r.c1.c0 = a.z2;
  r.c0.c2 = a.z3;
  r.c0.c1 = a.z4;
  r.c1.c2 = a.z5;
}

}  // namespace elp
