// G1 = E(Fp): y^2 = x^3 + b and G2 = E'(Fp2): y^2 = x^3 + b' (sextic twist), Jacobian coordinates.
// Replaces mcl G1/G2 ::add/::sub/::mul (call sites: src/ps-verifier.cc:21-29,72-108,220-227,
// src/ps-signer.cc:35-52,83-94,121-143, src/ps-requester.cc:38-66,109-110,144-146,165-261).
#pragma once
#include "tower.h"

namespace elp {

// limb-for-limb copy between the field types of a curve and of its Paired<> twin (same modulus, same limbs)
template <class C, class B>
ELP_INL Fp<C> fp_cast(const Fp<B>& a) {
  static_assert(C::NL == B::NL && C::LB == B::LB, "same field expected");
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) r.v[i] = a.v[i];
  return r;
}
// a plain-layout Fp2 in memory -> the layout of C (paired: this lane's component only)
template <class C>
ELP_INL Fp2<C> fp2_from_mem(const Fp2<typename PairInfo<C>::Base>& m) {
  Fp2<C> r;
  if constexpr (is_paired<C>()) {
    r.c = fp_cast<C>(pair_odd() ? m.c1 : m.c0);
  } else {
    r = m;
  }
  return r;
}

// ... and back: this lane's component of every coefficient into a plain-layout value in memory (the two lanes of a pair write disjoint halves)
template <class C>
ELP_INL void fp2_to_mem(Fp2<typename PairInfo<C>::Base>& m, const Fp2<C>& r) {
  if constexpr (is_paired<C>()) {
    if (pair_odd()) m.c1 = fp_cast<typename PairInfo<C>::Base>(r.c);
    else m.c0 = fp_cast<typename PairInfo<C>::Base>(r.c);
  } else {
    m = r;
  }
}
template <class C>
ELP_INL void fp12_to_mem(Fp12<typename PairInfo<C>::Base>& m, const Fp12<C>& r) {
  fp2_to_mem<C>(m.c0.c0, r.c0.c0);
  fp2_to_mem<C>(m.c0.c1, r.c0.c1);
  fp2_to_mem<C>(m.c0.c2, r.c0.c2);
  fp2_to_mem<C>(m.c1.c0, r.c1.c0);
  fp2_to_mem<C>(m.c1.c1, r.c1.c1);
  fp2_to_mem<C>(m.c1.c2, r.c1.c2);
}
template <class C>
ELP_INL void fp12_from_mem(Fp12<C>& r, const Fp12<typename PairInfo<C>::Base>& m) {
  r.c0.c0 = fp2_from_mem<C>(m.c0.c0);
  r.c0.c1 = fp2_from_mem<C>(m.c0.c1);
  r.c0.c2 = fp2_from_mem<C>(m.c0.c2);
  r.c1.c0 = fp2_from_mem<C>(m.c1.c0);
  r.c1.c1 = fp2_from_mem<C>(m.c1.c1);
  r.c1.c2 = fp2_from_mem<C>(m.c1.c2);
}

template <class F>
struct Aff;
// ---- field-operation adaptors so that the group law is written once for Fp and Fp2.
// MemF: the adaptor of the plain (unpaired) layout, in which key tables and bases are kept in HBM whatever layout the kernel computes in.
template <class C>
struct F1 {
  typedef Fp<C> T;
  typedef C Curve;
  typedef F1<typename PairInfo<C>::Base> MemF;
  static constexpr bool IS_EXT = false;
  ELP_INL static T from_mem(const typename MemF::T& m) { return fp_cast<C>(m); }
  ELP_INL static T mul(const T& a, const T& b) { return fp_mul<C>(a, b); }
  ELP_INL static T sqr(const T& a) { return fp_sqr<C>(a); }
  ELP_INL static T mul_pair(const T& a, const T& b, const T& c, const T& d) {   // a*b + c*d
    if constexpr (C::HEADROOM >= 3) return fp_mul_pair<C>(a, b, c, d);
    else return fp_add(fp_mul<C>(a, b), fp_mul<C>(c, d));
  }
  ELP_INL static T add(const T& a, const T& b) { return fp_add(a, b); }
  ELP_INL static T sub(const T& a, const T& b) { return fp_sub(a, b); }
  ELP_INL static T dbl(const T& a) { return fp_dbl(a); }
  // lazy forms (no carry pass) where the field has the headroom for them, the carried forms otherwise; carry() closes a lazy chain
  ELP_INL static T addl(const T& a, const T& b) { return fp_roomy<C>() ? fp_add_lazy(a, b) : fp_add(a, b); }
  ELP_INL static T subl(const T& a, const T& b) { return fp_roomy<C>() ? fp_sub_lazy(a, b) : fp_sub(a, b); }
  ELP_INL static T dbll(const T& a) { return fp_roomy<C>() ? fp_add_lazy(a, a) : fp_dbl(a); }
  ELP_INL static T carry(T a) {
    if (fp_roomy<C>()) fp_carry_fast(a);
    return a;
  }
  ELP_INL static T neg(const T& a) { return fp_neg(a); }
  ELP_INL static bool is_zero(const T& a) { return fp_is_zero<C>(a); }
  ELP_INL static bool is_zero_exact(const T& a) { return fp_is_zero_exact(a); }
  ELP_INL static bool eq(const T& a, const T& b) { return fp_eq(a, b); }
  ELP_INL static T zero() { return fp_zero<C>(); }
  ELP_INL static T one() { return fp_one<C>(); }
  ELP_INL static T inv(const T& a) { return fp_inv<C>(a); }
  ELP_INL static T select(bool c, const T& a, const T& b) { return fp_select(c, a, b); }
  ELP_INL static T curve_b() {
    T b;
    ELP_LOAD_FP(b, C::curve_b(i_));
    return b;
  }
};
template <class C>
struct F2 {
  typedef Fp2<C> T;
  typedef C Curve;
  typedef F2<typename PairInfo<C>::Base> MemF;
  static constexpr bool IS_EXT = true;
  ELP_INL static T from_mem(const typename MemF::T& m) { return fp2_from_mem<C>(m); }
  ELP_INL static T mul(const T& a, const T& b) { return fp2_mulv<C>(a, b); }
  ELP_INL static T sqr(const T& a) { return fp2_sqrv<C>(a); }
  ELP_INL static T mul_pair(const T& a, const T& b, const T& c, const T& d) {   // a*b + c*d
    T r;
    fp2_mul_pair<C>(r, a, b, c, d);
    return r;
  }
  ELP_INL static T add(const T& a, const T& b) { return fp2_add(a, b); }
  ELP_INL static T sub(const T& a, const T& b) { return fp2_sub(a, b); }
  ELP_INL static T dbl(const T& a) { return fp2_dbl(a); }
  ELP_INL static T addl(const T& a, const T& b) { return fp_roomy<C>() ? fp2_add_lazy(a, b) : fp2_add(a, b); }
  ELP_INL static T subl(const T& a, const T& b) { return fp_roomy<C>() ? fp2_sub_lazy(a, b) : fp2_sub(a, b); }
  ELP_INL static T dbll(const T& a) { return fp_roomy<C>() ? fp2_add_lazy(a, a) : fp2_dbl(a); }
  ELP_INL static T carry(const T& a) { return fp_roomy<C>() ? fp2_carry_fast(a) : a; }
  ELP_INL static T neg(const T& a) { return fp2_neg(a); }
  ELP_INL static bool is_zero(const T& a) { return fp2_is_zero(a); }
  ELP_INL static bool is_zero_exact(const T& a) { return fp2_is_zero_exact(a); }
  ELP_INL static bool eq(const T& a, const T& b) { return fp2_eq(a, b); }
  ELP_INL static T zero() { return fp2_zero<C>(); }
  ELP_INL static T one() { return fp2_one<C>(); }
  ELP_INL static T inv(const T& a) {
    T r;
    fp2_inv<C>(r, a);
    return r;
  }
  ELP_INL static T select(bool c, const T& a, const T& b) { return fp2_select(c, a, b); }
  ELP_INL static T curve_b() {
    T b;
    ELP_LOAD_FP2(b, C::twist_b(c_, i_));
    return b;
  }
};

template <class F>
struct Aff {  // affine point; (0,0) encodes the point at infinity (never on the curve since b != 0)
  typename F::T x, y;
};
template <class F>
struct Jac {  // Jacobian: (X/Z^2, Y/Z^3); Z == 0 encodes infinity
  typename F::T X, Y, Z;
};

// a table / base entry (plain layout in HBM) -> the kernel's layout
template <class F>
ELP_INL Aff<F> aff_from_mem(const Aff<typename F::MemF>& m) {
  Aff<F> r;
  r.x = F::from_mem(m.x);
  r.y = F::from_mem(m.y);
  return r;
}
template <class F>
ELP_INL bool aff_is_inf(const Aff<F>& p) {
  return F::is_zero_exact(p.x) && F::is_zero_exact(p.y);
}
template <class F>
ELP_INL void aff_set_inf(Aff<F>& p) {
  p.x = F::zero();
  p.y = F::zero();
}
template <class F>
ELP_INL bool jac_is_inf(const Jac<F>& p) {   // infinity is always stored with a literally zero Z
  return F::is_zero_exact(p.Z);
}
template <class F>
ELP_INL void jac_set_inf(Jac<F>& p) {
  p.X = F::one();
  p.Y = F::one();
  p.Z = F::zero();
}
template <class F>
ELP_INL void jac_from_aff(Jac<F>& r, const Aff<F>& p) {
  if (aff_is_inf(p)) {
    jac_set_inf(r);
  } else {
    r.X = p.x;
    r.Y = p.y;
    r.Z = F::one();
  }
}
template <class F>
ELP_INL void jac_neg(Jac<F>& r, const Jac<F>& p) {
  r.X = p.X;
  r.Y = F::neg(p.Y);
  r.Z = p.Z;
}
template <class F>
ELP_INL void aff_neg(Aff<F>& r, const Aff<F>& p) {
  r.x = p.x;
  r.y = F::neg(p.y);
}
template <class F>
ELP_HEAVY bool aff_on_curve(const Aff<F>& p) {
  if (aff_is_inf(p)) return true;
  typename F::T l = F::sqr(p.y);
  typename F::T r = F::add(F::mul(F::sqr(p.x), p.x), F::curve_b());
  return F::eq(l, r);
}

// dbl-2009-l (a = 0) with the two squaring tricks undone (a square costs 0.9 products here and the tricks cost carried sums) and
// Y3 as one inner product:  A = X^2, B = Y^2, D = 4 X B, E = 3 A, X3 = E^2 - 2 D, Y3 = E (D - X3) - (4B)(2B), Z3 = 2 Y Z.
// Operands and results carried; the comments give limb magnitudes in carried units where sums stay lazy (roomy fields).
template <class F>
ELP_HEAVY void jac_dbl(Jac<F>& r, const Jac<F>& p);
template <class F>
ELP_INL void jac_dbl_inl(Jac<F>& r, const Jac<F>& p) {
  typedef typename F::T T;
  T A = F::sqr(p.X);
  T B = F::sqr(p.Y);
  T D = F::mul(F::dbll(F::dbll(p.X)), B);                      // 4 x 1
  T E = F::carry(F::addl(F::dbll(A), A));                      // 3 -> 1 (squared next)
  T Fq = F::sqr(E);
  T Z3 = F::mul(F::dbll(p.Y), p.Z);                            // 2 x 1
  T X3 = F::carry(F::subl(Fq, F::dbll(D)));                    // 3 -> 1
  T B4 = F::carry(F::dbll(F::dbll(B)));                        // 4 -> 1
  T Y3 = F::mul_pair(E, F::subl(D, X3), F::neg(B4), F::dbll(B));   // E (D - X3) - (4B)(2B): 1 x 2 + 1 x 2, one reduction, no C = B^2
  r.Y = Y3;
  r.X = X3;
  r.Z = Z3;  // Y == 0 never happens on prime-order curves; Z == 0 stays 0
}

template <class F>
ELP_HEAVY void jac_dbl(Jac<F>& r, const Jac<F>& p) {
  jac_dbl_inl<F>(r, p);
}

// madd-2007-bl: Jacobian + affine, 7M + 4S, exceptional cases included.  The test "H == 0 (mod p)" (P == +-Q) is free in
// this representation: H^2 is needed anyway, and a Montgomery product of two multiples of p is LITERALLY zero
// ((kp * lp + m p)/R with m = -kl p exactly), while H != 0 (mod p) gives H^2 != 0.  So no modular reduction or
// comparison is needed on the hot path.
template <class F, bool DBL_INLINE>
ELP_INL void jac_madd_inl_t(Jac<F>& r, const Jac<F>& p, const Aff<F>& q);
template <class F>
ELP_INL void jac_madd_inl(Jac<F>& r, const Jac<F>& p, const Aff<F>& q) {
  jac_madd_inl_t<F, false>(r, p, q);
}
template <class F, bool DBL_INLINE>      // DBL_INLINE: the exceptional doubling inlined too (kernels with a register bound of their own: a call brings the callee's allocation)
ELP_INL void jac_madd_inl_t(Jac<F>& r, const Jac<F>& p, const Aff<F>& q) {
  typedef typename F::T T;
  if (aff_is_inf(q)) {
    r = p;
    return;
  }
  if (jac_is_inf(p)) {
    r.X = q.x;
    r.Y = q.y;
    r.Z = F::one();
    return;
  }
  T Z1Z1 = F::sqr(p.Z);
  T U2 = F::mul(q.x, Z1Z1);
  T S2 = F::mul(F::mul(q.y, p.Z), Z1Z1);
  T H = F::carry(F::subl(U2, p.X));                            // 2 -> 1 (squared next)
  T rr = F::subl(S2, p.Y);                                     // 2
  T HH = F::sqr(H);
  if (F::is_zero_exact(HH)) {            // H == 0 (mod p): same x-coordinate
    if (F::is_zero_exact(F::sqr(F::carry(rr)))) {  // and same y: doubling
      if constexpr (DBL_INLINE) jac_dbl_inl<F>(r, p);
      else jac_dbl<F>(r, p);
    } else {
      jac_set_inf(r);
    }
    return;
  }
  rr = F::carry(F::dbll(rr));                                  // 4 -> 1
  T I = F::dbll(F::dbll(HH));                                  // 4
  T J = F::mul(H, I);                                          // 1 x 4
  T V = F::mul(p.X, I);
  T X3 = F::carry(F::subl(F::subl(F::sqr(rr), J), F::dbll(V)));             // 4 -> 1
  T Y3 = F::mul_pair(rr, F::subl(V, X3), F::neg(F::dbll(p.Y)), J);   // rr (V - X3) - 2 Y1 J: 1 x 2 + 2 x 1, one reduction
  T Z3 = F::mul(F::dbll(p.Z), H);                              // (Z + H)^2 - Z^2 - H^2 = 2 Z H
  r.X = X3;
  r.Y = Y3;
  r.Z = Z3;
}

template <class F>
ELP_HEAVY void jac_madd(Jac<F>& r, const Jac<F>& p, const Aff<F>& q) {
  jac_madd_inl<F>(r, p, q);
}

// add-2007-bl: Jacobian + Jacobian, 11M + 5S
template <class F>
ELP_INL void jac_add_inl(Jac<F>& r, const Jac<F>& p, const Jac<F>& q) {
  typedef typename F::T T;
  if (jac_is_inf(q)) {
    r = p;
    return;
  }
  if (jac_is_inf(p)) {
    r = q;
    return;
  }
  T Z1Z1 = F::sqr(p.Z);
  T Z2Z2 = F::sqr(q.Z);
  T U1 = F::mul(p.X, Z2Z2);
  T U2 = F::mul(q.X, Z1Z1);
  T S1 = F::mul(F::mul(p.Y, q.Z), Z2Z2);
  T S2 = F::mul(F::mul(q.Y, p.Z), Z1Z1);
  T H = F::subl(U2, U1);                                       // 2
  T rr = F::subl(S2, S1);                                      // 2
  T I = F::sqr(F::carry(F::dbll(H)));                          // (2H)^2, operand 4 -> 1
  if (F::is_zero_exact(I)) {             // H == 0 (mod p), see jac_madd
    if (F::is_zero_exact(F::sqr(F::carry(rr)))) {
      jac_dbl<F>(r, p);
    } else {
      jac_set_inf(r);
    }
    return;
  }
  rr = F::carry(F::dbll(rr));                                  // 4 -> 1
  T J = F::mul(H, I);                                          // 2 x 1
  T V = F::mul(U1, I);
  T X3 = F::carry(F::subl(F::subl(F::sqr(rr), J), F::dbll(V)));             // 4 -> 1
  T Y3 = F::mul_pair(rr, F::subl(V, X3), F::neg(F::dbll(S1)), J);   // rr (V - X3) - 2 S1 J, one reduction
  T Z3 = F::mul(F::mul(F::dbll(p.Z), q.Z), H);                 // ((Z1 + Z2)^2 - Z1Z1 - Z2Z2) H = 2 Z1 Z2 H; 1 x 2
  r.X = X3;
  r.Y = Y3;
  r.Z = Z3;
}

template <class F>
ELP_HEAVY void jac_add(Jac<F>& r, const Jac<F>& p, const Jac<F>& q) {
  jac_add_inl<F>(r, p, q);
}

// Jacobian -> affine given zinv = 1/Z (or anything when Z == 0)
template <class F>
ELP_INL void jac_to_aff_with_zinv(Aff<F>& r, const Jac<F>& p, const typename F::T& zinv) {
  if (jac_is_inf(p)) {
    aff_set_inf(r);
    return;
  }
  typename F::T zi2 = F::sqr(zinv);
  r.x = F::mul(p.X, zi2);
  r.y = F::mul(F::mul(p.Y, zi2), zinv);
}
template <class F>
ELP_HEAVY void jac_to_aff(Aff<F>& r, const Jac<F>& p) {
  if (jac_is_inf(p)) {
    aff_set_inf(r);
    return;
  }
  jac_to_aff_with_zinv<F>(r, p, F::inv(p.Z));
}

// ---- scalars: 256-bit little-endian limbs (values < r; any 256-bit value is handled correctly)
struct Scalar {
  u32 v[8];
};
ELP_INL int scalar_window(const Scalar& k, int bit, int w) {  // bits [bit, bit+w)
  int limb = bit >> 5, sh = bit & 31;
  u64 t = k.v[limb];
  if (limb + 1 < 8) t |= (u64)k.v[limb + 1] << 32;
  return (int)((t >> sh) & ((1u << w) - 1));
}
ELP_INL bool scalar_is_zero(const Scalar& k) {
  u32 t = 0;
  for (int i = 0; i < 8; i++) t |= k.v[i];
  return t == 0;
}

// Variable-base scalar multiplication: fixed 4-bit windows, uniform control flow across lanes
// (64 x (4 dbl + 1 table add)); the per-lane table lives in private memory.
template <class F>
ELP_HEAVY void jac_mul_var(Jac<F>& r, const Aff<F>& p, const Scalar& k, int nwin = 64) {   // nwin 4-bit windows (scalar < 2^(4 nwin))
  Jac<F> tbl[16];
  jac_set_inf(tbl[0]);
  jac_from_aff(tbl[1], p);
  ELP_NOUNROLL
  for (int i = 2; i < 16; i++) {
    if (i & 1)
      jac_madd<F>(tbl[i], tbl[i - 1], p);
    else
      jac_dbl<F>(tbl[i], tbl[i >> 1]);
  }
  Jac<F> acc;
  jac_set_inf(acc);
  ELP_NOUNROLL
  for (int w = nwin - 1; w >= 0; w--) {
    if (w != nwin - 1) {
      jac_dbl<F>(acc, acc);
      jac_dbl<F>(acc, acc);
      jac_dbl<F>(acc, acc);
      jac_dbl<F>(acc, acc);
    }
    int d = scalar_window(k, 4 * w, 4);
    jac_add<F>(acc, acc, tbl[d]);
  }
  r = acc;
}

// ---- GLV / GLS scalar decomposition (constants from tools/glv.py): k -> D signed sub-scalars with sum_i k_i lam^i == k (mod r).
// c_j = round(k * G_j / 2^256), k_i = [i == 0] k - sum_j c_j B_ji, all modulo 2^(32 NW) (the true values are far smaller).
template <class C>
ELP_INL Scalar scalar_mod_r(Scalar k) {   // k mod r for any 256-bit k (r > 2^253: at most 7 subtractions)
  ELP_NOUNROLL
  for (int it = 0; it < 8; it++) {
    u64 br = 0;
    u32 d[8];
    for (int i = 0; i < 8; i++) {
      u64 t = (u64)k.v[i] - C::rmod(i) - br;
      d[i] = (u32)t;
      br = (t >> 32) & 1;
    }
    if (br) break;
    for (int i = 0; i < 8; i++) k.v[i] = d[i];
  }
  return k;
}
template <int D, int NG, int NW, class GL>
ELP_HEAVY void lattice_split(const Scalar& k, u32 (*mag)[NW], bool* neg, GL lat) {
  u32 c[D][NW];
  for (int j = 0; j < D; j++) {
    u32 prod[8 + NG + 1];
    for (int i = 0; i < 8 + NG + 1; i++) prod[i] = 0;
    for (int a = 0; a < 8; a++) {
      u64 carry = 0;
      for (int b = 0; b < NG; b++) {
        u64 t = (u64)k.v[a] * lat.g(j, b) + prod[a + b] + carry;
        prod[a + b] = (u32)t;
        carry = t >> 32;
      }
      prod[a + NG] = (u32)carry;
    }
    u64 t = (u64)prod[7] + 0x80000000u;   // + 2^255 (rounding)
    u64 carry = t >> 32;
    for (int i = 8; i < 8 + NG + 1 && carry; i++) {
      t = (u64)prod[i] + carry;
      prod[i] = (u32)t;
      carry = t >> 32;
    }
    for (int l = 0; l < NW; l++) c[j][l] = (8 + l < 8 + NG + 1) ? prod[8 + l] : 0;
  }
  for (int i = 0; i < D; i++) {
    u32 acc[NW];
    for (int l = 0; l < NW; l++) acc[l] = (i == 0 && l < 8) ? k.v[l] : 0;
    for (int j = 0; j < D; j++) {
      u32 t[NW];
      for (int l = 0; l < NW; l++) t[l] = 0;
      for (int a = 0; a < NW; a++) {
        u64 carry = 0;
        for (int b = 0; a + b < NW; b++) {
          u64 x = (u64)c[j][a] * lat.b(j, i, b) + t[a + b] + carry;
          t[a + b] = (u32)x;
          carry = x >> 32;
        }
      }
      const bool add = lat.gneg(j) != lat.bneg(j, i);   // product c_j * B_ji negative -> subtracting it adds
      u64 cy = add ? 0 : 1;
      for (int l = 0; l < NW; l++) {
        u64 x = (u64)acc[l] + (add ? t[l] : ~t[l]) + cy;
        acc[l] = (u32)x;
        cy = x >> 32;
      }
    }
    neg[i] = (acc[NW - 1] >> 31) != 0;
    if (neg[i]) {
      u64 cy = 1;
      for (int l = 0; l < NW; l++) {
        u64 x = (u64)(~acc[l]) + cy;
        acc[l] = (u32)x;
        cy = x >> 32;
      }
    }
    for (int l = 0; l < NW; l++) mag[i][l] = acc[l];
  }
}
template <class C>
struct Glv1Lat {
  ELP_HD u32 g(int j, int l) const { return C::glv1_g(j, l); }
  ELP_HD bool gneg(int j) const { return C::glv1_gneg(j); }
  ELP_HD u32 b(int j, int i, int l) const { return C::glv1_b(j, i, l); }
  ELP_HD bool bneg(int j, int i) const { return C::glv1_bneg(j, i); }
};
template <class C>
struct Gls2Lat {
  ELP_HD u32 g(int j, int l) const { return C::gls2_g(j, l); }
  ELP_HD bool gneg(int j) const { return C::gls2_gneg(j); }
  ELP_HD u32 b(int j, int i, int l) const { return C::gls2_b(j, i, l); }
  ELP_HD bool bneg(int j, int i) const { return C::gls2_bneg(j, i); }
};
template <int NW>
ELP_INL int limbs_window(const u32* m, int bit, int w) {
  int limb = bit >> 5, sh = bit & 31;
  u64 t = m[limb];
  if (limb + 1 < NW) t |= (u64)m[limb + 1] << 32;
  return (int)((t >> sh) & ((1u << w) - 1));
}

// Signed 4-bit recoding of a sub-scalar magnitude m < 16^NWIN / 4: with K' = m + 0x88...8 (NWIN nibbles), m = sum_i (nib_i(K') - 8) 16^i
// and every digit lies in [-8, 7], so the table holds 1P .. 8P only (half the table, half the set-up additions).
template <int NW, int NWIN>
ELP_INL void limbs_add_eights(u32* m) {
  u64 cy = 0;
  for (int l = 0; l < NW; l++) {
    const int lo = 8 * l;                                   // first nibble of this limb
    u32 cst = 0;
    for (int q = 0; q < 8; q++)
      if (lo + q < NWIN) cst |= 8u << (4 * q);
    u64 x = (u64)m[l] + cst + cy;
    m[l] = (u32)x;
    cy = x >> 32;
  }
}

// ---- where the per-item tables of small multiples (1P .. 8P) live.
// In private memory a read with a lane-dependent index touches one 256-byte row per DISTINCT index in the wave (private memory is interleaved by lane), eight
// rows per loaded word: an 8x read amplification that is harmless while the rows sit in L2 and dominates once they do not (the variable-base multiplications
// ran 1.6-2x slower on a full chip than on an idle one, and at full speed when every lane used the same scalar: DESIGN.md section 5).  The verification
// kernels therefore give every lane a slice of a launch workspace in HBM (KeyCtx::vtab): entries are contiguous per lane, padded to 16 bytes and moved as
// 128-bit words, so a lookup costs each lane its own one or two cache lines.  PrivTab is the fallback (host twin, kernels without a workspace).
template <class F>
ELP_HD constexpr int vtab_entry_words() { return (int)(((sizeof(Aff<F>) / 4 + 3) / 4) * 4); }
template <class F>
struct PrivTab {
  const Aff<F>* p;
  ELP_INL Aff<F> operator()(int i) const { return p[i]; }
};
template <class F>
struct WsTab {
  const u32* base;                                         // 16-byte aligned, 8 entries of vtab_entry_words<F>() words
  ELP_INL Aff<F> operator()(int i) const {
    constexpr int EW = vtab_entry_words<F>();
    union {
      u32 w[EW];
      Aff<F> a;
    } u;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4* q = reinterpret_cast<const uint4*>(base + (size_t)i * EW);
    ELP_UNROLL
    for (int k = 0; k < EW / 4; k++) {
      const uint4 v = q[k];
      u.w[4 * k] = v.x;
      u.w[4 * k + 1] = v.y;
      u.w[4 * k + 2] = v.z;
      u.w[4 * k + 3] = v.w;
    }
#else
    for (int k = 0; k < EW; k++) u.w[k] = base[(size_t)i * EW + k];
#endif
    return u.a;
  }
};
template <class F>
ELP_INL void vtab_store(u32* base, int i, const Aff<F>& a) {
  constexpr int EW = vtab_entry_words<F>();
  union {
    u32 w[EW];
    Aff<F> a;
  } u;
  for (int k = 0; k < EW; k++) u.w[k] = 0;
  u.a = a;
#if defined(__HIP_DEVICE_COMPILE__)
  uint4* q = reinterpret_cast<uint4*>(base + (size_t)i * EW);
  ELP_UNROLL
  for (int k = 0; k < EW / 4; k++) q[k] = make_uint4(u.w[4 * k], u.w[4 * k + 1], u.w[4 * k + 2], u.w[4 * k + 3]);
#else
  for (int k = 0; k < EW; k++) base[(size_t)i * EW + k] = u.w[k];
#endif
}

// ---- GLV / GLS multiplication over AFFINE tables 1P .. 8P (mixed additions: 11 / 29 instead of 16 / 43 products).
// [k]P for P in the order-r subgroup of G1: k = k1 + k2 lam with phi(x, y) = (beta x, y) = [lam](x, y); one shared chain of 132 doublings,
// two table additions per 4-bit window (the second through phi).  [k]Q in G2: k = k0 + k1 lam + k2 lam^2 + k3 lam^3 with psi(Q) = [lam]Q,
// lam = p mod r (psi = twist o Frobenius o untwist); 68 shared doublings, four table additions per window.  The caller builds the
// multiples with jac_multiples8 and makes them affine with one shared inversion (g1_mul_glv / g2_mul_gls / verify_id_nizk in pipeline.h).
template <class F>
ELP_HEAVY void jac_multiples8(Jac<F>* t, const Aff<F>& p) {   // t[i] = (i + 1) P
  jac_from_aff(t[0], p);
  jac_dbl<F>(t[1], t[0]);
  jac_madd<F>(t[2], t[1], p);
  jac_dbl<F>(t[3], t[1]);
  jac_madd<F>(t[4], t[3], p);
  jac_dbl<F>(t[5], t[2]);
  jac_madd<F>(t[6], t[5], p);
  jac_dbl<F>(t[7], t[3]);
}
// The entry of the NEXT (window, sub-scalar) step is loaded before the current addition is computed (its index only depends on the scalar), so the
// latency of the lookup -- an HBM round trip once the tables are out of L2 -- is covered by the ~3 000 instructions of a mixed addition.
template <class C, class Tab>
ELP_HEAVY void g1_mul_glv_with(Jac<F1<C>>& r, const Tab& tab, const Scalar& k_in) {
  typedef F1<C> F;
  u32 m[2][5];
  bool neg[2];
  lattice_split<2, 5, 5>(scalar_mod_r<C>(k_in), m, neg, Glv1Lat<C>());
  limbs_add_eights<5, 33>(m[0]);
  limbs_add_eights<5, 33>(m[1]);
  Fp<C> beta;
  ELP_LOAD_FP(beta, C::glv_beta(i_));
  Jac<F> acc;
  jac_set_inf(acc);
  int dg = limbs_window<5>(m[0], 4 * 32, 4) - 8;
  Aff<F> t = tab(dg == 0 ? 0 : (dg < 0 ? -dg : dg) - 1);
  ELP_NOUNROLL
  for (int step = 0; step < 66; step++) {
    const int w = 32 - (step >> 1), j = step & 1;
    if (j == 0 && w != 32) {
      ELP_NOUNROLL
      for (int d = 0; d < 4; d++) jac_dbl_inl<F>(acc, acc);
    }
    int dgn = 0;
    Aff<F> tn = t;
    if (step + 1 < 66) {
      const int wn = 32 - ((step + 1) >> 1), jn = (step + 1) & 1;
      dgn = limbs_window<5>(m[jn], 4 * wn, 4) - 8;
      tn = tab(dgn == 0 ? 0 : (dgn < 0 ? -dgn : dgn) - 1);
    }
    if (dg != 0 && !aff_is_inf(t)) {
      if (j == 1) t.x = fp_mul<C>(t.x, beta);
      if (neg[j] != (dg < 0)) t.y = fp_neg(t.y);
      jac_madd_inl<F>(acc, acc, t);
    }
    dg = dgn;
    t = tn;
  }
  r = acc;
}
// [a]P + [b]phi(P) for two 64-bit multipliers given as such (a = k[0..1], b = k[2..3]): the multiplier a + b lam of aggregated verification, drawn
// directly in decomposed form -- the map (a, b) -> a + b lam mod r is injective on [0, 2^64)^2 (the shortest vector of the GLV lattice has sup-norm
// 2^126), so 128 random bits still select one of 2^128 distinct multipliers -- and walked with 64 shared doublings instead of 128.  Same table of
// affine multiples 1P .. 8P, same signed 4-bit windows and one-step look-ahead as g1_mul_glv_with.
template <class C, class Tab>
ELP_HEAVY void g1_mul_pair64_with(Jac<F1<C>>& r, const Tab& tab, const Scalar& k) {
  typedef F1<C> F;
  u32 m[2][3];
  for (int j = 0; j < 2; j++) {
    m[j][0] = k.v[2 * j];
    m[j][1] = k.v[2 * j + 1];
    m[j][2] = 0;
    limbs_add_eights<3, 17>(m[j]);
  }
  Fp<C> beta;
  ELP_LOAD_FP(beta, C::glv_beta(i_));
  Jac<F> acc;
  jac_set_inf(acc);
  int dg = limbs_window<3>(m[0], 4 * 16, 4) - 8;
  Aff<F> t = tab(dg == 0 ? 0 : (dg < 0 ? -dg : dg) - 1);
  ELP_NOUNROLL
  for (int step = 0; step < 34; step++) {
    const int w = 16 - (step >> 1), j = step & 1;
    if (j == 0 && w != 16) {
      ELP_NOUNROLL
      for (int d = 0; d < 4; d++) jac_dbl_inl<F>(acc, acc);
    }
    int dgn = 0;
    Aff<F> tn = t;
    if (step + 1 < 34) {
      const int wn = 16 - ((step + 1) >> 1), jn = (step + 1) & 1;
      dgn = limbs_window<3>(m[jn], 4 * wn, 4) - 8;
      tn = tab(dgn == 0 ? 0 : (dgn < 0 ? -dgn : dgn) - 1);
    }
    if (dg != 0 && !aff_is_inf(t)) {
      if (j == 1) t.x = fp_mul<C>(t.x, beta);
      if (dg < 0) t.y = fp_neg(t.y);
      jac_madd_inl<F>(acc, acc, t);
    }
    dg = dgn;
    t = tn;
  }
  r = acc;
}
template <class C>
ELP_INL void g1_mul_glv_tab(Jac<F1<C>>& r, const Aff<F1<C>>* tab, const Scalar& k_in) {
  g1_mul_glv_with<C, PrivTab<F1<C>>>(r, PrivTab<F1<C>>{tab}, k_in);
}
// the table of g2_mul_gls_with in two pieces: d k (8 entries) and psi^j (d k), j = 1, 2, 3 (24 entries, entry (j - 1) * 8 + d - 1)
template <class F>
struct WsTabPsi {
  const u32* base;
  const u32* psi;
  ELP_INL Aff<F> operator()(int j, int i) const { return j == 0 ? WsTab<F>{base}(i) : WsTab<F>{psi}((j - 1) * 8 + i); }
};
// psi^j on an affine point of the twist: (conj^j x * gx_j, conj^j y * gy_j), j = 1, 2, 3
template <class C>
ELP_INL void g2_psi_aff(Aff<F2<C>>& t, int j) {
  Fp2<C> gx, gy;
  if (j == 1) {
    ELP_LOAD_FP2(gx, C::g2frob(1, 0, c_, i_));
    ELP_LOAD_FP2(gy, C::g2frob(1, 1, c_, i_));
  } else if (j == 2) {
    ELP_LOAD_FP2(gx, C::g2frob(2, 0, c_, i_));
    ELP_LOAD_FP2(gy, C::g2frob(2, 1, c_, i_));
  } else {
    ELP_LOAD_FP2(gx, C::g2frob(3, 0, c_, i_));
    ELP_LOAD_FP2(gy, C::g2frob(3, 1, c_, i_));
  }
  if (j & 1) {
    t.x = fp2_conj(t.x);
    t.y = fp2_conj(t.y);
  }
  fp2_mul<C>(t.x, t.x, gx);
  fp2_mul<C>(t.y, t.y, gy);
}
// PSI: `tab(j, i)` delivers psi^j ((i + 1) P) (WsTabPsi); otherwise `tab(i)` delivers (i + 1) P and psi^j is applied at every addition that needs it
template <class C, class Tab, bool PSI = false>
ELP_HEAVY void g2_mul_gls_with(Jac<F2<C>>& r, const Tab& tab, const Scalar& k_in) {
  typedef F2<C> F;
  u32 m[4][3];
  bool neg[4];
  lattice_split<4, 7, 3>(scalar_mod_r<C>(k_in), m, neg, Gls2Lat<C>());
  for (int j = 0; j < 4; j++) limbs_add_eights<3, 17>(m[j]);
  Jac<F> acc;
  jac_set_inf(acc);
  auto fetch = [&](int jj, int d) -> Aff<F> {
    const int e = d == 0 ? 0 : (d < 0 ? -d : d) - 1;
    if constexpr (PSI) return tab(jj, e);
    else return tab(e);
  };
  int dg = limbs_window<3>(m[0], 4 * 16, 4) - 8;
  Aff<F> t = fetch(0, dg);
  ELP_NOUNROLL
  for (int step = 0; step < 68; step++) {
    const int w = 16 - (step >> 2), j = step & 3;
    if (j == 0 && w != 16) {
      ELP_NOUNROLL
      for (int d = 0; d < 4; d++) jac_dbl_inl<F>(acc, acc);
    }
    int dgn = 0;
    Aff<F> tn = t;
    if (step + 1 < 68) {
      const int wn = 16 - ((step + 1) >> 2), jn = (step + 1) & 3;
      dgn = limbs_window<3>(m[jn], 4 * wn, 4) - 8;
      tn = fetch(jn, dgn);
    }
    if (dg != 0 && !aff_is_inf(t)) {
      if constexpr (!PSI) {
        if (j != 0) g2_psi_aff<C>(t, j);
      }
      if (neg[j] != (dg < 0)) t.y = fp2_neg(t.y);
      jac_madd_inl<F>(acc, acc, t);
    }
    dg = dgn;
    t = tn;
  }
  r = acc;
}
// ONE dimension of the same multiplication: r = [m_j] psi^j(P) with the digits of g2_mul_gls_with, so that the sum over j = 0 .. 3 is [k]P.  Four lanes of an item run one
// dimension each (64 doublings + 17 additions instead of 64 + 68) and add their results (the G2 job of the smallest batches, round 5: elpasso_impl.h vid_job_g2_quad).
template <class C, class Tab>
ELP_HEAVY void g2_mul_gls_dim(Jac<F2<C>>& r, const Tab& tab, const Scalar& k_in, int j) {
  typedef F2<C> F;
  u32 m[4][3];
  bool neg[4];
  lattice_split<4, 7, 3>(scalar_mod_r<C>(k_in), m, neg, Gls2Lat<C>());
  for (int q = 0; q < 4; q++) limbs_add_eights<3, 17>(m[q]);
  u32 mj[3] = {0, 0, 0};
  bool negj = false;
  for (int q = 0; q < 4; q++)
    if (q == j) {
      for (int i = 0; i < 3; i++) mj[i] = m[q][i];
      negj = neg[q];
    }
  Jac<F> acc;
  jac_set_inf(acc);
  auto fetch = [&](int d) -> Aff<F> { return tab(j, d == 0 ? 0 : (d < 0 ? -d : d) - 1); };
  int dg = limbs_window<3>(mj, 4 * 16, 4) - 8;
  Aff<F> t = fetch(dg);
  ELP_NOUNROLL
  for (int w = 16; w >= 0; w--) {
    if (w != 16) {
      ELP_NOUNROLL
      for (int d = 0; d < 4; d++) jac_dbl_inl<F>(acc, acc);
    }
    int dgn = 0;
    Aff<F> tn = t;
    if (w > 0) {
      dgn = limbs_window<3>(mj, 4 * (w - 1), 4) - 8;
      tn = fetch(dgn);
    }
    if (dg != 0 && !aff_is_inf(t)) {
      if (negj != (dg < 0)) t.y = fp2_neg(t.y);
      jac_madd_inl<F>(acc, acc, t);
    }
    dg = dgn;
    t = tn;
  }
  r = acc;
}
template <class C>
ELP_INL void g2_mul_gls_tab(Jac<F2<C>>& r, const Aff<F2<C>>* tab, const Scalar& k_in) {
  g2_mul_gls_with<C, PrivTab<F2<C>>>(r, PrivTab<F2<C>>{tab}, k_in);
}

// Fixed-base tables with SIGNED digits: for base B and window width W, entry [j][d-1] = d * 2^(W j) * B (affine), d = 1 .. 2^(W-1),
// j = 0 .. ceil(256/W)-1 -- half the entries of an unsigned table.  A scalar (reduced mod r first, so that the top window cannot overflow)
// is recoded on the fly into digits in (-2^(W-1), 2^(W-1)]:  d_j = window_j + carry;  d_j > 2^(W-1)  =>  d_j -= 2^W, carry into window j+1;
// a negative digit adds the NEGATED entry |d_j|.  Accumulating a scalar costs ceil(256/W) mixed additions and no doublings.
ELP_HD constexpr int fixed_base_entries(int W) { return 1 << (W - 1); }
ELP_INL int fixed_base_digit(const Scalar& k, int j, int W, int& carry) {   // call with j = 0, 1, 2, ... in order, carry starting at 0
  const int bit = j * W;
  int d = scalar_window(k, bit, (bit + W <= 256) ? W : 256 - bit) + carry;
  carry = d > (1 << (W - 1)) ? 1 : 0;
  if (carry) d -= 1 << W;
  return d;
}
template <class F>
ELP_HEAVY void jac_acc_fixed(Jac<F>& acc, const Aff<typename F::MemF>* table, int W, const Scalar& k_in, u32* hot = nullptr) {
  const int nwin = (256 + W - 1) / W;
  const int per = fixed_base_entries(W);
  const Scalar k = scalar_mod_r<typename F::Curve>(k_in);
  Jac<F>* ah = hot_as<Jac<F>, typename F::Curve>(hot);   // the running sum lives in the hot slot while the windows are added
  Jac<F>& a = ah ? *ah : acc;
  if (ah) a = acc;
  int carry = 0;
  // The entries are random reads of a table far larger than any cache (one HBM round trip each, with nothing else in the wave to hide it): the entry of
  // the NEXT window is requested before the current addition is computed.  Digit 0 reads entry 0 of the window and ignores it.
  if constexpr (is_paired<typename F::Curve>()) {
    // paired kernels (256 registers per lane): the look-ahead costs more in spills than it hides (measured: +1.5 %); plain loop
    ELP_NOUNROLL
    for (int j = 0; j < nwin; j++) {
      const int d = fixed_base_digit(k, j, W, carry);
      if (d != 0) {
        Aff<F> e = aff_from_mem<F>(table[(size_t)j * per + ((d < 0 ? -d : d) - 1)]);
        if (d < 0) e.y = F::neg(e.y);
        if (F::IS_EXT) jac_madd_inl<F>(a, a, e); else jac_madd<F>(a, a, e);
      }
    }
  } else {
    int d = fixed_base_digit(k, 0, W, carry);
    Aff<F> e = aff_from_mem<F>(table[d ? (d < 0 ? -d : d) - 1 : 0]);
    ELP_NOUNROLL
    for (int j = 0; j < nwin; j++) {
      int dn = 0;
      Aff<F> en = e;
      if (j + 1 < nwin) {
        dn = fixed_base_digit(k, j + 1, W, carry);
        en = aff_from_mem<F>(table[(size_t)(j + 1) * per + (dn ? (dn < 0 ? -dn : dn) - 1 : 0)]);
      }
      if (d != 0) {
        if (d < 0) e.y = F::neg(e.y);
        // the mixed addition is part of this loop for the Fp2 group (as a routine of its own it saves and restores ~170 registers per call)
        if (F::IS_EXT) jac_madd_inl<F>(a, a, e); else jac_madd<F>(a, a, e);
      }
      d = dn;
      e = en;
    }
  }
  if (ah) acc = a;
}

// psi^n = twist o Frobenius^n o untwist on G2 (affine): (x, y) -> (conj^n(x) g_x, conj^n(y) g_y), constants per twist type
template <class C>
ELP_HEAVY void g2_frob(Aff<F2<C>>& r, const Aff<F2<C>>& q, int n) {
  Fp2<C> x = (n & 1) ? fp2_conj(q.x) : q.x;
  Fp2<C> y = (n & 1) ? fp2_conj(q.y) : q.y;
  Fp2<C> gx, gy;
  if (n == 1) {
    ELP_LOAD_FP2(gx, C::g2frob(1, 0, c_, i_));
    ELP_LOAD_FP2(gy, C::g2frob(1, 1, c_, i_));
  } else if (n == 2) {
    ELP_LOAD_FP2(gx, C::g2frob(2, 0, c_, i_));
    ELP_LOAD_FP2(gy, C::g2frob(2, 1, c_, i_));
  } else {
    ELP_LOAD_FP2(gx, C::g2frob(3, 0, c_, i_));
    ELP_LOAD_FP2(gy, C::g2frob(3, 1, c_, i_));
  }
  fp2_mul<C>(r.x, x, gx);
  fp2_mul<C>(r.y, y, gy);
}

}  // namespace elp
