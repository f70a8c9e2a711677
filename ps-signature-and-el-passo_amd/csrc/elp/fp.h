// Base-field arithmetic Fp: Montgomery form, saturated 32-bit limbs (N = 8 for BN254, 12 for BLS12-381).
//
// Replaces mcl::FpT (third-parties/mcl, used through G1/G2/pairing at e.g. src/ps-verifier.cc:73-137).
// The inner MAC  (hi,lo) = a*b + c + d  is written so that hipcc lowers it to v_mad_u64_u32 on gfx950.
#pragma once
#include "common.h"

namespace elp {

template <class C>
struct Fp {
  u32 v[C::N];
};

template <class C>
ELP_INL Fp<C> fp_zero() {
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) r.v[i] = 0;
  return r;
}
template <class C>
ELP_INL Fp<C> fp_one() {
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) r.v[i] = C::one(i);
  return r;
}
template <class C>
ELP_INL bool fp_is_zero(const Fp<C>& a) {
  u32 t = 0;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) t |= a.v[i];
  return t == 0;
}
template <class C>
ELP_INL bool fp_eq(const Fp<C>& a, const Fp<C>& b) {
  u32 t = 0;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) t |= a.v[i] ^ b.v[i];
  return t == 0;
}

// r = a - p if a >= p else a   (a < 2p)
template <class C>
ELP_INL void fp_reduce_once(Fp<C>& a) {
  u32 d[C::N];
  u64 br = 0;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) {
    u64 t = (u64)a.v[i] - C::mod(i) - br;
    d[i] = (u32)t;
    br = (t >> 32) & 1;
  }
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) a.v[i] = br ? a.v[i] : d[i];
}

template <class C>
ELP_INL Fp<C> fp_add(const Fp<C>& a, const Fp<C>& b) {
  Fp<C> r;
  u64 c = 0;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) {
    u64 t = (u64)a.v[i] + b.v[i] + c;
    r.v[i] = (u32)t;
    c = t >> 32;
  }
  fp_reduce_once(r);  // p has spare top bits: a + b < 2p < 2^(32N), no carry out
  return r;
}
template <class C>
ELP_INL Fp<C> fp_sub(const Fp<C>& a, const Fp<C>& b) {
  Fp<C> r;
  u64 br = 0;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) {
    u64 t = (u64)a.v[i] - b.v[i] - br;
    r.v[i] = (u32)t;
    br = (t >> 32) & 1;
  }
  u32 mask = (u32)0 - (u32)br;
  u64 c = 0;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) {
    u64 t = (u64)r.v[i] + (C::mod(i) & mask) + c;
    r.v[i] = (u32)t;
    c = t >> 32;
  }
  return r;
}
template <class C>
ELP_INL Fp<C> fp_neg(const Fp<C>& a) {
  return fp_sub(fp_zero<C>(), a);
}
template <class C>
ELP_INL Fp<C> fp_dbl(const Fp<C>& a) {
  return fp_add(a, a);
}
template <class C>
ELP_INL Fp<C> fp_select(bool c, const Fp<C>& a, const Fp<C>& b) {  // c ? a : b
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}

// Montgomery product a*b*R^-1 mod p.  CIOS, "no-carry" variant (valid because the top word of p has spare
// bits: BN254 254/256, BLS12-381 381/384): 2N^2 + N multiply-accumulates, no extra accumulator words.
template <class C>
ELP_FPMUL Fp<C> fp_mul(Fp<C> a, Fp<C> b) {
  constexpr int N = C::N;
  u32 t[N];
  ELP_UNROLL
  for (int i = 0; i < N; i++) t[i] = 0;
  ELP_UNROLL
  for (int i = 0; i < N; i++) {
    u64 acc = (u64)a.v[0] * b.v[i] + t[0];
    u32 A = (u32)(acc >> 32);
    u32 t0 = (u32)acc;
    u32 m = t0 * C::INV;
    acc = (u64)m * C::mod(0) + t0;
    u32 Cc = (u32)(acc >> 32);
    ELP_UNROLL
    for (int j = 1; j < N; j++) {
      acc = (u64)a.v[j] * b.v[i] + t[j] + A;
      A = (u32)(acc >> 32);
      acc = (u64)m * C::mod(j) + (u32)acc + Cc;
      t[j - 1] = (u32)acc;
      Cc = (u32)(acc >> 32);
    }
    t[N - 1] = Cc + A;
  }
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < N; i++) r.v[i] = t[i];
  fp_reduce_once(r);
  return r;
}

template <class C>
ELP_FPMUL Fp<C> fp_sqr(Fp<C> a) {
  return fp_mul<C>(a, a);
}

// a^e for a public exponent given as N limbs through a constexpr accessor (square-and-multiply, MSB first).
template <class C, class E>
ELP_HEAVY Fp<C> fp_pow_const(const Fp<C>& a, E expo) {
  Fp<C> r = fp_one<C>();
  bool started = false;
  ELP_NOUNROLL
  for (int i = C::N * 32 - 1; i >= 0; i--) {
    if (started) r = fp_sqr<C>(r);
    if ((expo(i >> 5) >> (i & 31)) & 1) {
      r = started ? fp_mul<C>(r, a) : a;
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

template <class C>
ELP_HEAVY Fp<C> fp_inv(const Fp<C>& a) {  // a^(p-2); inv(0) = 0
  return fp_pow_const<C>(a, ExpPm2<C>());
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
  if (fp_is_zero(a)) return 0;
  Fp<C> t = fp_pow_const<C>(a, ExpPm1d2<C>());
  return fp_eq(t, fp_one<C>()) ? 1 : -1;
}

// ---- conversions.  "std" = canonical integer in [0,p) as N little-endian 32-bit limbs.
template <class C>
ELP_INL Fp<C> fp_from_std(const Fp<C>& s) {
  Fp<C> r2;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) r2.v[i] = C::r2(i);
  return fp_mul<C>(s, r2);
}
template <class C>
ELP_INL Fp<C> fp_to_std(const Fp<C>& m) {
  Fp<C> one = fp_zero<C>();
  one.v[0] = 1;
  return fp_mul<C>(m, one);
}
// load a generated constant: ELP_LOAD_FP(x, C::curve_b(i_))
#define ELP_LOAD_FP(dst, expr_i)                          \
  do {                                                    \
    ELP_UNROLL                                            \
    for (int i_ = 0; i_ < C::N; i_++) (dst).v[i_] = (expr_i); \
  } while (0)
// little-endian bytes (FBYTES) -> limbs (no reduction)
template <class C>
ELP_INL Fp<C> fp_load_le(const uint8_t* b) {
  Fp<C> r;
  for (int i = 0; i < C::N; i++)
    r.v[i] = (u32)b[4 * i] | ((u32)b[4 * i + 1] << 8) | ((u32)b[4 * i + 2] << 16) | ((u32)b[4 * i + 3] << 24);
  return r;
}
template <class C>
ELP_INL void fp_store_le(uint8_t* b, const Fp<C>& a) {
  for (int i = 0; i < C::N; i++) {
    b[4 * i] = (uint8_t)a.v[i];
    b[4 * i + 1] = (uint8_t)(a.v[i] >> 8);
    b[4 * i + 2] = (uint8_t)(a.v[i] >> 16);
    b[4 * i + 3] = (uint8_t)(a.v[i] >> 24);
  }
}
// a < p ?
template <class C>
ELP_INL bool fp_std_in_range(const Fp<C>& a) {
  u64 br = 0;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) {
    u64 t = (u64)a.v[i] - C::mod(i) - br;
    br = (t >> 32) & 1;
  }
  return br != 0;
}

}  // namespace elp
