// The pairing check e(sig1, K) e(-sig2, gg) == 1 on FOUR lanes per item (elp/quad.h): Miller loop + final exponentiation with the Fp12 value spread over a
// DPP quad, everything below Fp12 (G2 point arithmetic, line evaluation, Fp2 inversions) in the two-lane code of tower.h / pairing.h run by both lane pairs.
// Replaces pairing() + GT == of src/ps-verifier.cc:31-34 (PSVerifier::verify) and :132-137 (el_passo_verify_id) like pairing.h does; same verdicts.
// The formulas are those of pairing.h (miller_loop<C, 1, 1>, final_exp<C, false>); what differs is the layout and three choices that follow from it:
//   * the two lines of a Miller step are applied one after the other (two six-term sparse products: 3 402 multiply-adds per lane on BN254) instead of being
//     multiplied with each other first (a sparse x sparse product and a full product: 3 645, and three more reductions);
//   * the powers f^|z| run on compressed (Karabina) squarings with each lane pair owning one of the two Fp4 squarings of a step;
//   * nothing crosses a function call inside the loops: the Miller value, the running point and the exponentiation accumulator stay in registers
//     (an Fp12 is 27 / 42 registers per lane); the straight-line part of the hard part goes through out-of-line copies of the product / squaring.
#pragma once
#include "pairing.h"
#include "quad.h"

namespace elp {

#if defined(__HIP_DEVICE_COMPILE__)
ELP_INL i32 quad_swap_i32(i32 v) { return quad_dpp_i32<0x4E>(v); }
#else
inline i32 quad_swap_i32(i32 v) {
  i32 all[4];
  elp_quad_gather_hook(&v, all, sizeof v);
  return all[elp_quad_lane ^ 2];
}
#endif
ELP_INL bool quad_and(bool b) {      // true iff true on both lane pairs (b already agrees inside a pair)
  const i32 o = quad_swap_i32(b ? 1 : 0);
  return b & (o != 0);
}

// a^(p^n), n = 1, 2, 3: coefficient i of a lane pair is the coefficient of w^(2 i + j) (j = 0 low, 1 high): conjugated n times, scaled by gamma_{n, 2 i + j}
template <class C>
ELP_INL void fp12q_frob(Fp12Q<C>& r, const Fp12Q<C>& a, int n) {
  const bool cj = (n & 1) != 0, hi = quad_hi();
  const Fp2<C> t0 = cj ? fp2_conj(a.h.c0) : a.h.c0, t1 = cj ? fp2_conj(a.h.c1) : a.h.c1, t2 = cj ? fp2_conj(a.h.c2) : a.h.c2;
  const Fp2<C> g0 = fp2_select(hi, fp2_frob_coeff<C>(n, 1), fp2_one<C>());
  const Fp2<C> g1 = fp2_select(hi, fp2_frob_coeff<C>(n, 3), fp2_frob_coeff<C>(n, 2));
  const Fp2<C> g2 = fp2_select(hi, fp2_frob_coeff<C>(n, 5), fp2_frob_coeff<C>(n, 4));
  Fp6<C> o;
  fp2_mul<C>(o.c0, t0, g0);
  fp2_mul<C>(o.c1, t1, g1);
  fp2_mul<C>(o.c2, t2, g2);
  r.h = o;
}
// 1 / a = (a0 - a1 w) / (a0^2 - v a1^2): every pair squares its own half, the norm and its inverse are computed by both pairs alike
template <class C>
ELP_INL void fp12q_inv(Fp12Q<C>& r, const Fp12Q<C>& a) {
  const bool hi = quad_hi();
  Fp6<C> sq, vs, t, ti, o;
  fp6_sqr<C>(sq, a.h);
  fp6_mul_by_v(vs, sq);
  const Fp6<C> m = fp6_select(hi, vs, sq);                      // low: a0^2, high: v a1^2
  const Fp6<C> p = fp6_quad_swap(m);
  fp6_sub(t, fp6_select(hi, p, m), fp6_select(hi, m, p));
  fp6_inv<C>(ti, t);
  fp6_mul<C>(o, a.h, ti);
  r.h.c0.c = fp_cneg(hi, o.c0.c);
  r.h.c1.c = fp_cneg(hi, o.c1.c);
  r.h.c2.c = fp_cneg(hi, o.c2.c);
}
template <class C>
ELP_INL bool fp12q_is_one(const Fp12Q<C>& a) {
  const Fp2<C> want0 = fp2_select(quad_hi(), fp2_zero<C>(), fp2_one<C>());
  const bool mine = fp2_eq(a.h.c0, want0) & fp2_is_zero(a.h.c1) & fp2_is_zero(a.h.c2);      // no short cut: every lane takes part in every exchange
  return quad_and(mine);
}
// out-of-line copies for the straight-line part of the final exponentiation (one copy of the code each; the operands cross the call through private memory,
// 108 / 168 bytes per lane a few dozen times per item)
template <class C>
ELP_HEAVY void fp12q_mul_call(Fp12Q<C>& r, const Fp12Q<C>& a, const Fp12Q<C>& b) {
  fp12q_mul<C>(r, a, b);
}
template <class C>
ELP_HEAVY void fp12q_cyc_sqr_call(Fp12Q<C>& r, const Fp12Q<C>& a) {
  fp12q_cyc_sqr<C>(r, a);
}
template <class C>
ELP_HEAVY void fp12q_frob_call(Fp12Q<C>& r, const Fp12Q<C>& a, int n) {
  fp12q_frob<C>(r, a, n);
}

// The full element from a compressed one held in full by both pairs (tower.h: cyc_decompress) and ninv = 1 / N(c): z0 + z1 s = (b^2 - conj(c)) conj(c) / N(c).
// The low pair computes z0, the high pair z1 (one two-term Fp2 inner product each), and each keeps its three coefficients.
template <class C>
ELP_INL void cyc_decompress_q(Fp12Q<C>& r, const CycComp<C>& a, const Fp2<C>& ninv) {
  const bool hi = quad_hi();
  Fp2<C> B0, B1;
  fp4_sqr<C>(B0, B1, a.z2, a.z3);
  const Fp2<C> n0 = fp2_sub(B0, a.z4), n1 = fp2_add(B1, a.z5);                        // b^2 - conj(c)
  // (n0 + n1 s)(z4 - z5 s) = (n0 z4 - xi n1 z5) + (n1 z4 - n0 z5) s
  const Fp2<C> U = fp2_select(hi, n1, n0), V = fp2_select(hi, n0, fp2_mul_xi(n1));
  const Fp2<C> A2[2] = {U, fp2_neg(V)}, B2[2] = {a.z4, a.z5};
  const Fp2<C> m = fp2_dot<C, 2>(A2, B2);
  Fp2<C> z;
  fp2_mul<C>(z, m, ninv);                                                              // low: z0, high: z1
  // Fp12Q: low (c0, c1, c2) = (z0, z4, z3), high (c0, c1, c2) = (z2, z1, z5)
  r.h.c0 = fp2_select(hi, a.z2, z);
  r.h.c1 = fp2_select(hi, z, a.z4);
  r.h.c2 = fp2_select(hi, a.z5, a.z3);
}

// a^E for a in the cyclotomic subgroup and a compile-time exponent with few set bits (|z|): pairing.h fp12_exp_u64 on the quad -- runs of compressed
// squarings, a snapshot per set bit, ONE shared Fp2 inversion for the decompressions, the snapshots multiplied together.  `ok` is cleared if a snapshot
// cannot be decompressed (c = 0: never seen on pairing values; the caller falls back to the Granger-Scott chain).
// position of the k-th set bit of e above bit 0 (k = 0: the lowest)
ELP_HD constexpr int exp_set_bit(u64 e, int k) {
  int pos = 1;
  for (;; pos++)
    if ((e >> pos) & 1) {
      if (k == 0) return pos;
      k--;
    }
}
template <class C, u64 E>
ELP_HEAVY void fp12q_exp_comp(Fp12Q<C>& r, const Fp12Q<C>& a, bool& ok) {
  ELP_NONLEAF();
  constexpr int NSET = __builtin_popcountll(E >> 1);
  static_assert(NSET >= 1 && NSET <= 6, "exponent shape");
  CycCompQ<C> snap[NSET];
  CycCompQ<C> cur;
  fp12q_to_comp<C>(cur, a);
  ELP_UNROLL
  for (int k = 0; k < NSET; k++) {                 // one run of squarings per set bit above bit 0, a snapshot at its end
    const int from = k ? exp_set_bit(E, k - 1) + 1 : 1, to = exp_set_bit(E, k);
    ELP_NOUNROLL
    for (int s = from; s <= to; s++) cyc_compq_sqr<C>(cur, cur);
    snap[k] = cur;
  }
  // norms N(c) = z4^2 - xi z5^2 of the snapshots (c lives on the high pair: both pairs receive it) and their shared inversion (Montgomery's trick over Fp2)
  Fp2<C> nrm[NSET], pre[NSET];
  Fp2<C> acc = fp2_one<C>();
  bool bad = false;
  ELP_UNROLL
  for (int j = 0; j < NSET; j++) {
    CycComp<C> full;
    compq_to_paired<C>(full, snap[j]);
    nrm[j] = cyc_comp_norm<C>(full);
    bad |= fp2_is_zero<C>(nrm[j]);
    pre[j] = acc;
    fp2_mul<C>(acc, acc, nrm[j]);
  }
  ok = !bad;
  Fp2<C> inv;
  fp2_inv<C>(inv, acc);
  Fp12Q<C> prod = a;
  bool have = (E & 1) != 0;
  ELP_UNROLL
  for (int j = NSET - 1; j >= 0; j--) {
    Fp2<C> ninv;
    fp2_mul<C>(ninv, inv, pre[j]);
    fp2_mul<C>(inv, inv, nrm[j]);
    CycComp<C> full;
    compq_to_paired<C>(full, snap[j]);
    Fp12Q<C> t;
    cyc_decompress_q<C>(t, full, ninv);
    if (have) {
      fp12q_mul<C>(prod, prod, t);
    } else {
      prod = t;
      have = true;
    }
  }
  if (ok) r = prod;          // r may alias a: left untouched for the caller's fall-back otherwise
}
template <class C>
ELP_HEAVY void fp12q_exp_gs_call(Fp12Q<C>& r, const Fp12Q<C>& a, u64 e) {
  fp12q_exp_u64_gs<C>(r, a, e);
}
// a^z (signed z) in the cyclotomic subgroup
template <class C>
ELP_INL void fp12q_exp_z(Fp12Q<C>& r, const Fp12Q<C>& a) {
  bool ok = true;
  fp12q_exp_comp<C, C::ZABS>(r, a, ok);
  if (!ok) fp12q_exp_gs_call<C>(r, a, C::ZABS);        // quad-uniform: `ok` derives from values all four lanes agree on
  if (C::Z_NEG) fp12q_conj(r, r);
}

// f^(k (p^12 - 1) / r) == 1 with the small multiples k of pairing.h final_exp<C, false> (FKR on BN curves, the cube of the hard part on BLS12 curves)
template <class C>
ELP_INL bool final_exp_is_one4(const Fp12Q<C>& f_in) {
  Fp12Q<C> f, t0, t1, r;
  fp12q_inv<C>(t0, f_in);
  fp12q_conj(t1, f_in);
  fp12q_mul_call<C>(f, t1, t0);          // f^(p^6 - 1)
  fp12q_frob_call<C>(t0, f, 2);
  fp12q_mul_call<C>(f, t0, f);           // ^(p^2 + 1)
  if constexpr (C::IS_BN) {
    // Fuentes-Castaneda, Knapp, Rodriguez-Henriquez (pairing.h): l0 + l1 p + l2 p^2 + l3 p^3
    Fp12Q<C> fz, f2z, f6z, f6z2, f12z3, a, b, t;
    fp12q_exp_z<C>(fz, f);
    fp12q_cyc_sqr_call<C>(f2z, fz);
    fp12q_cyc_sqr_call<C>(t, f2z);              // f^4z
    fp12q_mul_call<C>(f6z, t, f2z);
    fp12q_exp_z<C>(f6z2, f6z);
    fp12q_cyc_sqr_call<C>(t, f6z2);             // f^12z^2
    fp12q_exp_z<C>(f12z3, t);
    fp12q_mul_call<C>(a, f12z3, f6z2);
    fp12q_mul_call<C>(a, a, f6z);               // f^l2
    fp12q_conj(t, f2z);
    fp12q_mul_call<C>(b, a, t);                 // f^l1
    fp12q_mul_call<C>(r, a, f6z2);
    fp12q_mul_call<C>(r, r, f);                 // f^l0
    fp12q_frob_call<C>(t, b, 1);
    fp12q_mul_call<C>(r, r, t);                 // (f^l1)^p
    fp12q_frob_call<C>(t, a, 2);
    fp12q_mul_call<C>(r, r, t);                 // (f^l2)^p^2
    fp12q_conj(t, f);
    fp12q_mul_call<C>(b, b, t);                 // f^l3
    fp12q_frob_call<C>(t, b, 3);
    fp12q_mul_call<C>(r, r, t);                 // (f^l3)^p^3
  } else {
    // BLS12, hard part cubed: (z-1)^2 (z+p) (z^2+p^2-1) + 3
    Fp12Q<C> a, b, c, t;
    fp12q_exp_z<C>(t, f);
    fp12q_conj(b, f);
    fp12q_mul_call<C>(a, t, b);                 // f^(z-1)
    fp12q_exp_z<C>(t, a);
    fp12q_conj(b, a);
    fp12q_mul_call<C>(a, t, b);                 // a^(z-1)
    fp12q_exp_z<C>(t, a);
    fp12q_frob_call<C>(b, a, 1);
    fp12q_mul_call<C>(b, b, t);                 // a^(z+p)
    fp12q_exp_z<C>(t, b);
    fp12q_exp_z<C>(t, t);
    fp12q_frob_call<C>(c, b, 2);
    fp12q_mul_call<C>(c, c, t);
    fp12q_conj(t, b);
    fp12q_mul_call<C>(c, c, t);                 // b^(z^2+p^2-1)
    fp12q_cyc_sqr_call<C>(t, f);
    fp12q_mul_call<C>(t, t, f);                 // f^3
    fp12q_mul_call<C>(r, c, t);
  }
  return fp12q_is_one<C>(r);
}

// f <- f * line(P) (pairing.h: ml_apply_line_inl)
template <class C>
ELP_INL void ml_apply_line4(Fp12Q<C>& f, const LineCoef<C>& l, const Fp<C>& xp, const Fp<C>& yp) {
  const Fp2<C> a = fp2_mul_fp(l.a, yp), b = fp2_mul_fp(l.b, xp);
  if constexpr (C::TWIST_D)
    fp12q_mul_by_line<C>(f, a, b, l.c);
  else
    fp12q_mul_by_line<C>(f, l.c, b, a);
}
// f = f_{s,Q}(P1) * f_{s,gg}(P2) for one run-time Q and the precomputed lines of gg (pairing.h: miller_loop<C, 1, 1>); pairs with a point at infinity
// contribute 1.  All four lanes hold the same points (G1 points whole, G2 points in the two-lane layout).
template <class C>
ELP_INL void miller_loop4(Fp12Q<C>& f, const Aff<F1<C>>& p1, const Aff<F2<C>>& q1, const Aff<F1<C>>& p2, const LineMem<C>* lines) {
  const bool live_v = !(aff_is_inf(p1) || aff_is_inf(q1)), live_f = !aff_is_inf(p2);
  G2Proj<C> T;
  T.X = q1.x;
  T.Y = q1.y;
  T.Z = fp2_one<C>();
  const Fp2<C> nqy = fp2_neg(q1.y);
  LineCoef<C> l;
  int n = 0;
  fp12q_set_one(f);
  ELP_NOUNROLL
  for (int i = 0; i < C::ATE_LEN; i++) {
    if (i != 0) fp12q_sqr<C>(f, f);
    const int d = C::ate_naf(i);
    ELP_NOUNROLL
    for (int half = 0; half < 2; half++) {          // 0: doubling step, 1: addition step (only for a non-zero digit)
      if (half == 1 && d == 0) break;
      if (live_v) {
        if (half == 0) {
          ml_dbl_step_inl<C>(T, l);
        } else {
          const Fp2<C> yq = fp2_select(d > 0, q1.y, nqy);
          ml_add_step_inl<C>(T, l, q1.x, yq);
        }
        ml_apply_line4<C>(f, l, p1.x, p1.y);
      }
      if (live_f) {
        const LineCoef<C> lf = line_from_mem<C>(lines[n]);
        ml_apply_line4<C>(f, lf, p2.x, p2.y);
      }
      n++;
    }
  }
  if (C::Z_NEG) fp12q_conj(f, f);
  if constexpr (C::IS_BN) {
    if (live_v) {
      if (C::Z_NEG) T.Y = fp2_neg(T.Y);
      Aff<F2<C>> f1, f2;
      g2_frob<C>(f1, q1, 1);
      g2_frob<C>(f2, q1, 2);
      ml_add_step<C>(T, l, f1.x, f1.y);
      ml_apply_line4<C>(f, l, p1.x, p1.y);
      ml_add_step<C>(T, l, f2.x, fp2_neg(f2.y));
      ml_apply_line4<C>(f, l, p1.x, p1.y);
    }
    if (live_f) {
      ml_apply_line4<C>(f, line_from_mem<C>(lines[n]), p2.x, p2.y);
      ml_apply_line4<C>(f, line_from_mem<C>(lines[n + 1]), p2.x, p2.y);
    }
  }
}

// Signature half on four lanes: e(sig1, K) * e(-sig2, gg) == 1       (src/ps-verifier.cc:133-137; pipeline.h ps_pairing_check)
template <class C>
ELP_HEAVY bool ps_pairing_check4(const LineMem<C>* gg_lines, const Aff<F1<C>>& sig1, const Aff<F1<C>>& sig2, const Aff<F2<C>>& aK) {
  ELP_NONLEAF();             // with the Fp6 routines inlined this body can pass 128 KB (common.h; the BLS12-381 unit is built with -DELP_NONLEAF_GUARD=1)
  Aff<F1<C>> nsig2;
  aff_neg(nsig2, sig2);
  if (aff_is_inf(sig2)) aff_set_inf(nsig2);
  Fp12Q<C> f;
  miller_loop4<C>(f, sig1, aK, nsig2, gg_lines);
  return final_exp_is_one4<C>(f);
}

}  // namespace elp
