// FOUR LANES PER ITEM: the Fp12 level of the tower on a DPP quad (round 5).
//
// Replaces the same mcl::Fp12T arithmetic as tower.h (pairing() / GT == at src/ps-verifier.cc:31-34,132-137), for batches that leave SIMDs idle at
// two lanes per item.  Layout: the four lanes 4k .. 4k+3 of a wave work on one item.  Lanes (0, 1) -- the LOW pair -- hold the Fp6 coefficient c0 of
// f = c0 + c1 w in the paired layout of common.h (even lane: real parts, odd lane: imaginary parts of the three Fp2 coefficients), lanes (2, 3) -- the
// HIGH pair -- hold c1.  An Fp12 value is 3 base-field elements per lane (27 registers on BN254, 42 on BLS12-381).  Everything below Fp12 (Fp2, Fp6, G2
// points) is the Paired<> code of tower.h run by both pairs; what is new here is the level at which the two pairs cooperate:
//   * fp12q_mul: schoolbook over w with both products of a pair fused into ONE Karatsuba pass of two-term Fp2 inner products
//     (low pair: a0 b0 + (v a1) b1, high pair: a1 b0 + a0 b1 -- six fp_mul_quad per lane where two lanes need eighteen fp_mul_pair);
//   * fp4q_mul: the product in Fp4 = Fp2[s]/(s^2 - xi) of two elements spread over the quad as ONE four-term inner product per lane (x-operands
//     broadcast inside the quad, y-operands fetched by lane-xor patterns) -- the building block of the Granger-Scott / Karabina squarings;
//   * fp12q_mul_by_line: f * (a + b w + c w^3) as three six-term inner products per lane.
// Exchanges are DPP quad_perm moves (no LDS): [1,0,3,2] inside a pair, [2,3,0,1] between the pairs, [3,2,1,0] diagonal, [k,k,k,k] broadcasts.
// Rules as for paired code: every lane of a quad executes every exchange.
#pragma once
#include "tower.h"

namespace elp {

#if defined(__HIP_DEVICE_COMPILE__)
ELP_INL bool quad_hi() { return (threadIdx.x & 2u) != 0; }
template <int CTRL>
ELP_INL i32 quad_dpp_i32(i32 v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// value of lane (q ^ M) of the quad, M = 0..3
template <class C, int M>
ELP_INL Fp<C> fp_quad_xor(const Fp<C>& a) {
  if constexpr (M == 0) {
    return a;
  } else {
    constexpr int CTRL = M == 1 ? 0xB1 : M == 2 ? 0x4E : 0x1B;   // quad_perm [1,0,3,2] / [2,3,0,1] / [3,2,1,0]
    Fp<C> r;
    ELP_UNROLL
    for (int i = 0; i < C::NL; i++) r.v[i] = quad_dpp_i32<CTRL>(a.v[i]);
    return r;
  }
}
// value of lane L of the quad on all four lanes
template <class C, int L>
ELP_INL Fp<C> fp_quad_bcast(const Fp<C>& a) {
  Fp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) r.v[i] = quad_dpp_i32<L * 0x55>(a.v[i]);
  return r;
}
#else
// Host twin (tests only): the four lanes of a quad are four threads; the twin installs a hook that returns the values of all four lanes (a rendezvous).
inline thread_local int elp_quad_lane = 0;
inline void (*elp_quad_gather_hook)(const void* own, void* all4, size_t bytes) = nullptr;
inline bool quad_hi() { return (elp_quad_lane & 2) != 0; }
template <class C, int M>
inline Fp<C> fp_quad_xor(const Fp<C>& a) {
  Fp<C> all[4];
  elp_quad_gather_hook(&a, all, sizeof a);
  return all[elp_quad_lane ^ M];
}
template <class C, int L>
inline Fp<C> fp_quad_bcast(const Fp<C>& a) {
  Fp<C> all[4];
  elp_quad_gather_hook(&a, all, sizeof a);
  return all[L];
}
#endif

// ---- N-term inner product with ONE Montgomery reduction: (sum_t a[t] b[t]) R^-1 (mod p).  Operand magnitudes, in units of a carried limb, must
// satisfy sum_t A_t B_t <= C::HEADROOM - 1 (asserted by the host twin under ELP_BOUND_CHECK); N (NL^2) + NL^2 multiply-adds.
template <class C, int N>
ELP_INL Fp<C> fp_dot(const Fp<C> (&a)[N], const Fp<C> (&b)[N]) {
  constexpr int NL = C::NL;
  static_assert(N >= 1 && N <= C::HEADROOM - 1, "too many terms for one 64-bit column");
#if defined(ELP_BOUND_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
  {
    long double tot = 0;
    for (int t = 0; t < N; t++) {
      long double ma = 0, mb = 0;
      for (int i = 0; i < NL - 1; i++) {
        if (llabs((long long)a[t].v[i]) > ma) ma = llabs((long long)a[t].v[i]);
        if (llabs((long long)b[t].v[i]) > mb) mb = llabs((long long)b[t].v[i]);
      }
      tot += ma * mb;
      assert(llabs((long long)a[t].v[NL - 1]) < (1LL << (C::LB - 1)) && llabs((long long)b[t].v[NL - 1]) < (1LL << (C::LB - 1)));
    }
    assert(tot * NL + (long double)NL * (long double)((i64)1 << (2 * C::LB - 2)) + 1.0e18L < 9223372036854775807.0L);
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
    for (int t = 0; t < N; t++) {
      ELP_UNROLL
      for (int i = 0; i <= k; i++) ELP_MAC(acc, a[t].v[i], b[t].v[k - i]);
    }
    ELP_UNROLL
    for (int i = 0; i < k; i++) ELP_MAC_S(acc, m[i], pl[k - i]);
    m[k] = elp_balanced30((u32)acc * C::INVL);
    ELP_MAC_S(acc, m[k], pl[0]);
    acc >>= ELP_LIMB_BITS;
  }
  ELP_UNROLL
  for (int k = NL; k < 2 * NL - 1; k++) {
    ELP_UNROLL
    for (int t = 0; t < N; t++) {
      ELP_UNROLL
      for (int i = k - NL + 1; i < NL; i++) ELP_MAC(acc, a[t].v[i], b[t].v[k - i]);
    }
    ELP_UNROLL
    for (int i = k - NL + 1; i < NL; i++) ELP_MAC_S(acc, m[i], pl[k - i]);
    r.v[k - NL] = elp_balanced30((u32)acc);
    acc = (acc + ELP_LIMB_HALF) >> ELP_LIMB_BITS;
  }
  r.v[NL - 1] = (i32)acc;
  return r;
}

// ---- N-term inner product in Fp2 on a lane pair: sum_t A_t B_t as ONE 2N-term base-field inner product per lane.  With (pa, pb) the partner's copies,
//    even lane:  re = sum own_a own_b - pa pb          odd lane:  im = sum own_a pb + pa own_b
template <class C, int N>
ELP_INL Fp2<C> fp2_dot(const Fp2<C> (&A)[N], const Fp2<C> (&B)[N]) {
  static_assert(is_paired<C>(), "paired layout");
  const bool odd = pair_odd();
  Fp<C> a[2 * N], b[2 * N];
  ELP_UNROLL
  for (int t = 0; t < N; t++) {
    const Fp<C> pa = fp_pair_swap(A[t].c), pb = fp_pair_swap(B[t].c);
    a[2 * t] = A[t].c;
    b[2 * t] = fp_select(odd, pb, B[t].c);
    a[2 * t + 1] = pa;
    b[2 * t + 1] = fp_select(odd, B[t].c, fp_neg(pb));
  }
  Fp2<C> r;
  r.c = fp_dot<C, 2 * N>(a, b);
  return r;
}

// ------------------------------------------------------------------ Fp12 on a quad
template <class C>
struct Fp12Q {   // C is a Paired<> traits class
  Fp6<C> h;      // low pair: c0, high pair: c1
};
template <class C>
ELP_INL Fp2<C> fp2_quad_swap(const Fp2<C>& a) {   // the other pair's coefficient (same parity)
  Fp2<C> r;
  r.c = fp_quad_xor<C, 2>(a.c);
  return r;
}
template <class C>
ELP_INL Fp6<C> fp6_quad_swap(const Fp6<C>& a) {
  Fp6<C> r;
  r.c0 = fp2_quad_swap(a.c0);
  r.c1 = fp2_quad_swap(a.c1);
  r.c2 = fp2_quad_swap(a.c2);
  return r;
}
template <class C>
ELP_INL Fp6<C> fp6_select(bool c, const Fp6<C>& a, const Fp6<C>& b) {
  Fp6<C> r;
  r.c0 = fp2_select(c, a.c0, b.c0);
  r.c1 = fp2_select(c, a.c1, b.c1);
  r.c2 = fp2_select(c, a.c2, b.c2);
  return r;
}
// plain-layout value (memory, or the registers of a one-lane routine) -> this lane's share, and back (every lane receives the whole value)
template <class C>
ELP_INL void fp12q_from_plain(Fp12Q<C>& r, const Fp12<typename PairInfo<C>::Base>& m) {
  const bool hi = quad_hi();
  r.h.c0 = fp2_from_mem<C>(hi ? m.c1.c0 : m.c0.c0);
  r.h.c1 = fp2_from_mem<C>(hi ? m.c1.c1 : m.c0.c1);
  r.h.c2 = fp2_from_mem<C>(hi ? m.c1.c2 : m.c0.c2);
}
template <class C>
ELP_INL void fp12q_to_plain(Fp12<typename PairInfo<C>::Base>& m, const Fp12Q<C>& a) {
  typedef typename PairInfo<C>::Base B;
  auto put = [&](Fp2<B>& lo, Fp2<B>& hi, const Fp2<C>& x) {
    lo.c0 = fp_cast<B>(fp_quad_bcast<C, 0>(x.c));
    lo.c1 = fp_cast<B>(fp_quad_bcast<C, 1>(x.c));
    hi.c0 = fp_cast<B>(fp_quad_bcast<C, 2>(x.c));
    hi.c1 = fp_cast<B>(fp_quad_bcast<C, 3>(x.c));
  };
  put(m.c0.c0, m.c1.c0, a.h.c0);
  put(m.c0.c1, m.c1.c1, a.h.c1);
  put(m.c0.c2, m.c1.c2, a.h.c2);
}
template <class C>
ELP_INL void fp12q_set_one(Fp12Q<C>& r) {
  r.h.c0 = fp2_select(quad_hi(), fp2_zero<C>(), fp2_one<C>());
  r.h.c1 = fp2_zero<C>();
  r.h.c2 = fp2_zero<C>();
}
template <class C>
ELP_INL void fp12q_conj(Fp12Q<C>& r, const Fp12Q<C>& a) {   // a^(p^6): c1 -> -c1
  const bool hi = quad_hi();
  r.h.c0.c = fp_cneg(hi, a.h.c0.c);
  r.h.c1.c = fp_cneg(hi, a.h.c1.c);
  r.h.c2.c = fp_cneg(hi, a.h.c2.c);
}

// r = a b + c d in Fp6 (paired layout) in ONE Karatsuba pass: six two-term Fp2 inner products (one fp_mul_quad per lane each).  a, b, c, d carried.
template <class C>
ELP_INL void fp6_mul2(Fp6<C>& r, const Fp6<C>& a, const Fp6<C>& b, const Fp6<C>& c, const Fp6<C>& d) {
  static_assert(is_paired<C>() && fp_roomy<C>(), "paired layout over a field with lazy-sum headroom");
  Fp2<C> t0, t1, t2, s, r0, r1, r2;
  fp2_mul_pair<C>(t0, a.c0, b.c0, c.c0, d.c0);
  fp2_mul_pair<C>(t1, a.c1, b.c1, c.c1, d.c1);
  fp2_mul_pair<C>(t2, a.c2, b.c2, c.c2, d.c2);
  // operand sums: lazy on the left, carried on the right -- four base-field terms of magnitude 2 x 1 per lane (<= 13)
  fp2_mul_pair<C>(s, fp2_add_lazy(a.c1, a.c2), fp2_add(b.c1, b.c2), fp2_add_lazy(c.c1, c.c2), fp2_add(d.c1, d.c2));
  r0 = fp2_carry(fp2_add_lazy(t0, fp2_mul_xi_lazy(fp2_sub_lazy(fp2_sub_lazy(s, t1), t2))));             // 1 + 2 * 3 = 7
  fp2_mul_pair<C>(s, fp2_add_lazy(a.c0, a.c1), fp2_add(b.c0, b.c1), fp2_add_lazy(c.c0, c.c1), fp2_add(d.c0, d.c1));
  r1 = fp2_carry_fast(fp2_add_lazy(fp2_sub_lazy(fp2_sub_lazy(s, t0), t1), fp2_mul_xi_lazy(t2)));        // 3 + 2
  fp2_mul_pair<C>(s, fp2_add_lazy(a.c0, a.c2), fp2_add(b.c0, b.c2), fp2_add_lazy(c.c0, c.c2), fp2_add(d.c0, d.c2));
  r2 = fp2_carry_fast(fp2_add_lazy(fp2_sub_lazy(fp2_sub_lazy(s, t0), t2), t1));                         // 4
  r.c0 = r0;
  r.c1 = r1;
  r.c2 = r2;
}

// r = a b.   low pair: a0 b0 + (v a1) b1,   high pair: a1 b0 + a0 b1  -- the same instruction stream, operands chosen by the pair
template <class C>
ELP_INL void fp12q_mul(Fp12Q<C>& r, const Fp12Q<C>& a, const Fp12Q<C>& b) {
  const bool hi = quad_hi();
  const Fp6<C> pa = fp6_quad_swap(a.h), pb = fp6_quad_swap(b.h);
  const Fp6<C> Y = fp6_select(hi, pb, b.h), W = fp6_select(hi, b.h, pb);
  Fp6<C> Z;                                        // high: a0 (= pa); low: v a1 = (xi pa.c2, pa.c0, pa.c1)
  Z.c0 = fp2_select(hi, pa.c0, fp2_mul_xi(pa.c2));
  Z.c1 = fp2_select(hi, pa.c1, pa.c0);
  Z.c2 = fp2_select(hi, pa.c2, pa.c1);
  Fp6<C> t;
  fp6_mul2<C>(t, a.h, Y, Z, W);
  r.h = t;
}
// r = a^2 by complex squaring: t = a0 a1, u = (a0 + a1)(a0 + v a1), c0 = u - t - v t, c1 = 2 t.  The low pair computes u, the high pair t -- ONE Fp6 product
// (six fp_mul_pair per lane) instead of the fused pair of the general product -- and t crosses to the low pair once.
template <class C>
ELP_INL void fp12q_sqr(Fp12Q<C>& r, const Fp12Q<C>& a) {
  const bool hi = quad_hi();
  const Fp6<C> p = fp6_quad_swap(a.h);
  Fp6<C> X, Y, vo;                                  // X = a0 + a1 (the same on both pairs); low: Y = a0 + v a1 = own + v p
  fp6_add(X, a.h, p);
  fp6_mul_by_v(vo, p);
  fp6_add(Y, a.h, vo);
  const Fp6<C> U = fp6_select(hi, a.h, X), V = fp6_select(hi, p, Y);
  Fp6<C> m;
  fp6_mul<C>(m, U, V);                              // low: u, high: t
  const Fp6<C> t = fp6_quad_swap(m);               // low: t
  Fp6<C> lo;
  lo.c0 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(m.c0, t.c0), fp2_mul_xi_lazy(t.c2)));      // u - t - v t: 1 + 1 + 2
  lo.c1 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(m.c1, t.c1), t.c0));
  lo.c2 = fp2_carry_fast(fp2_sub_lazy(fp2_sub_lazy(m.c2, t.c2), t.c1));
  r.h.c0 = fp2_select(hi, fp2_dbl(m.c0), lo.c0);
  r.h.c1 = fp2_select(hi, fp2_dbl(m.c1), lo.c1);
  r.h.c2 = fp2_select(hi, fp2_dbl(m.c2), lo.c2);
}

// ---- Fp4 = Fp2[s]/(s^2 - xi) on a quad.  An element X = x0 + x1 s has the components (x0.re, x0.im, x1.re, x1.im) = X[0..3]; component c lives on
// the lane q with q ^ 2 FX = c (FX = 0: x0 on the low pair; FX = 1: x0 on the high pair), likewise Y with FY; lane q returns component q ^ 2 FO of
//     X Y = (x0 y0 + xi x1 y1) + (x0 y1 + x1 y0) s
//   out0 = X0 Y0 - X1 Y1 + (X2 - X3) Y2 - (X2 + X3) Y3          out2 = X0 Y2 - X1 Y3 + X2 Y0 - X3 Y1
//   out1 = X0 Y1 + X1 Y0 + (X2 - X3) Y3 + (X2 + X3) Y2          out3 = X0 Y3 + X1 Y2 + X2 Y1 + X3 Y0
// i.e. out_j = sum_k xs_j[k] * sgn(j, k) * Y[j ^ k]: the x-operands are broadcasts (with the xi-twist for j < 2), the y-operands lane-xor fetches, the
// sign is minus for (j even, k odd).  x, y carried.  One four-term inner product per lane (magnitudes 1 + 1 + 2 + 2).
template <class C, int FX, int FY, int FO>
ELP_INL Fp<C> fp4q_mul(const Fp<C>& x, const Fp<C>& y) {
  const bool odd = pair_odd(), hi = quad_hi();
  const bool jlo = FO ? hi : !hi;                                // this lane computes out0 / out1
  const i32 ml = jlo ? -1 : 0;
  Fp<C> a[4], b[4];
  a[0] = fp_quad_bcast<C, 0 ^ (2 * FX)>(x);
  a[1] = fp_quad_bcast<C, 1 ^ (2 * FX)>(x);
  const Fp<C> X2 = fp_quad_bcast<C, 2 ^ (2 * FX)>(x), X3 = fp_quad_bcast<C, 3 ^ (2 * FX)>(x);
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) {
    a[2].v[i] = X2.v[i] - (X3.v[i] & ml);
    a[3].v[i] = X3.v[i] + (X2.v[i] & ml);
  }
  constexpr int D = 2 * (FO ^ FY);
  b[0] = fp_quad_xor<C, 0 ^ D>(y);
  b[1] = fp_cneg(!odd, fp_quad_xor<C, 1 ^ D>(y));
  b[2] = fp_quad_xor<C, 2 ^ D>(y);
  b[3] = fp_cneg(!odd, fp_quad_xor<C, 3 ^ D>(y));
  return fp_dot<C, 4>(a, b);
}

// Granger-Scott squaring in the cyclotomic subgroup (tower.h: fp12_cyc_sqr_inl).  The three Fp4 blocks (z0, z1) = (c0.c0, c1.c1), (z2, z3) = (c1.c0, c0.c2),
// (z4, z5) = (c0.c1, c1.c2) each pair a coefficient of the low pair with one of the high pair: a block is one fp4q_mul, the low pair receives
// A0 = x0^2 + xi x1^2, the high pair A1 = 2 x0 x1.  Then z' = 3 A -+ 2 z on the lane that holds z, and the weak reduction.
template <class C>
ELP_INL void fp12q_cyc_sqr(Fp12Q<C>& r, const Fp12Q<C>& a) {
  const bool hi = quad_hi();
  const Fp<C> xa = fp_select(hi, a.h.c1.c, a.h.c0.c);           // (z0 | z1)
  const Fp<C> xb = fp_select(hi, a.h.c0.c, a.h.c2.c);           // (z3 | z2): x0 = z2 on the high pair
  const Fp<C> xc = fp_select(hi, a.h.c2.c, a.h.c1.c);           // (z4 | z5)
  Fp2<C> oA, oB, oC;
  oA.c = fp4q_mul<C, 0, 0, 0>(xa, xa);                          // (A0 | A1)
  oB.c = fp4q_mul<C, 1, 1, 0>(xb, xb);                          // (B0 | B1)
  oC.c = fp4q_mul<C, 0, 0, 0>(xc, xc);                          // (C0 | C1)
  // low:  c0 = z0' = 3 A0 - 2 z0    c1 = z4' = 3 B0 - 2 z4    c2 = z3' = 3 C0 - 2 z3
  // high: c0 = z2' = 3 xi C1 + 2 z2  c1 = z1' = 3 A1 + 2 z1    c2 = z5' = 3 B1 + 2 z5
  const Fp2<C> xC = fp2_carry_fast(fp2_mul_xi_lazy(oC));
  const Fp2<C> T0 = fp2_select(hi, xC, oA), T1 = fp2_select(hi, oA, oB), T2 = fp2_select(hi, oB, oC);
  auto upd = [&](const Fp2<C>& T, const Fp2<C>& z) {
    Fp2<C> o = fp2_add_lazy(fp2_add_lazy(fp2_add_lazy(T, T), T), fp2_add_lazy(z, z));   // high: 3 T + 2 z
    Fp2<C> l = fp2_sub_lazy(fp2_add_lazy(fp2_add_lazy(T, T), T), fp2_add_lazy(z, z));   // low:  3 T - 2 z
    Fp2<C> s = fp2_select(hi, o, l);
    fp2_reduce_weak(s);
    return s;
  };
  Fp6<C> t;
  t.c0 = upd(T0, a.h.c0);
  t.c1 = upd(T1, a.h.c1);
  t.c2 = upd(T2, a.h.c2);
  r.h = t;
}

// Karabina compressed squaring (tower.h: cyc_comp_sqr_inl): only b = z2 + z3 s and c = z4 + z5 s are carried.  The two Fp4 squarings of a step are
// INDEPENDENT, so here each lane pair owns one of them whole -- the low pair b, the high pair c, as two-lane squarings (fp4_sqr: three fp2_sqr, one
// base-field product per lane each) -- and only the results cross:   b' = 3 s C + 2 conj(b),   c' = 3 B - 2 conj(c)   (s C = xi C1 + C0 s).
template <class C>
struct CycCompQ {
  Fp2<C> x0, x1;     // low pair: (z2, z3); high pair: (z4, z5)
};
template <class C>
ELP_INL void fp12q_to_comp(CycCompQ<C>& r, const Fp12Q<C>& a) {
  // Fp12Q: low pair (c0, c1, c2) = (z0, z4, z3), high pair (c0, c1, c2) = (z2, z1, z5).  low wants (z2, z3): z2 from the high pair's c0; high wants (z4, z5): z4 from the low pair's c1
  const bool hi = quad_hi();
  const Fp2<C> give = fp2_select(hi, a.h.c0, a.h.c1);          // low gives z4, high gives z2
  r.x0 = fp2_quad_swap(give);
  r.x1 = a.h.c2;
}
template <class C>
ELP_INL void cyc_compq_sqr(CycCompQ<C>& r, const CycCompQ<C>& a) {
  const bool hi = quad_hi();
  Fp2<C> S0, S1;
  fp4_sqr<C>(S0, S1, a.x0, a.x1);                               // low: B, high: C
  const Fp2<C> P0 = fp2_quad_swap(S0), P1 = fp2_quad_swap(S1);  // low: C, high: B
  // low:  x0' = z2' = 3 xi C1 + 2 z2,  x1' = z3' = 3 C0 - 2 z3        high: x0' = z4' = 3 B0 - 2 z4,  x1' = z5' = 3 B1 + 2 z5
  const Fp2<C> xP1 = fp2_carry_fast(fp2_mul_xi_lazy(P1));
  const Fp2<C> T0 = fp2_select(hi, P0, xP1), T1 = fp2_select(hi, P1, P0);
  auto upd = [&](const Fp2<C>& T, const Fp2<C>& z, bool plus) {
    const Fp2<C> t3 = fp2_add_lazy(fp2_add_lazy(T, T), T), z2 = fp2_add_lazy(z, z);
    Fp2<C> s = fp2_select(plus, fp2_add_lazy(t3, z2), fp2_sub_lazy(t3, z2));
    fp2_reduce_weak(s);
    return s;
  };
  const Fp2<C> n0 = upd(T0, a.x0, !hi), n1 = upd(T1, a.x1, hi);
  r.x0 = n0;
  r.x1 = n1;
}
// the whole (z2, z3, z4, z5) on every lane pair, in the two-lane form of tower.h (for norms and decompression, which run on both pairs alike)
template <class C>
ELP_INL void compq_to_paired(CycComp<C>& r, const CycCompQ<C>& a) {
  const bool hi = quad_hi();
  const Fp2<C> p0 = fp2_quad_swap(a.x0), p1 = fp2_quad_swap(a.x1);
  r.z2 = fp2_select(hi, p0, a.x0);
  r.z3 = fp2_select(hi, p1, a.x1);
  r.z4 = fp2_select(hi, a.x0, p0);
  r.z5 = fp2_select(hi, a.x1, p1);
}
// this lane pair's half of a value both pairs hold in full
template <class C>
ELP_INL void fp12q_from_paired(Fp12Q<C>& r, const Fp12<C>& a) {
  r.h = fp6_select(quad_hi(), a.c1, a.c0);
}

// f <- f * l for a D-type line l = a + b w + c w^3 = (a, 0, 0) + (b, c, 0) w  (a, b, c carried Fp2 values, the same on both pairs):
//   low pair:  f0 a + (v f1)(b + c v),   high pair:  f1 a + f0 (b + c v)    =    X a + Z (b + c v),  X = own, Z = the other half (times v on the low pair)
//   c0 = X0 a + Z0 b + xi Z2 c,   c1 = X1 a + Z1 b + Z0 c,   c2 = X2 a + Z2 b + Z1 c        -- three three-term Fp2 inner products
// M-type line l = a + b v + c v w = (a, b, 0) + (0, c, 0) w:
//   low pair:  f0 (a + b v) + (v f1)(c v),  high pair:  f1 (a + b v) + f0 (c v)     =    X (a + b v) + Z' c with Z' = v Z
//   c0 = X0 a + xi X2 b + Z'0 c,   c1 = X1 a + X0 b + Z'1 c,   c2 = X2 a + X1 b + Z'2 c
template <class C>
ELP_INL void fp12q_mul_by_line(Fp12Q<C>& f, const Fp2<C>& a, const Fp2<C>& b, const Fp2<C>& c) {
  const bool hi = quad_hi();
  const Fp6<C> p = fp6_quad_swap(f.h);
  Fp6<C> Z;                                        // high: f0 (= p); low: v f1 = (xi p.c2, p.c0, p.c1)
  Z.c0 = fp2_select(hi, p.c0, fp2_mul_xi(p.c2));
  Z.c1 = fp2_select(hi, p.c1, p.c0);
  Z.c2 = fp2_select(hi, p.c2, p.c1);
  Fp2<C> r0, r1, r2;
  if constexpr (C::TWIST_D) {
    const Fp2<C> xz2 = fp2_mul_xi(Z.c2);
    const Fp2<C> A0[3] = {f.h.c0, Z.c0, xz2}, A1[3] = {f.h.c1, Z.c1, Z.c0}, A2[3] = {f.h.c2, Z.c2, Z.c1};
    const Fp2<C> Bv[3] = {a, b, c};
    r0 = fp2_dot<C, 3>(A0, Bv);
    r1 = fp2_dot<C, 3>(A1, Bv);
    r2 = fp2_dot<C, 3>(A2, Bv);
  } else {
    const Fp2<C> xx2 = fp2_mul_xi(f.h.c2), xz2 = fp2_mul_xi(Z.c2);     // Z' = v Z = (xi Z2, Z0, Z1)
    const Fp2<C> A0[3] = {f.h.c0, xx2, xz2}, A1[3] = {f.h.c1, f.h.c0, Z.c0}, A2[3] = {f.h.c2, f.h.c1, Z.c1};
    const Fp2<C> Bv[3] = {a, b, c};
    r0 = fp2_dot<C, 3>(A0, Bv);
    r1 = fp2_dot<C, 3>(A1, Bv);
    r2 = fp2_dot<C, 3>(A2, Bv);
  }
  f.h.c0 = r0;
  f.h.c1 = r1;
  f.h.c2 = r2;
}

// a^e (e > 0) in the cyclotomic subgroup: square-and-multiply with Granger-Scott squarings (pairing.h: fp12_exp_u64_gs)
template <class C>
ELP_INL void fp12q_exp_u64_gs(Fp12Q<C>& r, const Fp12Q<C>& a, u64 e) {
  Fp12Q<C> acc = a;
  int top = 63;
  while (!((e >> top) & 1)) top--;
  ELP_NOUNROLL
  for (int i = top - 1; i >= 0; i--) {
    fp12q_cyc_sqr<C>(acc, acc);
    if ((e >> i) & 1) fp12q_mul<C>(acc, acc, a);
  }
  r = acc;
}

}  // namespace elp
