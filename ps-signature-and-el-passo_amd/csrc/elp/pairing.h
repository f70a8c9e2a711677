// Optimal ate multi-pairing: shared-squaring Miller loop over several (P, Q) pairs + one final exponentiation.
// Replaces mcl pairing()/GT== (src/ps-verifier.cc:31-34,134-137,208-211; src/ps-requester.cc:133-136):
// the reference computes e(sig1,K) == e(sig2,gg) with two full pairings; GT never leaves the library, so the
// product form  e(sig1,K) * e(-sig2,gg) == 1  with one final exponentiation is observationally identical.
#pragma once
#include "curve.h"

namespace elp {

template <class C>
struct G2Proj {  // homogeneous projective point on the twist (x = X/Z, y = Y/Z)
  Fp2<C> X, Y, Z;
};
template <class C>
struct LineCoef {  // un-evaluated line: (a * y_P, b * x_P, c)
  Fp2<C> a, b, c;
};

template <class C>
ELP_INL Fp2<C> fp2_twist_3b() {
  Fp2<C> b;
  ELP_LOAD_FP2(b, C::twist_3b(c_, i_));
  return b;
}
// precomputed lines are kept in the plain layout in HBM (built once per key by the unpaired set-up kernel)
template <class C>
using LineMem = LineCoef<typename PairInfo<C>::Base>;
template <class C>
ELP_INL decltype(auto) line_from_mem(const LineMem<C>& m) {   // plain layout: the entry itself (no copy); paired: this lane's components
  if constexpr (is_paired<C>()) {
    LineCoef<C> l;
    l.a = fp2_from_mem<C>(m.a);
    l.b = fp2_from_mem<C>(m.b);
    l.c = fp2_from_mem<C>(m.c);
    return l;
  } else {
    return (m);
  }
}

// Tangent line at T and T <- 2T.   line = 2YZ * y_P  -  3X^2 * x_P  +  (Y^2 - 3b'Z^2)
template <class C>
ELP_INL void ml_dbl_step_inl(G2Proj<C>& T, LineCoef<C>& l) {
  Fp<C> inv2;
  ELP_LOAD_FP(inv2, C::inv2(i_));
  Fp2<C> A, B, Cz, E, Fq, G, H, J, t;
  if constexpr (fp_roomy<C>()) {
    // same formulas, sums left lazy (limb magnitudes in carried units in the comments); T stays carried, the line coefficients are
    // handed on lazily (a, b: 3; c: 2), which the sparse product in ml_apply_line is sized for
    fp2_mul<C>(A, T.X, T.Y);
    A = fp2_mul_fp(A, inv2);
    fp2_sqr<C>(B, T.Y);
    fp2_sqr<C>(Cz, T.Z);
    fp2_mul<C>(E, Cz, fp2_twist_3b<C>());
    Fq = fp2_add_lazy(fp2_add_lazy(E, E), E);                                  // 3
    G = fp2_mul_fp(fp2_add_lazy(B, Fq), inv2);                                 // 4 x 1
    fp2_sqr<C>(H, fp2_add(T.Y, T.Z));                                          // squared: carried
    H = fp2_sub_lazy(fp2_sub_lazy(H, B), Cz);                                  // 3
    fp2_sqr<C>(J, T.X);
    l.a = H;
    l.b = fp2_neg(fp2_add_lazy(fp2_add_lazy(J, J), J));                        // 3
    l.c = fp2_sub_lazy(B, E);                                                  // 2
    fp2_mul<C>(T.X, A, fp2_sub_lazy(B, Fq));                                   // 1 x 4
    fp2_sqr<C>(t, G);
    Fp2<C> E2;
    fp2_sqr<C>(E2, E);
    T.Y = fp2_carry_fast(fp2_sub_lazy(t, fp2_add_lazy(fp2_add_lazy(E2, E2), E2)));   // 4 -> 1
    fp2_mul<C>(T.Z, B, H);                                                     // 1 x 3
    return;
  }
  fp2_mul<C>(A, T.X, T.Y);
  A = fp2_mul_fp(A, inv2);
  fp2_sqr<C>(B, T.Y);
  fp2_sqr<C>(Cz, T.Z);
  fp2_mul<C>(E, Cz, fp2_twist_3b<C>());
  Fq = fp2_add(fp2_dbl(E), E);
  G = fp2_mul_fp(fp2_add(B, Fq), inv2);
  fp2_sqr<C>(H, fp2_add(T.Y, T.Z));
  H = fp2_sub(H, fp2_add(B, Cz));
  fp2_sqr<C>(J, T.X);
  l.a = H;
  l.b = fp2_neg(fp2_add(fp2_dbl(J), J));
  l.c = fp2_sub(B, E);
  fp2_mul<C>(T.X, A, fp2_sub(B, Fq));
  fp2_sqr<C>(t, G);
  Fp2<C> E2;
  fp2_sqr<C>(E2, E);
  T.Y = fp2_sub(t, fp2_add(fp2_dbl(E2), E2));
  fp2_mul<C>(T.Z, B, H);
}

template <class C>
ELP_HEAVY void ml_dbl_step(G2Proj<C>& T, LineCoef<C>& l) {
  ml_dbl_step_inl<C>(T, l);
}

// Chord through T and Q (affine) and T <- T + Q.   line = mu * y_P - theta * x_P + (theta x_Q - mu y_Q)
template <class C>
ELP_INL void ml_add_step_inl(G2Proj<C>& T, LineCoef<C>& l, const Fp2<C>& xq, const Fp2<C>& yq) {
  Fp2<C> theta, mu, t, Cc, D, E, Fq, G, H;
  if constexpr (fp_roomy<C>()) {
    fp2_mul<C>(t, yq, T.Z);
    theta = fp2_sub(T.Y, t);                                                   // squared below: carried
    fp2_mul<C>(t, xq, T.Z);
    mu = fp2_sub(T.X, t);
    l.a = mu;
    l.b = fp2_neg(theta);
    fp2_mul<C>(t, theta, xq);
    fp2_mul<C>(Cc, mu, yq);
    l.c = fp2_sub_lazy(t, Cc);                                                 // 2
    fp2_sqr<C>(Cc, theta);
    fp2_sqr<C>(D, mu);
    fp2_mul<C>(E, mu, D);
    fp2_mul<C>(Fq, T.Z, Cc);
    fp2_mul<C>(G, T.X, D);
    H = fp2_sub_lazy(fp2_add_lazy(E, Fq), fp2_add_lazy(G, G));                 // 4
    fp2_mul<C>(T.X, mu, H);                                                    // 1 x 4
    fp2_mul<C>(t, theta, fp2_sub_lazy(G, H));                                  // 1 x 5
    fp2_mul<C>(Cc, E, T.Y);
    T.Y = fp2_carry_fast(fp2_sub_lazy(t, Cc));                                 // 2 -> 1
    fp2_mul<C>(T.Z, T.Z, E);
    return;
  }
  fp2_mul<C>(t, yq, T.Z);
  theta = fp2_sub(T.Y, t);
  fp2_mul<C>(t, xq, T.Z);
  mu = fp2_sub(T.X, t);
  l.a = mu;
  l.b = fp2_neg(theta);
  fp2_mul<C>(t, theta, xq);
  fp2_mul<C>(Cc, mu, yq);
  l.c = fp2_sub(t, Cc);
  fp2_sqr<C>(Cc, theta);
  fp2_sqr<C>(D, mu);
  fp2_mul<C>(E, mu, D);
  fp2_mul<C>(Fq, T.Z, Cc);
  fp2_mul<C>(G, T.X, D);
  H = fp2_sub(fp2_add(E, Fq), fp2_dbl(G));
  fp2_mul<C>(T.X, mu, H);
  fp2_mul<C>(t, theta, fp2_sub(G, H));
  fp2_mul<C>(Cc, E, T.Y);
  T.Y = fp2_sub(t, Cc);
  fp2_mul<C>(T.Z, T.Z, E);
}

// f <- f * line(P).  One leaf routine (evaluation at P and the sparse product together): a separate wrapper around the product would
// be a non-leaf function and save / restore its callee-saved registers on every call.
template <class C>
ELP_HEAVY void ml_add_step(G2Proj<C>& T, LineCoef<C>& l, const Fp2<C>& xq, const Fp2<C>& yq) {
  ml_add_step_inl<C>(T, l, xq, yq);
}

template <class C>
ELP_INL void ml_apply_line_inl(Fp12<C>& f, const LineCoef<C>& l, const Fp<C>& xp, const Fp<C>& yp) {
  Fp2<C> a = fp2_mul_fp(l.a, yp);
  Fp2<C> b = fp2_mul_fp(l.b, xp);
  if (C::TWIST_D)
    fp12_mul_by_line_inl<C>(f, a, b, l.c);
  else
    fp12_mul_by_line_inl<C>(f, l.c, b, a);
}
template <class C>
ELP_HEAVY void ml_apply_line(Fp12<C>& f, const LineCoef<C>& l, const Fp<C>& xp, const Fp<C>& yp) {
  ml_apply_line_inl<C>(f, l, xp, yp);
}

// number of line coefficients produced for one fixed G2 argument
template <class C>
ELP_HD constexpr int ml_num_lines() {
  int n = 0;
  for (int i = 0; i < C::ATE_LEN; i++) n += 1 + (C::ate_naf(i) != 0 ? 1 : 0);
  return n + (C::IS_BN ? 2 : 0);
}

// Walk the Miller loop for a fixed Q and record every line (used once per public key for gg).
template <class C>
ELP_HEAVY void ml_precompute(LineCoef<C>* out, const Aff<F2<C>>& q) {
  G2Proj<C> T;
  T.X = q.x;
  T.Y = q.y;
  T.Z = fp2_one<C>();
  Fp2<C> nqy = fp2_neg(q.y);
  int n = 0;
  ELP_NOUNROLL
  for (int i = 0; i < C::ATE_LEN; i++) {
    ml_dbl_step<C>(T, out[n++]);
    int d = C::ate_naf(i);
    if (d != 0) ml_add_step<C>(T, out[n++], q.x, d > 0 ? q.y : nqy);
  }
  if (C::IS_BN) {
    if (C::Z_NEG) T.Y = fp2_neg(T.Y);
    Aff<F2<C>> q1, q2;
    g2_frob<C>(q1, q, 1);
    g2_frob<C>(q2, q, 2);
    ml_add_step<C>(T, out[n++], q1.x, q1.y);
    ml_add_step<C>(T, out[n++], q2.x, fp2_neg(q2.y));
  }
}

// f = prod_i f_{s,Q_i}(P_i) over NV pairs with run-time Q and NF pairs whose lines were precomputed.
// Pairs with P or Q at infinity contribute 1 (e(O, Q) = e(P, O) = 1).
#ifndef ELP_MILLER_REG_ON
#define ELP_MILLER_REG_ON 1
#endif
template <class C, int NV, int NF>
ELP_HEAVY void miller_loop(Fp12<C>& f, const Aff<F1<C>>* pv, const Aff<F2<C>>* qv, const Aff<F1<C>>* pf,
                           const LineMem<C>* const* lines) {
  ELP_NONLEAF();
  G2Proj<C> T[NV > 0 ? NV : 1];
  Fp2<C> nqy[NV > 0 ? NV : 1];
  bool live_v[NV > 0 ? NV : 1];
  bool live_f[NF > 0 ? NF : 1];
  for (int k = 0; k < NV; k++) {
    live_v[k] = !(aff_is_inf(pv[k]) || aff_is_inf(qv[k]));
    T[k].X = qv[k].x;
    T[k].Y = qv[k].y;
    T[k].Z = fp2_one<C>();
    nqy[k] = fp2_neg(qv[k].y);
  }
  for (int k = 0; k < NF; k++) live_f[k] = !aff_is_inf(pf[k]);
  LineCoef<C> l;
  int n = 0;
  // For the verification shape (one variable pair, one pair with precomputed lines) the step routines are inlined into this loop
  // (FUSE): as separate routines each call saves and restores the callee-saved registers it uses, which was the bulk of the
  // private-memory traffic of the whole verification.  The half-step loop keeps one copy of each routine per pair kind.  Other
  // shapes keep the calls.  Callers of a fused shape pass real pointers for arrays it does not read: constant null arguments
  // propagated into the fused body trip an illegal-instruction bug of the compiler.
  constexpr bool FUSE = (NV == 1 && NF <= 1) && fp_roomy<C>();   // 9-limb field only: for 13 limbs it doubles the compile time for ~2 %
  constexpr bool MREG = FUSE && (ELP_MILLER_REG_ON != 0);         // the Miller value stays in registers through the fused loop (+1-2 %)
  Fp12<C> fr;
  Fp12<C>& fw = MREG ? fr : f;
  fp12_set_one(fw);
  ELP_NOUNROLL
  for (int i = 0; i < C::ATE_LEN; i++) {
    if (i != 0) {
      if (FUSE) fp12_sqr_inl<C>(fw, fw); else fp12_sqr<C>(f, f);
    }
    const int d = C::ate_naf(i);
    ELP_NOUNROLL
    for (int half = 0; half < 2; half++) {          // 0: doubling step, 1: addition step (only for a non-zero digit)
      if (half == 1 && d == 0) break;
      if constexpr (NV == 1 && NF == 1 && FUSE) {
        if (live_v[0] && live_f[0]) {        // both pairs live: multiply the two lines first, then f once
          if (half == 0) {
            ml_dbl_step_inl<C>(T[0], l);
          } else {
            const Fp2<C> yq = fp2_select(d > 0, qv[0].y, nqy[0]);
            ml_add_step_inl<C>(T[0], l, qv[0].x, yq);
          }
          const LineCoef<C>& lf = line_from_mem<C>(lines[0][n]);
          if constexpr (C::TWIST_D)
            fp12_mul_by_two_lines_inl<C>(fw, fp2_mul_fp(l.a, pv[0].y), fp2_mul_fp(l.b, pv[0].x), l.c, fp2_mul_fp(lf.a, pf[0].y),
                                         fp2_mul_fp(lf.b, pf[0].x), lf.c);
          else
            fp12_mul_by_two_lines_m_inl<C>(fw, l.c, fp2_mul_fp(l.b, pv[0].x), fp2_mul_fp(l.a, pv[0].y), lf.c, fp2_mul_fp(lf.b, pf[0].x),
                                           fp2_mul_fp(lf.a, pf[0].y));
          n++;
          continue;
        }
      }
      ELP_UNROLL
      for (int k = 0; k < NV; k++) {
        if (!live_v[k]) continue;
        if (half == 0) {
          if (FUSE) ml_dbl_step_inl<C>(T[k], l); else ml_dbl_step<C>(T[k], l);
        } else {
          const Fp2<C> yq = fp2_select(d > 0, qv[k].y, nqy[k]);   // by value: no select between a private and a generic pointer
          if (FUSE) ml_add_step_inl<C>(T[k], l, qv[k].x, yq); else ml_add_step<C>(T[k], l, qv[k].x, yq);
        }
        if (FUSE) ml_apply_line_inl<C>(fw, l, pv[k].x, pv[k].y); else ml_apply_line<C>(f, l, pv[k].x, pv[k].y);
      }
      ELP_UNROLL
      for (int k = 0; k < NF; k++) {
        if (!live_f[k]) continue;
        if (FUSE) ml_apply_line_inl<C>(fw, line_from_mem<C>(lines[k][n]), pf[k].x, pf[k].y); else ml_apply_line<C>(f, line_from_mem<C>(lines[k][n]), pf[k].x, pf[k].y);
      }
      n++;
    }
  }
  if (MREG) f = fr;
  if (C::Z_NEG) fp12_conj(f, f);
  if (C::IS_BN) {
    for (int k = 0; k < NV; k++)
      if (live_v[k]) {
        if (C::Z_NEG) T[k].Y = fp2_neg(T[k].Y);
        Aff<F2<C>> q1, q2;
        g2_frob<C>(q1, qv[k], 1);
        g2_frob<C>(q2, qv[k], 2);
        ml_add_step<C>(T[k], l, q1.x, q1.y);
        ml_apply_line<C>(f, l, pv[k].x, pv[k].y);
        ml_add_step<C>(T[k], l, q2.x, fp2_neg(q2.y));
        ml_apply_line<C>(f, l, pv[k].x, pv[k].y);
      }
    for (int k = 0; k < NF; k++)
      if (live_f[k]) {
        ml_apply_line<C>(f, line_from_mem<C>(lines[k][n]), pf[k].x, pf[k].y);
        ml_apply_line<C>(f, line_from_mem<C>(lines[k][n + 1]), pf[k].x, pf[k].y);
      }
  }
}

// The Miller values of TWO variable pairs as one product, f = f_{Q0}(P0) f_{Q1}(P1) (aggregated verification with two items per lane, round 5; D-type twist over the roomy
// field, i.e. BN254): ONE squaring of the accumulator per step serves both pairs and their two lines are multiplied together before they meet f (fp12_mul_by_two_lines_inl) --
// per pair a squaring of Fp12 and about a quarter of a sparse product less per step than two separate loops.  A pair that is not `live` (rejected by the NIZK half, or with a
// point at infinity) walks the loop on a copy of the other pair's points, so that the lanes of a wave stay in step, and contributes the neutral line (1, 0, 0).
template <class C>
struct MillerTwoStash {      // what the loop reads in its addition steps only: parked in the lane's LDS hot slot (exactly its 108 words on BN254) instead of registers
  Fp2<C> qx[2], qy[2], nqy[2];
};
template <class C>
ELP_HEAVY void miller_loop_two(Fp12<C>& f, const Aff<F1<C>>* P, const Aff<F2<C>>* Q, const bool* live_in, u32* hot = nullptr) {
  static_assert(C::TWIST_D && C::IS_BN && fp_roomy<C>(), "written for BN254");
  const bool live0 = live_in[0] && !aff_is_inf(P[0]) && !aff_is_inf(Q[0]);
  const bool live1 = live_in[1] && !aff_is_inf(P[1]) && !aff_is_inf(Q[1]);
  // stand-ins for a dead pair: the other pair's points (both dead: whatever pair 1 holds -- nothing of it reaches f)
  Aff<F1<C>> p[2];
  p[0].x = fp_select(live0, P[0].x, P[1].x);
  p[0].y = fp_select(live0, P[0].y, P[1].y);
  p[1].x = fp_select(live1, P[1].x, P[0].x);
  p[1].y = fp_select(live1, P[1].y, P[0].y);
  MillerTwoStash<C> st_priv;
  MillerTwoStash<C>* sh = hot_as<MillerTwoStash<C>, C>(hot);
  MillerTwoStash<C>& st = sh ? *sh : st_priv;
  st.qx[0] = fp2_select(live0, Q[0].x, Q[1].x);
  st.qy[0] = fp2_select(live0, Q[0].y, Q[1].y);
  st.qx[1] = fp2_select(live1, Q[1].x, Q[0].x);
  st.qy[1] = fp2_select(live1, Q[1].y, Q[0].y);
  const bool live[2] = {live0, live1};
  G2Proj<C> T[2];
  for (int k = 0; k < 2; k++) {
    T[k].X = st.qx[k];
    T[k].Y = st.qy[k];
    T[k].Z = fp2_one<C>();
    st.nqy[k] = fp2_neg(st.qy[k]);
  }
  const Fp2<C> one = fp2_one<C>(), zero = fp2_zero<C>();
  Fp12<C> fr;
  fp12_set_one(fr);
  // the two lines of a step, evaluated at their P (or neutral), times the accumulator
  auto apply = [&](const LineCoef<C>* l) {
    Fp2<C> a[2], b[2], c[2];
    ELP_UNROLL
    for (int k = 0; k < 2; k++) {
      a[k] = fp2_select(live[k], fp2_mul_fp(l[k].a, p[k].y), one);
      b[k] = fp2_select(live[k], fp2_mul_fp(l[k].b, p[k].x), zero);
      c[k] = fp2_select(live[k], l[k].c, zero);
    }
    fp12_mul_by_two_lines_inl<C>(fr, a[0], b[0], c[0], a[1], b[1], c[1]);
  };
  ELP_NOUNROLL
  for (int i = 0; i < C::ATE_LEN; i++) {
    if (i != 0) fp12_sqr_inl<C>(fr, fr);
    const int d = C::ate_naf(i);
    ELP_NOUNROLL
    for (int half = 0; half < 2; half++) {
      if (half == 1 && d == 0) break;
      LineCoef<C> l[2];
      ELP_UNROLL          // the pair index must be a compile-time constant: indexed at run time, T and l live in private memory and every access is a round trip
      for (int k = 0; k < 2; k++) {
        if (half == 0) {
          ml_dbl_step_inl<C>(T[k], l[k]);
        } else {
          const Fp2<C> yq = fp2_select(d > 0, st.qy[k], st.nqy[k]);
          ml_add_step_inl<C>(T[k], l[k], st.qx[k], yq);
        }
      }
      apply(l);
    }
  }
  f = fr;
  if (C::Z_NEG) fp12_conj(f, f);
  // the two closing additions of the BN loop (Q1 = psi(Q), -Q2 = -psi^2(Q)), again as two-line products
  Aff<F2<C>> q1[2], q2[2];
  for (int k = 0; k < 2; k++) {
    if (C::Z_NEG) T[k].Y = fp2_neg(T[k].Y);
    Aff<F2<C>> qk;
    qk.x = st.qx[k];
    qk.y = st.qy[k];
    g2_frob<C>(q1[k], qk, 1);
    g2_frob<C>(q2[k], qk, 2);
  }
  fr = f;
  {
    LineCoef<C> l[2];
    for (int k = 0; k < 2; k++) ml_add_step<C>(T[k], l[k], q1[k].x, q1[k].y);
    apply(l);
    for (int k = 0; k < 2; k++) ml_add_step<C>(T[k], l[k], q2[k].x, fp2_neg(q2[k].y));
    apply(l);
  }
  f = fr;
}

// a^e (e > 0, 64-bit) for a in the cyclotomic subgroup: plain square-and-multiply with Granger-Scott squarings
template <class C>
ELP_HEAVY void fp12_exp_u64_gs(Fp12<C>& r, const Fp12<C>& a, u64 e, u32* hot = nullptr) {
  Fp12<C> acc_priv;
  Fp12<C>* ah = hot_as<Fp12<C>, C>(hot);
  Fp12<C>& acc = ah ? *ah : acc_priv;
  acc = a;
  int top = 63;
  while (!((e >> top) & 1)) top--;
  ELP_NOUNROLL
  for (int i = top - 1; i >= 0; i--) {
    fp12_cyc_sqr<C>(acc, acc);     // stays a call: with the squaring inlined and the accumulator in registers the loop spills more than it saves
    if ((e >> i) & 1) fp12_mul<C>(acc, acc, a);
  }
  r = acc;
}
// The same power with COMPRESSED squarings (tower.h, Karabina): a^e = prod over the set bits i of a^(2^i); the run a^(2^i) is carried in compressed form
// (6 instead of 9 Fp2 squarings per step), snapshots are taken at the set bits, decompressed with ONE shared Fp2 inversion and multiplied together.  The
// exponents of the final exponentiation have 3 (BN254: |z|) to 6 set bits.  If some snapshot has c = 0 (it cannot be decompressed this way; never seen on
// pairing values) the plain loop above is used instead, so the result is always the same element.
#ifndef ELP_COMPRESSED_SQR
#define ELP_COMPRESSED_SQR 1
#endif
template <class C>
ELP_HEAVY void fp12_exp_u64(Fp12<C>& r, const Fp12<C>& a, u64 e, u32* hot = nullptr) {
#if ELP_COMPRESSED_SQR
  // snapshots per exponent: |z| of BN254 has 2 set bits below the top, the BLS12-381 |z| five (+ the top bit itself: one run ends there); an exponent with more
  // set bits (the dense (z - 1) / 3 of the exact BLS12 chain) takes the plain loop below.  Sized per curve: the arrays are part of the hottest private frame.
  constexpr int MAXS = C::IS_BN ? 2 : 6;
  int top = 63;
  while (!((e >> top) & 1)) top--;
  int nset = 0;
  for (int i = 1; i <= top; i++) nset += (int)((e >> i) & 1);
  if (nset >= 1 && nset <= MAXS) {
    CycComp<C> snap[MAXS];
    Fp2<C> nrm[MAXS], pre[MAXS];
    CycComp<C> cur;
    fp12_to_comp<C>(cur, a);
    int k = 0;
    ELP_NOUNROLL
    for (int i = 1; i <= top;) {                 // one run of squarings per set bit (bit `top` is set, so the last run ends there)
      int j = i;
      while (!((e >> j) & 1)) j++;
      cyc_comp_sqr_n<C>(cur, j - i + 1);
      snap[k++] = cur;
      i = j + 1;
    }
    // shared inversion of the norms (Montgomery's trick over Fp2)
    Fp2<C> acc = fp2_one<C>();
    bool bad = false;
    ELP_NOUNROLL
    for (int j = 0; j < nset; j++) {
      nrm[j] = cyc_comp_norm<C>(snap[j]);
      bad |= fp2_is_zero<C>(nrm[j]);
      pre[j] = acc;
      fp2_mul<C>(acc, acc, nrm[j]);
    }
    if (!bad) {
      Fp2<C> inv;
      fp2_inv<C>(inv, acc);
      Fp12<C> prod, t;
      bool have = false;
      if (e & 1) {
        prod = a;
        have = true;
      }
      ELP_NOUNROLL
      for (int j = nset - 1; j >= 0; j--) {
        Fp2<C> ninv;
        fp2_mul<C>(ninv, inv, pre[j]);
        fp2_mul<C>(inv, inv, nrm[j]);
        cyc_decompress<C>(t, snap[j], ninv);
        if (have) {
          fp12_mul<C>(prod, prod, t);
        } else {
          prod = t;
          have = true;
        }
      }
      r = prod;
      return;
    }
  }
#endif
  fp12_exp_u64_gs<C>(r, a, e, hot);
}
// a^|z| for a in the cyclotomic subgroup
template <class C>
ELP_INL void fp12_exp_absz(Fp12<C>& r, const Fp12<C>& a, u32* hot = nullptr) {
  fp12_exp_u64<C>(r, a, C::ZABS, hot);
}
// a^z (signed z) in the cyclotomic subgroup, where inversion is conjugation
template <class C>
ELP_INL void fp12_exp_z(Fp12<C>& r, const Fp12<C>& a, u32* hot = nullptr) {
  fp12_exp_absz<C>(r, a, hot);
  if (C::Z_NEG) fp12_conj(r, r);
}

// f^((p^12 - 1)/r): easy part (p^6 - 1)(p^2 + 1), then the hard part (p^4 - p^2 + 1)/r.
// `hot` may hold f_in itself: it is consumed by the first two statements and reused as the accumulator of the exponentiations.
// EXACT = false (callers that only test the result against 1) computes a small multiple of the exponent: on BN curves the multiple below (FKR), on BLS12
// curves the hard part is taken to the THIRD power,
//     3 (p^4-p^2+1)/r = (z-1)^2 (z+p) (z^2+p^2-1) + 3,
// which replaces the one dense exponent of the exact chain, (z-1)/3 (32 of 63 bits set), by a fifth sparse power of z.  GT has prime order r and 3 does not
// divide r, so the cube is 1 exactly when the pairing product is.  Values that leave the library as GT bytes (elp_pairing) use EXACT = true.
template <class C, bool EXACT = true>
ELP_HEAVY void final_exp(Fp12<C>& r, const Fp12<C>& f_in, u32* hot = nullptr) {
  Fp12<C> f, t0, t1;
  fp12_inv<C>(t0, f_in);
  fp12_conj(t1, f_in);
  fp12_mul<C>(f, t1, t0);      // f^(p^6 - 1)
  fp12_frob<C>(t0, f, 2);
  fp12_mul<C>(f, t0, f);       // ^(p^2 + 1)
  if (ELP_BISECT_AT(3)) {
    r = f;
    return;
  }
  if constexpr (C::IS_BN && !EXACT) {
    // Fuentes-Castaneda, Knapp, Rodriguez-Henriquez: the multiple 2z(6z^2+3z+1) of the hard part (a non-zero integer below r, hence prime to it),
    //     l0 + l1 p + l2 p^2 + l3 p^3,   l2 = 6z + 6z^2 + 12z^3,  l1 = l2 - 2z,  l0 = l2 + 6z^2 + 1,  l3 = l1 - 1:
    // three powers of z, three squarings, ten products, three Frobenius maps (the exact chain below: thirteen products, four squarings, seven maps).
    Fp12<C> fz, f2z, f6z, f6z2, f12z3, a, b, t;
    fp12_exp_z<C>(fz, f, hot);
    fp12_cyc_sqr<C>(f2z, fz);
    fp12_cyc_sqr<C>(t, f2z);                // f^4z
    fp12_mul<C>(f6z, t, f2z);
    fp12_exp_z<C>(f6z2, f6z, hot);
    fp12_cyc_sqr<C>(t, f6z2);               // f^12z^2
    fp12_exp_z<C>(f12z3, t, hot);
    fp12_mul<C>(a, f12z3, f6z2);
    fp12_mul<C>(a, a, f6z);                 // f^l2
    fp12_conj(t, f2z);
    fp12_mul<C>(b, a, t);                   // f^l1
    fp12_mul<C>(r, a, f6z2);
    fp12_mul<C>(r, r, f);                   // f^l0
    fp12_frob<C>(t, b, 1);
    fp12_mul<C>(r, r, t);                   // (f^l1)^p
    fp12_frob<C>(t, a, 2);
    fp12_mul<C>(r, r, t);                   // (f^l2)^p^2
    fp12_conj(t, f);
    fp12_mul<C>(b, b, t);                   // f^l3
    fp12_frob<C>(t, b, 3);
    fp12_mul<C>(r, r, t);                   // (f^l3)^p^3
  } else if constexpr (C::IS_BN) {
    // Devegili-Scott-Dahab: (p^4-p^2+1)/r = p^3 + (6z^2+1) p^2 + (-36z^3-18z^2-12z+1) p + (-36z^3-30z^2-18z-2)
    // evaluated with the vectorial addition chain  y0 y1^2 y2^6 y3^12 y4^18 y5^30 y6^36.
    Fp12<C> fz, fz2, fz3, y0, y1, y2, y3, y4, y5, y6, T0, T1, t;
    fp12_exp_z<C>(fz, f, hot);
    fp12_exp_z<C>(fz2, fz, hot);
    fp12_exp_z<C>(fz3, fz2, hot);
    fp12_frob<C>(y0, f, 1);
    fp12_frob<C>(t, f, 2);
    fp12_mul<C>(y0, y0, t);
    fp12_frob<C>(t, f, 3);
    fp12_mul<C>(y0, y0, t);                 // f^p f^p2 f^p3
    fp12_conj(y1, f);                       // f^-1
    fp12_frob<C>(y2, fz2, 2);               // (f^z2)^p2
    fp12_frob<C>(y3, fz, 1);
    fp12_conj(y3, y3);                      // ((f^z)^p)^-1
    fp12_frob<C>(t, fz2, 1);
    fp12_mul<C>(y4, fz, t);
    fp12_conj(y4, y4);                      // (f^z (f^z2)^p)^-1
    fp12_conj(y5, fz2);                     // (f^z2)^-1
    fp12_frob<C>(t, fz3, 1);
    fp12_mul<C>(y6, fz3, t);
    fp12_conj(y6, y6);                      // (f^z3 (f^z3)^p)^-1
    fp12_cyc_sqr<C>(T0, y6);
    fp12_mul<C>(T0, T0, y4);
    fp12_mul<C>(T0, T0, y5);
    fp12_mul<C>(T1, y3, y5);
    fp12_mul<C>(T1, T1, T0);
    fp12_mul<C>(T0, T0, y2);
    fp12_cyc_sqr<C>(T1, T1);
    fp12_mul<C>(T1, T1, T0);
    fp12_cyc_sqr<C>(T1, T1);
    fp12_mul<C>(T0, T1, y1);
    fp12_mul<C>(T1, T1, y0);
    fp12_cyc_sqr<C>(T0, T0);
    fp12_mul<C>(r, T0, T1);
  } else {
    // BLS12 (Hayashida-Hayasaka-Teruya): (p^4-p^2+1)/r = ((z-1)^2/3) (z+p) (z^2+p^2-1) + 1, evaluated exactly:
    // a = f^((z-1)/3), a = a^(z-1), b = a^(z+p), c = b^(z^2+p^2-1), result = c * f.
    Fp12<C> a, b, c, t;
    if constexpr (EXACT) {
      fp12_exp_u64<C>(a, f, C::ZM1D3_ABS, hot);
      if (C::Z_NEG) fp12_conj(a, a);        // z - 1 < 0 as well
    } else {
      fp12_exp_z<C>(t, f, hot);
      if (ELP_BISECT_AT(4)) {
        r = t;
        return;
      }
      fp12_conj(b, f);
      fp12_mul<C>(a, t, b);                 // f^(z-1)
    }
    fp12_exp_z<C>(t, a, hot);
    fp12_conj(b, a);
    fp12_mul<C>(a, t, b);                   // a^(z-1)
    fp12_exp_z<C>(t, a, hot);
    fp12_frob<C>(b, a, 1);
    fp12_mul<C>(b, b, t);                   // a^(z+p)
    if (ELP_BISECT_AT(5)) {
      r = b;
      return;
    }
    fp12_exp_z<C>(t, b, hot);
    fp12_exp_z<C>(t, t, hot);
    fp12_frob<C>(c, b, 2);
    fp12_mul<C>(c, c, t);
    fp12_conj(t, b);
    fp12_mul<C>(c, c, t);                   // b^(z^2+p^2-1)
    if constexpr (EXACT) {
      fp12_mul<C>(r, c, f);
    } else {
      fp12_cyc_sqr<C>(t, f);
      fp12_mul<C>(t, t, f);                 // f^3
      fp12_mul<C>(r, c, t);
    }
  }
}
// final_exp(f) == 1, by the cheaper chain where there is one
template <class C>
ELP_INL bool final_exp_is_one(const Fp12<C>& f_in, u32* hot = nullptr) {
  Fp12<C> g;
  final_exp<C, false>(g, f_in, hot);
  return fp12_is_one(g);
}

}  // namespace elp
