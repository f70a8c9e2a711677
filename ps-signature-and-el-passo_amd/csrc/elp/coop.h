// Cooperative pairing check: one item on 16 or 32 lane pairs with an Fp2 register file in LDS, interpreting the level-scheduled programs that
// tools/gen_coop.py generates (elp/coop_prog_<curve>.h).  Replaces, for SMALL batches and for the tail of aggregated verification, the
// one-lane(-pair)-per-item evaluation of  pairing() + GT==  (src/ps-verifier.cc:31-34,134-137):
//     check: [ f_Q(P1) f_gg(P2) ]^((p^12-1)/r) == 1      (P1 = sig1, Q = K, P2 = -sig2; the lines of gg come precomputed from the key)
//     tail : [ F f_gg(P2) ]^((p^12-1)/r) == 1            (F = product of the batch's Miller values, P2 = -sum d_i sig2_i)
// A Miller loop followed by a final exponentiation is one dependency chain of ~600 Fp12-level operations; a lane pair that walks it alone
// needs ~8 ms whatever the batch size.  Inside every Fp12 operation 6 to 18 Fp2 products are independent: the program lists them level by
// level, lane pair q executes slot q of a level, a barrier separates levels (all lanes of an item sit in one wave, so the barrier is the
// in-order LDS pipeline).  Two classes of levels: products (one fp_mul_pair per lane: the lane of parity c computes component c) and linear
// steps (Fp-linear combinations with small integer 2x2 matrices on (re, im), accumulated in 64-bit limbs and weakly reduced; loads of the
// fixed lines; the one base-field inversion of the final exponentiation).  The value computed is the exact GT element (Devegili-Scott-Dahab
// hard part): gen_coop.py validates the scheduled program against the big-int model bit for bit, tests/ compare the kernels with the oracle.
#pragma once
#include "pairing.h"
#if !defined(__HIP_DEVICE_COMPILE__)
#include <vector>
#endif

namespace elp {

struct CoopProg {        // one scheduled program (device pointers on the device, plain arrays on the host twin)
  const u32* prog;       // nsteps x NP descriptors of two words
  const uint8_t* cls;    // nsteps: 1 = product step
  const u32* terms;      // term table of the LIN operations
  int nsteps;
  int out[6];            // registers of the result (c0.c0 c0.c1 c0.c2 c1.c0 c1.c1 c1.c2)
  const u32* chunk_off = nullptr;   // first term of every chunk of COOP_CHUNK steps (+ end): the kernels stage descriptors and terms through LDS chunk by chunk
  int np = 16;                      // lane pairs per item the program was scheduled for (descriptors per step); the kernels know it at compile time
};

// The register file and the staged program live in LDS: on the device the interpreter takes LDS-address-space pointers, so that every access is a DS
// instruction (through generic pointers they become FLAT loads: measured 53 % of the kernel's wave-cycles waiting on them)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) i32 coop_i32;
typedef __attribute__((address_space(3))) u32 coop_u32;
#else
typedef i32 coop_i32;
typedef u32 coop_u32;
#endif

enum { COOP_OP_MUL = 0, COOP_OP_MULC = 1, COOP_OP_MULS = 2, COOP_OP_LIN = 3, COOP_OP_LDL = 4, COOP_OP_INV = 5, COOP_OP_NOP = 15 };

// register file of one item: R[reg][component][limb]
template <class C>
ELP_HD constexpr int coop_reg_words() { return 2 * C::NL; }
template <class C>
ELP_INL Fp<C> coop_ld(const coop_i32* R, int reg, int comp) {
  Fp<C> r;
  const coop_i32* p = R + (reg * 2 + comp) * C::NL;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) r.v[i] = p[i];
  return r;
}
template <class C>
ELP_INL void coop_st(coop_i32* R, int reg, int comp, const Fp<C>& a) {
  coop_i32* p = R + (reg * 2 + comp) * C::NL;
  ELP_UNROLL
  for (int i = 0; i < C::NL; i++) p[i] = a.v[i];
}

// sum of 64-bit limb accumulators -> carried limbs of a value in (-1.5 p, 1.5 p): sequential carry, quotient by p estimated from the top limb
// (the generator bounds the coefficient mass of a combination by 200, so the top limb stays below 2^31 and top * QK fits 64 bits), q p subtracted
template <class C>
ELP_INL Fp<C> coop_lin_finish(const i64* acc) {
  constexpr int NL = C::NL;
  static_assert(C::QI == NL - 1, "the quotient estimate reads the top limb only");
  i64 w[NL];
  i64 t = 0;
  ELP_UNROLL
  for (int i = 0; i < NL - 1; i++) {
    t += acc[i];
    const i32 lo = elp_balanced<C::LB>((u32)t);
    w[i] = lo;
    t = (t - lo) >> C::LB;
  }
  w[NL - 1] = t + acc[NL - 1];
  const i64 q = (w[NL - 1] * (i64)C::QK + ((i64)1 << (C::QS - 1))) >> C::QS;
  Fp<C> r;
  t = 0;
  ELP_UNROLL
  for (int i = 0; i < NL - 1; i++) {
    t += w[i] - q * (i64)C::modl(i);
    const i32 lo = elp_balanced<C::LB>((u32)t);
    r.v[i] = lo;
    t = (t - lo) >> C::LB;
  }
  r.v[NL - 1] = (i32)(t + w[NL - 1] - q * (i64)C::modl(NL - 1));
  return r;
}

// constant `id` of the MULC operations, component `comp` (Montgomery form), from the curve parameters
template <class C>
ELP_HEAVY Fp<C> coop_const(const uint8_t (*kind)[3], int id, int comp) {
  const int k0 = kind[id][0], n = kind[id][1], k = kind[id][2];
  Fp<C> r = fp_zero<C>();
  if (k0 == 0) {
    if (comp == 0) ELP_LOAD_FP(r, C::inv2(i_));
  } else if (k0 == 1) {
    for (int i = 0; i < C::NL; i++) r.v[i] = C::twist_3b(comp, i);
  } else if (k0 == 2) {
    for (int i = 0; i < C::NL; i++) r.v[i] = n == 1 ? C::frob1(k, comp, i) : (n == 2 ? C::frob2(k, comp, i) : C::frob3(k, comp, i));
  } else {
    for (int i = 0; i < C::NL; i++) r.v[i] = C::g2frob(n, k, comp, i);
  }
  return r;
}

// One slot of one step for the lane of parity `comp`: computes the component `comp` of the slot's result.  Returns the destination register or -1
// (empty slot).  `consts`: the COOP_NCONST Fp2 constants in the layout of the register file (the kernels keep them in LDS); `lines`: the precomputed
// lines of the fixed argument as a flat array of Fp2 (a, b, c per line; stored un-carried by ml_precompute, carried here).
// (inlined into the kernels' step loop: as a call, the result would travel through the lane's private memory -- a global-memory round trip per step)
template <class C>
ELP_INL int coop_exec_desc(u32 d0, u32 d1, const coop_u32* terms, int comp, const coop_i32* R, const coop_i32* consts, const Fp2<C>* lines, Fp<C>& out);
#if !defined(__HIP_DEVICE_COMPILE__)
template <class C>
ELP_INL int coop_exec_slot(const CoopProg& P, int step, int slot, int comp, const i32* R, const i32* consts, const Fp2<C>* lines, Fp<C>& out) {
  return coop_exec_desc<C>(P.prog[((size_t)step * P.np + slot) * 2], P.prog[((size_t)step * P.np + slot) * 2 + 1], P.terms, comp, R, consts, lines, out);
}
#endif
// the same from the two descriptor words; `terms[d1 + t]` must be term t of a LIN descriptor (the kernels pass a pointer into their LDS copy of the chunk)
template <class C>
ELP_INL int coop_exec_desc(u32 d0, u32 d1, const coop_u32* terms, int comp, const coop_i32* R, const coop_i32* consts, const Fp2<C>* lines, Fp<C>& out) {
  const int op = (int)(d0 >> 28);
  if (op == COOP_OP_NOP) return -1;
  const int dst = (int)((d0 >> 20) & 255);
  if (op <= COOP_OP_MULS) {
    const int ra = (int)((d0 >> 12) & 255), rb = (int)((d0 >> 4) & 255), x = (int)(d0 & 15);
    const Fp<C> a0 = coop_ld<C>(R, ra, 0), a1 = coop_ld<C>(R, ra, 1);
    Fp<C> b0, b1;
    if (op == COOP_OP_MUL) {
      b0 = coop_ld<C>(R, rb, 0);
      b1 = coop_ld<C>(R, rb, 1);
    } else if (op == COOP_OP_MULC) {
      b0 = coop_ld<C>(consts, (rb << 4) | x, 0);
      b1 = coop_ld<C>(consts, (rb << 4) | x, 1);
    } else {                                     // a * (Fp scalar): b = (s, 0)
      b0 = coop_ld<C>(R, rb, x);
      b1 = fp_zero<C>();
    }
    // component 0: a0 b0 - a1 b1;  component 1: a0 b1 + a1 b0  -- one two-term inner product with a single reduction per lane
    const Fp<C> y = comp ? b1 : b0;
    const Fp<C> w = comp ? b0 : fp_neg(b1);
    out = fp_mul_pair<C>(a0, y, a1, w);
    return dst;
  }
  if (op == COOP_OP_LIN) {
    const int nt = (int)(d0 & 0xFFFFF);
    i64 acc[C::NL];
    for (int i = 0; i < C::NL; i++) acc[i] = 0;
    const int sh = comp ? 0 : 8;                  // row of the matrix that produces this lane's component: (m00 m01) or (m10 m11)
    // four terms per round: the term words first, then all their operands (unconditionally: a zero coefficient costs nothing but the load), then the
    // multiply-adds -- two LDS round trips per FOUR terms instead of two per term (a lone wave has nothing else to hide them behind)
    ELP_NOUNROLL
    for (int t = 0; t < nt; t += 4) {
      u32 tw[4];
      ELP_UNROLL
      for (int q = 0; q < 4; q++) tw[q] = (t + q < nt) ? terms[d1 + t + q] : 0u;      // a zero word: register 0, coefficients 0
      Fp<C> v0[4], v1[4];
      ELP_UNROLL
      for (int q = 0; q < 4; q++) {
        const int r = (int)((tw[q] >> 16) & 255);
        v0[q] = coop_ld<C>(R, r, 0);
        v1[q] = coop_ld<C>(R, r, 1);
      }
      ELP_UNROLL
      for (int q = 0; q < 4; q++) {
        const i32 c0 = ((i32)((tw[q] >> (sh + 4)) << 28)) >> 28, c1 = ((i32)((tw[q] >> sh) << 28)) >> 28;
        ELP_UNROLL
        for (int i = 0; i < C::NL; i++) acc[i] += (i64)c0 * v0[q].v[i] + (i64)c1 * v1[q].v[i];
      }
    }
    out = coop_lin_finish<C>(acc);
    return dst;
  }
  if (op == COOP_OP_LDL) {
    const int k = (int)(d0 & 0xFFF);
    out = comp ? lines[k].c1 : lines[k].c0;
    fp_carry(out);
    return dst;
  }
  // COOP_OP_INV: (1 / re(a), 0)
  const int ra = (int)((d0 >> 12) & 255);
  out = comp ? fp_zero<C>() : fp_inv<C>(coop_ld<C>(R, ra, 0));
  return dst;
}

// Host-side sequential execution of a program over one register file (the host twin; also the reference for the device kernel's lane mapping):
// every slot of a step reads the registers as they were BEFORE the step.
#if !defined(__HIP_DEVICE_COMPILE__)
template <class C>
inline void coop_run_host(const CoopProg& P, i32* R, const Fp2<C>* consts2, int nconst, const Fp2<C>* lines) {
  std::vector<i32> cw((size_t)nconst * coop_reg_words<C>());
  for (int k = 0; k < nconst; k++) {
    coop_st<C>(cw.data(), k, 0, consts2[k].c0);
    coop_st<C>(cw.data(), k, 1, consts2[k].c1);
  }
  const i32* consts = cw.data();
  for (int s = 0; s < P.nsteps; s++) {
    Fp<C> res[32][2];
    int dst[32][2];
    for (int q = 0; q < P.np; q++)
      for (int c = 0; c < 2; c++) dst[q][c] = coop_exec_slot<C>(P, s, q, c, R, consts, lines, res[q][c]);
    for (int q = 0; q < P.np; q++)
      for (int c = 0; c < 2; c++)
        if (dst[q][c] >= 0) coop_st<C>(R, dst[q][c], c, res[q][c]);
  }
}
#endif

}  // namespace elp
