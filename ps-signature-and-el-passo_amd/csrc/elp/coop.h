// Cooperative pairing check: one item on 16 or 32 lane pairs with an Fp2 register file in LDS, interpreting the level-scheduled programs that
// tools/gen_coop.py generates (elp/coop_prog_<curve>.h).  Replaces, for SMALL batches and for the tail of aggregated verification, the
// one-lane(-pair)-per-item evaluation of  pairing() + GT==  (src/ps-verifier.cc:31-34,134-137):
//     check: [ f_Q(P1) f_gg(P2) ]^((p^12-1)/r) == 1      (P1 = sig1, Q = K, P2 = -sig2; the lines of gg come precomputed from the key)
//     tail : [ F f_gg(P2) ]^((p^12-1)/r) == 1            (F = product of the batch's Miller values, P2 = -sum d_i sig2_i)
// A Miller loop followed by a final exponentiation is one dependency chain of ~600 Fp12-level operations; a lane pair that walks it alone
// needs ~8 ms whatever the batch size.  Inside every Fp12 operation 6 to 18 Fp2 products are independent: the program lists them level by
// level, lane pair q executes slot q of a level, a barrier separates levels (all lanes of an item sit in one wave, so the barrier is the
// in-order LDS pipeline).  Two classes of levels: products (one fp_mul_pair per lane: the lane of parity c computes component c) and linear
// steps (Fp-linear combinations with small integer 2x2 matrices on (re, im), accumulated in 64-bit limbs and finished by ONE parallel carry pass where the
// generator's magnitude analysis allows -- the products in between pull the values back towards p -- and by a reduction modulo p where not; loads of the
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
  const u32* terms;      // the 16-bit entries of the LIN operations, two per word; the entries of a chunk start at a word
  int nsteps;
  int out[6];            // registers of the result (c0.c0 c0.c1 c0.c2 c1.c0 c1.c1 c1.c2)
  const u32* chunk_off = nullptr;   // first entry WORD of every chunk of COOP_CHUNK steps (+ end): the kernels stage descriptors and entries through LDS chunk by chunk
  int np = 16;                      // lane pairs per item the program was scheduled for (descriptors per step); the kernels know it at compile time
  const u32* chunk_line = nullptr;  // per chunk: first | count << 16 of the line coefficients (Fp2 values of the fixed argument's lines) its LDL operations read
};

// The register file and the staged program live in LDS: on the device the interpreter takes LDS-address-space pointers, so that every access is a DS
// instruction (through generic pointers they become FLAT loads: measured 53 % of the kernel's wave-cycles waiting on them)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) i32 coop_i32;
typedef __attribute__((address_space(3))) u32 coop_u32;
typedef __attribute__((address_space(3))) uint16_t coop_u16;
typedef __attribute__((address_space(3))) unsigned long long coop_u64;
#else
typedef i32 coop_i32;
typedef u32 coop_u32;
typedef uint16_t coop_u16;
typedef unsigned long long coop_u64;
#endif

enum { COOP_OP_MUL = 0, COOP_OP_MULC = 1, COOP_OP_MULS = 2, COOP_OP_LIN = 3, COOP_OP_LDL = 4, COOP_OP_INV = 5, COOP_OP_SQR = 6, COOP_OP_NOP = 15 };
constexpr int COOP_IN_ONE = 4;             // register pinned to the constant (1, 0) in every program (tools/gen_coop.py IN_ONE; CoopTables<C>::IN_ONE is checked against it)
constexpr u32 COOP_LIN_LIGHT = 1u << 19;   // LIN descriptor: the combination keeps the light finish (tools/gen_coop.py analyse)

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
// (the generator bounds sum |coefficient| |value| of a combination by 600 p -- 8 000 p on BLS12-381 --, so the top limb stays below 2^31 and top * QK fits 64 bits), q p subtracted
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

// the light finish: ONE parallel carry pass, the integer value untouched.  The generator proves |value| < 48 p (BN254; 256 p on BLS12-381) for every register (top limb below 2^(LB-1),
// what a product asks of its operands) and that the products in between pull the magnitudes back (tools/gen_coop.py analyse); the other limbs come out
// carried (|limb| <= 2^(LB-1) + sum of |coefficients|).
template <class C>
ELP_INL Fp<C> coop_lin_carry(const i64* acc) {
  constexpr int NL = C::NL;
  Fp<C> r;
  i32 cy[NL];
  ELP_UNROLL
  for (int i = 0; i < NL - 1; i++) {
    const i32 lo = elp_balanced<C::LB>((u32)acc[i]);
    cy[i] = (i32)((acc[i] - lo) >> C::LB);
    r.v[i] = lo;
  }
  ELP_UNROLL
  for (int i = 1; i < NL - 1; i++) r.v[i] += cy[i - 1];
  const i64 top = acc[NL - 1] + cy[NL - 2];
#if defined(ELP_BOUND_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
  assert(top > -((i64)1 << (C::LB - 1)) && top < ((i64)1 << (C::LB - 1)));
#endif
  r.v[NL - 1] = (i32)top;
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
// (empty slot).  `consts`: the COOP_NCONST Fp2 constants in the layout of the register file (the kernels keep them in LDS); `lines_w`: the precomputed
// lines of the fixed argument as a flat array of Fp2 (a, b, c per line; stored un-carried by ml_precompute, carried here), seen as words, from coefficient
// `line_base` on (the kernels stage the coefficients a chunk of steps reads through LDS together with the chunk: read in place, each costs the step that
// needs it a global-memory round trip; the host twin passes the whole array and 0).
// (inlined into the kernels' step loop: as a call, the result would travel through the lane's private memory -- a global-memory round trip per step)
template <class C>
ELP_INL int coop_exec_desc(u32 d0, u32 d1, const coop_u16* terms, int comp, const coop_i32* R, const coop_i32* consts, const coop_i32* lines_w, int line_base, Fp<C>& out);
#if !defined(__HIP_DEVICE_COMPILE__)
template <class C>
ELP_INL int coop_exec_slot(const CoopProg& P, int step, int slot, int comp, const i32* R, const i32* consts, const Fp2<C>* lines, Fp<C>& out) {
  return coop_exec_desc<C>(P.prog[((size_t)step * P.np + slot) * 2], P.prog[((size_t)step * P.np + slot) * 2 + 1], reinterpret_cast<const uint16_t*>(P.terms), comp, R, consts,
                           reinterpret_cast<const i32*>(lines), 0, out);
}
#endif
// the same from the two descriptor words; `terms[d1 + t]` must be entry t of a LIN descriptor (the kernels pass a pointer into their LDS copy of the chunk)
template <class C>
ELP_INL int coop_exec_desc(u32 d0, u32 d1, const coop_u16* terms, int comp, const coop_i32* R, const coop_i32* consts, const coop_i32* lines_w, int line_base, Fp<C>& out) {
  const int op = (int)(d0 >> 28);
  if (op == COOP_OP_NOP) return -1;
  const int dst = (int)((d0 >> 20) & 255);
  if (op <= COOP_OP_MULS) {
    // component 0: a0 b0 - a1 b1;  component 1: a0 b1 + a1 b0  -- one two-term inner product  a0 y + a1 w  with a single reduction per lane.  The three kinds
    // differ only in WHERE y and w are read (selected by address, one instruction stream: the lanes of a step mix them freely):
    //   MUL  : b = register rb              y = b[comp], w = b[1 - comp]
    //   MULC : b = constant (rb << 4) | x   the same, from the constants
    //   MULS : b = (s, 0), s = component x of register rb:  y = comp ? 0 : s,  w = comp ? s : 0   (0 = the imaginary part of the pinned input ONE)
    // and w is negated on the lane of component 0.
    const int ra = (int)((d0 >> 12) & 255), rb = (int)((d0 >> 4) & 255), x = (int)(d0 & 15);
    const bool isc = op == COOP_OP_MULC, iss = op == COOP_OP_MULS;
    const coop_i32* pb = isc ? consts : R;
    const int regb = isc ? ((rb << 4) | x) : rb;
    const int so = (rb * 2 + x) * C::NL, zo = (COOP_IN_ONE * 2 + 1) * C::NL;
    const int oy = iss ? (comp ? zo : so) : (regb * 2 + comp) * C::NL;
    const int ow = iss ? (comp ? so : zo) : (regb * 2 + (comp ^ 1)) * C::NL;
    const Fp<C> a0 = coop_ld<C>(R, ra, 0), a1 = coop_ld<C>(R, ra, 1);
    Fp<C> y, w;
    ELP_UNROLL
    for (int i = 0; i < C::NL; i++) y.v[i] = pb[oy + i];
    ELP_UNROLL
    for (int i = 0; i < C::NL; i++) w.v[i] = pb[ow + i];
    const i32 sg = comp ? 0 : -1;                   // (v ^ sg) - sg = comp ? v : -v
    ELP_UNROLL
    for (int i = 0; i < C::NL; i++) w.v[i] = (w.v[i] ^ sg) - sg;
    out = fp_mul_pair<C>(a0, y, a1, w);
    return dst;
  }
  if (op == COOP_OP_SQR) {
    // a step of squarings only (the generator never mixes them with other products: lanes on two paths would pay for both):
    // component 0: (a0 + a1)(a0 - a1);  component 1: (2 a0) a1  -- ONE product of lazy sums per lane, a third fewer multiply-adds than the inner product
    const int ra = (int)((d0 >> 12) & 255);
    const Fp<C> a0 = coop_ld<C>(R, ra, 0), a1 = coop_ld<C>(R, ra, 1);
    Fp<C> x, y;
    ELP_UNROLL
    for (int i = 0; i < C::NL; i++) {
      x.v[i] = a0.v[i] + (comp ? a0.v[i] : a1.v[i]);
      y.v[i] = comp ? a1.v[i] : a0.v[i] - a1.v[i];
    }
    out = fp_mul<C>(x, y);
    return dst;
  }
  if (op == COOP_OP_LIN) {
    // this lane's entries: n0 of them for the real component, then n1 for the imaginary one; an entry names ONE source component (its word offset in the
    // register file) and a coefficient -- a diagonal term costs a lane one load, a xi-multiple two
    const int n0 = (int)(d0 & 127), n1 = (int)((d0 >> 7) & 127);
    const int n = comp ? n1 : n0;
    const coop_u16* ent = terms + d1 + (comp ? n0 : 0);
    i64 acc[C::NL];
    for (int i = 0; i < C::NL; i++) acc[i] = 0;
    // four entries per round (the lists are padded to multiples of four and start at multiples of 8 bytes): ONE 64-bit load of the entries, then all their
    // operands, then the multiply-adds -- two LDS round trips per FOUR entries (a lone wave has nothing else to hide them behind)
    const coop_u64* ent4 = reinterpret_cast<const coop_u64*>(ent);
    unsigned long long nw = ent4[0];               // the entries of round r + 1 are read while round r runs (past the last list: two words of padding)
    ELP_NOUNROLL
    for (int t = 0; t < n; t += 4) {
      const unsigned long long ew = nw;
      nw = ent4[(t >> 2) + 1];
      u32 e[4];
      ELP_UNROLL
      for (int q = 0; q < 4; q++) e[q] = (u32)(ew >> (16 * q)) & 0xFFFFu;
      Fp<C> v[4];
      ELP_UNROLL
      for (int q = 0; q < 4; q++) {
        const coop_i32* p = R + (e[q] >> 4);
        ELP_UNROLL
        for (int i = 0; i < C::NL; i++) v[q].v[i] = p[i];
      }
      ELP_UNROLL
      for (int q = 0; q < 4; q++) {
        const i32 c = ((i32)(e[q] << 28)) >> 28;
        ELP_UNROLL
        for (int i = 0; i < C::NL; i++) acc[i] += (i64)c * v[q].v[i];
      }
    }
    out = (d0 & COOP_LIN_LIGHT) ? coop_lin_carry<C>(acc) : coop_lin_finish<C>(acc);
    return dst;
  }
  if (op == COOP_OP_LDL) {
    const int k = (int)(d0 & 0xFFF) - line_base;
    const coop_i32* p = lines_w + (k * 2 + comp) * C::NL;
    ELP_UNROLL
    for (int i = 0; i < C::NL; i++) out.v[i] = p[i];
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
