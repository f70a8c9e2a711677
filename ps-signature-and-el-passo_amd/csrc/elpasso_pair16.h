// Kernel of the pairing check with ONE ITEM PER 16-LANE ROW of a wave (round 6): included by the translation unit elpasso_bn254_pair16.hip only.
//
// Replaces, for small batches of PS verifications, the same  pairing() + GT==  of src/ps-verifier.cc:31-34 as the other layouts:  e(sig1, K) e(-sig2, gg) == 1.
// Twelve of the sixteen lanes of a row hold one base-field coefficient each of the Fp12 value f = sum_k f_k w^k (lane q = 2k + c).  The whole check is a flat PROGRAM
// of 691 steps (tools/gen_row16.py -> elpasso_pair16_prog.h, generated and SIMULATED against the big-int model there): in every step each lane evaluates ONE inner
// product  sum_t coeff_t * slot[a_t] * slot[b_t]  with a single Montgomery reduction (quad.h fp_dot) over operands that live in the row's LDS slots -- lane-dependent
// addresses and small integer coefficients from a table, one instruction stream for all lanes, no interpretation of operations: sums, xi-multiples, conjugations and the
// constant of the twist were expanded symbolically into the terms by the generator.  An Fp12 product is one step of twelve terms per lane (3.5 us for a lone wave
// against 7.5 us on four lanes per item, tools/ubench_row16.hip), a Granger-Scott squaring one step of four.
// A workgroup is one wave = four items; all exchanges go through LDS, the barriers are wave-local.
#pragma once
#include "elpasso_impl.h"
#include "elp/quad.h"
#define ROW16_DEV __constant__
// one translation unit per curve: elpasso_bn254_pair16.hip, elpasso_bls12_381_pair16.hip (the latter defines R16_BLS); tables and program from tools/gen_row16.py --curve ...
#ifdef R16_BLS
#include "elpasso_pair16_prog_bls12_381.h"
namespace r16t = row16_bls;
#else
#include "elpasso_pair16_prog.h"
namespace r16t = row16;
#endif

namespace elp {

#ifdef R16_BLS
typedef BLS12_381 R16C;
#else
typedef BN254 R16C;
#endif
constexpr int R16_NL = R16C::NL;
constexpr int R16_NLP = (R16_NL + 3) & ~3;                           // limbs of a slot padded to 16-byte words
constexpr int R16_REG0 = r16t::NSLOT;                               // the stored Fp12 values ("registers" of the program) follow the generator's slots
constexpr int R16_ROW_SLOTS = r16t::NSLOT + 12 * r16t::NREG;
constexpr int R16_ROWS = 4;
constexpr int R16_LDS_WORDS = (R16_ROWS * R16_ROW_SLOTS + r16t::NCONST) * R16_NLP;

// A workgroup is ONE wave: its LDS instructions are issued and executed in program order, so a step's writes cannot overtake the reads that precede them and the next
// step's reads see them -- no s_barrier and, above all, no s_waitcnt on the table loads that are in flight for the NEXT step.  Only the compiler must keep the order.
#if defined(__HIP_DEVICE_COMPILE__)
#define R16_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#else
#define R16_SYNC() ((void)0)
#endif

ELP_INL Fp<R16C> r16_ld(const i32* L, const i32* K, int s) {         // slot s of the row (below CONST_BASE) or of the shared constants
  const i32* p = s < r16t::CONST_BASE ? L + s * R16_NLP : K + (s - r16t::CONST_BASE) * R16_NLP;
  Fp<R16C> r;
  ELP_UNROLL
  for (int i = 0; i < R16_NL; i++) r.v[i] = p[i];
  return r;
}
ELP_INL Fp<R16C> r16_ld_row(const i32* L, int s) {                    // a slot of the row by its index, whatever the index (the stored values sit above CONST_BASE)
  Fp<R16C> r;
  ELP_UNROLL
  for (int i = 0; i < R16_NL; i++) r.v[i] = L[s * R16_NLP + i];
  return r;
}
ELP_INL void r16_st(i32* L, int s, const Fp<R16C>& a) {
  ELP_UNROLL
  for (int i = 0; i < R16_NL; i++) L[s * R16_NLP + i] = a.v[i];
}
ELP_INL Fp<R16C> r16_pair_swap(const Fp<R16C>& a) {                  // the other component of the lane's Fp2 coefficient: DPP quad_perm [1,0,3,2]
#if defined(__HIP_DEVICE_COMPILE__)
  Fp<R16C> r;
  ELP_UNROLL
  for (int i = 0; i < R16_NL; i++) r.v[i] = __builtin_amdgcn_update_dpp(0, a.v[i], 0xB1, 0xF, 0xF, true);
  return r;
#else
  return a;                                                          // (the host pass only parses the kernel)
#endif
}
// One inner product of NT terms.  Two forms with the same multiply-adds and the same bounds (quad.h fp_dot): COLUMN-scanning keeps all 2 NT operands in registers and
// walks the columns with one accumulator (fp_dot itself); OPERAND-scanning keeps 2 NL - 1 column accumulators and takes the terms one at a time -- 2 x NL + 2 (2 NL - 1)
// registers instead of 2 NT NL, which is what the 14-limb field needs (336 operand registers for a twelve-term product otherwise: 27 spilled, 212 parked in AGPRs).
#ifndef R16_DOT_OS
#ifdef R16_BLS
#define R16_DOT_OS 1
#else
#define R16_DOT_OS 0
#endif
#endif
template <int NT>
ELP_INL Fp<R16C> r16_dot(const i32* L, const i32* K, const u32 (&tb)[12]) {
#if R16_DOT_OS
  typedef R16C C;
  constexpr int NL = R16_NL;
  i64 col[2 * NL - 1];
  ELP_UNROLL
  for (int k = 0; k < 2 * NL - 1; k++) col[k] = 0;
  ELP_UNROLL
  for (int t = 0; t < NT; t++) {
    const Fp<R16C> a = r16_ld(L, K, (int)(tb[t] & 0xFF));
    Fp<R16C> b = r16_ld(L, K, (int)((tb[t] >> 8) & 0xFF));
    const i32 cf = (i32)(tb[t] << 8) >> 24;
    ELP_UNROLL
    for (int i = 0; i < NL; i++) b.v[i] *= cf;
    ELP_UNROLL
    for (int i = 0; i < NL; i++) {
      ELP_UNROLL
      for (int j = 0; j < NL; j++) ELP_MAC(col[i + j], a.v[i], b.v[j]);
    }
  }
  i32 m[NL], pl[NL];
  ELP_UNROLL
  for (int i = 0; i < NL; i++) pl[i] = elp_opaque(C::modl(i));
  Fp<R16C> r;
  i64 acc = 0;
  ELP_UNROLL
  for (int k = 0; k < NL; k++) {
    acc += col[k];
    ELP_UNROLL
    for (int i = 0; i < k; i++) ELP_MAC_S(acc, m[i], pl[k - i]);
    m[k] = elp_balanced30((u32)acc * C::INVL);
    ELP_MAC_S(acc, m[k], pl[0]);
    acc >>= ELP_LIMB_BITS;
  }
  ELP_UNROLL
  for (int k = NL; k < 2 * NL - 1; k++) {
    acc += col[k];
    ELP_UNROLL
    for (int i = k - NL + 1; i < NL; i++) ELP_MAC_S(acc, m[i], pl[k - i]);
    r.v[k - NL] = elp_balanced30((u32)acc);
    acc = (acc + ELP_LIMB_HALF) >> ELP_LIMB_BITS;
  }
  r.v[NL - 1] = (i32)acc;
  return r;
#else
  Fp<R16C> a[NT], b[NT];
  ELP_UNROLL
  for (int t = 0; t < NT; t++) {
    a[t] = r16_ld(L, K, (int)(tb[t] & 0xFF));
    b[t] = r16_ld(L, K, (int)((tb[t] >> 8) & 0xFF));
    const i32 cf = (i32)(tb[t] << 8) >> 24;                          // the signed 8-bit coefficient: +-1, 2, 3, 4, 6, 12
    ELP_UNROLL
    for (int i = 0; i < R16_NL; i++) b[t].v[i] *= cf;
  }
  return fp_dot<R16C, NT>(a, b);
#endif
}

// The program loop shared by the two kernels (TAIL: the closing step of aggregated verification).  Returns the row's verdict.
template <bool TAIL>
ELP_INL bool r16_run(i32* lds, i32* L, const i32* K, int row, int q, int c, bool live1, bool live2, const LineMem<R16C>* gg_lines, const Fp12<R16C>* Fglob, int dbg_pc,
                     i32* dbg) {
  const u32* const prog = TAIL ? r16t::PROG_TAIL : r16t::PROG;
  const int nprog = TAIL ? r16t::NPROG_TAIL : r16t::NPROG;
  u32 tb[12];
  u32 e = prog[0];
  {
    const int sid = (int)(e >> 8) & 0xFF;
    ELP_UNROLL
    for (int t = 0; t < 12; t++) tb[t] = (e & 0xFF) == 0 ? r16t::STEP_TERMS[sid][q][t] : 0u;
  }
  bool verdict = false;
  ELP_NOUNROLL
  for (int pc = 0; pc < nprog; pc++) {
    const u32 e_next = pc + 1 < nprog ? prog[pc + 1] : 5u;
    u32 tb_next[12];
    {                                                               // the table of the NEXT dot step travels while this step computes
      const int sidn = (int)(e_next >> 8) & 0xFF;
      const bool dotn = (e_next & 0xFF) == 0;
      ELP_UNROLL
      for (int t = 0; t < 12; t++) tb_next[t] = dotn ? r16t::STEP_TERMS[sidn][q][t] : 0u;
    }
    const int op = (int)(e & 0xFF), a0 = (int)(e >> 8) & 0xFF, a1 = (int)(e >> 16) & 0xFF;
    if (dbg && pc == dbg_pc) {                                      // the state BEFORE entry pc of the program: every slot of every row of workgroup 0
      if (blockIdx.x == 0)
        for (int s = (int)threadIdx.x; s < R16_ROWS * R16_ROW_SLOTS * R16_NLP; s += 64) dbg[s] = lds[s];
      return false;
    }
    if (op == 0) {
      const int nt = r16t::STEP_NT[a0], fl = r16t::STEP_FLAGS[a0];
      const u32 dst = r16t::STEP_DEST[a0][q];
      Fp<R16C> r;
      switch (nt) {
        case 1: r = r16_dot<1>(L, K, tb); break;
        case 2: r = r16_dot<2>(L, K, tb); break;
        case 3: r = r16_dot<3>(L, K, tb); break;
        case 4: r = r16_dot<4>(L, K, tb); break;
        case 6: r = r16_dot<6>(L, K, tb); break;
        case 8: r = r16_dot<8>(L, K, tb); break;
        default: r = r16_dot<12>(L, K, tb); break;
      }
      if (dst & (1u << 9)) {                                        // Granger-Scott: 3 * (inner product) +- 2 * (the lane's coefficient), weakly reduced
        const Fp<R16C> own = r16_ld(L, K, (int)(dst & 0xFF));
        const i32 s2 = (dst & (1u << 10)) ? 2 : -2;
        ELP_UNROLL
        for (int w = 0; w < R16_NL; w++) r.v[w] = 3 * r.v[w] + s2 * own.v[w];
        fp_carry<R16C>(r);
        fp_reduce_weak<R16C>(r);
      }
      const int tag = fl >> 1;
      const bool dead = (tag == 1 && !live1) || (tag == 2 && !live2);
      const bool wr = (dst & (1u << 8)) && !(dead && q < 12);
      Fp<R16C> xr;
      if (fl & 1) {                                                 // uniform: the exchange is executed by every lane
        const Fp<R16C> p = r16_pair_swap(r);
        xr = c == 0 ? fp_sub<R16C>(r, p) : fp_add<R16C>(p, r);
      }
      R16_SYNC();
      if (wr) r16_st(L, (int)(dst & 0xFF), r);
      if ((fl & 1) && wr && q < 12) r16_st(L, r16t::SLOT_X0 + q, xr);
      R16_SYNC();
    } else if (op == 1) {                                           // the fixed line a0: six base-field values of gg's precomputed line -> LF
      if (q < 6) r16_st(L, r16t::SLOT_LF + q, reinterpret_cast<const Fp<R16C>*>(&gg_lines[a0])[q]);
      R16_SYNC();
    } else if (op == 2) {                                           // one base-field inversion (lane 0 of the row)
      const Fp<R16C> x = r16_ld(L, K, a0);
      const Fp<R16C> y = fp_inv<R16C>(x);
      if (q == 0) r16_st(L, a1, y);
      R16_SYNC();
    } else if (op == 3) {                                           // stored value a0 -> W0 (with xi * value -> X0) or W1
      Fp<R16C> v = r16_ld_row(L, R16_REG0 + 12 * a0 + (q < 12 ? q : 0));
      const Fp<R16C> p = r16_pair_swap(v);
      if (q < 12) {
        if (a1 == 0) {
          r16_st(L, r16t::SLOT_W0 + q, v);
          r16_st(L, r16t::SLOT_X0 + q, c == 0 ? fp_sub<R16C>(v, p) : fp_add<R16C>(p, v));
        } else {
          r16_st(L, r16t::SLOT_W1 + q, v);
        }
      }
      R16_SYNC();
    } else if (op == 4) {                                           // W0 -> stored value a0
      if (q < 12) r16_st(L, R16_REG0 + 12 * a0 + q, r16_ld(L, K, r16t::SLOT_W0 + q));
      R16_SYNC();
    } else if (op == 6) {                                           // W1 <- F, the product of the batch's Miller values (plain layout: c0 <-> w^0, w^2, w^4; c1 <-> w^1, w^3, w^5)
      if (q < 12) r16_st(L, r16t::SLOT_W1 + q, reinterpret_cast<const Fp<R16C>*>(Fglob)[((q >> 1) & 1) * 6 + (q >> 2) * 2 + (q & 1)]);
      R16_SYNC();
    } else {                                                        // the result is 1 <=> every coefficient of W0 - 1 is zero modulo p
      Fp<R16C> v = r16_ld(L, K, r16t::SLOT_W0 + (q < 12 ? q : 1));
      if (q == 0) v = fp_sub<R16C>(v, r16_ld(L, K, r16t::SLOT_ONE));
      const bool z = fp_is_zero<R16C>(v);
      const unsigned long long bz = __ballot(z);
      verdict = (((unsigned)(bz >> (16u * (unsigned)row))) & 0xFFFu) == 0xFFFu;
    }
    e = e_next;
    ELP_UNROLL
    for (int t = 0; t < 12; t++) tb[t] = tb_next[t];
  }
  return verdict;
}

// flags[i] = todo[i] && sig1, sig2 decode && e(sig1, K) e(-sig2, gg) == 1 for items [4 blockIdx.x, 4 blockIdx.x + 4): one per row
__global__ void __launch_bounds__(64, 1) k_pair16(const LineMem<R16C>* gg_lines, const u32* recs, int rec_words, const uint8_t* todo, const u32* kws, size_t kstride,
                                                  uint8_t* flags, unsigned long long* accepted, size_t n, int dbg_pc = -1, i32* dbg = nullptr) {      // dbg: tools/pair16_check.hip
  __shared__ __attribute__((aligned(16))) i32 lds[R16_LDS_WORDS];
  const int row = (int)(threadIdx.x >> 4), q = (int)(threadIdx.x & 15), c = q & 1;
  i32* const L = lds + row * R16_ROW_SLOTS * R16_NLP;
  i32* const K = lds + R16_ROWS * R16_ROW_SLOTS * R16_NLP;
  const size_t item0 = (size_t)blockIdx.x * R16_ROWS + row;
  const bool in_range = item0 < n;
  const size_t i = in_range ? item0 : n - 1;                       // rows beyond the batch walk the program on a copy of the last item and publish nothing
  // ---- set-up: constants, zeroed slots, the item's points
  for (int s = (int)threadIdx.x; s < r16t::NCONST; s += 64) {
    ELP_UNROLL
    for (int w = 0; w < R16_NL; w++) K[s * R16_NLP + w] = r16t::CONSTS[s][w];
  }
  for (int s = q; s < R16_ROW_SLOTS; s += 16) {
    ELP_UNROLL
    for (int w = 0; w < R16_NLP; w++) L[s * R16_NLP + w] = 0;
  }
  __syncthreads();
  bool ok_in = todo[i] != 0;
  bool live1 = false, live2 = false;
  {
    const u32* rec = recs + i * (size_t)rec_words;
    // lane 0: sig1 -> P1, lane 1: sig2 -> P2 = (x, -y) (decoded and checked by the lane: range, on the curve); lanes 4..7: the four base-field words of K from the
    // launch workspace (pipeline.h vid_store_k: word w of the plain-layout Aff<F2> at kws[w * kstride + i]) -> T.X, T.Y and Q; T.Z = 1; f = 1
    bool ok_pt = true, inf_pt = false;
    if (q < 2) {
      Aff<F1<R16C>> pt;
      ok_pt = g1_load<R16C>(pt, rec + q * 2 * R16C::N);
      inf_pt = ok_pt && aff_is_inf(pt);
      if (q == 1) pt.y = fp_neg(pt.y);
      r16_st(L, (q == 0 ? r16t::SLOT_P1 : r16t::SLOT_P2), pt.x);
      r16_st(L, (q == 0 ? r16t::SLOT_P1 : r16t::SLOT_P2) + 1, pt.y);
    }
    bool k_zero = true;
    if (q >= 4 && q < 8) {
      Fp<R16C> v;
      u32 any = 0;
      ELP_UNROLL
      for (int w = 0; w < R16_NL; w++) {
        v.v[w] = (i32)kws[(size_t)((q - 4) * R16_NL + w) * kstride + i];
        any |= (u32)v.v[w];
      }
      k_zero = any == 0;
      r16_st(L, r16t::SLOT_T + (q - 4), v);
      r16_st(L, r16t::SLOT_Q + (q - 4), v);
    }
    if (q == 8) {
      const Fp<R16C> one = r16_ld(L, K, r16t::SLOT_ONE);
      r16_st(L, r16t::SLOT_T + 4, one);
      r16_st(L, r16t::SLOT_W0, one);
      r16_st(L, r16t::SLOT_X0, one);                                // xi * 1 = 1 + i
      r16_st(L, r16t::SLOT_X0 + 1, one);
    }
    if (q >= 12) r16_st(L, r16t::SLOT_E + (q - 12), r16_ld(L, K, r16t::CONST_E_INIT + (q - 12)));      // E = 3 b', E3 = 9 b' for Z = 1
    // the row's verdict on its inputs and which pairs take part: ballots over the wave, this row's 16 bits
    const unsigned long long b_bad = __ballot(!ok_pt), b_inf = __ballot(inf_pt), b_kz = __ballot(!k_zero);
    const unsigned sh = 16u * (unsigned)row;
    const unsigned bad = (unsigned)(b_bad >> sh) & 0x3u, inf = (unsigned)(b_inf >> sh) & 0x3u, knz = (unsigned)(b_kz >> sh) & 0xF0u;
    ok_in = ok_in && bad == 0;
    live1 = !(inf & 1u) && knz != 0;                                 // e(O, K) = e(sig1, O) = 1
    live2 = !(inf & 2u);
  }
  __syncthreads();
  const bool verdict = r16_run<false>(lds, L, K, row, q, c, live1, live2, gg_lines, nullptr, dbg_pc, dbg);
  const bool ok = ok_in && verdict && in_range;
  if (in_range && q == 0) flags[item0] = ok ? 1 : 0;
  if (accepted) {
    const unsigned long long b = __ballot(ok && q == 0);
    if (threadIdx.x == 0 && b != 0) atomicAdd(accepted, (unsigned long long)__popcll(b));
  }
}

// The closing step of aggregated verification on one row: agg_ok = [ F f_gg(-S2) ]^((p^12-1)/r) == 1     (S2 = sum d_i sig2_i, std affine; F: plain layout)
__global__ void __launch_bounds__(64, 1) k_agg_final16(const LineMem<R16C>* gg_lines, const Fp12<R16C>* F, const u32* s2_std, int* agg_ok) {
  __shared__ __attribute__((aligned(16))) i32 lds[R16_LDS_WORDS];
  const int row = (int)(threadIdx.x >> 4), q = (int)(threadIdx.x & 15), c = q & 1;
  i32* const L = lds + row * R16_ROW_SLOTS * R16_NLP;
  i32* const K = lds + R16_ROWS * R16_ROW_SLOTS * R16_NLP;
  for (int s = (int)threadIdx.x; s < r16t::NCONST; s += 64) {
    ELP_UNROLL
    for (int w = 0; w < R16_NL; w++) K[s * R16_NLP + w] = r16t::CONSTS[s][w];
  }
  for (int s = q; s < R16_ROW_SLOTS; s += 16) {
    ELP_UNROLL
    for (int w = 0; w < R16_NLP; w++) L[s * R16_NLP + w] = 0;
  }
  __syncthreads();
  bool ok_pt = true, inf_pt = false;
  if (q == 1) {
    Aff<F1<R16C>> pt;
    ok_pt = g1_load<R16C>(pt, s2_std);
    inf_pt = ok_pt && aff_is_inf(pt);
    r16_st(L, r16t::SLOT_P2, pt.x);
    r16_st(L, r16t::SLOT_P2 + 1, fp_neg(pt.y));
  }
  if (q == 8) {
    const Fp<R16C> one = r16_ld(L, K, r16t::SLOT_ONE);
    r16_st(L, r16t::SLOT_W0, one);
    r16_st(L, r16t::SLOT_X0, one);
    r16_st(L, r16t::SLOT_X0 + 1, one);
  }
  const unsigned long long b_bad = __ballot(!ok_pt), b_inf = __ballot(inf_pt);
  const bool ok_in = ((unsigned)b_bad & 0x2u) == 0;                     // row 0 decides (the other rows walk the program on zeros)
  const bool live2 = ((unsigned)(b_inf >> (16u * (unsigned)row)) & 0x2u) == 0;
  __syncthreads();
  const bool verdict = r16_run<true>(lds, L, K, row, q, c, false, live2, gg_lines, F, -1, nullptr);
  if (threadIdx.x == 0) *agg_ok = (ok_in && verdict) ? 1 : 0;
}

// The product tree of aggregated verification on rows: out[b] = prod in[32 b .. min(n, 32 b + 32)).  One wave per 32 values: every row takes eight of them -- seven
// times the step W0 <- W0 * W1 of the program (twelve terms per lane, one Montgomery reduction) -- and two more steps fold the four rows: nine products in sequence
// where a lane of k_fp12_reduce walks 63 full Fp12 products (65 536 items: 1 024 values -> 32 -> 1 in two launches; timing in profiles/r06_aggregated.md).
// Values are Fp12 in the plain layout (c0 <-> w^0, w^2, w^4; c1 <-> w^1, w^3, w^5), as k_verify_id_agg writes and k_agg_final16 reads them.
ELP_INL int r16_plain_index(int q) { return ((q >> 1) & 1) * 6 + (q >> 2) * 2 + (q & 1); }
ELP_INL void r16_mul_step(i32* L, const i32* K, const u32 (&tb)[12], int q, int c) {
  const Fp<R16C> r = r16_dot<12>(L, K, tb);
  const Fp<R16C> p = r16_pair_swap(r);
  const Fp<R16C> xr = c == 0 ? fp_sub<R16C>(r, p) : fp_add<R16C>(p, r);      // xi * (W0 * W1) -> X0, xi = 1 + i
  R16_SYNC();
  if (q < 12) {
    r16_st(L, r16t::SLOT_W0 + q, r);
    r16_st(L, r16t::SLOT_X0 + q, xr);
  }
  R16_SYNC();
}
__global__ void __launch_bounds__(64, 1) k_fp12_reduce16(const Fp12<R16C>* in, size_t n, Fp12<R16C>* out) {
  __shared__ __attribute__((aligned(16))) i32 lds[R16_LDS_WORDS];
  const int row = (int)(threadIdx.x >> 4), q = (int)(threadIdx.x & 15), c = q & 1, ql = q < 12 ? q : 0;
  i32* const L = lds + row * R16_ROW_SLOTS * R16_NLP;
  i32* const K = lds + R16_ROWS * R16_ROW_SLOTS * R16_NLP;
  for (int s = (int)threadIdx.x; s < r16t::NCONST; s += 64) {
    ELP_UNROLL
    for (int w = 0; w < R16_NL; w++) K[s * R16_NLP + w] = r16t::CONSTS[s][w];
  }
  for (int s = q; s < R16_ROW_SLOTS; s += 16) {
    ELP_UNROLL
    for (int w = 0; w < R16_NLP; w++) L[s * R16_NLP + w] = 0;
  }
  __syncthreads();
  u32 tb[12];
  ELP_UNROLL
  for (int t = 0; t < 12; t++) tb[t] = r16t::STEP_TERMS[r16t::STEP_MUL_W0_W1][q][t];
  Fp<R16C> one_q = r16_ld(L, K, r16t::SLOT_ONE);                       // the lane's coefficient of the value 1
  if (q != 0) {
    ELP_UNROLL
    for (int w = 0; w < R16_NL; w++) one_q.v[w] = 0;
  }
  const size_t base = (size_t)blockIdx.x * 32 + (size_t)row * 8;
  const int pi = r16_plain_index(ql);
  {                                                                    // W0 <- the row's first value (1 beyond the end), X0 <- xi * W0
    const Fp<R16C> v = base < n ? reinterpret_cast<const Fp<R16C>*>(in + base)[pi] : one_q;
    const Fp<R16C> p = r16_pair_swap(v);
    if (q < 12) {
      r16_st(L, r16t::SLOT_W0 + q, v);
      r16_st(L, r16t::SLOT_X0 + q, c == 0 ? fp_sub<R16C>(v, p) : fp_add<R16C>(p, v));
    }
  }
  const size_t wg_left = n - (size_t)blockIdx.x * 32;
  const int longest = wg_left < 8 ? (int)wg_left : 8;                  // values of row 0, the longest row (values fill the rows in order): uniform over the wave
  Fp<R16C> nxt = base + 1 < n ? reinterpret_cast<const Fp<R16C>*>(in + base + 1)[pi] : one_q;
  ELP_NOUNROLL
  for (int j = 1; j < longest; j++) {
    if (q < 12) r16_st(L, r16t::SLOT_W1 + q, nxt);
    R16_SYNC();
    nxt = (j + 1 < 8 && base + (size_t)j + 1 < n) ? reinterpret_cast<const Fp<R16C>*>(in + base + j + 1)[pi] : one_q;      // travels while the step computes
    r16_mul_step(L, K, tb, q, c);
  }
  // rows 0, 2 <- * rows 1, 3; row 0 <- * row 2 (the rows that only lend their value multiply by 1: the wave stays in one instruction stream)
  ELP_NOUNROLL
  for (int lvl = 1; lvl <= 2; lvl++) {
    const bool take = (row & (2 * lvl - 1)) == 0;
    const i32* const Lo = lds + (row + lvl) * R16_ROW_SLOTS * R16_NLP;
    const Fp<R16C> v = take ? r16_ld_row(Lo, r16t::SLOT_W0 + ql) : one_q;
    if (q < 12) r16_st(L, r16t::SLOT_W1 + q, v);
    R16_SYNC();
    r16_mul_step(L, K, tb, q, c);
  }
  if (row == 0 && q < 12) reinterpret_cast<Fp<R16C>*>(out + blockIdx.x)[pi] = r16_ld(L, K, r16t::SLOT_W0 + q);
}

}  // namespace elp

template <class B>
void launch_fp12_reduce16(hipStream_t stream, const void* in, size_t n, void* out) {
  static_assert(std::is_same<B, elp::R16C>::value, "this translation unit holds the other curve's tables");
  if (n == 0) return;
  hipLaunchKernelGGL(elp::k_fp12_reduce16, dim3((unsigned)((n + 31) / 32)), dim3(64), 0, stream, (const elp::Fp12<elp::R16C>*)in, n, (elp::Fp12<elp::R16C>*)out);
}
template <class B>
void launch_agg_final16(hipStream_t stream, const void* gg_lines, const void* F, const void* s2_std, int* agg_ok) {
  static_assert(std::is_same<B, elp::R16C>::value, "this translation unit holds the other curve's tables");
  hipLaunchKernelGGL(elp::k_agg_final16, dim3(1), dim3(64), 0, stream, (const elp::LineMem<elp::R16C>*)gg_lines, (const elp::Fp12<elp::R16C>*)F, (const u32*)s2_std, agg_ok);
}
template <class B>
void launch_pair16(hipStream_t stream, const void* gg_lines, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags,
                   void* d_accepted) {
  static_assert(std::is_same<B, elp::R16C>::value, "this translation unit holds the other curve's tables");
  if (n == 0) return;
  hipLaunchKernelGGL(elp::k_pair16, dim3((unsigned)((n + elp::R16_ROWS - 1) / elp::R16_ROWS)), dim3(64), 0, stream, (const elp::LineMem<elp::R16C>*)gg_lines,
                     (const u32*)d_records, words, todo, kws, kstride, d_flags, (unsigned long long*)d_accepted, n);
}
