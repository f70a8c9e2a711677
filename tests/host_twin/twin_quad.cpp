// TEST INFRASTRUCTURE ONLY: host (g++) build of the FOUR-LANES-PER-ITEM layer (csrc/elp/quad.h, pair4.h).  The four lanes of a quad are four threads;
// exchanges inside a lane pair and across the quad are rendezvous (a divergence between lanes shows up as a hang reported after 60 s, or as a size
// mismatch), so the quad code is unit-tested against the one-lane formulas of tower.h / pairing.h in a container without a GPU -- also under
// -DELP_BOUND_CHECK, which redoes every limb operation in 64 bits.  Never linked into, or called from, the product.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "elp/pipeline.h"
#include "elp/pair4.h"
#include "elp/params_bls12_381.h"
#include "elp/params_bn254.h"

using namespace elp;

struct Bus {
  std::atomic<int> count{0};
  std::atomic<int> sense{0};
  const void* slot[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t bytes[4] = {0, 0, 0, 0};
};
static thread_local Bus* tl_pair_bus = nullptr;
static thread_local Bus* tl_quad_bus = nullptr;
static void bus_barrier(Bus* b, int parties) {
  const int s = b->sense.load(std::memory_order_acquire);
  if (b->count.fetch_add(1, std::memory_order_acq_rel) == parties - 1) {
    b->count.store(0, std::memory_order_relaxed);
    b->sense.store(1 - s, std::memory_order_release);
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (b->sense.load(std::memory_order_acquire) == s) {
    if ((++spins & 0xfff) == 0) {
      std::this_thread::yield();
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
        fprintf(stderr, "quad twin: the lanes of a %s diverged around an exchange (a partner never arrived)\n", parties == 2 ? "pair" : "quad");
        abort();
      }
    }
  }
}
static void pair_exchange(void* buf, size_t bytes) {
  Bus* b = tl_pair_bus;
  const int me = elp_pair_parity;
  b->slot[me] = buf;
  b->bytes[me] = bytes;
  bus_barrier(b, 2);
  if (b->bytes[1 - me] != bytes) {
    fprintf(stderr, "quad twin: exchange size mismatch between the lanes of a pair\n");
    abort();
  }
  unsigned char tmp[512];
  memcpy(tmp, b->slot[1 - me], bytes);
  bus_barrier(b, 2);
  memcpy(buf, tmp, bytes);
}
static void quad_gather(const void* own, void* all4, size_t bytes) {
  Bus* b = tl_quad_bus;
  const int me = elp_quad_lane;
  b->slot[me] = own;
  b->bytes[me] = bytes;
  bus_barrier(b, 4);
  for (int l = 0; l < 4; l++) {
    if (b->bytes[l] != bytes) {
      fprintf(stderr, "quad twin: exchange size mismatch between the lanes of a quad\n");
      abort();
    }
    memcpy((unsigned char*)all4 + l * bytes, b->slot[l], bytes);
  }
  bus_barrier(b, 4);
}
template <class Fn>
static int run_quad(Fn fn) {   // fn(lane) -> int; all four lanes must agree
  Bus quad, pairs[2];
  int res[4] = {-1, -2, -3, -4};
  auto lane = [&](int l) {
    elp_quad_lane = l;
    elp_pair_parity = l & 1;
    tl_pair_bus = &pairs[l >> 1];
    tl_quad_bus = &quad;
    elp_pair_exchange_hook = pair_exchange;
    elp_quad_gather_hook = quad_gather;
    res[l] = fn(l);
  };
  std::thread t1(lane, 1), t2(lane, 2), t3(lane, 3);
  lane(0);
  t1.join();
  t2.join();
  t3.join();
  elp_quad_lane = 0;
  elp_pair_parity = 0;
  if (res[0] != res[1] || res[0] != res[2] || res[0] != res[3]) {
    fprintf(stderr, "quad twin: the lanes of a quad returned different results (%d, %d, %d, %d)\n", res[0], res[1], res[2], res[3]);
    return -100;
  }
  return res[0];
}

template <class B>
static void load12(Fp12<B>& f, const u32* w) {
  Fp2<B>* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  for (int k = 0; k < 6; k++) {
    c[k]->c0 = fp_from_std<B>(fp_load_w<B>(w + (2 * k) * B::N));
    c[k]->c1 = fp_from_std<B>(fp_load_w<B>(w + (2 * k + 1) * B::N));
  }
}
template <class B>
static void store12(u32* w, const Fp12<B>& f) {
  const Fp2<B>* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  for (int k = 0; k < 6; k++) {
    fp_store_w<B>(w + (2 * k) * B::N, fp_to_std<B>(c[k]->c0));
    fp_store_w<B>(w + (2 * k + 1) * B::N, fp_to_std<B>(c[k]->c1));
  }
}
template <class B>
static void to_cyclotomic(Fp12<B>& f) {
  Fp12<B> t0, t1, g;
  fp12_inv<B>(t0, f);
  fp12_conj(t1, f);
  fp12_mul<B>(g, t1, t0);
  fp12_frob<B>(t0, g, 2);
  fp12_mul<B>(f, t0, g);
}
// every lane writes its own share of the plain-layout value
template <class C>
static void store_share(Fp12<typename PairInfo<C>::Base>& m, const Fp12Q<C>& a) {
  typedef typename PairInfo<C>::Base B;
  Fp6<B>& half = quad_hi() ? m.c1 : m.c0;
  (pair_odd() ? half.c0.c1 : half.c0.c0) = fp_cast<B>(a.h.c0.c);
  (pair_odd() ? half.c1.c1 : half.c1.c0) = fp_cast<B>(a.h.c1.c);
  (pair_odd() ? half.c2.c1 : half.c2.c0) = fp_cast<B>(a.h.c2.c);
}

// One Fp12-level operation on the quad and on one lane: op 0 product, 1 squaring, 2 Granger-Scott squaring, 3 `n` compressed squarings (b, c blocks compared), 4 Frobenius^n,
// 5 inverse, 6 f^z by the compressed chain, 7 sparse line product with (la, lb, lc) = the first three Fp2 coefficients of g, 8 conjugate, 9 f^|z| by the GS chain.
// Inputs f, g: 12 N canonical words each; ops 2, 3, 6, 9 first move f into the cyclotomic subgroup.  out_quad / out_plain: 12 N canonical words.
template <class B>
static int quad_op(int op, int n, const u32* fw, const u32* gw, u32* out_quad, u32* out_plain) {
  typedef Paired<B> C;
  Fp12<B> f, g, ref;
  load12<B>(f, fw);
  load12<B>(g, gw);
  if (op == 2 || op == 3 || op == 6 || op == 9) to_cyclotomic<B>(f);
  // one lane
  ref = f;
  if (op == 0) fp12_mul<B>(ref, f, g);
  else if (op == 1) fp12_sqr<B>(ref, f);
  else if (op == 2) fp12_cyc_sqr<B>(ref, f);
  else if (op == 3) {
    CycComp<B> k;
    fp12_to_comp<B>(k, f);
    for (int i = 0; i < n; i++) cyc_comp_sqr_inl<B>(k, k);
    ref.c1.c0 = k.z2; ref.c0.c2 = k.z3; ref.c0.c1 = k.z4; ref.c1.c2 = k.z5;
  } else if (op == 4) fp12_frob<B>(ref, f, n);
  else if (op == 5) fp12_inv<B>(ref, f);
  else if (op == 6) fp12_exp_z<B>(ref, f, nullptr);
  else if (op == 7) fp12_mul_by_line<B>(ref, g.c0.c0, g.c0.c1, g.c0.c2);
  else if (op == 8) fp12_conj(ref, f);
  else if (op == 9) fp12_exp_u64_gs<B>(ref, f, B::ZABS, nullptr);
  store12<B>(out_plain, ref);
  // four lanes
  Fp12<B> res = f;      // op 3 leaves (z0, z1) as they were
  const int rc = run_quad([&](int) {
    Fp12Q<C> a, b, r;
    fp12q_from_plain<C>(a, f);
    fp12q_from_plain<C>(b, g);
    r = a;
    if (op == 0) fp12q_mul<C>(r, a, b);
    else if (op == 1) fp12q_sqr<C>(r, a);
    else if (op == 2) fp12q_cyc_sqr<C>(r, a);
    else if (op == 3) {
      CycCompQ<C> k;
      fp12q_to_comp<C>(k, a);
      for (int i = 0; i < n; i++) cyc_compq_sqr<C>(k, k);
      CycComp<C> full;
      compq_to_paired<C>(full, k);
      // back into the Fp12Q slots: low (c1, c2) = (z4, z3), high (c0, c2) = (z2, z5)
      const bool hi = quad_hi();
      r.h.c0 = fp2_select(hi, full.z2, a.h.c0);
      r.h.c1 = fp2_select(hi, a.h.c1, full.z4);
      r.h.c2 = fp2_select(hi, full.z5, full.z3);
    } else if (op == 4) fp12q_frob<C>(r, a, n);
    else if (op == 5) fp12q_inv<C>(r, a);
    else if (op == 6) fp12q_exp_z<C>(r, a);
    else if (op == 7) fp12q_mul_by_line<C>(r, fp2_from_mem<C>(g.c0.c0), fp2_from_mem<C>(g.c0.c1), fp2_from_mem<C>(g.c0.c2));
    else if (op == 8) fp12q_conj(r, a);
    else if (op == 9) fp12q_exp_u64_gs<C>(r, a, C::ZABS);
    store_share<C>(res, r);
    return 1;
  });
  store12<B>(out_quad, res);
  return rc;
}

// e(sig1, K) e(-sig2, gg) == 1 on four lanes (pair4.h) and on one (pipeline.h ps_pairing_check's two steps); the points as canonical words, gg's lines made here.
// Returns verdict_quad | verdict_plain << 1, or -1 if a point does not decode.
template <class B>
static int pair_check(const u32* sig1w, const u32* sig2w, const u32* Kw, const u32* ggw) {
  typedef Paired<B> C;
  Aff<F1<B>> s1, s2;
  Aff<F2<B>> K, gg;
  if (!g1_load<B>(s1, sig1w) || !g1_load<B>(s2, sig2w) || !g2_load<B>(K, Kw) || !g2_load<B>(gg, ggw)) return -1;
  std::vector<LineCoef<B>> lines(ml_num_lines<B>());
  ml_precompute<B>(lines.data(), gg);
  int plain;
  {
    Aff<F1<B>> ns2;
    aff_neg(ns2, s2);
    if (aff_is_inf(s2)) aff_set_inf(ns2);
    Fp12<B> f;
    const LineMem<B>* lp[1] = {lines.data()};
    miller_loop<B, 1, 1>(f, &s1, &K, &ns2, lp);
    plain = final_exp_is_one<B>(f, nullptr) ? 1 : 0;
  }
  const int quad = run_quad([&](int) {
    Aff<F1<C>> q1, q2;
    Aff<F2<C>> qK;
    q1.x = fp_cast<C>(s1.x); q1.y = fp_cast<C>(s1.y);
    q2.x = fp_cast<C>(s2.x); q2.y = fp_cast<C>(s2.y);
    qK.x = fp2_from_mem<C>(K.x);
    qK.y = fp2_from_mem<C>(K.y);
    return ps_pairing_check4<C>(lines.data(), q1, q2, qK) ? 1 : 0;
  });
  if (quad < 0) return quad;
  return quad | (plain << 1);
}

// debugging aid: the Miller value f_K(sig1) f_gg(-sig2) on four lanes and on one (canonical words)
template <class B>
static int miller_both(const u32* sig1w, const u32* sig2w, const u32* Kw, const u32* ggw, u32* out_quad, u32* out_plain) {
  typedef Paired<B> C;
  Aff<F1<B>> s1, s2, ns2;
  Aff<F2<B>> K, gg;
  if (!g1_load<B>(s1, sig1w) || !g1_load<B>(s2, sig2w) || !g2_load<B>(K, Kw) || !g2_load<B>(gg, ggw)) return -1;
  std::vector<LineCoef<B>> lines(ml_num_lines<B>());
  ml_precompute<B>(lines.data(), gg);
  aff_neg(ns2, s2);
  if (aff_is_inf(s2)) aff_set_inf(ns2);
  Fp12<B> f, res;
  const LineMem<B>* lp[1] = {lines.data()};
  miller_loop<B, 1, 1>(f, &s1, &K, &ns2, lp);
  store12<B>(out_plain, f);
  const int rc = run_quad([&](int) {
    Aff<F1<C>> q1, q2;
    Aff<F2<C>> qK;
    q1.x = fp_cast<C>(s1.x); q1.y = fp_cast<C>(s1.y);
    q2.x = fp_cast<C>(ns2.x); q2.y = fp_cast<C>(ns2.y);
    qK.x = fp2_from_mem<C>(K.x);
    qK.y = fp2_from_mem<C>(K.y);
    Fp12Q<C> fq;
    miller_loop4<C>(fq, q1, qK, q2, lines.data());
    store_share<C>(res, fq);
    return 1;
  });
  store12<B>(out_quad, res);
  return rc;
}
// ... and the verdict of the final exponentiation on a given value, four lanes | one lane << 1
template <class B>
static int final_both(const u32* fw) {
  typedef Paired<B> C;
  Fp12<B> f;
  load12<B>(f, fw);
  const int plain = final_exp_is_one<B>(f, nullptr) ? 1 : 0;
  const int quad = run_quad([&](int) {
    Fp12Q<C> a;
    fp12q_from_plain<C>(a, f);
    return final_exp_is_one4<C>(a) ? 1 : 0;
  });
  return quad < 0 ? quad : (quad | (plain << 1));
}

extern "C" {
#if !defined(TWINQ_CURVE) || TWINQ_CURVE == 0
int twinq_bn254_op(int op, int n, const u32* f, const u32* g, u32* oq, u32* op_) { return quad_op<BN254>(op, n, f, g, oq, op_); }
int twinq_bn254_miller(const u32* s1, const u32* s2, const u32* K, const u32* gg, u32* oq, u32* op_) { return miller_both<BN254>(s1, s2, K, gg, oq, op_); }
int twinq_bn254_final(const u32* f) { return final_both<BN254>(f); }
int twinq_bn254_pair_check(const u32* s1, const u32* s2, const u32* K, const u32* gg) { return pair_check<BN254>(s1, s2, K, gg); }
#endif
#if !defined(TWINQ_CURVE) || TWINQ_CURVE == 1
int twinq_bls_op(int op, int n, const u32* f, const u32* g, u32* oq, u32* op_) { return quad_op<BLS12_381>(op, n, f, g, oq, op_); }
int twinq_bls_miller(const u32* s1, const u32* s2, const u32* K, const u32* gg, u32* oq, u32* op_) { return miller_both<BLS12_381>(s1, s2, K, gg, oq, op_); }
int twinq_bls_final(const u32* f) { return final_both<BLS12_381>(f); }
int twinq_bls_pair_check(const u32* s1, const u32* s2, const u32* K, const u32* gg) { return pair_check<BLS12_381>(s1, s2, K, gg); }
#endif
}
