// TEST INFRASTRUCTURE ONLY: host (g++) build of the device arithmetic headers under
// ps-signature-and-el-passo_amd/csrc/elp/, so the formulas can be unit-tested against the oracle in a container
// that has no GPU.  This library is never linked into, or called from, the product (libelpasso_hip.so).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "elp/pipeline.h"
#include "elp/params_bls12_381.h"
#include "elp/params_bn254.h"
#include "elp/coop.h"
#include "elp/coop_prog_bn254.h"
#if !defined(TWIN_PART) || TWIN_PART == 2
#include "elp/coop_prog_bls12_381.h"
#endif

using namespace elp;

template <class C>
struct TwinCtx {
  KeyCtx<C> key;
  std::vector<Aff<F1<C>>> t1, b1;
  std::vector<Aff<F2<C>>> t2, b2;
  std::vector<LineCoef<C>> lines;
  std::vector<u32> hot;   // stands in for the lane's LDS hot slot so the aliasing rules of KeyCtx::hot are exercised on the host
  std::vector<u32> vtab;  // stands in for the lane's slice of the launch workspace (KeyCtx::vtab); ELP_TWIN_NO_VTAB=1 exercises the private-memory fallback
  void *t1s = nullptr, *t2s = nullptr;   // sparse (calloc-backed, lazily committed) tables of ctx_new_sparse
  ~TwinCtx() {
    free(t1s);
    free(t2s);
  }
};

// miller_loop_two (pairing.h; BN254 only) beside the product of the two single loops: o_two / o_prod = the two Miller values in canonical form
template <class C>
static int twin_miller_two(const u32* P0, const u32* Q0, const u32* P1, const u32* Q1, int live_mask, u32* o_two, u32* o_prod) {
  if constexpr (C::TWIST_D && C::IS_BN && fp_roomy<C>()) {
    Aff<F1<C>> p[2];
    Aff<F2<C>> q[2];
    if (!g1_load<C>(p[0], P0) || !g2_load<C>(q[0], Q0) || !g1_load<C>(p[1], P1) || !g2_load<C>(q[1], Q1)) return 0;
    const bool live[2] = {(live_mask & 1) != 0, (live_mask & 2) != 0};
    Fp12<C> f2, fa, fb;
    miller_loop_two<C>(f2, p, q, live);
    fp12_set_one(fa);
    fp12_set_one(fb);
    if (live[0]) miller_loop<C, 1, 0>(fa, &p[0], &q[0], (const Aff<F1<C>>*)0, (const LineCoef<C>* const*)0);
    if (live[1]) miller_loop<C, 1, 0>(fb, &p[1], &q[1], (const Aff<F1<C>>*)0, (const LineCoef<C>* const*)0);
    fp12_mul<C>(fa, fa, fb);
    gt_store<C>(o_two, f2);
    gt_store<C>(o_prod, fa);
    return 1;
  } else {
    return -1;
  }
}
// one table entry per window for the digits of `k`: entry (j, d_j) = d_j * 2^(W j) * base, exactly what jac_acc_fixed will read
template <class F>
static void touch_entries(Aff<F>* base_tbl, const Aff<F>& base, int W, int nwin, int per, const Scalar& k) {
  if (aff_is_inf(base)) return;
  std::vector<Aff<F>> bj(nwin);
  table_window_bases<F>(bj.data(), base, W, nwin);
  const Scalar kr = scalar_mod_r<typename F::Curve>(k);       // the same signed recoding as jac_acc_fixed
  int carry = 0;
  for (int j = 0; j < nwin; j++) {
    int d = fixed_base_digit(kr, j, W, carry);
    if (d == 0) continue;
    if (d < 0) d = -d;
    Scalar s;
    for (int i = 0; i < 8; i++) s.v[i] = 0;
    s.v[0] = (u32)d;
    Jac<F> acc;
    jac_mul_var<F>(acc, bj[j], s);
    jac_to_aff<F>(base_tbl[(size_t)j * per + (d - 1)], acc);
  }
}

template <class F>
static void build_tables(std::vector<Aff<F>>& tbl, const std::vector<Aff<F>>& bases, int W, int nwin, int per) {
  tbl.resize(bases.size() * (size_t)nwin * per);
  std::vector<Aff<F>> bj(nwin);
  for (size_t b = 0; b < bases.size(); b++) {
    Aff<F>* base_tbl = tbl.data() + b * (size_t)nwin * per;
    if (aff_is_inf(bases[b])) {
      for (size_t i = 0; i < (size_t)nwin * per; i++) aff_set_inf(base_tbl[i]);
      continue;
    }
    table_window_bases<F>(bj.data(), bases[b], W, nwin);
    for (int j = 0; j < nwin; j++) table_fill_chunk<F>(base_tbl + (size_t)j * per, bj[j], 1, per);
  }
}

#define TWIN(C, pfx)                                                                                                   \
  extern "C" {                                                                                                         \
  void pfx##_fp_mul(const u32* a, const u32* b, u32* o) {                                                              \
    Fp<C> x = fp_from_std<C>(fp_load_w<C>(a)), y = fp_from_std<C>(fp_load_w<C>(b));                                    \
    fp_store_w<C>(o, fp_to_std<C>(fp_mul<C>(x, y)));                                                                   \
  }                                                                                                                    \
  void pfx##_fp_inv(const u32* a, u32* o) {                                                                            \
    fp_store_w<C>(o, fp_to_std<C>(fp_inv<C>(fp_from_std<C>(fp_load_w<C>(a)))));                                        \
  }                                                                                                                    \
  int pfx##_fp_sqrt(const u32* a, u32* o) {                                                                            \
    Fp<C> r;                                                                                                           \
    bool ok = fp_sqrt<C>(r, fp_from_std<C>(fp_load_w<C>(a)));                                                          \
    fp_store_w<C>(o, fp_to_std<C>(r));                                                                                 \
    return ok;                                                                                                         \
  }                                                                                                                    \
  int pfx##_fp2_sqrt(const u32* a, u32* o) {                                                                           \
    Fp2<C> x, r;                                                                                                       \
    r.c0 = r.c1 = fp_zero<C>();      /* a rejected input leaves r alone */                                             \
    x.c0 = fp_from_std<C>(fp_load_w<C>(a));                                                                            \
    x.c1 = fp_from_std<C>(fp_load_w<C>(a + C::N));                                                                     \
    bool ok = fp2_sqrt<C>(r, x);                                                                                       \
    fp_store_w<C>(o, fp_to_std<C>(r.c0));                                                                              \
    fp_store_w<C>(o + C::N, fp_to_std<C>(r.c1));                                                                       \
    return ok;                                                                                                         \
  }                                                                                                                    \
  int pfx##_g1_mul(const u32* P, const u32* k, u32* o) {                                                               \
    Aff<F1<C>> p, r;                                                                                                   \
    if (!g1_load<C>(p, P)) return 0;                                                                                   \
    Jac<F1<C>> j;                                                                                                      \
    jac_mul_var<F1<C>>(j, p, scalar_load_w(k));                                                                        \
    jac_to_aff<F1<C>>(r, j);                                                                                           \
    g1_store<C>(o, r);                                                                                                 \
    return 1;                                                                                                          \
  }                                                                                                                    \
  int pfx##_g2_mul(const u32* P, const u32* k, u32* o) {                                                               \
    Aff<F2<C>> p, r;                                                                                                   \
    if (!g2_load<C>(p, P)) return 0;                                                                                   \
    Jac<F2<C>> j;                                                                                                      \
    jac_mul_var<F2<C>>(j, p, scalar_load_w(k));                                                                        \
    jac_to_aff<F2<C>>(r, j);                                                                                           \
    g2_store<C>(o, r);                                                                                                 \
    return 1;                                                                                                          \
  }                                                                                                                    \
  int pfx##_g1_mul_glv(const u32* P, const u32* k, u32* o) {                                                           \
    Aff<F1<C>> p, r;                                                                                                   \
    if (!g1_load<C>(p, P)) return 0;                                                                                   \
    Jac<F1<C>> j;                                                                                                      \
    g1_mul_glv<C>(j, p, scalar_load_w(k));                                                                             \
    jac_to_aff<F1<C>>(r, j);                                                                                           \
    g1_store<C>(o, r);                                                                                                 \
    return 1;                                                                                                          \
  }                                                                                                                    \
  int pfx##_g2_mul_gls(const u32* P, const u32* k, u32* o) {                                                           \
    Aff<F2<C>> p, r;                                                                                                   \
    if (!g2_load<C>(p, P)) return 0;                                                                                   \
    Jac<F2<C>> j;                                                                                                      \
    g2_mul_gls<C>(j, p, scalar_load_w(k));                                                                             \
    jac_to_aff<F2<C>>(r, j);                                                                                           \
    g2_store<C>(o, r);                                                                                                 \
    return 1;                                                                                                          \
  }                                                                                                                    \
  /* the form the G2 job of small batches runs: multiples 1P .. 8P and their psi^j images laid out as k_vid_ktab writes them (WsTabPsi), psi read not recomputed */ \
  int pfx##_g2_mul_gls_psi(const u32* P, const u32* k, u32* o) {                                                       \
    typedef F2<C> G;                                                                                                   \
    Aff<G> p, r;                                                                                                       \
    if (!g2_load<C>(p, P)) return 0;                                                                                   \
    constexpr int EW = vtab_entry_words<G>();                                                                          \
    alignas(16) static thread_local u32 tab[8 * EW], psi[24 * EW];                                                     \
    Jac<G> jk[8];                                                                                                      \
    jac_multiples8<G>(jk, p);                                                                                          \
    for (int q = 0; q < 8; q++) {                                                                                      \
      Aff<G> a = p;                                                                                                    \
      if (q) jac_to_aff<G>(a, jk[q]);                                                                                  \
      vtab_store<G>(tab, q, a);                                                                                        \
      for (int j = 1; j < 4; j++) {                                                                                    \
        Aff<G> t = a;                                                                                                  \
        g2_psi_aff<C>(t, j);                                                                                           \
        vtab_store<G>(psi, (j - 1) * 8 + q, t);                                                                        \
      }                                                                                                                \
    }                                                                                                                  \
    Jac<G> j;                                                                                                          \
    g2_mul_gls_with<C, WsTabPsi<G>, true>(j, WsTabPsi<G>{tab, psi}, scalar_load_w(k));                                 \
    jac_to_aff<G>(r, j);                                                                                               \
    g2_store<C>(o, r);                                                                                                 \
    /* the four-lane form: one dimension per lane (g2_mul_gls_dim), folded pairwise as the lanes of a quad do: must give the same point */ \
    Jac<G> d[4];                                                                                                       \
    for (int q = 0; q < 4; q++) g2_mul_gls_dim<C, WsTabPsi<G>>(d[q], WsTabPsi<G>{tab, psi}, scalar_load_w(k), q);      \
    jac_add<G>(d[0], d[0], d[1]);                                                                                      \
    jac_add<G>(d[2], d[2], d[3]);                                                                                      \
    jac_add<G>(d[0], d[0], d[2]);                                                                                      \
    Aff<G> r4;                                                                                                         \
    jac_to_aff<G>(r4, d[0]);                                                                                           \
    u32 o4[4 * C::N];                                                                                                  \
    g2_store<C>(o4, r4);                                                                                               \
    return memcmp(o4, o, sizeof o4) == 0 ? 1 : -4;                                                                     \
  }                                                                                                                    \
  int pfx##_g1_add(const u32* P, const u32* Q, u32* o) {                                                               \
    Aff<F1<C>> p, q, r;                                                                                                \
    if (!g1_load<C>(p, P) || !g1_load<C>(q, Q)) return 0;                                                              \
    Jac<F1<C>> j;                                                                                                      \
    jac_from_aff(j, p);                                                                                                \
    jac_madd<F1<C>>(j, j, q);                                                                                          \
    jac_to_aff<F1<C>>(r, j);                                                                                           \
    g1_store<C>(o, r);                                                                                                 \
    return 1;                                                                                                          \
  }                                                                                                                    \
  int pfx##_g2_add(const u32* P, const u32* Q, u32* o) {                                                               \
    Aff<F2<C>> p, q, r;                                                                                                \
    if (!g2_load<C>(p, P) || !g2_load<C>(q, Q)) return 0;                                                              \
    Jac<F2<C>> j;                                                                                                      \
    jac_from_aff(j, p);                                                                                                \
    jac_madd<F2<C>>(j, j, q);                                                                                          \
    jac_to_aff<F2<C>>(r, j);                                                                                           \
    g2_store<C>(o, r);                                                                                                 \
    return 1;                                                                                                          \
  }                                                                                                                    \
  /* order-r subgroup membership of a point of E(Fp) (csrc/elp/pipeline.h g1_in_subgroup); -1 = not on the curve */    \
  int pfx##_g1_in_subgroup(const u32* P) {                                                                             \
    Aff<F1<C>> p;                                                                                                      \
    if (!g1_load<C>(p, P)) return -1;                                                                                  \
    return g1_in_subgroup<C>(p) ? 1 : 0;                                                                               \
  }                                                                                                                    \
  int pfx##_g1_decompress(const uint8_t* in, u32* o) {                                                                 \
    Aff<F1<C>> p;                                                                                                      \
    if (!g1_deserialize<C>(p, in)) return 0;                                                                           \
    g1_store<C>(o, p);                                                                                                 \
    return 1;                                                                                                          \
  }                                                                                                                    \
  int pfx##_g2_decompress(const uint8_t* in, u32* o) {                                                                 \
    Aff<F2<C>> p;                                                                                                      \
    if (!g2_deserialize<C>(p, in)) return 0;                                                                           \
    g2_store<C>(o, p);                                                                                                 \
    return 1;                                                                                                          \
  }                                                                                                                    \
  void pfx##_hash_to_g1(const uint8_t* msg, size_t len, u32* o) {                                                      \
    Aff<F1<C>> p;                                                                                                      \
    hash_and_map_to_g1<C>(p, msg, len);                                                                                \
    g1_store<C>(o, p);                                                                                                 \
  }                                                                                                                    \
  /* single pairing e(P,Q) -> GT (12 std field elements) ; mode 1: miller loop only; mode 2: cyc-sqr self-test */     \
  int pfx##_pairing(const u32* P, const u32* Q, u32* o, int mode) {                                                    \
    Aff<F1<C>> p;                                                                                                      \
    Aff<F2<C>> q;                                                                                                      \
    if (!g1_load<C>(p, P) || !g2_load<C>(q, Q)) return 0;                                                              \
    Fp12<C> f, g;                                                                                                      \
    miller_loop<C, 1, 0>(f, &p, &q, (const Aff<F1<C>>*)0, (const LineCoef<C>* const*)0);                               \
    if (mode == 1) {                                                                                                   \
      gt_store<C>(o, f);                                                                                               \
      return 1;                                                                                                        \
    }                                                                                                                  \
    final_exp<C>(g, f);                                                                                                \
    if (mode == 2) {                                                                                                   \
      Fp12<C> a, b;                                                                                                    \
      fp12_cyc_sqr<C>(a, g);                                                                                           \
      fp12_sqr<C>(b, g);                                                                                               \
      return fp12_eq(a, b) ? 1 : -1;                                                                                   \
    }                                                                                                                  \
    gt_store<C>(o, g);                                                                                                 \
    return 1;                                                                                                          \
  }                                                                                                                    \
  int pfx##_miller_two(const u32* P0, const u32* Q0, const u32* P1, const u32* Q1, int live_mask, u32* o_two, u32* o_prod) {        \
    return twin_miller_two<C>(P0, Q0, P1, Q1, live_mask, o_two, o_prod);                                               \
  }                                                                                                                    \
  /* same pairing but with Q's lines precomputed (checks the fixed-argument path) */                                   \
  int pfx##_pairing_fixedq(const u32* P, const u32* Q, u32* o) {                                                       \
    Aff<F1<C>> p;                                                                                                      \
    Aff<F2<C>> q;                                                                                                      \
    if (!g1_load<C>(p, P) || !g2_load<C>(q, Q)) return 0;                                                              \
    std::vector<LineCoef<C>> ln(ml_num_lines<C>());                                                                    \
    ml_precompute<C>(ln.data(), q);                                                                                    \
    const LineCoef<C>* lines[1] = {ln.data()};                                                                         \
    Fp12<C> f, g;                                                                                                      \
    miller_loop<C, 0, 1>(f, (const Aff<F1<C>>*)0, (const Aff<F2<C>>*)0, &p, lines);                                    \
    final_exp<C>(g, f);                                                                                                \
    gt_store<C>(o, g);                                                                                                 \
    return 1;                                                                                                          \
  }                                                                                                                    \
  /* Fp12 unary/binary ops on std-form tower elements (debug / unit tests): op 0 mul, 1 sqr, 2 inv, 3 frob(n=arg), 4 cyc_sqr, 5 conj, 6 final_exp */ \
  void pfx##_fp12_op(int op, int arg, const u32* x, const u32* y, u32* o) {                                            \
    Fp12<C> a, b, r;                                                                                                   \
    Fp2<C>* ea[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};                                     \
    Fp2<C>* eb[6] = {&b.c0.c0, &b.c0.c1, &b.c0.c2, &b.c1.c0, &b.c1.c1, &b.c1.c2};                                     \
    for (int i = 0; i < 6; i++) {                                                                                      \
      ea[i]->c0 = fp_from_std<C>(fp_load_w<C>(x + (2 * i) * C::N));                                                    \
      ea[i]->c1 = fp_from_std<C>(fp_load_w<C>(x + (2 * i + 1) * C::N));                                                \
      eb[i]->c0 = fp_from_std<C>(fp_load_w<C>(y + (2 * i) * C::N));                                                    \
      eb[i]->c1 = fp_from_std<C>(fp_load_w<C>(y + (2 * i + 1) * C::N));                                                \
    }                                                                                                                  \
    if (op == 0) fp12_mul<C>(r, a, b);                                                                                 \
    else if (op == 1) fp12_sqr<C>(r, a);                                                                               \
    else if (op == 2) fp12_inv<C>(r, a);                                                                               \
    else if (op == 3) fp12_frob<C>(r, a, arg);                                                                         \
    else if (op == 4) fp12_cyc_sqr<C>(r, a);                                                                           \
    else if (op == 5) fp12_conj(r, a);                                                                                 \
    else final_exp<C>(r, a);                                                                                           \
    gt_store<C>(o, r);                                                                                                 \
  }                                                                                                                    \
  void* pfx##_ctx_new(int A, int W, const u32* g1b, const u32* g2b) {                                                  \
    TwinCtx<C>* c = new TwinCtx<C>();                                                                                  \
    c->b1.resize(A + 6);                                                                                               \
    c->b2.resize(A + 2);                                                                                               \
    for (int i = 0; i < A + 6; i++)                                                                                    \
      if (!g1_load<C>(c->b1[i], g1b + i * 2 * C::N)) return 0;                                                         \
    for (int i = 0; i < A + 2; i++)                                                                                    \
      if (!g2_load<C>(c->b2[i], g2b + i * 4 * C::N)) return 0;                                                         \
    int nwin = (256 + W - 1) / W, per = fixed_base_entries(W);                                                                  \
    build_tables<F1<C>>(c->t1, c->b1, W, nwin, per);                                                                   \
    build_tables<F2<C>>(c->t2, c->b2, W, nwin, per);                                                                   \
    c->lines.resize(ml_num_lines<C>());                                                                                \
    ml_precompute<C>(c->lines.data(), c->b2[0]);                                                                       \
    c->key.A = A;                                                                                                      \
    c->key.W = W;                                                                                                      \
    c->key.nwin = nwin;                                                                                                \
    c->key.per = per;                                                                                                  \
    c->key.t1 = c->t1.data();                                                                                          \
    c->key.t2 = c->t2.data();                                                                                          \
    c->key.b1 = c->b1.data();                                                                                          \
    c->key.b2 = c->b2.data();                                                                                          \
    c->key.gg_lines = c->lines.data();                                                                                 \
    c->hot.assign(ELP_HOT_WORDS, 0xdeadbeefu);                                                                         \
    c->key.hot = getenv("ELP_TWIN_NO_HOT") ? nullptr : c->hot.data();                                                  \
    c->vtab.assign(vtab_words<C>(), 0xdeadbeefu);                                                                      \
    c->key.vtab = getenv("ELP_TWIN_NO_VTAB") ? nullptr : c->vtab.data();                                              \
    return c;                                                                                                          \
  }                                                                                                                    \
  /* Tables of ANY window width without building them: zero-filled virtual memory (an all-zero entry reads as infinity) into which   \
     table_touch() writes exactly the entries one item will read.  Used by tools/count_ops.py to count the W = 16 kernel. */       \
  void* pfx##_ctx_new_sparse(int A, int W, const u32* g1b, const u32* g2b) {                                           \
    TwinCtx<C>* c = new TwinCtx<C>();                                                                                  \
    c->b1.resize(A + 6);                                                                                               \
    c->b2.resize(A + 2);                                                                                               \
    for (int i = 0; i < A + 6; i++)                                                                                    \
      if (!g1_load<C>(c->b1[i], g1b + i * 2 * C::N)) return 0;                                                         \
    for (int i = 0; i < A + 2; i++)                                                                                    \
      if (!g2_load<C>(c->b2[i], g2b + i * 4 * C::N)) return 0;                                                         \
    int nwin = (256 + W - 1) / W, per = fixed_base_entries(W);                                                                  \
    c->t1s = calloc((size_t)(A + 6) * nwin * per, sizeof(Aff<F1<C>>));                                                 \
    c->t2s = calloc((size_t)(A + 2) * nwin * per, sizeof(Aff<F2<C>>));                                                 \
    if (!c->t1s || !c->t2s) return 0;                                                                                  \
    c->lines.resize(ml_num_lines<C>());                                                                                \
    ml_precompute<C>(c->lines.data(), c->b2[0]);                                                                       \
    c->key.A = A;                                                                                                      \
    c->key.W = W;                                                                                                      \
    c->key.nwin = nwin;                                                                                                \
    c->key.per = per;                                                                                                  \
    c->key.t1 = (const Aff<F1<C>>*)c->t1s;                                                                             \
    c->key.t2 = (const Aff<F2<C>>*)c->t2s;                                                                             \
    c->key.b1 = c->b1.data();                                                                                          \
    c->key.b2 = c->b2.data();                                                                                          \
    c->key.gg_lines = c->lines.data();                                                                                 \
    c->hot.assign(ELP_HOT_WORDS, 0xdeadbeefu);                                                                         \
    c->key.hot = c->hot.data();                                                                                        \
    c->vtab.assign(vtab_words<C>(), 0xdeadbeefu);                                                                      \
    c->key.vtab = c->vtab.data();                                                                                      \
    return c;                                                                                                          \
  }                                                                                                                    \
  void pfx##_table_touch(void* cv, int group, int base, const u32* k) {                                                \
    TwinCtx<C>* c = (TwinCtx<C>*)cv;                                                                                   \
    const size_t stride = (size_t)c->key.nwin * c->key.per;                                                            \
    if (group == 1)                                                                                                    \
      touch_entries<F1<C>>((Aff<F1<C>>*)c->t1s + base * stride, c->b1[base], c->key.W, c->key.nwin, c->key.per, scalar_load_w(k)); \
    else                                                                                                               \
      touch_entries<F2<C>>((Aff<F2<C>>*)c->t2s + base * stride, c->b2[base], c->key.W, c->key.nwin, c->key.per, scalar_load_w(k)); \
  }                                                                                                                    \
  /* replaces ONE G1 base (the relying party's H1(service_name) is base A + 1) and rebuilds its table only: what elp_set_rp does on the device */ \
  int pfx##_ctx_set_g1_base(void* cv, int idx, const u32* pt) {                                                        \
    TwinCtx<C>* c = (TwinCtx<C>*)cv;                                                                                   \
    if (idx < 0 || idx >= (int)c->b1.size() || c->t1.empty() || !g1_load<C>(c->b1[idx], pt)) return 0;                 \
    std::vector<Aff<F1<C>>> one(1, c->b1[idx]), tbl;                                                                   \
    build_tables<F1<C>>(tbl, one, c->key.W, c->key.nwin, c->key.per);                                                  \
    std::copy(tbl.begin(), tbl.end(), c->t1.begin() + (size_t)idx * c->key.nwin * c->key.per);                         \
    return 1;                                                                                                          \
  }                                                                                                                    \
  void pfx##_ctx_free(void* c) { delete (TwinCtx<C>*)c; }                                                              \
  void pfx##_ctx_set_flags(void* c, int flags) { ((TwinCtx<C>*)c)->key.flags = flags; } /* KEY_STRICT_SIG = 1, KEY_NO_SUBGROUP_CHECK = 2 */ \
  int pfx##_verify_id(void* c, const u32* rec, uint64_t mask, int retr, const uint8_t* ad, size_t adlen) {             \
    return verify_id_item<C>(((TwinCtx<C>*)c)->key, rec, mask, retr != 0, ad, adlen) ? 1 : 0;                          \
  }                                                                                                                    \
  /* the two-phase form (k_vid_nizk / k_vid_pair): the two job roles in turn, K through the workspace layout */          \
  int pfx##_verify_id_split(void* cv, const u32* rec, uint64_t mask, int retr, const uint8_t* ad, size_t adlen) {        \
    TwinCtx<C>* c = (TwinCtx<C>*)cv;                                                                                   \
    VidShared<C> sh;                                                                                                   \
    memset(&sh, 0xa5, sizeof sh);                                                                                      \
    VidNizkState<C> st0, st1;                                                                                          \
    Aff<F2<C>> aK, unused;                                                                                             \
    vid_nizk_jobs<C>(c->key, 0, rec, mask, retr != 0, sh, st0, unused);                                                \
    vid_nizk_jobs<C>(c->key, 1, rec, mask, retr != 0, sh, st1, aK);                                                    \
    if (!vid_nizk_finish<C>(sh, st0, retr != 0, ad, adlen)) return 0;                                                  \
    std::vector<u32> ws((size_t)vid_k_words<C>() * 3, 0xdeadbeefu);                                                    \
    vid_store_k<C>(ws.data(), 3, 1, aK);                                                                               \
    Aff<F2<C>> k2;                                                                                                     \
    vid_load_k<C>(k2, ws.data(), 3, 1);                                                                                \
    return vid_pair_item<C>(c->key, rec, k2) ? 1 : 0;                                                                  \
  }                                                                                                                    \
  /* the small-batch form (k_vid_fixed_coop + k_vid_nizk4 + pairing): the fixed-base sums computed first (here sequentially), the four job roles in turn */ \
  int pfx##_verify_id_jobs4(void* cv, const u32* rec, uint64_t mask, int retr, const uint8_t* ad, size_t adlen) {        \
    TwinCtx<C>* c = (TwinCtx<C>*)cv;                                                                                   \
    Jac<F2<C>> pre[2];                                                                                                 \
    jac_set_inf(pre[0]);                                                                                               \
    jac_set_inf(pre[1]);                                                                                               \
    {                                                                                                                  \
      PairedRecordSrc<C> ps;                                                                                           \
      ps.init(rec, mask, c->key.A, retr != 0);                                                                         \
      int jh = 0, jr = 0;                                                                                              \
      for (int i = 0; i < c->key.A; i++) {                                                                             \
        if ((mask >> i) & 1) acc_fixed_g2<C>(pre[0], c->key, G2_BASE_YY0 + i, ps.rs(jh++));                             \
        else acc_fixed_g2<C>(pre[1], c->key, G2_BASE_YY0 + i, scalar_load_w(ps.w_ms_ + 8 * jr++));                      \
      }                                                                                                                \
      acc_fixed_g2<C>(pre[0], c->key, G2_BASE_GG, ps.rs(retr ? ps.nrs() - 2 : ps.nrs() - 1));                           \
      acc_fixed_g2<C>(pre[0], c->key, G2_BASE_XX, scalar_one_minus<C>(scalar_load_w(ps.w_k_ + 4 * C::N)));              \
    }                                                                                                                  \
    VidShared<C> sh;                                                                                                   \
    memset(&sh, 0xa5, sizeof sh);                                                                                      \
    VidNizkState<C> st[4];                                                                                             \
    Aff<F2<C>> aK, unused;                                                                                             \
    for (int role = 0; role < 4; role++) vid_nizk_jobs4<C>(c->key, role, rec, mask, retr != 0, sh, st[role], role == 1 ? aK : unused, pre); \
    if (!vid_nizk_finish<C>(sh, st[0], retr != 0, ad, adlen)) return 0;                                                \
    return vid_pair_item<C>(c->key, rec, aK) ? 1 : 0;                                                                  \
  }                                                                                                                    \
  /* [a]P + [b]phi(P) with a = k[0..1], b = k[2..3] (curve.h g1_mul_pair64_with: the multiplier of aggregated verification); 0 = P is not on the curve */ \
  int pfx##_g1_mul_pair64(const u32* P, const u32* k, u32* out) {                                                      \
    Aff<F1<C>> p;                                                                                                      \
    if (!g1_load<C>(p, P)) return 0;                                                                                   \
    Aff<F1<C>> tab[8];                                                                                                 \
    Jac<F1<C>> jm[8];                                                                                                  \
    jac_multiples8<F1<C>>(jm, p);                                                                                      \
    for (int i = 0; i < 8; i++) jac_to_aff<F1<C>>(tab[i], jm[i]);                                                      \
    Scalar d;                                                                                                          \
    for (int i = 0; i < 8; i++) d.v[i] = i < 4 ? k[i] : 0;                                                             \
    Jac<F1<C>> r;                                                                                                      \
    g1_mul_pair64_with<C, PrivTab<F1<C>>>(r, PrivTab<F1<C>>{tab}, d);                                                  \
    Aff<F1<C>> a;                                                                                                      \
    jac_to_aff<F1<C>>(a, r);                                                                                           \
    g1_store<C>(out, a);                                                                                               \
    return 1;                                                                                                          \
  }                                                                                                                    \
  int pfx##_verify_id_wire(void* c, const uint8_t* msg, size_t len, int retr, const uint8_t* ad, size_t adlen) {          \
    return verify_id_wire_item<C>(((TwinCtx<C>*)c)->key, msg, len, retr != 0, ad, adlen) ? 1 : 0;                      \
  }                                                                                                                    \
  int pfx##_wire_decode(const uint8_t* msg, size_t len, int A, int retr, u32* rec, uint64_t* mask) {                      \
    int ok = 1;                                                                                                         \
    u64 m = 0;                                                                                                          \
    for (int job = 0; job < WIRE_DECODE_JOBS; job++) ok &= wire_decode_job<C>(job, A, msg, len, retr != 0, rec, &m) ? 1 : 0; \
    *mask = m;                                                                                                          \
    return ok;                                                                                                          \
  }                                                                                                                    \
  int pfx##_prove_id(void* c, const u32* rec, uint64_t mask, int retr, const uint8_t* ad, size_t adlen, u32* out) {      \
    return prove_id_item<C>(((TwinCtx<C>*)c)->key, rec, mask, retr != 0, ad, adlen, out) ? 1 : 0;                      \
  }                                                                                                                    \
  void pfx##_request_id(void* c, const u32* rec, uint64_t mask, const uint8_t* ad, size_t adlen, u32* out) {             \
    request_id_item<C>(((TwinCtx<C>*)c)->key, rec, mask, ad, adlen, out);                                              \
  }                                                                                                                    \
  int pfx##_ps_verify(void* c, const u32* rec, int nattr) {                                                            \
    return ps_verify_item<C>(((TwinCtx<C>*)c)->key, rec, nattr) ? 1 : 0;                                               \
  }                                                                                                                    \
  int pfx##_provide_id(void* c, const u32* rec, uint64_t mask, const uint8_t* ad, size_t adlen, u32* out) {            \
    return provide_id_item<C>(((TwinCtx<C>*)c)->key, rec, mask, ad, adlen, out) ? 1 : 0;                               \
  }                                                                                                                    \
  }

// ---- lane pairs on the host: the two lanes of a pair are two threads; every exchange is a rendezvous (two barriers), so paired code
// whose lanes take different paths around an exchange hangs here (reported after 60 s) instead of silently reading garbage.
struct PairBus {
  std::atomic<int> count{0};
  std::atomic<int> sense{0};
  void* slot[2] = {nullptr, nullptr};
  size_t bytes[2] = {0, 0};
};
static thread_local PairBus* tl_bus = nullptr;
static void pair_barrier(PairBus* b) {
  const int s = b->sense.load(std::memory_order_acquire);
  if (b->count.fetch_add(1, std::memory_order_acq_rel) == 1) {
    b->count.store(0, std::memory_order_relaxed);
    b->sense.store(1 - s, std::memory_order_release);
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (b->sense.load(std::memory_order_acquire) == s) {
    if ((++spins & 0xffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
      fprintf(stderr, "host twin: the lanes of a pair diverged around an exchange (partner never arrived)\n");
      abort();
    }
  }
}
static void pair_exchange(void* buf, size_t bytes) {
  PairBus* b = tl_bus;
  const int me = elp_pair_parity;
  b->slot[me] = buf;
  b->bytes[me] = bytes;
  pair_barrier(b);
  if (b->bytes[1 - me] != bytes) {
    fprintf(stderr, "host twin: exchange size mismatch between the lanes of a pair\n");
    abort();
  }
  unsigned char tmp[256];
  memcpy(tmp, b->slot[1 - me], bytes);
  pair_barrier(b);
  memcpy(buf, tmp, bytes);
}
template <class Fn>
static int run_pair(Fn fn) {   // fn(parity) -> int; both lanes must agree
  PairBus bus;
  int res[2] = {-1, -2};
  auto lane = [&](int parity) {
    elp_pair_parity = parity;
    tl_bus = &bus;
    elp_pair_exchange_hook = pair_exchange;
    res[parity] = fn(parity);
  };
  std::thread t1(lane, 1);
  lane(0);
  t1.join();
  elp_pair_parity = 0;
  if (res[0] != res[1]) {
    fprintf(stderr, "host twin: the lanes of a pair returned different verdicts (%d, %d)\n", res[0], res[1]);
    return -100;
  }
  return res[0];
}

template <class C>
static KeyCtx<Paired<C>> paired_key(const TwinCtx<C>* c, u32* hot) {
  KeyCtx<Paired<C>> k;
  k.A = c->key.A;
  k.W = c->key.W;
  k.nwin = c->key.nwin;
  k.per = c->key.per;
  k.t1 = c->key.t1;
  k.t2 = c->key.t2;
  k.b1 = c->key.b1;
  k.b2 = c->key.b2;
  k.gg_lines = c->key.gg_lines;
  k.hot = hot;
  k.flags = c->key.flags;
  return k;
}
// ELP_OPT_SPLIT_PHASES = 3 on the host: the four G1 jobs of an item one after the other in the plain layout (stride-1 workspace), then the paired kernel's body
// over their output -- the same two functions the kernels k_vid_g1jobs / k_verify_id_paired_g1 wrap.
template <class B>
static int twin_verify_id_g1split(const TwinCtx<B>* c, const u32* rec, uint64_t mask, int retr, const uint8_t* ad, size_t adlen) {
  std::vector<u32> ws(g1jobs_ws_words<B>(), 0xdeadbeefu);
  for (int job = 0; job < 4; job++) {
    KeyCtx<B> k = c->key;
    std::vector<u32> hot(ELP_HOT_WORDS, 0xdeadbeefu);
    k.hot = hot.data();
    std::vector<u32> slot(8 * vtab_entry_words<F1<B>>(), 0xdeadbeefu);
    vid_g1_job<B>(k, job, rec, mask, retr != 0, getenv("ELP_TWIN_NO_VTAB") ? nullptr : slot.data(), ws.data(), 1, 0);
  }
  return run_pair([&](int) {
    std::vector<u32> hot(ELP_HOT_WORDS_PAIRED, 0xdeadbeefu);
    KeyCtx<Paired<B>> k = paired_key<B>(c, hot.data());
    std::vector<u32> vt(vtab_words<Paired<B>>(), 0xdeadbeefu);
    k.vtab = getenv("ELP_TWIN_NO_VTAB") ? nullptr : vt.data();
    G1JobsOut pre{ws.data(), 1, 0, B::FBYTES / 4};
    return verify_id_item_paired_g1done<Paired<B>>(k, rec, mask, retr != 0, ad, adlen, pre) ? 1 : 0;
  });
}
#if !defined(TWIN_PART) || TWIN_PART == 2
typedef Paired<BLS12_381> BLSP;
extern "C" {
int twin_blsp_verify_id(void* cv, const u32* rec, uint64_t mask, int retr, const uint8_t* ad, size_t adlen) {
  const TwinCtx<BLS12_381>* c = (const TwinCtx<BLS12_381>*)cv;
  return run_pair([&](int) {
    std::vector<u32> hot(ELP_HOT_WORDS_PAIRED, 0xdeadbeefu);
    KeyCtx<BLSP> k = paired_key<BLS12_381>(c, hot.data());
    std::vector<u32> vt(vtab_words<BLSP>(), 0xdeadbeefu);
    k.vtab = getenv("ELP_TWIN_NO_VTAB") ? nullptr : vt.data();
    return verify_id_item_paired<BLSP>(k, rec, mask, retr != 0, ad, adlen, (k.flags & KEY_PHASE_MIX) != 0) ? 1 : 0;      /* flags 4: the pairing check first */
  });
}
/* the item of aggregated verification: one lane (k_verify_id_agg) and the lane pair (k_verify_id_agg_paired); f_out = the twelve coefficients of the Miller value */
int twin_bls_agg_item(void* cv, const u32* rec, uint64_t mask, int retr, const uint8_t* ad, size_t adlen, const uint8_t* seed, uint64_t index, u32* f_out, u32* delta,
                      u32* sig2) {
  TwinCtx<BLS12_381>* c = (TwinCtx<BLS12_381>*)cv;
  Fp12<BLS12_381> f;
  const bool ok = verify_id_agg_item<BLS12_381>(c->key, rec, mask, retr != 0, ad, adlen, seed, index, f, delta, sig2);
  gt_store<BLS12_381>(f_out, f);
  return ok ? 1 : 0;
}
int twin_blsp_agg_item(void* cv, const u32* rec, uint64_t mask, int retr, const uint8_t* ad, size_t adlen, const uint8_t* seed, uint64_t index, u32* f_out, u32* delta,
                       u32* sig2) {
  const TwinCtx<BLS12_381>* c = (const TwinCtx<BLS12_381>*)cv;
  Fp12<BLS12_381> mem;
  memset(&mem, 0x5a, sizeof mem);
  const int r = run_pair([&](int) {
    std::vector<u32> hot(ELP_HOT_WORDS_PAIRED, 0xdeadbeefu);
    KeyCtx<BLSP> k = paired_key<BLS12_381>(c, hot.data());
    std::vector<u32> vt(vtab_words<BLSP>(), 0xdeadbeefu);
    k.vtab = getenv("ELP_TWIN_NO_VTAB") ? nullptr : vt.data();
    Fp12<BLSP> f;
    const bool ok = verify_id_agg_item_paired<BLSP>(k, rec, mask, retr != 0, ad, adlen, seed, index, f, delta, sig2);
    fp12_to_mem<BLSP>(mem, f);
    return ok ? 1 : 0;
  });
  gt_store<BLS12_381>(f_out, mem);
  return r;
}
int twin_blsp_verify_id_g1split(void* cv, const u32* rec, uint64_t mask, int retr, const uint8_t* ad, size_t adlen) {
  return twin_verify_id_g1split<BLS12_381>((const TwinCtx<BLS12_381>*)cv, rec, mask, retr, ad, adlen);
}
int twin_blsp_ps_verify(void* cv, const u32* rec, int nattr) {
  const TwinCtx<BLS12_381>* c = (const TwinCtx<BLS12_381>*)cv;
  return run_pair([&](int) {
    std::vector<u32> hot(ELP_HOT_WORDS_PAIRED, 0xdeadbeefu);
    KeyCtx<BLSP> k = paired_key<BLS12_381>(c, hot.data());
    std::vector<u32> vt(vtab_words<BLSP>(), 0xdeadbeefu);
    k.vtab = getenv("ELP_TWIN_NO_VTAB") ? nullptr : vt.data();
    return ps_verify_item<BLSP>(k, rec, nattr) ? 1 : 0;
  });
}
int twin_blsp_pairing(const u32* P, const u32* Q, u32* o) {
  return run_pair([&](int) {
    Aff<F1<BLSP>> p;
    Aff<F2<BLSP>> q;
    if (!g1_load<BLSP>(p, P) || !g2_load<BLSP>(q, Q)) return 0;
    Fp12<BLSP> f, g;
    miller_loop<BLSP, 1, 0>(f, &p, &q, (const Aff<F1<BLSP>>*)0, (const LineMem<BLSP>* const*)0);
    final_exp<BLSP>(g, f);
    gt_store<BLSP>(o, g);
    return 1;
  });
}
}
#endif
// Cooperative pairing check (elp/coop.h): the level-scheduled program executed slot by slot on the host.  mode 0: [f_K(sig1) f_gg(-sig2)]^e by the 16-pair
// program, mode 2: the same by the 32-pair program, mode 1 (the tail of aggregated verification, 32 pairs): [F f_gg(-sig2)]^e with F = f_K(sig1) computed here
// by the ordinary Miller loop.  Writes the GT bytes; returns 1 iff the value is 1.
#define TWIN_PAIR_COOP(FN, CURVE, NS) \
extern "C" int FN(void* cv, const u32* sig1w, const u32* sig2w, const u32* Kw, int mode, u32* gt_out) { \
  typedef CURVE C; \
  using namespace elp::NS; \
  TwinCtx<C>* c = (TwinCtx<C>*)cv; \
  Aff<F1<C>> s1, s2; \
  Aff<F2<C>> K; \
  if (!g1_load<C>(s1, sig1w) || !g1_load<C>(s2, sig2w) || !g2_load<C>(K, Kw)) return -1; \
  std::vector<Fp2<C>> consts(COOP_NCONST); \
  for (int i = 0; i < COOP_NCONST; i++) { \
    consts[i].c0 = coop_const<C>(CONST_KIND, i, 0); \
    consts[i].c1 = coop_const<C>(CONST_KIND, i, 1); \
  } \
  std::vector<i32> R((size_t)COOP_NREG * coop_reg_words<C>(), 0x5a5a5a5); \
  coop_st<C>(R.data(), IN_P2, 0, s2.x); \
  coop_st<C>(R.data(), IN_P2, 1, fp_neg(s2.y)); \
  coop_st<C>(R.data(), IN_ONE, 0, fp_one<C>()); \
  coop_st<C>(R.data(), IN_ONE, 1, fp_zero<C>()); \
  CoopProg P; \
  if (mode == 0 || mode == 2) { \
    coop_st<C>(R.data(), IN_P1, 0, s1.x); \
    coop_st<C>(R.data(), IN_P1, 1, s1.y); \
    coop_st<C>(R.data(), IN_QX, 0, K.x.c0); \
    coop_st<C>(R.data(), IN_QX, 1, K.x.c1); \
    coop_st<C>(R.data(), IN_QY, 0, K.y.c0); \
    coop_st<C>(R.data(), IN_QY, 1, K.y.c1); \
    if (mode == 0) { \
      P = CoopProg{CHECK_PROG, CHECK_CLASS, CHECK_TERMS, CHECK_NSTEPS, {CHECK_OUT[0], CHECK_OUT[1], CHECK_OUT[2], CHECK_OUT[3], CHECK_OUT[4], CHECK_OUT[5]}}; \
      P.np = CHECK_NP; \
    } else { \
      P = CoopProg{CHECK32_PROG, CHECK32_CLASS, CHECK32_TERMS, CHECK32_NSTEPS, {CHECK32_OUT[0], CHECK32_OUT[1], CHECK32_OUT[2], CHECK32_OUT[3], CHECK32_OUT[4], CHECK32_OUT[5]}}; \
      P.np = CHECK32_NP; \
    } \
  } else { \
    Fp12<C> F; \
    miller_loop<C, 1, 0>(F, &s1, &K, &s1, (const LineMem<C>* const*)0); \
    const Fp2<C>* e[6] = {&F.c0.c0, &F.c0.c1, &F.c0.c2, &F.c1.c0, &F.c1.c1, &F.c1.c2}; \
    for (int j = 0; j < 6; j++) { \
      coop_st<C>(R.data(), IN_F0 + j, 0, e[j]->c0); \
      coop_st<C>(R.data(), IN_F0 + j, 1, e[j]->c1); \
    } \
    P = CoopProg{TAIL_PROG, TAIL_CLASS, TAIL_TERMS, TAIL_NSTEPS, {TAIL_OUT[0], TAIL_OUT[1], TAIL_OUT[2], TAIL_OUT[3], TAIL_OUT[4], TAIL_OUT[5]}}; \
    P.np = TAIL_NP; \
  } \
  coop_run_host<C>(P, R.data(), consts.data(), (int)consts.size(), reinterpret_cast<const Fp2<C>*>(c->lines.data())); \
  Fp12<C> g; \
  Fp2<C>* o[6] = {&g.c0.c0, &g.c0.c1, &g.c0.c2, &g.c1.c0, &g.c1.c1, &g.c1.c2}; \
  for (int j = 0; j < 6; j++) { \
    o[j]->c0 = coop_ld<C>(R.data(), P.out[j], 0); \
    o[j]->c1 = coop_ld<C>(R.data(), P.out[j], 1); \
  } \
  gt_store<C>(gt_out, g); \
  return fp12_is_one(g) ? 1 : 0; \
}
#if !defined(TWIN_PART) || TWIN_PART == 1
TWIN_PAIR_COOP(twin_bn254_pair_coop, BN254, coop_bn254)
// debug aid for tools/gen_coop.py: the register file (canonical values) after `nsteps` steps of the check program
extern "C" int twin_bn254_coop_debug(void* cv, const u32* sig1w, const u32* sig2w, const u32* Kw, int nsteps, u32* regs_out) {
  typedef BN254 C;
  using namespace elp::coop_bn254;
  TwinCtx<C>* c = (TwinCtx<C>*)cv;
  Aff<F1<C>> s1, s2;
  Aff<F2<C>> K;
  if (!g1_load<C>(s1, sig1w) || !g1_load<C>(s2, sig2w) || !g2_load<C>(K, Kw)) return -1;
  std::vector<Fp2<C>> consts(COOP_NCONST);
  for (int i = 0; i < COOP_NCONST; i++) {
    consts[i].c0 = coop_const<C>(CONST_KIND, i, 0);
    consts[i].c1 = coop_const<C>(CONST_KIND, i, 1);
  }
  std::vector<i32> R((size_t)COOP_NREG * coop_reg_words<C>(), 0);
  coop_st<C>(R.data(), IN_P2, 0, s2.x);
  coop_st<C>(R.data(), IN_P2, 1, fp_neg(s2.y));
  coop_st<C>(R.data(), IN_ONE, 0, fp_one<C>());
  coop_st<C>(R.data(), IN_ONE, 1, fp_zero<C>());
  coop_st<C>(R.data(), IN_P1, 0, s1.x);
  coop_st<C>(R.data(), IN_P1, 1, s1.y);
  coop_st<C>(R.data(), IN_QX, 0, K.x.c0);
  coop_st<C>(R.data(), IN_QX, 1, K.x.c1);
  coop_st<C>(R.data(), IN_QY, 0, K.y.c0);
  coop_st<C>(R.data(), IN_QY, 1, K.y.c1);
  CoopProg P{CHECK_PROG, CHECK_CLASS, CHECK_TERMS, nsteps, {0, 0, 0, 0, 0, 0}};
  coop_run_host<C>(P, R.data(), consts.data(), (int)consts.size(), reinterpret_cast<const Fp2<C>*>(c->lines.data()));
  for (int r = 0; r < COOP_NREG; r++)
    for (int cc = 0; cc < 2; cc++) fp_store_w<C>(regs_out + (r * 2 + cc) * C::N, fp_to_std<C>(coop_ld<C>(R.data(), r, cc)));
  return 0;
}
typedef Paired<BN254> BN254P;
extern "C" {
// paired-layout verification on a context made by twin_bn254_ctx_new (tables are shared: plain layout in memory)
int twin_bn254p_verify_id(void* cv, const u32* rec, uint64_t mask, int retr, const uint8_t* ad, size_t adlen) {
  const TwinCtx<BN254>* c = (const TwinCtx<BN254>*)cv;
  return run_pair([&](int) {
    std::vector<u32> hot(ELP_HOT_WORDS_PAIRED, 0xdeadbeefu);
    KeyCtx<BN254P> k = paired_key<BN254>(c, getenv("ELP_TWIN_NO_HOT") ? nullptr : hot.data());
    std::vector<u32> vt(vtab_words<BN254P>(), 0xdeadbeefu);
    k.vtab = getenv("ELP_TWIN_NO_VTAB") ? nullptr : vt.data();
    return verify_id_item_paired<BN254P>(k, rec, mask, retr != 0, ad, adlen, (k.flags & KEY_PHASE_MIX) != 0) ? 1 : 0;    /* flags 4: the pairing check first */
  });
}
int twin_bn254p_verify_id_g1split(void* cv, const u32* rec, uint64_t mask, int retr, const uint8_t* ad, size_t adlen) {
  return twin_verify_id_g1split<BN254>((const TwinCtx<BN254>*)cv, rec, mask, retr, ad, adlen);
}
int twin_bn254p_verify_id_wire(void* cv, const uint8_t* msg, size_t len, int retr, const uint8_t* ad, size_t adlen) {
  const TwinCtx<BN254>* c = (const TwinCtx<BN254>*)cv;
  return run_pair([&](int) {
    std::vector<u32> hot(ELP_HOT_WORDS_PAIRED, 0xdeadbeefu);
    KeyCtx<BN254P> k = paired_key<BN254>(c, hot.data());
    std::vector<u32> vt(vtab_words<BN254P>(), 0xdeadbeefu);
    k.vtab = getenv("ELP_TWIN_NO_VTAB") ? nullptr : vt.data();
    return verify_id_wire_item_paired<BN254P>(k, msg, len, retr != 0, ad, adlen) ? 1 : 0;
  });
}
int twin_bn254p_ps_verify(void* cv, const u32* rec, int nattr) {
  const TwinCtx<BN254>* c = (const TwinCtx<BN254>*)cv;
  return run_pair([&](int) {
    std::vector<u32> hot(ELP_HOT_WORDS_PAIRED, 0xdeadbeefu);
    KeyCtx<BN254P> k = paired_key<BN254>(c, hot.data());
    std::vector<u32> vt(vtab_words<BN254P>(), 0xdeadbeefu);
    k.vtab = getenv("ELP_TWIN_NO_VTAB") ? nullptr : vt.data();
    return ps_verify_item<BN254P>(k, rec, nattr) ? 1 : 0;
  });
}
// e(P, Q) computed by a lane pair; GT bytes as twin_bn254_pairing
int twin_bn254p_pairing(const u32* P, const u32* Q, u32* o) {
  return run_pair([&](int) {
    Aff<F1<BN254P>> p;
    Aff<F2<BN254P>> q;
    if (!g1_load<BN254P>(p, P) || !g2_load<BN254P>(q, Q)) return 0;
    Fp12<BN254P> f, g;
    miller_loop<BN254P, 1, 0>(f, &p, &q, (const Aff<F1<BN254P>>*)0, (const LineMem<BN254P>* const*)0);
    final_exp<BN254P>(g, f);
    gt_store<BN254P>(o, g);
    return 1;
  });
}
int twin_bn254p_g2_mul_gls(const u32* P, const u32* k, u32* o) {
  return run_pair([&](int) {
    Aff<F2<BN254P>> p, r;
    if (!g2_load<BN254P>(p, P)) return 0;
    Jac<F2<BN254P>> j;
    g2_mul_gls<BN254P>(j, p, scalar_load_w(k));
    jac_to_aff<F2<BN254P>>(r, j);
    g2_store<BN254P>(o, r);
    return 1;
  });
}
int twin_bn254p_g2_decompress(const uint8_t* in, u32* o) {
  return run_pair([&](int) {
    Aff<F2<BN254P>> p;
    if (!g2_deserialize<BN254P>(p, in)) return 0;
    g2_store<BN254P>(o, p);
    return 1;
  });
}
}
#endif

#if !defined(TWIN_PART) || TWIN_PART == 2
// the same interpreter over the BLS12-381 programs (M-type twist, Hayashida-Hayasaka-Teruya chain taken to the third power: the value is the CUBE of the GT element)
TWIN_PAIR_COOP(twin_bls_pair_coop, BLS12_381, coop_bls12_381)
#endif
// TWIN_PART selects one curve so the two halves can be compiled in parallel (tests/elp_testlib.py); default: both
#if !defined(TWIN_PART) || TWIN_PART == 1
TWIN(BN254, twin_bn254)
#endif
#if !defined(TWIN_PART) || TWIN_PART == 2
TWIN(BLS12_381, twin_bls)
#endif

#if defined(ELP_COUNT_OPS) && (!defined(TWIN_PART) || TWIN_PART == 1)
extern "C" void twin_op_counts(unsigned long long* out, int reset) {
  out[0] = elp::elp_op_counts[0];
  out[1] = elp::elp_op_counts[1];
  out[2] = elp::elp_op_counts[2];
  out[3] = elp::elp_op_counts[3];
  if (reset) elp::elp_op_counts[0] = elp::elp_op_counts[1] = elp::elp_op_counts[2] = elp::elp_op_counts[3] = 0;
}
#endif
