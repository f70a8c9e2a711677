// Per-item bodies of the batched hot path.  Each function processes ONE independent item (credential / proof);
// ../elpasso_impl.h wraps them in __global__ kernels (one item per lane, or per lane pair in the paired layout), tests/host_twin wraps
// them in host loops.
//
// Reference call stacks restated here (optimised structure: fixed-base tables, one shared final exponentiation):
//   verify_id_item        <- PSVerifier::el_passo_verify_id                       src/ps-verifier.cc:37-138
//   verify_id_item(noretr)<- PSVerifier::el_passo_verify_id_without_id_retrieval  src/ps-verifier.cc:140-212
//   ps_verify_item        <- PSVerifier::verify / PSRequester::verify             src/ps-verifier.cc:13-35
//   provide_id_item       <- PSSigner::el_passo_provide_id                        src/ps-signer.cc:63-146
#pragma once
#include "encode.h"
#include "pairing.h"

namespace elp {

// Wire-independent record formats (all integers little-endian, "std" = canonical non-Montgomery):
//   G1 affine : x[F] y[F]            (all-zero = infinity)          F = C::FBYTES
//   G2 affine : x.a[F] x.b[F] y.a[F] y.b[F]
//   Fr        : 32 bytes
template <class C>
struct Sizes {
  static constexpr int F = C::FBYTES;
  static constexpr int G1 = 2 * F;
  static constexpr int G2 = 4 * F;
  static constexpr int FR = 32;
  static constexpr int GT = 12 * F;
};

// Device-resident key context (built once per public key by elp_set_pubkey / elp_set_rp).
template <class C>
struct KeyCtx {
  int A;                          // attributes in the key
  int W;                          // fixed-base window width
  int nwin, per;                  // windows per scalar, entries per window (signed digits: 2^(W-1), curve.h fixed_base_entries)
  // key material lives in HBM in the PLAIN layout (built by the unpaired set-up kernels) whatever layout C computes in
  const Aff<typename F1<C>::MemF>* t1;   // G1 tables: base b at t1 + b * nwin * per
  const Aff<typename F2<C>::MemF>* t2;   // G2 tables
  const Aff<typename F1<C>::MemF>* b1;   // G1 bases (affine):  0 = g, 1+i = Y_i, A+1 = H1(service), A+2 = g_eg, A+3 = authority_pk, A+4 = h, A+5 = X (signer secret)
  const Aff<typename F2<C>::MemF>* b2;   // G2 bases (affine):  0 = gg, 1 = XX, 2+i = YY_i
  const LineMem<C>* gg_lines;            // precomputed Miller lines of gg
  u32* hot = nullptr;             // this lane's LDS hot slot (ELP_HOT_WORDS words) or null, see common.h
  u32* vtab = nullptr;            // this lane's slice of the launch workspace for the tables of small multiples (vtab_words<C>() words, 16-byte aligned) or null:
                                  // the host hands the base of the workspace in, the kernel advances it to its lane (curve.h, WsTab)
  int flags = 0;                  // KEY_STRICT_SIG: proofs with sig1 == infinity are rejected (PS / EL PASSO require sigma_1 != 1)
  u32* vpsi = nullptr;            // small batches only: this item's psi^j images (j = 1, 2, 3) of the multiples 1k .. 8k, 24 entries of vtab_entry_words<F2<C>>() words beside
                                  // the table in vtab (k_vid_ktab writes them, the G2 job of k_vid_small reads them: curve.h WsTabPsi), or null
};
// The reference's el_passo_verify_id accepts sig1 = sig2 = infinity with a self-made NIZK (e(O,K) e(O,gg) = 1: a universal forgery;
// golden case "sig_both_zero", src/ps-verifier.cc:133-137 has no isZero test although PSVerifier::verify :16-18 has one).  The library
// rejects it by default (elp_set_option(ELP_OPT_STRICT_SIGNATURE)); reference-compatible behaviour is opt-in.
enum { KEY_STRICT_SIG = 1, KEY_NO_SUBGROUP_CHECK = 2, KEY_PHASE_MIX = 4, KEY_QUAD_G2 = 8 };      // KEY_QUAD_G2 (set by the launcher of k_vid_small for this launch only): NIZK workgroups of 16 items, the G2 job on four lanes per item   // KEY_PHASE_MIX: the second half of a two-lane launch's workgroups runs the pairing check BEFORE the NIZK half (verify_id_item_paired)
//   // KEY_NO_SUBGROUP_CHECK: skip g1_in_subgroup on prover-supplied points (ELP_OPT_SUBGROUP_CHECK = 0)
// words of workspace per lane: 1P .. 8P of k and of up to three G1 points
template <class C>
ELP_HD constexpr int vtab_words() { return 8 * vtab_entry_words<F2<C>>() + 3 * 8 * vtab_entry_words<F1<C>>(); }
enum { G1_BASE_G = 0, G1_BASE_Y0 = 1 };
enum { G2_BASE_GG = 0, G2_BASE_XX = 1, G2_BASE_YY0 = 2 };
template <class C>
ELP_INL int g1_base_hs(const KeyCtx<C>& k) { return k.A + 1; }
template <class C>
ELP_INL int g1_base_geg(const KeyCtx<C>& k) { return k.A + 2; }
template <class C>
ELP_INL int g1_base_apk(const KeyCtx<C>& k) { return k.A + 3; }
template <class C>
ELP_INL int g1_base_h(const KeyCtx<C>& k) { return k.A + 4; }
template <class C>
ELP_INL int g1_base_skx(const KeyCtx<C>& k) { return k.A + 5; }
template <class C>
ELP_INL int g1_num_bases(int A) { return A + 6; }
template <class C>
ELP_INL int g2_num_bases(int A) { return A + 2; }

// ---- word-wise loads/stores of std-form values (buffers are 4-byte aligned)
template <class C>
ELP_INL StdFp<C> fp_load_w(const u32* w) {
  StdFp<C> r;
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) r.w[i] = w[i];
  return r;
}
template <class C>
ELP_INL void fp_store_w(u32* w, const StdFp<C>& a) {
  ELP_UNROLL
  for (int i = 0; i < C::N; i++) w[i] = a.w[i];
}
ELP_INL Scalar scalar_load_w(const u32* w) {
  Scalar s;
  for (int i = 0; i < 8; i++) s.v[i] = w[i];
  return s;
}
// returns false if a coordinate is >= p or the point is not on the curve
template <class C>
ELP_HEAVY bool g1_load(Aff<F1<C>>& p, const u32* w) {
  StdFp<C> x = fp_load_w<C>(w), y = fp_load_w<C>(w + C::N);
  if (std_is_zero(x) && std_is_zero(y)) {
    aff_set_inf(p);
    return true;
  }
  if (!std_in_range<C>(x) || !std_in_range<C>(y)) return false;
  p.x = fp_from_std<C>(x);
  p.y = fp_from_std<C>(y);
  return aff_on_curve<F1<C>>(p);
}
// Membership of an E(Fp) point in the order-r subgroup G1 on curves whose G1 cofactor is not 1 (BLS12-381; on BN curves E(Fp) = G1).
// The reference never meets the question (it runs on BN254) and mcl's default does not check; this library REJECTS prover-supplied G1 points
// outside the subgroup: phi is the user's pseudonym at the relying party and (E1, E2) the identity-retrieval token, and a small-order
// component would give one user several pseudonyms / an undecryptable token, while the GLV multiplication used on them is only a scalar
// multiplication inside G1.  sig1 / sig2 are exempt: a cofactor component vanishes in the pairing.
// Test (Scott, "A note on group membership tests for G1, G2 and GT on BLS pairing-friendly curves"): P in G1  <=>  [z^2] P == -psi(P) for the
// endomorphism psi(x, y) = (w x, y), w the primitive cube root of unity whose eigenvalue on G1 is -z^2.  Both endomorphisms z^2 + psi and
// z^2 + psi^2 have degree z^4 - z^2 + 1 = r, so their kernels are cyclic of order r and contain no rational point outside G1: comparing with
// BOTH roots is equally sound and needs no convention about which root the parameter file calls beta.  ~126 doublings + 10 additions.
template <class C>
ELP_HEAVY bool g1_in_subgroup(const Aff<F1<C>>& p) {
  if constexpr (C::IS_BN) {
    (void)p;
    return true;
  } else {
    typedef F1<C> F;
    if (aff_is_inf(p)) return true;
    Jac<F> q, r;
    jac_from_aff(q, p);
    int top = 63;
    while (!((C::ZABS >> top) & 1)) top--;
    ELP_NOUNROLL
    for (int i = top - 1; i >= 0; i--) {          // q = [|z|] P
      jac_dbl<F>(q, q);
      if ((C::ZABS >> i) & 1) jac_madd<F>(q, q, p);
    }
    r = q;
    ELP_NOUNROLL
    for (int i = top - 1; i >= 0; i--) {          // r = [|z|] q = [z^2] P
      jac_dbl<F>(r, r);
      if ((C::ZABS >> i) & 1) jac_add<F>(r, r, q);
    }
    if (jac_is_inf(r)) return false;              // a non-zero point killed by z^2 has an order prime to r
    // r == -psi(P) = (w x, -y) in Jacobian form:  X == w x Z^2  and  Y == -y Z^3
    const Fp<C> z2 = fp_sqr<C>(r.Z), z3 = fp_mul<C>(z2, r.Z);
    if (!fp_is_zero<C>(fp_add_lazy(r.Y, fp_mul<C>(p.y, z3)))) return false;
    Fp<C> beta;
    ELP_LOAD_FP(beta, C::glv_beta(i_));
    const Fp<C> xz = fp_mul<C>(p.x, z2);
    const Fp<C> t1 = fp_mul<C>(xz, beta), t2 = fp_mul<C>(t1, beta);
    return fp_is_zero<C>(fp_sub_lazy(r.X, t1)) || fp_is_zero<C>(fp_sub_lazy(r.X, t2));
  }
}
// sigma_1 of a PS signature must not be the identity OF G1 (src/ps-verifier.cc:16-18 for PSVerifier::verify; for el_passo_verify_id under
// KEY_STRICT_SIG).  On a curve with a G1 cofactor "not the identity" has to be asked of the order-r component: a point T of E(Fp) whose order divides the
// cofactor pairs to e(T, Q) = 1 with everything, so sig1 = T, sig2 = O satisfies e(sig1, K) e(-sig2, gg) = 1 for ANY K although sig1 is not the point
// at infinity -- a universal forgery from the public key alone (round-3 advisor finding).  Rule: sig1 != O and (unless the caller vouches for its inputs with
// ELP_OPT_SUBGROUP_CHECK = 0) sig1 in G1; then sig1 has order exactly r and e(sig1, .) is non-degenerate.  sig2 needs no test: once sig1 is a generator of
// G1 the equation pins the order-r component of sig2, and its cofactor component is invisible to every party.  A no-op beyond the infinity test on BN curves.
template <class C>
ELP_HEAVY bool sig1_admissible(int key_flags, const Aff<F1<C>>& sig1) {
  if (aff_is_inf(sig1)) return false;
  if constexpr (!C::IS_BN) {
    if (!(key_flags & KEY_NO_SUBGROUP_CHECK) && !g1_in_subgroup<C>(sig1)) return false;
  }
  return true;
}
// the el_passo_verify_id form of the rule: only under KEY_STRICT_SIG (the reference accepts sig1 = sig2 = O there; the default of this library does not)
template <class C>
ELP_INL bool sig1_strict_ok(int key_flags, const Aff<F1<C>>& sig1) {
  return !(key_flags & KEY_STRICT_SIG) || sig1_admissible<C>(key_flags, sig1);
}
template <class C>
ELP_HEAVY bool g2_load(Aff<F2<C>>& p, const u32* w) {
  if constexpr (is_paired<C>()) {   // each lane takes its own components; validity is agreed on by the pair
    const int o = pair_odd() ? C::N : 0;
    StdFp<C> x = fp_load_w<C>(w + o), y = fp_load_w<C>(w + 2 * C::N + o);
    if (pair_and(std_is_zero(x) && std_is_zero(y))) {
      aff_set_inf(p);
      return true;
    }
    if (!pair_and(std_in_range<C>(x) && std_in_range<C>(y))) return false;
    p.x.c = fp_from_std<C>(x);
    p.y.c = fp_from_std<C>(y);
    return aff_on_curve<F2<C>>(p);
  } else {
  StdFp<C> a = fp_load_w<C>(w), b = fp_load_w<C>(w + C::N), c = fp_load_w<C>(w + 2 * C::N), d = fp_load_w<C>(w + 3 * C::N);
  if (std_is_zero(a) && std_is_zero(b) && std_is_zero(c) && std_is_zero(d)) {
    aff_set_inf(p);
    return true;
  }
  if (!std_in_range<C>(a) || !std_in_range<C>(b) || !std_in_range<C>(c) || !std_in_range<C>(d)) return false;
  p.x.c0 = fp_from_std<C>(a);
  p.x.c1 = fp_from_std<C>(b);
  p.y.c0 = fp_from_std<C>(c);
  p.y.c1 = fp_from_std<C>(d);
  return aff_on_curve<F2<C>>(p);
  }
}
template <class C>
ELP_HEAVY void g1_store(u32* w, const Aff<F1<C>>& p) {
  if (aff_is_inf(p)) {
    for (int i = 0; i < 2 * C::N; i++) w[i] = 0;
    return;
  }
  fp_store_w<C>(w, fp_to_std<C>(p.x));
  fp_store_w<C>(w + C::N, fp_to_std<C>(p.y));
}
template <class C>
ELP_HEAVY void g2_store(u32* w, const Aff<F2<C>>& p) {
  if constexpr (is_paired<C>()) {   // the two lanes write disjoint halves of the record
    const int o = pair_odd() ? C::N : 0;
    if (aff_is_inf(p)) {
      for (int i = 0; i < C::N; i++) w[o + i] = w[2 * C::N + o + i] = 0;
      return;
    }
    fp_store_w<C>(w + o, fp_to_std<C>(p.x.c));
    fp_store_w<C>(w + 2 * C::N + o, fp_to_std<C>(p.y.c));
    return;
  } else {
  if (aff_is_inf(p)) {
    for (int i = 0; i < 4 * C::N; i++) w[i] = 0;
    return;
  }
  fp_store_w<C>(w, fp_to_std<C>(p.x.c0));
  fp_store_w<C>(w + C::N, fp_to_std<C>(p.x.c1));
  fp_store_w<C>(w + 2 * C::N, fp_to_std<C>(p.y.c0));
  fp_store_w<C>(w + 3 * C::N, fp_to_std<C>(p.y.c1));
  }
}
template <class C>
ELP_HEAVY void gt_store(u32* w, const Fp12<C>& f) {  // order: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 (each a, b)
  const Fp2<C>* e[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  for (int i = 0; i < 6; i++) {
    if constexpr (is_paired<C>()) {
      fp_store_w<C>(w + (2 * i + (pair_odd() ? 1 : 0)) * C::N, fp_to_std<C>(e[i]->c));
    } else {
      fp_store_w<C>(w + (2 * i) * C::N, fp_to_std<C>(e[i]->c0));
      fp_store_w<C>(w + (2 * i + 1) * C::N, fp_to_std<C>(e[i]->c1));
    }
  }
}

// ---- fixed-base accumulation against the key tables
template <class C>
ELP_INL void acc_fixed_g1(Jac<F1<C>>& acc, const KeyCtx<C>& k, int base, const Scalar& s) {
  jac_acc_fixed<F1<C>>(acc, k.t1 + (size_t)base * k.nwin * k.per, k.W, s, k.hot);
}
template <class C>
ELP_INL void acc_fixed_g2(Jac<F2<C>>& acc, const KeyCtx<C>& k, int base, const Scalar& s) {
  jac_acc_fixed<F2<C>>(acc, k.t2 + (size_t)base * k.nwin * k.per, k.W, s, k.hot);
}

// 1/Z for several Jacobian Z's with ONE base-field inversion (Montgomery's trick); Z == 0 entries are skipped.
template <class C, int N1, int N2>
ELP_HEAVY void batch_zinv(Fp<C>* zi1, const Fp<C>* z1, Fp2<C>* zi2, const Fp2<C>* z2) {
  constexpr int NT = N1 + N2;
  Fp<C> v[NT], pre[NT];
  for (int i = 0; i < N1; i++) v[i] = fp_is_zero_exact(z1[i]) ? fp_one<C>() : z1[i];
  for (int i = 0; i < N2; i++) {
    Fp<C> n = fp2_norm<C>(z2[i]);   // norm (exactly 0 for the literal zero of infinity); paired: the same value on both lanes
    v[N1 + i] = fp_is_zero_exact(n) ? fp_one<C>() : n;
  }
  Fp<C> acc = fp_one<C>();
  for (int i = 0; i < NT; i++) {
    pre[i] = acc;
    acc = fp_mul<C>(acc, v[i]);
  }
  Fp<C> inv = fp_inv<C>(acc);
  for (int i = NT - 1; i >= 0; i--) {
    Fp<C> vi = fp_mul<C>(inv, pre[i]);
    inv = fp_mul<C>(inv, v[i]);
    if (i < N1) {
      zi1[i] = vi;
    } else {
      zi2[i - N1] = fp2_conj_mul_fp<C>(z2[i - N1], vi);
    }
  }
}

// One variable-base multiplication on its own: table of multiples, one (division-step) inversion for its seven non-trivial entries,
// mixed additions in the loop.
template <class C>
ELP_HEAVY void g1_mul_glv(Jac<F1<C>>& r, const Aff<F1<C>>& p, const Scalar& k, u32* hot = nullptr) {
  (void)hot;
  typedef F1<C> F;
  Jac<F> jt[8];
  Aff<F> tab[8];
  jac_multiples8<F>(jt, p);
  Fp<C> z[7], zi[7];
  for (int i = 1; i < 8; i++) z[i - 1] = jt[i].Z;
  batch_zinv<C, 7, 0>(zi, z, (Fp2<C>*)0, (const Fp2<C>*)0);
  tab[0] = p;
  for (int i = 1; i < 8; i++) jac_to_aff_with_zinv<F>(tab[i], jt[i], zi[i - 1]);
  g1_mul_glv_tab<C>(r, tab, k);
}
template <class C>
ELP_HEAVY void g2_mul_gls(Jac<F2<C>>& r, const Aff<F2<C>>& p, const Scalar& k, u32* hot = nullptr) {
  (void)hot;
  typedef F2<C> F;
  Jac<F> jt[8];
  Aff<F> tab[8];
  jac_multiples8<F>(jt, p);
  Fp2<C> z[7], zi[7];
  for (int i = 1; i < 8; i++) z[i - 1] = jt[i].Z;
  batch_zinv<C, 0, 7>((Fp<C>*)0, (const Fp<C>*)0, zi, z);
  tab[0] = p;
  for (int i = 1; i < 8; i++) jac_to_aff_with_zinv<F>(tab[i], jt[i], zi[i - 1]);
  g2_mul_gls_tab<C>(r, tab, k);
}

// Fiat-Shamir challenge: Fr::setHashOf( SHA256( hex(part_0) || ... || ad ) )  (SHA-256 applied twice)
struct Transcript {
  Sha256 s;
};
ELP_INL void transcript_init(Transcript& t) { sha256_init(t.s); }
template <class C>
ELP_INL void transcript_g1(Transcript& t, const Aff<F1<C>>& p) {
  uint8_t b[C::FBYTES];
  g1_serialize<C>(b, p);
  sha256_update_hex(t.s, b, C::FBYTES);
}
template <class C>
ELP_INL void transcript_g2(Transcript& t, const Aff<F2<C>>& p) {
  uint8_t b[2 * C::FBYTES];
  g2_serialize<C>(b, p);
  sha256_update_hex(t.s, b, 2 * C::FBYTES);
}
template <class C>
ELP_INL Scalar transcript_challenge(Transcript& t, const uint8_t* ad, size_t ad_len) {
  sha256_update(t.s, ad, ad_len);
  uint8_t d[32];
  sha256_final(t.s, d);
  Sha256 s2;
  sha256_init(s2);
  sha256_update(s2, d, 32);
  sha256_final(s2, d);
  return scalar_from_digest<C>(d);
}

// ------------------------------------------------------------------------------------------------------------
// EL PASSO VerifyID.  Record (words): sig1 | sig2 | phi | [E1 | E2] | k | c | rs[H+1 or H+2] | m[A-H]
//   hidden_mask bit i set <=> attribute i is hidden (proof.attributes[i] == "");  m = Fr::setHashOf(attribute)
//   for the revealed attributes in attribute order.
template <class C>
ELP_HD constexpr int verify_id_record_words(int A, int H, bool retr) {
  return ((retr ? 5 : 3) * 2 * C::N) + 4 * C::N + 8 * (1 + H + (retr ? 2 : 1) + (A - H));
}

// Where the fields of one proof come from.  RecordSrc: the fixed-stride std-form record of elp_verify_id_batch.
// WireSrc (further below): the reference's own T-L-V wire message (IdProof::toBufferString, src/ps-encoding.cc:451-467),
// with point decompression and attribute hashing done here on the device.
template <class C>
struct RecordSrc {
  const u32 *rs_, *ms_, *w_phi_, *w_k_;
  u64 mask_;
  int nrs_, jr_;
  bool sub_ = true;               // phi, E1, E2 must lie in the order-r subgroup (a no-op on BN curves); callers clear it from the key's flags
  ELP_HD bool open(const u32* rec, u64 hidden_mask, int A, bool retr, Aff<F1<C>>& sig1, Aff<F1<C>>& sig2, Aff<F1<C>>& phi,
                   Aff<F1<C>>& E1, Aff<F1<C>>& E2, Aff<F2<C>>& kk, Scalar& c) {
    int H = 0;
    for (int i = 0; i < A; i++) H += (int)((hidden_mask >> i) & 1);
    mask_ = hidden_mask;
    nrs_ = H + (retr ? 2 : 1);
    jr_ = 0;
    bool ok = true;
    const u32* p = rec;
    ok &= g1_load<C>(sig1, p); p += 2 * C::N;
    ok &= g1_load<C>(sig2, p); p += 2 * C::N;
    w_phi_ = p;
    ok &= g1_load<C>(phi, p);  p += 2 * C::N;
    if (retr) {
      ok &= g1_load<C>(E1, p); p += 2 * C::N;
      ok &= g1_load<C>(E2, p); p += 2 * C::N;
    }
    if constexpr (!C::IS_BN) {
      if (ok && sub_) ok = g1_in_subgroup<C>(phi) && (!retr || (g1_in_subgroup<C>(E1) && g1_in_subgroup<C>(E2)));
    }
    w_k_ = p;
    ok &= g2_load<C>(kk, p); p += 4 * C::N;
    c = scalar_load_w(p); p += 8;
    rs_ = p; p += 8 * nrs_;
    ms_ = p;
    return ok;
  }
  // the same view of a record whose points somebody else has validated: scalars and the transcript's input bytes only
  ELP_HD void open_lite(const u32* rec, u64 hidden_mask, int A, bool retr, Scalar& c) {
    int H = 0;
    for (int i = 0; i < A; i++) H += (int)((hidden_mask >> i) & 1);
    mask_ = hidden_mask;
    nrs_ = H + (retr ? 2 : 1);
    jr_ = 0;
    w_phi_ = rec + 4 * C::N;
    w_k_ = rec + (retr ? 5 : 3) * 2 * C::N;
    c = scalar_load_w(w_k_ + 4 * C::N);
    rs_ = w_k_ + 4 * C::N + 8;
    ms_ = rs_ + 8 * nrs_;
  }
  ELP_HD int nrs() const { return nrs_; }
  ELP_HD bool hidden(int i) const { return (mask_ >> i) & 1; }
  ELP_HD Scalar rs(int j) const { return scalar_load_w(rs_ + 8 * j); }
  ELP_HD Scalar next_revealed_hash(int) { return scalar_load_w(ms_ + 8 * jr_++); }   // revealed attributes in order
  // wire bytes of the transcript's input points straight from their canonical coordinates (a validated record has one encoding per point)
  ELP_HD void ser_k(uint8_t* out) const { g2_serialize_std<C>(out, w_k_); }
  ELP_HD void ser_g1(int which, uint8_t* out) const { g1_serialize_std<C>(out, w_phi_ + which * 2 * C::N); }   // 0 = phi, 1 = E1, 2 = E2
};

// NIZK half of VerifyID: recomputes V_k, V_phi, (V_E1, V_E2), the challenge, and K (returned in affine form for the pairing).
template <class C, class Src>
ELP_HEAVY bool verify_id_nizk(const KeyCtx<C>& key, Src& src, bool retr, const Aff<F1<C>>& phi, const Aff<F1<C>>& E1,
                              const Aff<F1<C>>& E2, const Aff<F2<C>>& kk, const Scalar& c, const uint8_t* ad, size_t ad_len,
                              Aff<F2<C>>& aK) {
  typedef F1<C> G1F;
  typedef F2<C> G2F;
  const int A = key.A;
  const int nrs = src.nrs();

  // V_k = k^c * prod_{hidden} YY_j^{r_j} * gg^{r_t} * XX^{1-c}          (src/ps-verifier.cc:72-88)
  // K   = k * prod_{revealed} YY_i^{m_i}                                 (src/ps-verifier.cc:214-229)
  // V_phi = phi^c * H1(service)^{r_0}                                    (src/ps-verifier.cc:91-96)
  // V_E1 = E1^c * g^{r_eps} ; V_E2 = E2^c * y^{r_eps} * h^{r_1}          (src/ps-verifier.cc:99-108)
  Jac<G2F> Vk, K;
  Jac<G1F> Vphi, VE1, VE2;
  const Scalar r_t = src.rs(retr ? nrs - 2 : nrs - 1);
  Scalar one;
  for (int i = 0; i < 8; i++) one.v[i] = 0;
  one.v[0] = 1;
  Scalar cred = c;  // 1 - c mod r (c >= r cannot match the recomputed challenge; reduce defensively)
  if (scalar_geq_r<C>(cred)) {
    Scalar rr;
    for (int i = 0; i < 8; i++) rr.v[i] = C::rmod(i);
    cred = scalar_sub_mod_r<C>(cred, rr);
  }
  const Scalar one_minus_c = scalar_sub_mod_r<C>(one, cred);
  // The four variable-base terms k^c, phi^c, E1^c, E2^c share the scalar: their tables of small multiples (1P .. 8P) are built in
  // Jacobian form, made affine with ONE inversion for all 28 entries that need it, and the GLS / GLV loops run on mixed additions.
  Aff<G2F> tabk[8];
  Aff<G1F> tab1[3][8];
  {
    Jac<G2F> jk[8];
    Jac<G1F> j1[3][8];
    jac_multiples8<G2F>(jk, kk);
    jac_multiples8<G1F>(j1[0], phi);
    if (retr) {
      jac_multiples8<G1F>(j1[1], E1);
      jac_multiples8<G1F>(j1[2], E2);
    }
    Fp<C> z1[21], zi1[21];
    Fp2<C> z2[7], zi2[7];
    for (int t = 0; t < 3; t++)
      for (int i = 1; i < 8; i++) z1[7 * t + i - 1] = (t == 0 || retr) ? j1[t][i].Z : fp_one<C>();
    for (int i = 1; i < 8; i++) z2[i - 1] = jk[i].Z;
    batch_zinv<C, 21, 7>(zi1, z1, zi2, z2);
    tabk[0] = kk;
    for (int i = 1; i < 8; i++) jac_to_aff_with_zinv<G2F>(tabk[i], jk[i], zi2[i - 1]);
    for (int t = 0; t < 3; t++) {
      if (t != 0 && !retr) continue;
      tab1[t][0] = t == 0 ? phi : (t == 1 ? E1 : E2);
      for (int i = 1; i < 8; i++) jac_to_aff_with_zinv<G1F>(tab1[t][i], j1[t][i], zi1[7 * t + i - 1]);
    }
  }
  // with a launch workspace the tables are read from this lane's slice of it (contiguous per lane), not from private memory: curve.h, WsTab
  u32* const wsk = key.vtab;
  u32* const ws1 = key.vtab ? key.vtab + 8 * vtab_entry_words<G2F>() : nullptr;
  if (wsk) {
    for (int i = 0; i < 8; i++) vtab_store<G2F>(wsk, i, tabk[i]);
    for (int t = 0; t < 3; t++) {
      if (t != 0 && !retr) continue;
      for (int i = 0; i < 8; i++) vtab_store<G1F>(ws1 + t * 8 * vtab_entry_words<G1F>(), i, tab1[t][i]);
    }
    g2_mul_gls_with<C, WsTab<G2F>>(Vk, WsTab<G2F>{wsk}, c);
  } else {
    g2_mul_gls_tab<C>(Vk, tabk, c);       // [c]k by the 4-dimensional GLS decomposition (k is expected in the order-r subgroup)
  }
  jac_from_aff(K, kk);
  {
    int jh = 0, jr = 0;
    for (int i = 0; i < A; i++) {
      if (src.hidden(i)) {
        acc_fixed_g2<C>(Vk, key, G2_BASE_YY0 + i, src.rs(jh));
        jh++;
      } else {
        acc_fixed_g2<C>(K, key, G2_BASE_YY0 + i, src.next_revealed_hash(i));
        jr++;
      }
    }
  }
  acc_fixed_g2<C>(Vk, key, G2_BASE_GG, r_t);
  acc_fixed_g2<C>(Vk, key, G2_BASE_XX, one_minus_c);
  if (ws1)
    g1_mul_glv_with<C, WsTab<G1F>>(Vphi, WsTab<G1F>{ws1 + 0 * 8 * vtab_entry_words<G1F>()}, c);
  else
    g1_mul_glv_tab<C>(Vphi, tab1[0], c);
  acc_fixed_g1<C>(Vphi, key, g1_base_hs(key), src.rs(0));
  if (retr) {
    const Scalar r_e = src.rs(nrs - 1);
    if (ws1)
      g1_mul_glv_with<C, WsTab<G1F>>(VE1, WsTab<G1F>{ws1 + 1 * 8 * vtab_entry_words<G1F>()}, c);
    else
      g1_mul_glv_tab<C>(VE1, tab1[1], c);
    acc_fixed_g1<C>(VE1, key, g1_base_geg(key), r_e);
    if (ws1)
      g1_mul_glv_with<C, WsTab<G1F>>(VE2, WsTab<G1F>{ws1 + 2 * 8 * vtab_entry_words<G1F>()}, c);
    else
      g1_mul_glv_tab<C>(VE2, tab1[2], c);
    acc_fixed_g1<C>(VE2, key, g1_base_apk(key), r_e);
    acc_fixed_g1<C>(VE2, key, g1_base_h(key), src.rs(1));
  }

  // canonical affine forms with one shared inversion
  Fp<C> z1[3], zi1[3];
  Fp2<C> z2[2], zi2[2];
  z1[0] = Vphi.Z;
  z1[1] = retr ? VE1.Z : fp_one<C>();
  z1[2] = retr ? VE2.Z : fp_one<C>();
  z2[0] = Vk.Z;
  z2[1] = K.Z;
  batch_zinv<C, 3, 2>(zi1, z1, zi2, z2);
  Aff<G2F> aVk;
  Aff<G1F> aVphi, aVE1, aVE2;
  jac_to_aff_with_zinv<G2F>(aVk, Vk, zi2[0]);
  jac_to_aff_with_zinv<G2F>(aK, K, zi2[1]);
  jac_to_aff_with_zinv<G1F>(aVphi, Vphi, zi1[0]);

  // c' = Hr(SHA256(hex k | hex phi | [hex E1 | hex E2] | hex V_k | hex V_phi | [hex V_E1 | hex V_E2] | ad))
  Transcript t;
  transcript_init(t);
  transcript_g2<C>(t, kk);
  transcript_g1<C>(t, phi);
  if (retr) {
    transcript_g1<C>(t, E1);
    transcript_g1<C>(t, E2);
  }
  transcript_g2<C>(t, aVk);
  transcript_g1<C>(t, aVphi);
  if (retr) {
    jac_to_aff_with_zinv<G1F>(aVE1, VE1, zi1[1]);
    jac_to_aff_with_zinv<G1F>(aVE2, VE2, zi1[2]);
    transcript_g1<C>(t, aVE1);
    transcript_g1<C>(t, aVE2);
  }
  const Scalar c2 = transcript_challenge<C>(t, ad, ad_len);
  return scalar_eq(c2, c);
}

// Signature half: e(sig1, K) == e(sig2, gg)  <=>  e(sig1, K) * e(-sig2, gg) == 1       (src/ps-verifier.cc:133-137)
template <class C>
ELP_HEAVY bool ps_pairing_check(const KeyCtx<C>& key, const Aff<F1<C>>& sig1, const Aff<F1<C>>& sig2, const Aff<F2<C>>& aK) {
  Aff<F1<C>> nsig2;
  aff_neg(nsig2, sig2);
  if (aff_is_inf(sig2)) aff_set_inf(nsig2);
  Fp12<C> f_priv;
  Fp12<C>* fh = hot_as<Fp12<C>, C>(key.hot);
  Fp12<C>& f = fh ? *fh : f_priv;
  const LineMem<C>* lines[1] = {key.gg_lines};
  miller_loop<C, 1, 1>(f, &sig1, &aK, &nsig2, lines);
  return final_exp_is_one<C>(f, key.hot);
}

template <class C, class Src>
ELP_HEAVY bool verify_id_core(const KeyCtx<C>& key, Src& src, bool retr, const Aff<F1<C>>& sig1, const Aff<F1<C>>& sig2,
                              const Aff<F1<C>>& phi, const Aff<F1<C>>& E1, const Aff<F1<C>>& E2, const Aff<F2<C>>& kk, const Scalar& c,
                              const uint8_t* ad, size_t ad_len) {
  Aff<F2<C>> aK;
  if (!sig1_strict_ok<C>(key.flags, sig1)) return false;
  if (!verify_id_nizk<C, Src>(key, src, retr, phi, E1, E2, kk, c, ad, ad_len, aK)) return false;
  return ps_pairing_check<C>(key, sig1, sig2, aK);
}

// Aggregated (random-linear-combination) verification, SURVEY.md section 8f rank 4.  For the items that pass the NIZK half,
//     prod_i [ e(sig1_i, K_i) e(-sig2_i, gg) ]^{d_i} == 1   <=>   prod_i e(d_i sig1_i, K_i) * e(-sum_i d_i sig2_i, gg) == 1
// with verifier-chosen 128-bit d_i: ONE final exponentiation and ONE Miller loop against gg for the whole batch; sum d_i sig2_i is
// a Pippenger MSM.  This per-item part returns the item's Miller value f_i = f(d_i sig1_i, K_i) (1 for rejected items), its
// multiplier d_i (as the pair (a_i, b_i) of d_i = a_i + b_i lam) and a copy of sig2_i.  If the batch equation fails, the caller falls back to the per-item check, so verdicts
// stay exact; a wrong accept needs a 2^-128 event.
ELP_HD inline Scalar agg_multiplier(const uint8_t seed[32], u64 index) {
  Sha256 s;
  sha256_init(s);
  sha256_update(s, seed, 32);
  for (int i = 0; i < 8; i++) sha256_put(s, (uint8_t)(index >> (8 * i)));
  uint8_t d[32];
  sha256_final(s, d);
  Scalar k;
  for (int i = 0; i < 8; i++) k.v[i] = 0;
  for (int i = 0; i < 4; i++) k.v[i] = (u32)d[4 * i] | ((u32)d[4 * i + 1] << 8) | ((u32)d[4 * i + 2] << 16) | ((u32)d[4 * i + 3] << 24);
  k.v[3] |= 0x80000000u;   // exactly 128 bits, never zero
  return k;
}
// aP = [d]sig1 (affine) for the multiplier d = a + b lam of aggregated verification (a = d.v[0..1], b = d.v[2..3], curve.h g1_mul_pair64_with): [a]sig1 + [b]phi(sig1)
// over the affine multiples 1 sig1 .. 8 sig1 (one inversion), kept in the lane's workspace slice (the NIZK half is done with it) rather than in lane-indexed private
// memory.  A function of its own for the two-lane item (verify_id_agg_item_paired): its 3 KB of tables then live in a frame that is not stacked on the NIZK half's
// (kernel frame 17 088 -> 14 064 B on BLS12-381, same time); the one-lane item keeps the same lines inline (see there).
template <class C>
ELP_HEAVY void agg_scaled_sig1(const KeyCtx<C>& key, const Aff<F1<C>>& sig1, const Scalar& d, Aff<F1<C>>& aP) {
  typedef F1<C> G1F;
  Jac<G1F> P;
  if (aff_is_inf(sig1)) {
    aff_set_inf(aP);
    return;
  }
  Aff<G1F> tab[8];
  {
    Jac<G1F> jm[8];
    jac_multiples8<G1F>(jm, sig1);
    // Montgomery's trick over Z(2P) .. Z(8P) with batch_zinv's zero guard: without KEY_STRICT_SIG (or with ELP_OPT_SUBGROUP_CHECK = 0) sig1 may have a
    // small order on a curve with a G1 cofactor and some multiple is the point at infinity (Z = 0); the guarded form keeps every other entry right and
    // the batch equation / per-item fallback then decide such an item as the per-item path does
    Fp<C> z[7], zi[7];
    for (int i = 1; i < 8; i++) z[i - 1] = jm[i].Z;
    batch_zinv<C, 7, 0>(zi, z, (Fp2<C>*)0, (const Fp2<C>*)0);
    tab[0] = sig1;
    for (int i = 1; i < 8; i++) jac_to_aff_with_zinv<G1F>(tab[i], jm[i], zi[i - 1]);
  }
  u32* const ws1 = key.vtab ? key.vtab + 8 * vtab_entry_words<F2<C>>() : nullptr;
  if (ws1) {
    for (int i = 0; i < 8; i++) vtab_store<G1F>(ws1, i, tab[i]);
    g1_mul_pair64_with<C, WsTab<G1F>>(P, WsTab<G1F>{ws1}, d);
  } else {
    g1_mul_pair64_with<C, PrivTab<G1F>>(P, PrivTab<G1F>{tab}, d);
  }
  jac_to_aff<F1<C>>(aP, P);
}
// everything of an item up to its Miller loop: NIZK half, multiplier, [d]sig1 (affine) and K.  false = the item is rejected (aP, aK at infinity, multiplier 0)
template <class C>
ELP_HEAVY bool verify_id_agg_prepare(const KeyCtx<C>& key, const u32* rec, u64 hidden_mask, bool retr, const uint8_t* ad, size_t ad_len,
                                     const uint8_t* seed, u64 index, Aff<F1<C>>& aP, Aff<F2<C>>& aK, u32* delta_out, u32* sig2_out) {
  Aff<F1<C>> sig1, sig2, phi, E1, E2;
  Aff<F2<C>> kk;
  Scalar c;
  RecordSrc<C> src;
  src.sub_ = (key.flags & KEY_NO_SUBGROUP_CHECK) == 0;
  aff_set_inf(aP);
  aff_set_inf(aK);
  for (int i = 0; i < 8; i++) delta_out[i] = 0;
  for (int i = 0; i < 2 * C::N; i++) sig2_out[i] = 0;
  if (!src.open(rec, hidden_mask, key.A, retr, sig1, sig2, phi, E1, E2, kk, c)) return false;
  if (!sig1_strict_ok<C>(key.flags, sig1)) return false;
  if (!verify_id_nizk<C, RecordSrc<C>>(key, src, retr, phi, E1, E2, kk, c, ad, ad_len, aK)) {
    aff_set_inf(aK);
    return false;
  }
  // multiplier d = a + b lam (a = d.v[0..1], b = d.v[2..3], curve.h g1_mul_pair64_with): [d]sig1 = [a]sig1 + [b]phi(sig1) over the affine multiples
  // 1 sig1 .. 8 sig1 (one inversion), kept in the lane's workspace slice (the NIZK half is done with it) rather than in lane-indexed private memory
  // multiplier d = a + b lam (a = d.v[0..1], b = d.v[2..3], curve.h g1_mul_pair64_with): [d]sig1 = [a]sig1 + [b]phi(sig1) over the affine multiples
  // 1 sig1 .. 8 sig1 (one inversion), kept in the lane's workspace slice (the NIZK half is done with it) rather than in lane-indexed private memory.
  // Inline here on purpose: as a function of its own (agg_scaled_sig1 below, which the two-lane form uses) the kernel's frame shrinks by 2 KB but the
  // BN254 kernel runs 0.2 ms slower (A/B in one lease, round 6)
  const Scalar d = agg_multiplier(seed, index);
  Jac<F1<C>> P;
  if (aff_is_inf(sig1)) {
    jac_set_inf(P);
  } else {
    typedef F1<C> G1F;
    Aff<G1F> tab[8];
    {
      Jac<G1F> jm[8];
      jac_multiples8<G1F>(jm, sig1);
      // Montgomery's trick over Z(2P) .. Z(8P) with batch_zinv's zero guard: without KEY_STRICT_SIG (or with ELP_OPT_SUBGROUP_CHECK = 0) sig1 may have a
      // small order on a curve with a G1 cofactor and some multiple is the point at infinity (Z = 0); the guarded form keeps every other entry right and
      // the batch equation / per-item fallback then decide such an item as the per-item path does
      Fp<C> z[7], zi[7];
      for (int i = 1; i < 8; i++) z[i - 1] = jm[i].Z;
      batch_zinv<C, 7, 0>(zi, z, (Fp2<C>*)0, (const Fp2<C>*)0);
      tab[0] = sig1;
      for (int i = 1; i < 8; i++) jac_to_aff_with_zinv<G1F>(tab[i], jm[i], zi[i - 1]);
    }
    u32* const ws1 = key.vtab ? key.vtab + 8 * vtab_entry_words<F2<C>>() : nullptr;
    if (ws1) {
      for (int i = 0; i < 8; i++) vtab_store<G1F>(ws1, i, tab[i]);
      g1_mul_pair64_with<C, WsTab<G1F>>(P, WsTab<G1F>{ws1}, d);
    } else {
      g1_mul_pair64_with<C, PrivTab<G1F>>(P, PrivTab<G1F>{tab}, d);
    }
  }
  jac_to_aff<F1<C>>(aP, P);
  for (int i = 0; i < 8; i++) delta_out[i] = d.v[i];
  g1_store<C>(sig2_out, sig2);
  return true;
}
template <class C>
ELP_HEAVY bool verify_id_agg_item(const KeyCtx<C>& key, const u32* rec, u64 hidden_mask, bool retr, const uint8_t* ad, size_t ad_len,
                                  const uint8_t* seed, u64 index, Fp12<C>& f, u32* delta_out, u32* sig2_out) {
  Aff<F1<C>> aP;
  Aff<F2<C>> aK;
  fp12_set_one(f);
  if (!verify_id_agg_prepare<C>(key, rec, hidden_mask, retr, ad, ad_len, seed, index, aP, aK, delta_out, sig2_out)) return false;
  Fp12<C>* fh = hot_as<Fp12<C>, C>(key.hot);
  Fp12<C>& fm = fh ? *fh : f;
  const LineMem<C>* no_lines[1] = {key.gg_lines};          // never read (no fixed pair); real pointers keep the fused loop compilable
  miller_loop<C, 1, 0>(fm, &aP, &aK, &aP, no_lines);
  if (fh) f = fm;
  return true;
}
// TWO items of one lane (batches of more than one full round: elp_verify_id_batch_aggregated_dev_t): both NIZK halves, then ONE Miller loop for the two pairs
// (pairing.h miller_loop_two); f = the product of the two Miller values, ok[k] = item k passed its NIZK half.  `rec1` may be null (an odd batch's last lane).
template <class C>
ELP_HEAVY void verify_id_agg_item2(const KeyCtx<C>& key, const u32* rec0, const u32* rec1, u64 hidden_mask, bool retr, const uint8_t* ad0, size_t ad_len0,
                                   const uint8_t* ad1, size_t ad_len1, const uint8_t* seed, u64 index0, Fp12<C>& f, u32* delta_out0, u32* sig2_out0,
                                   u32* delta_out1, u32* sig2_out1, bool* ok) {
  Aff<F1<C>> aP[2];
  Aff<F2<C>> aK[2];
  ok[0] = verify_id_agg_prepare<C>(key, rec0, hidden_mask, retr, ad0, ad_len0, seed, index0, aP[0], aK[0], delta_out0, sig2_out0);
  ok[1] = false;
  aff_set_inf(aP[1]);
  aff_set_inf(aK[1]);
  if (rec1) ok[1] = verify_id_agg_prepare<C>(key, rec1, hidden_mask, retr, ad1, ad_len1, seed, index0 + 1, aP[1], aK[1], delta_out1, sig2_out1);
  if constexpr (C::TWIST_D && C::IS_BN && fp_roomy<C>()) {
    miller_loop_two<C>(f, aP, aK, ok, key.hot);      // the hot slot is free between the NIZK half and the wave's product
  } else {                                  // other curves: two loops and a product (not dispatched to; kept so that the template is complete)
    const LineMem<C>* no_lines[1] = {key.gg_lines};
    Fp12<C> g;
    miller_loop<C, 1, 0>(f, &aP[0], &aK[0], &aP[0], no_lines);
    miller_loop<C, 1, 0>(g, &aP[1], &aK[1], &aP[1], no_lines);
    fp12_mul<C>(f, f, g);
  }
}

template <class C>
ELP_HEAVY bool verify_id_item(const KeyCtx<C>& key, const u32* rec, u64 hidden_mask, bool retr, const uint8_t* ad,
                              size_t ad_len) {
  Aff<F1<C>> sig1, sig2, phi, E1, E2;
  Aff<F2<C>> kk;
  Scalar c;
  RecordSrc<C> src;
  src.sub_ = (key.flags & KEY_NO_SUBGROUP_CHECK) == 0;
  if (!src.open(rec, hidden_mask, key.A, retr, sig1, sig2, phi, E1, E2, kk, c)) return false;
  return verify_id_core<C, RecordSrc<C>>(key, src, retr, sig1, sig2, phi, E1, E2, kk, c, ad, ad_len);
}

// ------------------------------------------------------------------------------------------------------------
// EL PASSO VerifyID as TWO PHASES (round 3; kernels k_vid_nizk / k_vid_pair in ../elpasso_impl.h).
//
// One item per lane at the headline batch gives the chip exactly one wave per SIMD, and a lone wave issues one vector instruction
// per ~5 cycles whatever it is; two resident waves share the SIMD at ~3.4 cycles per instruction of this mix and hide each other's
// memory latency.  The NIZK half of a verification consists of independent jobs over different groups, so phase 1
// gives every item TWO lanes in two different waves of one workgroup, each lane a whole job in the plain layout (no lane exchanges,
// no duplicated work):
//   G2 job : V_k = k^c prod_{hidden} YY_j^{r_j} gg^{r_t} XX^{1-c}                                      src/ps-verifier.cc:72-88
//   G1 job : V_phi, V_E1, V_E2 (three GLV multiplications + four fixed-base terms)  and  K = k prod_{revealed} YY_i^{m_i}
//                                                                                            src/ps-verifier.cc:91-108, 214-229
// The jobs meet in LDS (VidShared: the serialised commitments), the G2 lane hashes the transcript and publishes the verdict of the
// NIZK half; K goes to a per-item slot of the launch workspace.  Phase 2 (one lane per item, the whole register file) is the pairing
// check on (sig1, sig2, K).  Both phases fit 256 / 512 registers without the other's temporaries in their frames.
template <class C>
struct VidShared {
  u32 vk[2 * C::FBYTES / 4];       // serialised V_k                  (G2 job)
  u32 v1[3][C::FBYTES / 4];        // serialised V_phi, V_E1, V_E2    (G1 job)
  u32 ok_role[4];                  // per job lane: the record decoded and this lane's own admission test (round 4: one subgroup test per lane on BLS12-381) passed
};
ELP_INL void bytes_to_words(u32* w, const uint8_t* b, int nbytes) {
  for (int i = 0; i < nbytes / 4; i++) w[i] = (u32)b[4 * i] | ((u32)b[4 * i + 1] << 8) | ((u32)b[4 * i + 2] << 16) | ((u32)b[4 * i + 3] << 24);
}
ELP_INL void words_to_bytes(uint8_t* b, const u32* w, int nbytes) {
  for (int i = 0; i < nbytes / 4; i++) {
    const u32 x = w[i];
    b[4 * i] = (uint8_t)x;
    b[4 * i + 1] = (uint8_t)(x >> 8);
    b[4 * i + 2] = (uint8_t)(x >> 16);
    b[4 * i + 3] = (uint8_t)(x >> 24);
  }
}
template <class C>
ELP_INL Scalar scalar_one_minus(const Scalar& c) {   // 1 - c mod r (c >= r cannot match the recomputed challenge; reduced defensively)
  Scalar one;
  for (int i = 0; i < 8; i++) one.v[i] = 0;
  one.v[0] = 1;
  Scalar cred = c;
  if (scalar_geq_r<C>(cred)) {
    Scalar rr;
    for (int i = 0; i < 8; i++) rr.v[i] = C::rmod(i);
    cred = scalar_sub_mod_r<C>(cred, rr);
  }
  return scalar_sub_mod_r<C>(one, cred);
}
// G2 job.  `vk` receives the wire bytes of V_k (as words).
// `pre` (optional): the fixed-base part of V_k, sum rs_j YY_i + r_t gg + (1 - c) XX, as a Jacobian point computed beforehand by the cooperative kernel
// k_vid_fixed_coop (8 lanes per sum); without it the job walks the tables itself.
template <class C, class Src>
ELP_HEAVY void vid_job_g2(const KeyCtx<C>& key, Src& src, bool retr, const Aff<F2<C>>& kk, const Scalar& c, u32* vk, const Jac<F2<C>>* pre = nullptr,
                          bool table_ready = false) {      // table_ready: the multiples 1k .. 8k already sit in the lane's workspace slice (k_vid_ktab)
  typedef F2<C> G2F;
  const int A = key.A;
  const int nrs = src.nrs();
  const Scalar r_t = src.rs(retr ? nrs - 2 : nrs - 1);
  Jac<G2F> Vk;
  if (table_ready && key.vtab && key.vpsi) {
    g2_mul_gls_with<C, WsTabPsi<G2F>, true>(Vk, WsTabPsi<G2F>{key.vtab, key.vpsi}, c);      // psi^j (d k) read, not recomputed at each of the 51 additions that need one
  } else if (table_ready && key.vtab) {
    g2_mul_gls_with<C, WsTab<G2F>>(Vk, WsTab<G2F>{key.vtab}, c);
  } else {
    // 1k .. 8k: Jacobian multiples, one inversion (of the product of the norms) for the seven that need it
    Aff<G2F> tabk[8];
    {
      Jac<G2F> jk[8];
      jac_multiples8<G2F>(jk, kk);
      Fp2<C> z2[7], zi2[7];
      for (int i = 1; i < 8; i++) z2[i - 1] = jk[i].Z;
      batch_zinv<C, 0, 7>((Fp<C>*)0, (const Fp<C>*)0, zi2, z2);
      tabk[0] = kk;
      for (int i = 1; i < 8; i++) jac_to_aff_with_zinv<G2F>(tabk[i], jk[i], zi2[i - 1]);
    }
    u32* const wsk = key.vtab;
    if (wsk) {
      for (int i = 0; i < 8; i++) vtab_store<G2F>(wsk, i, tabk[i]);
      g2_mul_gls_with<C, WsTab<G2F>>(Vk, WsTab<G2F>{wsk}, c);
    } else {
      g2_mul_gls_tab<C>(Vk, tabk, c);
    }
  }
  if (pre) {
    Jac<G2F> S = *pre;
    jac_add<G2F>(Vk, Vk, S);
  } else {
    int jh = 0;
    for (int i = 0; i < A; i++)
      if (src.hidden(i)) {
        acc_fixed_g2<C>(Vk, key, G2_BASE_YY0 + i, src.rs(jh));
        jh++;
      }
    acc_fixed_g2<C>(Vk, key, G2_BASE_GG, r_t);
    acc_fixed_g2<C>(Vk, key, G2_BASE_XX, scalar_one_minus<C>(c));
  }
  Aff<G2F> aVk;
  jac_to_aff<G2F>(aVk, Vk);
  uint8_t b[2 * C::FBYTES];
  g2_serialize<C>(b, aVk);
  bytes_to_words(vk, b, 2 * C::FBYTES);
}
// K = k prod_{revealed} YY_i^{m_i}, affine (src/ps-verifier.cc:214-229): four fixed-base terms at A = 8, H = 4; whichever job has the time takes it
// `pre` (optional): sum over the revealed attributes of m_i YY_i as a Jacobian point (k_vid_fixed_coop)
template <class C, class Src>
ELP_HEAVY void vid_job_k(const KeyCtx<C>& key, Src& src, const Aff<F2<C>>& kk, Aff<F2<C>>& aK, const Jac<F2<C>>* pre = nullptr) {
  typedef F2<C> G2F;
  Jac<G2F> K;
  if (pre) {
    K = *pre;
    jac_madd<G2F>(K, K, kk);
  } else {
    jac_from_aff(K, kk);
    for (int i = 0; i < key.A; i++)
      if (!src.hidden(i)) acc_fixed_g2<C>(K, key, G2_BASE_YY0 + i, src.next_revealed_hash(i));
  }
  jac_to_aff<G2F>(aK, K);
}
// G1 job.  `v1` receives the wire bytes of V_phi, V_E1, V_E2 (as words).
template <class C, class Src>
ELP_HEAVY void vid_job_g1(const KeyCtx<C>& key, Src& src, bool retr, const Aff<F1<C>>& phi, const Aff<F1<C>>& E1, const Aff<F1<C>>& E2,
                          const Scalar& c, u32 (*v1)[C::FBYTES / 4]) {
  typedef F1<C> G1F;
  typedef F2<C> G2F;
  const int nrs = src.nrs();
  Jac<G1F> V[3];
  u32* const ws1 = key.vtab ? key.vtab + 8 * vtab_entry_words<G2F>() : nullptr;
  const int nmul = retr ? 3 : 1;
  {
    Aff<G1F> tab1[3][8];
    {
      Jac<G1F> j1[3][8];
      jac_multiples8<G1F>(j1[0], phi);
      if (retr) {
        jac_multiples8<G1F>(j1[1], E1);
        jac_multiples8<G1F>(j1[2], E2);
      }
      Fp<C> z1[21], zi1[21];
      for (int t = 0; t < 3; t++)
        for (int i = 1; i < 8; i++) z1[7 * t + i - 1] = (t < nmul) ? j1[t][i].Z : fp_one<C>();
      batch_zinv<C, 21, 0>(zi1, z1, (Fp2<C>*)0, (const Fp2<C>*)0);
      for (int t = 0; t < nmul; t++) {
        tab1[t][0] = t == 0 ? phi : (t == 1 ? E1 : E2);
        for (int i = 1; i < 8; i++) jac_to_aff_with_zinv<G1F>(tab1[t][i], j1[t][i], zi1[7 * t + i - 1]);
      }
    }
    for (int t = 0; t < nmul; t++) {
      if (ws1) {
        u32* const w = ws1 + t * 8 * vtab_entry_words<G1F>();
        for (int i = 0; i < 8; i++) vtab_store<G1F>(w, i, tab1[t][i]);
        g1_mul_glv_with<C, WsTab<G1F>>(V[t], WsTab<G1F>{w}, c);
      } else {
        g1_mul_glv_tab<C>(V[t], tab1[t], c);
      }
    }
  }
  acc_fixed_g1<C>(V[0], key, g1_base_hs(key), src.rs(0));
  if (retr) {
    const Scalar r_e = src.rs(nrs - 1);
    acc_fixed_g1<C>(V[1], key, g1_base_geg(key), r_e);
    acc_fixed_g1<C>(V[2], key, g1_base_apk(key), r_e);
    acc_fixed_g1<C>(V[2], key, g1_base_h(key), src.rs(1));
  }
  Fp<C> z1[3], zi1[3];
  for (int t = 0; t < 3; t++) z1[t] = (t < nmul) ? V[t].Z : fp_one<C>();
  batch_zinv<C, 3, 0>(zi1, z1, (Fp2<C>*)0, (const Fp2<C>*)0);
  for (int t = 0; t < nmul; t++) {
    Aff<G1F> a;
    jac_to_aff_with_zinv<G1F>(a, V[t], zi1[t]);
    uint8_t b[C::FBYTES];
    g1_serialize<C>(b, a);
    bytes_to_words(v1[t], b, C::FBYTES);
  }
}
// ONE of the three G1 commitments (which = 0: V_phi = [c]phi + [rs_0]hs; 1: V_E1 = [c]E1 + [r_e]g_eg; 2: V_E2 = [c]E2 + [r_e]apk + [rs_1]h) with its own
// table of multiples and inversions: the form in which the commitments spread over three lanes of an item (k_vid_nizk4).  `vtab_slot`: this commitment's
// 8-entry slice of the launch workspace or null.
template <class C, class Src>
ELP_HEAVY void vid_job_g1_one(const KeyCtx<C>& key, Src& src, int which, const Aff<F1<C>>& P, const Scalar& c, u32* vrow, u32* vtab_slot) {
  typedef F1<C> G1F;
  const int nrs = src.nrs();
  Jac<G1F> V;
  {
    Aff<G1F> tab[8];
    {
      Jac<G1F> jm[8];
      jac_multiples8<G1F>(jm, P);
      Fp<C> z[7], zi[7];
      for (int i = 1; i < 8; i++) z[i - 1] = jm[i].Z;
      batch_zinv<C, 7, 0>(zi, z, (Fp2<C>*)0, (const Fp2<C>*)0);
      tab[0] = P;
      for (int i = 1; i < 8; i++) jac_to_aff_with_zinv<G1F>(tab[i], jm[i], zi[i - 1]);
    }
    if (vtab_slot) {
      for (int i = 0; i < 8; i++) vtab_store<G1F>(vtab_slot, i, tab[i]);
      g1_mul_glv_with<C, WsTab<G1F>>(V, WsTab<G1F>{vtab_slot}, c);
    } else {
      g1_mul_glv_tab<C>(V, tab, c);
    }
  }
  if (which == 0) {
    acc_fixed_g1<C>(V, key, g1_base_hs(key), src.rs(0));
  } else {
    const Scalar r_e = src.rs(nrs - 1);
    if (which == 1) {
      acc_fixed_g1<C>(V, key, g1_base_geg(key), r_e);
    } else {
      acc_fixed_g1<C>(V, key, g1_base_apk(key), r_e);
      acc_fixed_g1<C>(V, key, g1_base_h(key), src.rs(1));
    }
  }
  Aff<G1F> a;
  jac_to_aff<G1F>(a, V);
  uint8_t b[C::FBYTES];
  g1_serialize<C>(b, a);
  bytes_to_words(vrow, b, C::FBYTES);
}
// closing step of phase 1: c == Hr(SHA256(hex k | hex phi | [hex E1 | hex E2] | hex V_k | hex V_phi | [hex V_E1 | hex V_E2] | ad))
// (src/ps-verifier.cc:111-130); the input points come from the source's own serialisations (ser_k / ser_g1)
template <class C, class Src>
ELP_HEAVY bool vid_challenge_ok(const Src& src, bool retr, const u32* vk, const u32 (*v1)[C::FBYTES / 4], const Scalar& c, const uint8_t* ad,
                                size_t ad_len) {
  Transcript t;
  transcript_init(t);
  uint8_t b[2 * C::FBYTES];
  src.ser_k(b);
  sha256_update_hex(t.s, b, 2 * C::FBYTES);
  for (int q = 0; q < (retr ? 3 : 1); q++) {
    src.ser_g1(q, b);
    sha256_update_hex(t.s, b, C::FBYTES);
  }
  words_to_bytes(b, vk, 2 * C::FBYTES);
  sha256_update_hex(t.s, b, 2 * C::FBYTES);
  for (int q = 0; q < (retr ? 3 : 1); q++) {
    words_to_bytes(b, v1[q], C::FBYTES);
    sha256_update_hex(t.s, b, C::FBYTES);
  }
  const Scalar c2 = transcript_challenge<C>(t, ad, ad_len);
  return scalar_eq(c2, c);
}
// K of one item in the launch workspace: word w of item i at ws[w * stride + i] (coalesced across the lanes of a wave)
template <class C>
ELP_HD constexpr int vid_k_words() { return (int)(sizeof(Aff<F2<C>>) / 4); }
template <class C>
ELP_INL void vid_store_k(u32* ws, size_t stride, size_t i, const Aff<F2<C>>& aK) {
  const u32* w = reinterpret_cast<const u32*>(&aK);
  ELP_UNROLL
  for (int q = 0; q < vid_k_words<C>(); q++) ws[(size_t)q * stride + i] = w[q];
}
template <class C>
ELP_INL void vid_load_k(Aff<F2<C>>& aK, const u32* ws, size_t stride, size_t i) {
  u32* w = reinterpret_cast<u32*>(&aK);
  ELP_UNROLL
  for (int q = 0; q < vid_k_words<C>(); q++) w[q] = ws[(size_t)q * stride + i];
}
// Phase 1 of one item from the fixed-stride record, role by role (role 0: G2 job + closing step, role 1: G1 job).  On the device the two
// roles are two waves and `sh` is in LDS with a workgroup barrier between vid_nizk_jobs and vid_nizk_finish; the host twin calls them in turn.
template <class C>
struct VidNizkState {      // what a lane keeps across the barrier
  RecordSrc<C> src;
  Scalar c;
  bool ok;
};
template <class C>
ELP_HEAVY void vid_nizk_jobs(const KeyCtx<C>& key, int role, const u32* rec, u64 hidden_mask, bool retr, VidShared<C>& sh, VidNizkState<C>& st,
                             Aff<F2<C>>& aK, const Jac<F2<C>>* pre = nullptr) {      // pre: {fixed part of V_k, fixed part of K} of this item, or null
  Aff<F1<C>> sig1, sig2, phi, E1, E2;
  Aff<F2<C>> kk;
  st.src.sub_ = (key.flags & KEY_NO_SUBGROUP_CHECK) == 0;
  st.ok = st.src.open(rec, hidden_mask, key.A, retr, sig1, sig2, phi, E1, E2, kk, st.c);
  if (st.ok && !sig1_strict_ok<C>(key.flags, sig1)) st.ok = false;
  if (role == 0) {
    sh.ok_role[0] = st.ok ? 1u : 0u;
    if (st.ok) vid_job_g2<C, RecordSrc<C>>(key, st.src, retr, kk, st.c, sh.vk, pre);
  } else {
    sh.ok_role[1] = sh.ok_role[2] = sh.ok_role[3] = st.ok ? 1u : 0u;
    aff_set_inf(aK);
    if (st.ok) {
      vid_job_g1<C, RecordSrc<C>>(key, st.src, retr, phi, E1, E2, st.c, sh.v1);
      vid_job_k<C, RecordSrc<C>>(key, st.src, kk, aK, pre ? pre + 1 : nullptr);
    }
  }
}
// The same phase with FOUR roles per item (small batches: the jobs of an item in parallel): 0 = G2 job + closing step, 1 = V_phi and K, 2 = V_E1, 3 = V_E2.
// `pre` (the fixed-base sums of k_vid_fixed_coop) is required.
template <class C>
ELP_HEAVY void vid_nizk_jobs4(const KeyCtx<C>& key, int role, const u32* rec, u64 hidden_mask, bool retr, VidShared<C>& sh, VidNizkState<C>& st,
                              Aff<F2<C>>& aK, const Jac<F2<C>>* pre, bool k_done = false, bool table_ready = false,
                              bool g2_by_caller = false) {      // k_done: K and the table of multiples of k were already made by the kernels that ran before (k_vid_fixed_coop, k_vid_ktab); g2_by_caller: role 0 leaves the G2 job to its caller (four lanes per item: elpasso_impl.h vid_job_g2_quad)
  Aff<F1<C>> sig1, sig2, phi, E1, E2;
  Aff<F2<C>> kk;
  // every lane decodes the record (range and on-curve tests: cheap); the SUBGROUP tests of a curve with a G1 cofactor are dealt out, one per lane -- role 0:
  // sig1 (under KEY_STRICT_SIG), 1: phi, 2: E1, 3: E2 -- and meet in sh.ok_role: four tests side by side instead of four in a row on every lane
  st.src.sub_ = false;
  st.ok = st.src.open(rec, hidden_mask, key.A, retr, sig1, sig2, phi, E1, E2, kk, st.c);
  if (st.ok && (key.flags & KEY_STRICT_SIG) && aff_is_inf(sig1)) st.ok = false;
  if constexpr (!C::IS_BN) {
    if (st.ok && !(key.flags & KEY_NO_SUBGROUP_CHECK)) {
      if (role == 0)
        st.ok = !(key.flags & KEY_STRICT_SIG) || g1_in_subgroup<C>(sig1);
      else if (role == 1 || retr)
        st.ok = g1_in_subgroup<C>(role == 1 ? phi : (role == 2 ? E1 : E2));
    }
  }
  sh.ok_role[role] = st.ok ? 1u : 0u;
  u32* const ws1 = key.vtab ? key.vtab + 8 * vtab_entry_words<F2<C>>() : nullptr;
  if (role == 0) {
    if (st.ok && !g2_by_caller) vid_job_g2<C, RecordSrc<C>>(key, st.src, retr, kk, st.c, sh.vk, pre, table_ready);
  } else if (role == 1) {
    aff_set_inf(aK);
    if (st.ok) {
      vid_job_g1_one<C, RecordSrc<C>>(key, st.src, 0, phi, st.c, sh.v1[0], ws1);
      if (!k_done) vid_job_k<C, RecordSrc<C>>(key, st.src, kk, aK, pre + 1);
    }
  } else if (retr && st.ok) {
    const int t = role - 1;
    vid_job_g1_one<C, RecordSrc<C>>(key, st.src, t, t == 1 ? E1 : E2, st.c, sh.v1[t], ws1 ? ws1 + t * 8 * vtab_entry_words<F1<C>>() : nullptr);
  }
}
template <class C>
ELP_HEAVY bool vid_nizk_finish(const VidShared<C>& sh, const VidNizkState<C>& st, bool retr, const uint8_t* ad, size_t ad_len) {
  if (!st.ok || !(sh.ok_role[0] & sh.ok_role[1] & sh.ok_role[2] & sh.ok_role[3])) return false;
  return vid_challenge_ok<C, RecordSrc<C>>(st.src, retr, sh.vk, sh.v1, st.c, ad, ad_len);
}
// Phase 2 of one item: the pairing check on the record's signature and the K of phase 1 (src/ps-verifier.cc:133-137).
template <class C>
ELP_HEAVY bool vid_pair_item(const KeyCtx<C>& key, const u32* rec, const Aff<F2<C>>& aK) {
  Aff<F1<C>> sig1, sig2;
  if (!g1_load<C>(sig1, rec) || !g1_load<C>(sig2, rec + 2 * C::N)) return false;
  return ps_pairing_check<C>(key, sig1, sig2, aK);
}

// ------------------------------------------------------------------------------------------------------------
// EL PASSO VerifyID in the PAIRED layout (C = Paired<B>, two lanes per proof; common.h "Lane pairs").  Same record, same verdicts.
//   * everything over Fp2 -- [c]k by GLS, the ten fixed-base G2 terms, K, the Miller loop and the final exponentiation -- is computed by
//     the pair together, each lane holding one component;
//   * the G1 work is split by job: the even lane computes V_phi and V_E1, the odd lane V_E2 (same instruction stream, different points,
//     bases and scalars), so the three GLV multiplications cost the time of two;
//   * the transcript is hashed by both lanes (identical bytes): inputs are serialised straight from the record words, V_k by
//     g2_serialize's exchange, the three G1 commitments are swapped between the lanes as wire bytes.
template <class C>
ELP_INL Scalar scalar_select(bool c, const Scalar& a, const Scalar& b) {
  Scalar r;
  for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
// the fixed-stride record as a source for the paired kernels (scalars and transcript bytes straight from the record words)
template <class C>
struct PairedRecordSrc {
  const u32 *w_phi_, *w_k_, *w_rs_, *w_ms_;
  u64 mask_;
  int nrs_, jr_;
  ELP_HD void init(const u32* rec, u64 hidden_mask, int A, bool retr) {
    int H = 0;
    for (int i = 0; i < A; i++) H += (int)((hidden_mask >> i) & 1);
    mask_ = hidden_mask;
    nrs_ = H + (retr ? 2 : 1);
    jr_ = 0;
    const int G1W = 2 * C::N, G2W = 4 * C::N;
    w_phi_ = rec + 2 * G1W;
    w_k_ = rec + (retr ? 5 : 3) * G1W;
    w_rs_ = w_k_ + G2W + 8;
    w_ms_ = w_rs_ + 8 * nrs_;
  }
  ELP_HD int nrs() const { return nrs_; }
  ELP_HD bool hidden(int i) const { return (mask_ >> i) & 1; }
  ELP_HD Scalar rs(int j) const { return scalar_load_w(w_rs_ + 8 * j); }
  ELP_HD Scalar next_revealed_hash(int) { return scalar_load_w(w_ms_ + 8 * jr_++); }
  ELP_HD void ser_k(uint8_t* out) const { g2_serialize_std<C>(out, w_k_); }
  ELP_HD void ser_g1(int which, uint8_t* out) const { g1_serialize_std<C>(out, w_phi_ + which * 2 * C::N); }
};
template <class C, class Src>
ELP_HEAVY bool verify_id_paired_nizk(const KeyCtx<C>& key, Src& src, bool retr, const Aff<F1<C>>& P0,
                                     const Aff<F1<C>>& P1, const Aff<F2<C>>& kk, const Scalar& c, const uint8_t* ad, size_t ad_len,
                                     Aff<F2<C>>& aK, bool with_k = true) {      // with_k = false: K was computed before (verify_id_paired_k); aK comes back as k
  static_assert(is_paired<C>(), "paired layout only");
  typedef F1<C> G1F;
  typedef F2<C> G2F;
  const bool odd = pair_odd();
  const int A = key.A;
  const int nrs = src.nrs();
  const Scalar r_t = src.rs(retr ? nrs - 2 : nrs - 1);
  const Scalar r_e = src.rs(nrs - 1);        // used only with retrieval
  Scalar one;
  for (int i = 0; i < 8; i++) one.v[i] = 0;
  one.v[0] = 1;
  Scalar cred = c;
  if (scalar_geq_r<C>(cred)) {
    Scalar rr;
    for (int i = 0; i < 8; i++) rr.v[i] = C::rmod(i);
    cred = scalar_sub_mod_r<C>(cred, rr);
  }
  const Scalar one_minus_c = scalar_sub_mod_r<C>(one, cred);
  // tables 1P .. 8P of k (pair), of this lane's first G1 point and (with retrieval) of its second one; ONE inversion per lane
  Aff<G2F> tabk[8];
  Aff<G1F> tab0[8], tab1[8];
  {
    Jac<G2F> jk[8];
    Jac<G1F> j0[8], j1[8];
    jac_multiples8<G2F>(jk, kk);
    jac_multiples8<G1F>(j0, P0);
    if (retr) jac_multiples8<G1F>(j1, P1);
    Fp<C> z1[14], zi1[14];
    Fp2<C> z2[7], zi2[7];
    for (int i = 1; i < 8; i++) {
      z1[i - 1] = j0[i].Z;
      z1[7 + i - 1] = retr ? j1[i].Z : fp_one<C>();
      z2[i - 1] = jk[i].Z;
    }
    batch_zinv<C, 14, 7>(zi1, z1, zi2, z2);
    tabk[0] = kk;
    tab0[0] = P0;
    tab1[0] = P1;
    for (int i = 1; i < 8; i++) {
      jac_to_aff_with_zinv<G2F>(tabk[i], jk[i], zi2[i - 1]);
      jac_to_aff_with_zinv<G1F>(tab0[i], j0[i], zi1[i - 1]);
      if (retr) jac_to_aff_with_zinv<G1F>(tab1[i], j1[i], zi1[7 + i - 1]);
    }
  }
  // G2, by the pair:  V_k = k^c prod_{hidden} YY_j^{r_j} gg^{r_t} XX^{1-c},  K = k prod_{revealed} YY_i^{m_i}     (src/ps-verifier.cc:72-88,214-229)
  Jac<G2F> Vk, K;
  u32* const wsk = key.vtab;
  u32* const ws1 = key.vtab ? key.vtab + 8 * vtab_entry_words<G2F>() : nullptr;
  if (wsk) {
    for (int i = 0; i < 8; i++) {
      vtab_store<G2F>(wsk, i, tabk[i]);
      vtab_store<G1F>(ws1, i, tab0[i]);
      if (retr) vtab_store<G1F>(ws1 + 8 * vtab_entry_words<G1F>(), i, tab1[i]);
    }
    g2_mul_gls_with<C, WsTab<G2F>>(Vk, WsTab<G2F>{wsk}, c);
  } else {
    g2_mul_gls_tab<C>(Vk, tabk, c);
  }
  jac_from_aff(K, kk);
  {
    int jh = 0;
    for (int i = 0; i < A; i++) {
      if (src.hidden(i)) {
        acc_fixed_g2<C>(Vk, key, G2_BASE_YY0 + i, src.rs(jh));
        jh++;
      } else if (with_k) {
        acc_fixed_g2<C>(K, key, G2_BASE_YY0 + i, src.next_revealed_hash(i));
      }
    }
  }
  acc_fixed_g2<C>(Vk, key, G2_BASE_GG, r_t);
  acc_fixed_g2<C>(Vk, key, G2_BASE_XX, one_minus_c);
  // G1, one job list per lane.  even: V_phi = phi^c H1(svc)^{r_0}, V_E1 = E1^c g^{r_eps};  odd: V_E2 = E2^c y^{r_eps} h^{r_1}   (:91-108)
  Jac<G1F> V0, V1;
  if (ws1)
    g1_mul_glv_with<C, WsTab<G1F>>(V0, WsTab<G1F>{ws1}, c);
  else
    g1_mul_glv_tab<C>(V0, tab0, c);
  if (!odd || retr) {
    const int b0 = odd ? g1_base_apk(key) : g1_base_hs(key);
    acc_fixed_g1<C>(V0, key, b0, scalar_select<C>(odd, r_e, src.rs(0)));
  }
  if (odd && retr) acc_fixed_g1<C>(V0, key, g1_base_h(key), src.rs(1));
  if (retr) {
    if (ws1)
      g1_mul_glv_with<C, WsTab<G1F>>(V1, WsTab<G1F>{ws1 + 8 * vtab_entry_words<G1F>()}, c);
    else
      g1_mul_glv_tab<C>(V1, tab1, c);
    if (!odd) acc_fixed_g1<C>(V1, key, g1_base_geg(key), r_e);
  } else {
    jac_set_inf(V1);
  }
  // canonical affine forms: one inversion per lane
  Fp<C> z1[2], zi1[2];
  Fp2<C> z2[2], zi2[2];
  z1[0] = V0.Z;
  z1[1] = V1.Z;
  z2[0] = Vk.Z;
  z2[1] = K.Z;
  batch_zinv<C, 2, 2>(zi1, z1, zi2, z2);
  Aff<G2F> aVk;
  Aff<G1F> a0, a1;
  jac_to_aff_with_zinv<G2F>(aVk, Vk, zi2[0]);
  jac_to_aff_with_zinv<G2F>(aK, K, zi2[1]);
  jac_to_aff_with_zinv<G1F>(a0, V0, zi1[0]);
  jac_to_aff_with_zinv<G1F>(a1, V1, zi1[1]);
  // this lane's two commitments as wire bytes, and the partner's
  uint8_t mine[2][C::FBYTES], theirs[2][C::FBYTES];
  g1_serialize<C>(mine[0], a0);
  g1_serialize<C>(mine[1], a1);
  for (int q = 0; q < 2; q++)
    for (int i = 0; i < C::FBYTES; i += 4) {
      const u32 wv = (u32)mine[q][i] | ((u32)mine[q][i + 1] << 8) | ((u32)mine[q][i + 2] << 16) | ((u32)mine[q][i + 3] << 24);
      const u32 pv = (u32)pair_swap_i32((int32_t)wv);
      theirs[q][i] = (uint8_t)pv;
      theirs[q][i + 1] = (uint8_t)(pv >> 8);
      theirs[q][i + 2] = (uint8_t)(pv >> 16);
      theirs[q][i + 3] = (uint8_t)(pv >> 24);
    }
  // c' = Hr(SHA256(hex k | hex phi | [hex E1 | hex E2] | hex V_k | hex V_phi | [hex V_E1 | hex V_E2] | ad))          (:111-122)
  uint8_t buf[2 * C::FBYTES];
  Transcript t;
  transcript_init(t);
  src.ser_k(buf);
  sha256_update_hex(t.s, buf, 2 * C::FBYTES);
  src.ser_g1(0, buf);
  sha256_update_hex(t.s, buf, C::FBYTES);
  if (retr) {
    src.ser_g1(1, buf);
    sha256_update_hex(t.s, buf, C::FBYTES);
    src.ser_g1(2, buf);
    sha256_update_hex(t.s, buf, C::FBYTES);
  }
  transcript_g2<C>(t, aVk);
  sha256_update_hex(t.s, odd ? theirs[0] : mine[0], C::FBYTES);          // V_phi (even lane's first job)
  if (retr) {
    sha256_update_hex(t.s, odd ? theirs[1] : mine[1], C::FBYTES);        // V_E1  (even lane's second job)
    sha256_update_hex(t.s, odd ? mine[0] : theirs[0], C::FBYTES);        // V_E2  (odd lane's first job)
  }
  const Scalar c2 = transcript_challenge<C>(t, ad, ad_len);
  return scalar_eq(c2, c);
}

// K = k prod_{revealed} YY_i^{m_i} alone (src/ps-verifier.cc:72-88), affine: what the pairing check needs of the NIZK half
template <class C, class Src>
ELP_HEAVY void verify_id_paired_k(const KeyCtx<C>& key, Src& src, const Aff<F2<C>>& kk, Aff<F2<C>>& aK) {
  typedef F2<C> G2F;
  Jac<G2F> K;
  jac_from_aff(K, kk);
  for (int i = 0; i < key.A; i++)
    if (!src.hidden(i)) acc_fixed_g2<C>(K, key, G2_BASE_YY0 + i, src.next_revealed_hash(i));
  Fp2<C> z2[1], zi2[1];
  z2[0] = K.Z;
  batch_zinv<C, 0, 1>((Fp<C>*)0, (const Fp<C>*)0, zi2, z2);
  jac_to_aff_with_zinv<G2F>(aK, K, zi2[0]);
}
// pair_first (wave-uniform; k_verify_id_paired under KEY_PHASE_MIX): the pairing check runs BEFORE the NIZK half.  The two halves are independent once K exists
// (the verdict is their AND), and a launch whose waves all walk them in the same order has every wave in the same phase at the same time -- the private-memory
// working sets of the Fp12 arithmetic of all waves compete for L2 together.  With half of the workgroups in the other order a SIMD's two waves are in different
// phases for most of the launch.
template <class C>
ELP_HEAVY bool verify_id_item_paired(const KeyCtx<C>& key, const u32* rec, u64 hidden_mask, bool retr, const uint8_t* ad, size_t ad_len, bool pair_first = false) {
  static_assert(is_paired<C>(), "paired layout only");
  const bool odd = pair_odd();
  const int G1W = 2 * C::N;
  Aff<F1<C>> sig1, sig2, P0, P1;
  Aff<F2<C>> kk, aK;
  // both lanes need sig1 / sig2 (line evaluation); the NIZK points go to the lane that works on them: even phi, E1; odd E2
  bool ok = g1_load<C>(sig1, rec);
  ok &= g1_load<C>(sig2, rec + G1W);
  aff_set_inf(P0);
  aff_set_inf(P1);
  if (!odd || retr) ok &= g1_load<C>(P0, rec + (odd ? 4 : 2) * G1W);
  if (!odd && retr) ok &= g1_load<C>(P1, rec + 3 * G1W);
  if constexpr (!C::IS_BN) {
    // order-r subgroup membership (g1_in_subgroup): the even lane, which has two of the three multiplications to do, checks phi; the odd lane checks
    // E2 and E1 (handed over by lane exchange)
    if (!(key.flags & KEY_NO_SUBGROUP_CHECK)) {
      // second test of a lane: the odd lane takes E1, the even lane sig1 under KEY_STRICT_SIG (sig1_admissible above; both lanes hold sig1) -- four tests in
      // the time of two
      Aff<F1<C>> Q1;
      Q1.x = fp_pair_swap(P1.x);
      Q1.y = fp_pair_swap(P1.y);
      if (!odd) {
        if (key.flags & KEY_STRICT_SIG) Q1 = sig1; else aff_set_inf(Q1);
      }
      if (ok) ok = g1_in_subgroup<C>(P0) && g1_in_subgroup<C>(Q1);
    }
  }
  ok = pair_and(ok);
  const bool okk = g2_load<C>(kk, rec + (retr ? 5 : 3) * G1W);
  if (!ok || !okk) return false;
  if ((key.flags & KEY_STRICT_SIG) && aff_is_inf(sig1)) return false;       // with the subgroup test above: sig1 has order exactly r
  const Scalar c = scalar_load_w(rec + (retr ? 5 : 3) * G1W + 4 * C::N);
  PairedRecordSrc<C> src;
  src.init(rec, hidden_mask, key.A, retr);
  if (pair_first) {
    verify_id_paired_k<C, PairedRecordSrc<C>>(key, src, kk, aK);
    const bool pok = ps_pairing_check<C>(key, sig1, sig2, aK);
    Aff<F2<C>> unused;
    src.init(rec, hidden_mask, key.A, retr);
    const bool nok = verify_id_paired_nizk<C, PairedRecordSrc<C>>(key, src, retr, P0, P1, kk, c, ad, ad_len, unused, false);
    return pok && nok;
  }
  if (!verify_id_paired_nizk<C, PairedRecordSrc<C>>(key, src, retr, P0, P1, kk, c, ad, ad_len, aK)) return false;
  return ps_pairing_check<C>(key, sig1, sig2, aK);
}

// The item of AGGREGATED verification in the paired layout (round 6; kernel k_verify_id_agg_paired): what verify_id_agg_item does on one lane -- the NIZK half, the
// multiplier d, [d]sig1 and ONE Miller loop f = f_K([d]sig1) -- on the lane pair of verify_id_item_paired.  The 14-limb field of BLS12-381 is what the two-lane
// layout exists for: one lane per item runs this kernel at 35.7 ms per 32 768 items, slower than the per-item path it is meant to beat.
// [d]sig1 is base-field work and does not split over the pair: both lanes compute it (both need the affine result for their halves of the line evaluations).
// Same admission rules as the per-item paired path (its opening lines are repeated here rather than shared, so that the headline kernel's code is untouched).
template <class C>
ELP_HEAVY bool verify_id_agg_item_paired(const KeyCtx<C>& key, const u32* rec, u64 hidden_mask, bool retr, const uint8_t* ad, size_t ad_len, const uint8_t* seed,
                                         u64 index, Fp12<C>& f, u32* delta_out, u32* sig2_out) {
  static_assert(is_paired<C>(), "paired layout only");
  const bool odd = pair_odd();
  const int G1W = 2 * C::N;
  Aff<F1<C>> sig1, sig2, P0, P1, aP;
  Aff<F2<C>> kk, aK;
  fp12_set_one(f);
  if (!odd) {
    for (int i = 0; i < 8; i++) delta_out[i] = 0;
    for (int i = 0; i < 2 * C::N; i++) sig2_out[i] = 0;
  }
  bool ok = g1_load<C>(sig1, rec);
  ok &= g1_load<C>(sig2, rec + G1W);
  aff_set_inf(P0);
  aff_set_inf(P1);
  if (!odd || retr) ok &= g1_load<C>(P0, rec + (odd ? 4 : 2) * G1W);
  if (!odd && retr) ok &= g1_load<C>(P1, rec + 3 * G1W);
  if constexpr (!C::IS_BN) {
    if (!(key.flags & KEY_NO_SUBGROUP_CHECK)) {
      Aff<F1<C>> Q1;
      Q1.x = fp_pair_swap(P1.x);
      Q1.y = fp_pair_swap(P1.y);
      if (!odd) {
        if (key.flags & KEY_STRICT_SIG) Q1 = sig1; else aff_set_inf(Q1);
      }
      if (ok) ok = g1_in_subgroup<C>(P0) && g1_in_subgroup<C>(Q1);
    }
  }
  ok = pair_and(ok);
  const bool okk = g2_load<C>(kk, rec + (retr ? 5 : 3) * G1W);
  if (!ok || !okk) return false;
  if ((key.flags & KEY_STRICT_SIG) && aff_is_inf(sig1)) return false;
  const Scalar c = scalar_load_w(rec + (retr ? 5 : 3) * G1W + 4 * C::N);
  PairedRecordSrc<C> src;
  src.init(rec, hidden_mask, key.A, retr);
  if (!verify_id_paired_nizk<C, PairedRecordSrc<C>>(key, src, retr, P0, P1, kk, c, ad, ad_len, aK)) return false;
  const Scalar d = agg_multiplier(seed, index);
  agg_scaled_sig1<C>(key, sig1, d, aP);
  if (!odd) {
    for (int i = 0; i < 8; i++) delta_out[i] = d.v[i];
    g1_store<C>(sig2_out, sig2);
  }
  Fp12<C>* fh = hot_as<Fp12<C>, C>(key.hot);               // the accumulator of the loop in the lane's LDS slot, as in ps_pairing_check
  Fp12<C>& fm = fh ? *fh : f;
  const LineMem<C>* no_lines[1] = {key.gg_lines};
  miller_loop<C, 1, 0>(fm, &aP, &aK, &aP, no_lines);
  if (fh) f = fm;
  return true;
}

// ------------------------------------------------------------------------------------------------------------
// G1 JOBS AS A KERNEL OF THEIR OWN (round 4; ELP_OPT_SPLIT_PHASES = 3, opt-in on both curves: measured no faster than the fused kernels).
//
// In the paired layout the base-field work of a verification does not split over the lane pair: the three commitments V_phi, V_E1, V_E2 (one GLV
// multiplication + fixed-base terms + an inversion each, src/ps-verifier.cc:91-108) and -- on a curve with a G1 cofactor -- the four subgroup tests (phi, E1,
// E2, sig1) are whole jobs of ONE lane, seven jobs on two lanes = four job slots, a third of the paired kernel's time on BLS12-381, executed at the issue rate
// of a kernel that holds 256 registers and an Fp12 frame per lane.  Here they are a kernel of their own in the plain layout: one lane per JOB, job-uniform waves
// (workgroup b works on job b mod 4 of items [64 (b / 4), 64 (b / 4) + 64)), nothing but Jacobian G1 points in registers, several waves per SIMD.  The paired
// kernel that follows (verify_id_item_paired_g1done) keeps everything over Fp2 -- [c]k, the fixed-base G2 sums, K, the transcript hash, Miller loop and final
// exponentiation -- and reads the commitments' wire bytes and the jobs' verdicts from the launch workspace.
//   job 0 / 1 / 2 : P = phi / E1 / E2:  on the curve, in G1;  V = [c]P + fixed-base terms, affine, serialised           -> row `job` of the workspace
//   job 3         : sig1, sig2 on the curve; sig1 admissible under KEY_STRICT_SIG (sig1_admissible: != O and in G1)      -> verdict only
// Workspace (32-bit words, word w of item i at ws[w * stride + i], coalesced across a wave): rows 0..2 of FBYTES / 4 words, then the four verdict words.
template <class C>
ELP_HD constexpr int g1jobs_ws_words() { return 3 * (C::FBYTES / 4) + 4; }
template <class C>
ELP_HEAVY void vid_g1_job(const KeyCtx<C>& key, int job, const u32* rec, u64 hidden_mask, bool retr, u32* vtab_slot, u32* ws, size_t stride, size_t i) {
  static_assert(!is_paired<C>(), "plain layout: one lane per job");
  constexpr int FBW = C::FBYTES / 4;
  const int G1W = 2 * C::N;
  u32* const okw = ws + (size_t)(3 * FBW + job) * stride + i;
  if (job == 3) {
    Aff<F1<C>> sig1, sig2;
    bool ok = g1_load<C>(sig1, rec) && g1_load<C>(sig2, rec + G1W);
    if (ok) ok = sig1_strict_ok<C>(key.flags, sig1);
    *okw = ok ? 1u : 0u;
    return;
  }
  if (job != 0 && !retr) {        // no E1 / E2 in the record (the condition is uniform over the launch)
    *okw = 1u;
    return;
  }
  Aff<F1<C>> P;
  bool ok = g1_load<C>(P, rec + (2 + job) * G1W);
  if constexpr (!C::IS_BN) {
    if (ok && !(key.flags & KEY_NO_SUBGROUP_CHECK)) ok = g1_in_subgroup<C>(P);
  }
  u32 v[FBW];
  for (int q = 0; q < FBW; q++) v[q] = 0;
  if (ok) {
    RecordSrc<C> src;
    Scalar c;
    src.open_lite(rec, hidden_mask, key.A, retr, c);
    vid_job_g1_one<C, RecordSrc<C>>(key, src, job, P, c, v, vtab_slot);
  }
  for (int q = 0; q < FBW; q++) ws[(size_t)(job * FBW + q) * stride + i] = v[q];
  *okw = ok ? 1u : 0u;
}
// what the paired kernel reads back
struct G1JobsOut {
  const u32* ws;
  size_t stride, i;
  ELP_HD bool ok() const {
    u32 a = 1;
    for (int j = 0; j < 4; j++) a &= ws[(size_t)(3 * FB4 + j) * stride + i];
    return a != 0;
  }
  int FB4;      // FBYTES / 4 of the curve
  ELP_HD void row(int t, uint8_t* out) const {        // wire bytes of commitment t
    for (int q = 0; q < FB4; q++) {
      const u32 x = ws[(size_t)(t * FB4 + q) * stride + i];
      out[4 * q] = (uint8_t)x;
      out[4 * q + 1] = (uint8_t)(x >> 8);
      out[4 * q + 2] = (uint8_t)(x >> 16);
      out[4 * q + 3] = (uint8_t)(x >> 24);
    }
  }
};
// NIZK half of the paired layout WITHOUT the G1 jobs: V_k and K by the pair, the commitments' bytes from the workspace.
template <class C, class Src>
ELP_HEAVY bool verify_id_paired_nizk_g2(const KeyCtx<C>& key, Src& src, bool retr, const Aff<F2<C>>& kk, const Scalar& c, const uint8_t* ad, size_t ad_len,
                                        const G1JobsOut& pre, Aff<F2<C>>& aK) {
  static_assert(is_paired<C>(), "paired layout only");
  typedef F2<C> G2F;
  const int A = key.A;
  const int nrs = src.nrs();
  const Scalar r_t = src.rs(retr ? nrs - 2 : nrs - 1);
  const Scalar one_minus_c = scalar_one_minus<C>(c);
  Aff<G2F> tabk[8];
  {
    Jac<G2F> jk[8];
    jac_multiples8<G2F>(jk, kk);
    Fp2<C> z2[7], zi2[7];
    for (int i = 1; i < 8; i++) z2[i - 1] = jk[i].Z;
    batch_zinv<C, 0, 7>((Fp<C>*)0, (const Fp<C>*)0, zi2, z2);
    tabk[0] = kk;
    for (int i = 1; i < 8; i++) jac_to_aff_with_zinv<G2F>(tabk[i], jk[i], zi2[i - 1]);
  }
  // V_k = k^c prod_{hidden} YY_j^{r_j} gg^{r_t} XX^{1-c},  K = k prod_{revealed} YY_i^{m_i}     (src/ps-verifier.cc:72-88,214-229)
  Jac<G2F> Vk, K;
  u32* const wsk = key.vtab;
  if (wsk) {
    for (int i = 0; i < 8; i++) vtab_store<G2F>(wsk, i, tabk[i]);
    g2_mul_gls_with<C, WsTab<G2F>>(Vk, WsTab<G2F>{wsk}, c);
  } else {
    g2_mul_gls_tab<C>(Vk, tabk, c);
  }
  jac_from_aff(K, kk);
  {
    int jh = 0;
    for (int i = 0; i < A; i++) {
      if (src.hidden(i)) {
        acc_fixed_g2<C>(Vk, key, G2_BASE_YY0 + i, src.rs(jh));
        jh++;
      } else {
        acc_fixed_g2<C>(K, key, G2_BASE_YY0 + i, src.next_revealed_hash(i));
      }
    }
  }
  acc_fixed_g2<C>(Vk, key, G2_BASE_GG, r_t);
  acc_fixed_g2<C>(Vk, key, G2_BASE_XX, one_minus_c);
  Fp2<C> z2[2], zi2[2];
  z2[0] = Vk.Z;
  z2[1] = K.Z;
  batch_zinv<C, 0, 2>((Fp<C>*)0, (const Fp<C>*)0, zi2, z2);
  Aff<G2F> aVk;
  jac_to_aff_with_zinv<G2F>(aVk, Vk, zi2[0]);
  jac_to_aff_with_zinv<G2F>(aK, K, zi2[1]);
  // c' = Hr(SHA256(hex k | hex phi | [hex E1 | hex E2] | hex V_k | hex V_phi | [hex V_E1 | hex V_E2] | ad))          (:111-122)
  uint8_t buf[2 * C::FBYTES];
  Transcript t;
  transcript_init(t);
  src.ser_k(buf);
  sha256_update_hex(t.s, buf, 2 * C::FBYTES);
  for (int q = 0; q < (retr ? 3 : 1); q++) {
    src.ser_g1(q, buf);
    sha256_update_hex(t.s, buf, C::FBYTES);
  }
  transcript_g2<C>(t, aVk);
  for (int q = 0; q < (retr ? 3 : 1); q++) {
    pre.row(q, buf);
    sha256_update_hex(t.s, buf, C::FBYTES);
  }
  const Scalar c2 = transcript_challenge<C>(t, ad, ad_len);
  return scalar_eq(c2, c);
}
template <class C>
ELP_HEAVY bool verify_id_item_paired_g1done(const KeyCtx<C>& key, const u32* rec, u64 hidden_mask, bool retr, const uint8_t* ad, size_t ad_len,
                                            const G1JobsOut& pre) {
  static_assert(is_paired<C>(), "paired layout only");
  const int G1W = 2 * C::N;
  Aff<F1<C>> sig1, sig2;
  Aff<F2<C>> kk, aK;
  // both lanes need sig1 / sig2 (line evaluation); their validity, the strict-signature rule and everything about phi / E1 / E2 were decided by the G1 jobs
  bool ok = g1_load<C>(sig1, rec);
  ok &= g1_load<C>(sig2, rec + G1W);
  ok &= pre.ok();
  const bool okk = g2_load<C>(kk, rec + (retr ? 5 : 3) * G1W);
  if (!ok || !okk) return false;
  const Scalar c = scalar_load_w(rec + (retr ? 5 : 3) * G1W + 4 * C::N);
  PairedRecordSrc<C> src;
  src.init(rec, hidden_mask, key.A, retr);
  if (!verify_id_paired_nizk_g2<C, PairedRecordSrc<C>>(key, src, retr, kk, c, ad, ad_len, pre, aK)) return false;
  return ps_pairing_check<C>(key, sig1, sig2, aK);
}

// ---- wire ingest (SURVEY.md section 8f rank 1 + 2): T-L-V parsing, point decompression (one Fp / Fp2 square root each) and
// Fr::setHashOf of the revealed attributes on the device.  Layout (src/ps-encoding.cc:451-467, Appendix C of SURVEY.md):
//   01 L sig1 | 01 L sig2 | 02 2L k | 01 L phi | 03 20 c | 06 m (20 r)* | 07 A (len str)* [| 01 L E1 | 01 L E2],  L = FBYTES
// var = 1 byte below 253, else FD hi lo.  Anything malformed or truncated rejects the item (the reference has undefined
// behaviour there); scalars must be < r as mcl's Fr::deserialize requires.
template <class C>
struct WireSrc {
  const uint8_t *b_, *rs_, *attr_;   // message, first response, cursor into the string list
  size_t len_;
  int nrs_, nattr_, attr_i_;
  u64 mask_;

  ELP_HD static bool var(const uint8_t* b, size_t len, size_t& off, size_t& v) {
    if (off >= len) return false;
    uint8_t f = b[off];
    if (f < 253) {
      v = f;
      off += 1;
      return true;
    }
    if (f == 253 && off + 2 < len) {
      v = ((size_t)b[off + 1] << 8) | b[off + 2];
      off += 3;
      return true;
    }
    return false;
  }
  ELP_HD static bool elem(const uint8_t* b, size_t len, size_t& off, uint8_t type, size_t want, const uint8_t*& body) {
    if (off >= len || b[off] != type) return false;
    off += 1;
    size_t n;
    if (!var(b, len, off, n) || n != want || off + n > len) return false;
    body = b + off;
    off += n;
    return true;
  }
  ELP_HD static bool scalar_ok(const uint8_t* p, Scalar& s) {
    s = scalar_load_le(p);
    return !scalar_geq_r<C>(s);
  }
  bool sub_ = true;                                       // as RecordSrc::sub_
  const uint8_t *p1_, *p2_, *pk_, *pphi_, *pe1_, *pe2_;   // bodies of the point elements (validated lengths)
  bool kflag_ = false;                                    // flag bit of k's canonical encoding (set by whoever decodes k)
  // structure, lengths and scalar ranges of the message; no point is decoded here
  ELP_HD bool parse(const uint8_t* msg, size_t len, int A, bool retr, Scalar& c) {
    constexpr size_t L = C::FBYTES;
    b_ = msg;
    len_ = len;
    size_t off = 0, n;
    const uint8_t* pc;
    pe1_ = pe2_ = nullptr;
    if (!elem(msg, len, off, 1, L, p1_) || !elem(msg, len, off, 1, L, p2_) || !elem(msg, len, off, 2, 2 * L, pk_) ||
        !elem(msg, len, off, 1, L, pphi_) || !elem(msg, len, off, 3, 32, pc))
      return false;
    // FrList
    if (off >= len || msg[off] != 6) return false;
    off += 1;
    if (!var(msg, len, off, n) || n > 64) return false;
    nrs_ = (int)n;
    rs_ = msg + off;
    for (int j = 0; j < nrs_; j++) {
      size_t l;
      if (!var(msg, len, off, l) || l != 32 || off + 32 > len) return false;
      Scalar t;
      if (!scalar_ok(msg + off, t)) return false;
      off += 32;
    }
    // StrList: record where it starts, derive the hidden mask
    if (off >= len || msg[off] != 7) return false;
    off += 1;
    if (!var(msg, len, off, n) || (int)n != A) return false;   // the reference indexes YYi[i] unchecked; we require the key's count
    nattr_ = A;
    attr_ = msg + off;
    attr_i_ = 0;
    mask_ = 0;
    int H = 0;
    for (int i = 0; i < A; i++) {
      size_t l;
      if (!var(msg, len, off, l) || off + l > len) return false;
      if (l == 0) {
        mask_ |= 1ull << i;
        H++;
      }
      off += l;
    }
    if (nrs_ != H + (retr ? 2 : 1) || H < (retr ? 2 : 1)) return false;
    if (retr) {
      if (!elem(msg, len, off, 1, L, pe1_) || !elem(msg, len, off, 1, L, pe2_)) return false;   // src/ps-verifier.cc:68-70
    }
    return scalar_ok(pc, c);
  }
  ELP_HD bool open(const uint8_t* msg, size_t len, int A, bool retr, Aff<F1<C>>& sig1, Aff<F1<C>>& sig2, Aff<F1<C>>& phi,
                   Aff<F1<C>>& E1, Aff<F1<C>>& E2, Aff<F2<C>>& kk, Scalar& c) {
    if (!parse(msg, len, A, retr, c)) return false;
    if (retr && (!g1_deserialize<C>(E1, pe1_) || !g1_deserialize<C>(E2, pe2_))) return false;
    if (!(g1_deserialize<C>(sig1, p1_) && g1_deserialize<C>(sig2, p2_) && g1_deserialize<C>(phi, pphi_) && g2_deserialize<C>(kk, pk_, &kflag_))) return false;
    if constexpr (!C::IS_BN) {
      if (sub_ && !(g1_in_subgroup<C>(phi) && (!retr || (g1_in_subgroup<C>(E1) && g1_in_subgroup<C>(E2))))) return false;
    }
    return true;
  }
  // wire bytes of the transcript's input points: the message's own bytes.  Every G1 encoding the decoder accepts is canonical (x < p, the
  // flag is the parity the decoder enforces on y != 0, infinity is all-zero); for k the flag is replaced by the canonical one (a set flag on
  // a point with y.a = 0 is accepted by the reference's decoder and re-serialised with the flag clear: g2_deserialize, encode.h)
  ELP_HD void ser_k(uint8_t* out) const {
    for (int i = 0; i < 2 * C::FBYTES; i++) out[i] = pk_[i];
    out[2 * C::FBYTES - 1] = (uint8_t)((out[2 * C::FBYTES - 1] & 0x7f) | (kflag_ ? 0x80 : 0));
  }
  ELP_HD void ser_g1(int which, uint8_t* out) const {   // 0 = phi, 1 = E1, 2 = E2
    const uint8_t* p = which == 0 ? pphi_ : (which == 1 ? pe1_ : pe2_);
    for (int i = 0; i < C::FBYTES; i++) out[i] = p[i];
  }
  ELP_HD int nrs() const { return nrs_; }
  ELP_HD bool hidden(int i) const { return (mask_ >> i) & 1; }
  // entry j of the FrList: each entry is var(32) | 32 bytes, where var is normally the single byte 20 but may be the 3-byte form
  // FD 00 20 (the reference's parseVar accepts both, src/ps-encoding.cc:149-162), so the list is walked (bounds validated in open())
  ELP_HD Scalar rs(int j) const {
    const uint8_t* p = rs_;
    for (int i = 0; i < j; i++) p += (p[0] == 253 ? 3 : 1) + 32;
    return scalar_load_le(p + (p[0] == 253 ? 3 : 1));
  }
  // Fr::setHashOf(attributes[i]) for the next revealed attribute i (called with increasing i)
  ELP_HD Scalar next_revealed_hash(int i) {
    size_t off = 0, l = 0;
    const size_t rem = (size_t)(b_ + len_ - attr_);
    while (attr_i_ <= i) {
      var(attr_, rem, off, l);     // bounds were validated in open()
      if (attr_i_ == i) break;
      off += l;
      attr_i_++;
    }
    Sha256 s;
    sha256_init(s);
    sha256_update(s, attr_ + off, l);
    uint8_t d[32];
    sha256_final(s, d);
    attr_ += off + l;
    attr_i_ = i + 1;
    return scalar_from_digest<C>(d);
  }
};

template <class C>
ELP_HEAVY bool verify_id_wire_item(const KeyCtx<C>& key, const uint8_t* msg, size_t len, bool retr, const uint8_t* ad, size_t ad_len) {
  Aff<F1<C>> sig1, sig2, phi, E1, E2;
  Aff<F2<C>> kk;
  Scalar c;
  WireSrc<C> src;
  src.sub_ = (key.flags & KEY_NO_SUBGROUP_CHECK) == 0;
  if (!src.open(msg, len, key.A, retr, sig1, sig2, phi, E1, E2, kk, c)) return false;
  return verify_id_core<C, WireSrc<C>>(key, src, retr, sig1, sig2, phi, E1, E2, kk, c, ad, ad_len);
}

// ---- wire message -> verify_id record, job by job (round 5: wire batches on the small / mid-size paths).  The record paths (interpreter, four lanes per item)
// take fixed-stride records; a relying party receives IdProof::toBufferString() messages (src/ps-encoding.cc:451-467).  k_wire_decode turns one into the other with
// job-uniform waves: job 0..4 decompress sig1, sig2, phi, E1, E2 (one Fp square root each), job 5 decompresses k (Fp2 square root), job 6 validates the structure,
// copies c and the responses, hashes the revealed attributes (Fr::setHashOf, src/ps-verifier.cc:224) and reports the hidden mask.  Every job parses the T-L-V
// structure itself (a few hundred instructions) and writes its own words of the record; ok[job] = 0 marks the item invalid.  The record's size does not depend on the
// number of hidden attributes (1 + (H + 2 | H + 1) + (A - H) scalars), only its interpretation does: the record kernels take ONE mask per launch, so the caller runs
// them only when every message of the batch has the same pattern.  Subgroup tests (BLS12-381) are left to the record path, which repeats them on the decoded points.
constexpr int WIRE_DECODE_JOBS = 7;
template <class C>
ELP_HEAVY bool wire_decode_job(int job, int A, const uint8_t* msg, size_t len, bool retr, u32* rec, u64* mask_out) {
  WireSrc<C> src;
  Scalar c;
  if (!src.parse(msg, len, A, retr, c)) return false;
  const int G1W = 2 * C::N;
  if (job < 5) {
    if (job >= 3 && !retr) return true;
    const uint8_t* body = job == 0 ? src.p1_ : job == 1 ? src.p2_ : job == 2 ? src.pphi_ : job == 3 ? src.pe1_ : src.pe2_;
    Aff<F1<C>> P;
    if (!g1_deserialize<C>(P, body)) return false;
    g1_store<C>(rec + job * G1W, P);
    return true;
  }
  u32* const wk = rec + (retr ? 5 : 3) * G1W;
  if (job == 5) {
    Aff<F2<C>> kk;
    bool kflag = false;
    if (!g2_deserialize<C>(kk, src.pk_, &kflag)) return false;
    g2_store<C>(wk, kk);
    return true;
  }
  // job 6: scalars
  u32* w = wk + 4 * C::N;
  for (int i = 0; i < 8; i++) w[i] = c.v[i];
  w += 8;
  for (int j = 0; j < src.nrs(); j++) {
    const Scalar r = src.rs(j);
    for (int i = 0; i < 8; i++) w[i] = r.v[i];
    w += 8;
  }
  for (int i = 0; i < A; i++)
    if (!src.hidden(i)) {
      const Scalar m = src.next_revealed_hash(i);
      for (int q = 0; q < 8; q++) w[q] = m.v[q];
      w += 8;
    }
  *mask_out = src.mask_;
  return true;
}

// The same in the paired layout: both lanes parse the message; the five G1 decompressions (one Fp square root each) are split between the
// lanes (even: sig1, phi, E1; odd: sig2, E2), sig1 / sig2 are then swapped so both lanes hold both, k is decompressed by the pair together.
template <class C>
ELP_HEAVY bool verify_id_wire_item_paired(const KeyCtx<C>& key, const uint8_t* msg, size_t len, bool retr, const uint8_t* ad, size_t ad_len) {
  static_assert(is_paired<C>(), "paired layout only");
  const bool odd = pair_odd();
  WireSrc<C> src;
  Scalar c;
  if (!src.parse(msg, len, key.A, retr, c)) return false;
  Aff<F1<C>> S, P0, P1;
  Aff<F2<C>> kk, aK;
  bool ok = g1_deserialize<C>(S, odd ? src.p2_ : src.p1_);
  aff_set_inf(P0);
  aff_set_inf(P1);
  if (!odd || retr) ok &= g1_deserialize<C>(P0, odd ? src.pe2_ : src.pphi_);
  if (!odd && retr) ok &= g1_deserialize<C>(P1, src.pe1_);
  if constexpr (!C::IS_BN) {
    if (!(key.flags & KEY_NO_SUBGROUP_CHECK)) {      // as in verify_id_item_paired: even lane phi and (strict) sig1, odd lane E2 and E1
      Aff<F1<C>> Q1;
      Q1.x = fp_pair_swap(P1.x);
      Q1.y = fp_pair_swap(P1.y);
      if (!odd) {
        if (key.flags & KEY_STRICT_SIG) Q1 = S; else aff_set_inf(Q1);        // S is sig1 on the even lane
      }
      if (ok) ok = g1_in_subgroup<C>(P0) && g1_in_subgroup<C>(Q1);
    }
  }
  ok = pair_and(ok);
  const bool okk = g2_deserialize<C>(kk, src.pk_, &src.kflag_);
  if (!ok || !okk) return false;
  Aff<F1<C>> T, sig1, sig2;
  T.x = fp_pair_swap(S.x);
  T.y = fp_pair_swap(S.y);
  sig1.x = fp_select(odd, T.x, S.x);
  sig1.y = fp_select(odd, T.y, S.y);
  sig2.x = fp_select(odd, S.x, T.x);
  sig2.y = fp_select(odd, S.y, T.y);
  if ((key.flags & KEY_STRICT_SIG) && aff_is_inf(sig1)) return false;
  if (!verify_id_paired_nizk<C, WireSrc<C>>(key, src, retr, P0, P1, kk, c, ad, ad_len, aK)) return false;
  return ps_pairing_check<C>(key, sig1, sig2, aK);
}

// ------------------------------------------------------------------------------------------------------------
// Plain PS verification.  Record: sig1 | sig2 | m[A]   (src/ps-verifier.cc:13-35)
template <class C>
ELP_HEAVY bool ps_verify_item(const KeyCtx<C>& key, const u32* rec, int nattr) {
  typedef F1<C> G1F;
  typedef F2<C> G2F;
  Aff<G1F> sig1, sig2;
  if (!g1_load<C>(sig1, rec)) return false;
  if (!g1_load<C>(sig2, rec + 2 * C::N)) return false;
  if (!sig1_admissible<C>(key.flags, sig1)) return false;   // src/ps-verifier.cc:16-18 -- asked of the order-r component (BLS12-381: both lanes of a pair hold sig1 and agree)
  const u32* ms = rec + 4 * C::N;
  Jac<G2F> K;
  jac_from_aff(K, aff_from_mem<G2F>(key.b2[G2_BASE_XX]));
  for (int i = 0; i < nattr; i++) acc_fixed_g2<C>(K, key, G2_BASE_YY0 + i, scalar_load_w(ms + 8 * i));
  Aff<G2F> aK;
  jac_to_aff<G2F>(aK, K);
  if (ELP_BISECT_AT(1)) return !aff_is_inf(aK);
  Aff<G1F> nsig2;
  aff_neg(nsig2, sig2);
  if (aff_is_inf(sig2)) aff_set_inf(nsig2);
  Fp12<C> f_priv;
  Fp12<C>* fh = hot_as<Fp12<C>, C>(key.hot);
  Fp12<C>& f = fh ? *fh : f_priv;
  const LineMem<C>* lines[1] = {key.gg_lines};
  miller_loop<C, 1, 1>(f, &sig1, &aK, &nsig2, lines);
  if (ELP_BISECT_AT(2)) return fp2_is_zero<C>(f.c0.c0);
  return final_exp_is_one<C>(f, key.hot);
}

// ------------------------------------------------------------------------------------------------------------
// EL PASSO ProvideID (IdP issuance).  Record: A | c | rs[H+1] | m[A-H] | u      Output: sig1 | sig2
//   (src/ps-signer.cc:63-146; the CSPRNG nonce u of sign_commitment is an input so results are reproducible)
template <class C>
ELP_HD constexpr int provide_id_record_words(int A, int H) {
  return 2 * C::N + 8 * (1 + (H + 1) + (A - H) + 1);
}
template <class C>
ELP_HEAVY bool provide_id_item(const KeyCtx<C>& key, const u32* rec, u64 hidden_mask, const uint8_t* ad, size_t ad_len,
                               u32* out) {
  typedef F1<C> G1F;
  const int A = key.A;
  int H = 0;
  for (int i = 0; i < A; i++) H += (int)((hidden_mask >> i) & 1);
  Aff<G1F> Ac;
  for (int i = 0; i < 4 * C::N; i++) out[i] = 0;
  if (!g1_load<C>(Ac, rec)) return false;
  if constexpr (!C::IS_BN) {
    if (!(key.flags & KEY_NO_SUBGROUP_CHECK) && !g1_in_subgroup<C>(Ac)) return false;      // the commitment to sign must lie in G1
  }
  const u32* p = rec + 2 * C::N;
  const Scalar c = scalar_load_w(p); p += 8;
  const u32* rs = p; p += 8 * (H + 1);
  const u32* ms = p; p += 8 * (A - H);
  const Scalar u = scalar_load_w(p);
  // V = A^c * g^{r_0} * prod_{hidden} Y_i^{r_j}                           (src/ps-signer.cc:82-94)
  Jac<G1F> V, Ap;
  g1_mul_glv<C>(V, Ac, c, key.hot);
  acc_fixed_g1<C>(V, key, G1_BASE_G, scalar_load_w(rs));
  jac_from_aff(Ap, Ac);
  {
    int jh = 1, jr = 0;
    for (int i = 0; i < A; i++) {
      if ((hidden_mask >> i) & 1) {
        acc_fixed_g1<C>(V, key, G1_BASE_Y0 + i, scalar_load_w(rs + 8 * jh));
        jh++;
      } else {
        // A' = A * prod_{revealed} Y_i^{m_i}; skipped entirely when the key has one attribute (src/ps-signer.cc:115-117)
        if (A != 1) acc_fixed_g1<C>(Ap, key, G1_BASE_Y0 + i, scalar_load_w(ms + 8 * jr));
        jr++;
      }
    }
  }
  Aff<G1F> aV;
  jac_to_aff<G1F>(aV, V);
  Transcript t;
  transcript_init(t);
  transcript_g1<C>(t, Ac);
  transcript_g1<C>(t, aV);
  const Scalar c2 = transcript_challenge<C>(t, ad, ad_len);
  if (!scalar_eq(c2, c)) return false;
  // sigma = (g^u, (X * A')^u)                                             (src/ps-signer.cc:132-146)
  Jac<G1F> s1, s2;
  jac_set_inf(s1);
  acc_fixed_g1<C>(s1, key, G1_BASE_G, u);
  jac_madd<G1F>(Ap, Ap, aff_from_mem<G1F>(key.b1[g1_base_skx(key)]));
  Aff<G1F> aAp;
  jac_to_aff<G1F>(aAp, Ap);
  g1_mul_glv<C>(s2, aAp, u, key.hot);
  Fp<C> z[2], zi[2];
  z[0] = s1.Z;
  z[1] = s2.Z;
  batch_zinv<C, 2, 0>(zi, z, (Fp2<C>*)0, (const Fp2<C>*)0);
  Aff<G1F> a1, a2;
  jac_to_aff_with_zinv<G1F>(a1, s1, zi[0]);
  jac_to_aff_with_zinv<G1F>(a2, s2, zi[1]);
  g1_store<C>(out, a1);
  g1_store<C>(out + 2 * C::N, a2);
  return true;
}

// ------------------------------------------------------------------------------------------------------------
// User side in batch.  Randomness is an input (the reference draws it with setByCSPRNG), in the
// reference's draw order, so outputs are reproducible and can be compared with the oracle bit for bit.

// rho - secret * c  (mod r): the Schnorr responses (src/ps-requester.cc:80-85,280-294); Fr as a second Montgomery field
template <class C>
ELP_HEAVY Scalar fr_response(const Scalar& rho, const Scalar& secret, const Scalar& c) {
  typedef typename FrOf<C>::type R;
  StdFp<R> a, b;
  for (int i = 0; i < 8; i++) {
    a.w[i] = secret.v[i];
    b.w[i] = c.v[i];
  }
  StdFp<R> t = fp_to_std<R>(fp_mul<R>(fp_from_std<R>(a), fp_from_std<R>(b)));
  Scalar prod;
  for (int i = 0; i < 8; i++) prod.v[i] = t.w[i];
  return scalar_sub_mod_r<C>(scalar_mod_r<C>(rho), prod);
}

// EL PASSO RequestID (src/ps-requester.cc:19-99).  Record: m[A] | t1 | rho0 | rho[H]     (m = Fr::setHashOf(attribute), all A)
// Output: A | c | rs[H+1]   (the first part of the provide_id record: append m_revealed and the IdP nonce to issue)
template <class C>
ELP_HD constexpr int request_id_record_words(int A, int H) { return 8 * (A + 2 + H); }
template <class C>
ELP_HD constexpr int request_id_out_words(int H) { return 2 * C::N + 8 * (2 + H); }
template <class C>
ELP_HEAVY void request_id_item(const KeyCtx<C>& key, const u32* rec, u64 hidden_mask, const uint8_t* ad, size_t ad_len, u32* out) {
  typedef F1<C> G1F;
  const int A = key.A;
  const u32* ms = rec;
  const Scalar t1 = scalar_load_w(rec + 8 * A), rho0 = scalar_load_w(rec + 8 * (A + 1));
  const u32* rhos = rec + 8 * (A + 2);
  Jac<G1F> Ac, V;
  jac_set_inf(Ac);
  jac_set_inf(V);
  acc_fixed_g1<C>(Ac, key, G1_BASE_G, t1);                       // A = g^t prod Y_i^{m_i}
  acc_fixed_g1<C>(V, key, G1_BASE_G, rho0);                      // V = g^rho0 prod Y_i^{rho_i}
  int j = 0;
  for (int i = 0; i < A; i++)
    if ((hidden_mask >> i) & 1) {
      acc_fixed_g1<C>(Ac, key, G1_BASE_Y0 + i, scalar_load_w(ms + 8 * i));
      acc_fixed_g1<C>(V, key, G1_BASE_Y0 + i, scalar_load_w(rhos + 8 * j));
      j++;
    }
  Fp<C> z[2], zi[2];
  z[0] = Ac.Z;
  z[1] = V.Z;
  batch_zinv<C, 2, 0>(zi, z, (Fp2<C>*)0, (const Fp2<C>*)0);
  Aff<G1F> aA, aV;
  jac_to_aff_with_zinv<G1F>(aA, Ac, zi[0]);
  jac_to_aff_with_zinv<G1F>(aV, V, zi[1]);
  Transcript t;
  transcript_init(t);
  transcript_g1<C>(t, aA);
  transcript_g1<C>(t, aV);
  const Scalar c = transcript_challenge<C>(t, ad, ad_len);
  g1_store<C>(out, aA);
  u32* o = out + 2 * C::N;
  for (int i = 0; i < 8; i++) o[i] = c.v[i];
  o += 8;
  Scalar r0 = fr_response<C>(rho0, t1, c);
  for (int i = 0; i < 8; i++) o[i] = r0.v[i];
  o += 8;
  j = 0;
  for (int i = 0; i < A; i++)
    if ((hidden_mask >> i) & 1) {
      Scalar r = fr_response<C>(scalar_load_w(rhos + 8 * j), scalar_load_w(ms + 8 * i), c);
      for (int q = 0; q < 8; q++) o[q] = r.v[q];
      o += 8;
      j++;
    }
}

// EL PASSO ProveID (src/ps-requester.cc:150-310, :312-432).  Record: sig1 | sig2 | m[A] | t | r | [eps] | rho[H] | rho_t | [rho_e]
// Output: the verify_id record  sig1' | sig2' | phi | [E1 | E2] | k | c | rs | m_revealed  -- prover output = verifier input.
template <class C>
ELP_HD constexpr int prove_id_record_words(int A, int H, bool retr) { return 4 * C::N + 8 * (A + 2 + (retr ? 1 : 0) + H + 1 + (retr ? 1 : 0)); }
template <class C>
ELP_HEAVY bool prove_id_item(const KeyCtx<C>& key, const u32* rec, u64 hidden_mask, bool retr, const uint8_t* ad, size_t ad_len,
                             u32* out) {
  typedef F1<C> G1F;
  typedef F2<C> G2F;
  const int A = key.A;
  int H = 0;
  for (int i = 0; i < A; i++) H += (int)((hidden_mask >> i) & 1);
  const int out_words = verify_id_record_words<C>(A, H, retr);
  for (int i = 0; i < out_words; i++) out[i] = 0;
  Aff<G1F> sig1, sig2;
  if (!g1_load<C>(sig1, rec) || !g1_load<C>(sig2, rec + 2 * C::N)) return false;
  if (!(hidden_mask & 1) || (retr && !(hidden_mask & 2))) return false;   // attribute 0 (and 1) must be hidden, SURVEY.md section 3D
  const u32* ms = rec + 4 * C::N;
  const u32* p = ms + 8 * A;
  const Scalar t = scalar_load_w(p), rr = scalar_load_w(p + 8);
  p += 16;
  Scalar eps;
  for (int i = 0; i < 8; i++) eps.v[i] = 0;
  if (retr) {
    eps = scalar_load_w(p);
    p += 8;
  }
  const u32* rhos = p;
  const Scalar rho_t = scalar_load_w(p + 8 * H);
  Scalar rho_e = eps;
  if (retr) rho_e = scalar_load_w(p + 8 * (H + 1));
  // randomised signature (sig1^r, (sig2 sig1^t)^r)                        src/ps-requester.cc:163-170
  Jac<G1F> s1, s2, tmp;
  g1_mul_glv<C>(s1, sig1, rr, key.hot);
  g1_mul_glv<C>(tmp, sig1, t, key.hot);
  jac_madd<G1F>(tmp, tmp, sig2);
  Aff<G1F> atmp;
  jac_to_aff<G1F>(atmp, tmp);
  g1_mul_glv<C>(s2, atmp, rr, key.hot);
  // phi, E1, E2 and the commitments                                        :172-187, :227-261
  Jac<G1F> phi, E1, E2, Vphi, VE1, VE2;
  jac_set_inf(phi); jac_set_inf(E1); jac_set_inf(E2); jac_set_inf(Vphi); jac_set_inf(VE1); jac_set_inf(VE2);
  acc_fixed_g1<C>(phi, key, g1_base_hs(key), scalar_load_w(ms));
  acc_fixed_g1<C>(Vphi, key, g1_base_hs(key), scalar_load_w(rhos));
  if (retr) {
    acc_fixed_g1<C>(E1, key, g1_base_geg(key), eps);
    acc_fixed_g1<C>(E2, key, g1_base_apk(key), eps);
    acc_fixed_g1<C>(E2, key, g1_base_h(key), scalar_load_w(ms + 8));
    acc_fixed_g1<C>(VE1, key, g1_base_geg(key), rho_e);
    acc_fixed_g1<C>(VE2, key, g1_base_apk(key), rho_e);
    acc_fixed_g1<C>(VE2, key, g1_base_h(key), scalar_load_w(rhos + 8));
  }
  // k = XX prod YY_j^{m_j} gg^t ; V_k = XX prod YY_j^{rho_j} gg^{rho_t}     :189-204, :227-246
  Jac<G2F> kk, Vk;
  jac_from_aff(kk, aff_from_mem<G2F>(key.b2[G2_BASE_XX]));
  jac_from_aff(Vk, aff_from_mem<G2F>(key.b2[G2_BASE_XX]));
  {
    int j = 0;
    for (int i = 0; i < A; i++)
      if ((hidden_mask >> i) & 1) {
        acc_fixed_g2<C>(kk, key, G2_BASE_YY0 + i, scalar_load_w(ms + 8 * i));
        acc_fixed_g2<C>(Vk, key, G2_BASE_YY0 + i, scalar_load_w(rhos + 8 * j));
        j++;
      }
  }
  acc_fixed_g2<C>(kk, key, G2_BASE_GG, t);
  acc_fixed_g2<C>(Vk, key, G2_BASE_GG, rho_t);
  // affine forms with one inversion
  Fp<C> z1[8], zi1[8];
  Fp2<C> z2[2], zi2[2];
  z1[0] = s1.Z; z1[1] = s2.Z; z1[2] = phi.Z; z1[3] = Vphi.Z;
  z1[4] = retr ? E1.Z : fp_one<C>(); z1[5] = retr ? E2.Z : fp_one<C>(); z1[6] = retr ? VE1.Z : fp_one<C>(); z1[7] = retr ? VE2.Z : fp_one<C>();
  z2[0] = kk.Z;
  z2[1] = Vk.Z;
  batch_zinv<C, 8, 2>(zi1, z1, zi2, z2);
  Aff<G1F> a1, a2, aphi, aVphi, aE1, aE2, aVE1, aVE2;
  Aff<G2F> ak, aVk;
  jac_to_aff_with_zinv<G1F>(a1, s1, zi1[0]);
  jac_to_aff_with_zinv<G1F>(a2, s2, zi1[1]);
  jac_to_aff_with_zinv<G1F>(aphi, phi, zi1[2]);
  jac_to_aff_with_zinv<G1F>(aVphi, Vphi, zi1[3]);
  jac_to_aff_with_zinv<G2F>(ak, kk, zi2[0]);
  jac_to_aff_with_zinv<G2F>(aVk, Vk, zi2[1]);
  Transcript tr;
  transcript_init(tr);
  transcript_g2<C>(tr, ak);
  transcript_g1<C>(tr, aphi);
  if (retr) {
    jac_to_aff_with_zinv<G1F>(aE1, E1, zi1[4]);
    jac_to_aff_with_zinv<G1F>(aE2, E2, zi1[5]);
    jac_to_aff_with_zinv<G1F>(aVE1, VE1, zi1[6]);
    jac_to_aff_with_zinv<G1F>(aVE2, VE2, zi1[7]);
    transcript_g1<C>(tr, aE1);
    transcript_g1<C>(tr, aE2);
  }
  transcript_g2<C>(tr, aVk);
  transcript_g1<C>(tr, aVphi);
  if (retr) {
    transcript_g1<C>(tr, aVE1);
    transcript_g1<C>(tr, aVE2);
  }
  const Scalar c = transcript_challenge<C>(tr, ad, ad_len);
  // emit the verifier's record
  u32* o = out;
  g1_store<C>(o, a1); o += 2 * C::N;
  g1_store<C>(o, a2); o += 2 * C::N;
  g1_store<C>(o, aphi); o += 2 * C::N;
  if (retr) {
    g1_store<C>(o, aE1); o += 2 * C::N;
    g1_store<C>(o, aE2); o += 2 * C::N;
  }
  g2_store<C>(o, ak); o += 4 * C::N;
  for (int i = 0; i < 8; i++) o[i] = c.v[i];
  o += 8;
  {
    int j = 0;
    for (int i = 0; i < A; i++)
      if ((hidden_mask >> i) & 1) {
        Scalar r = fr_response<C>(scalar_load_w(rhos + 8 * j), scalar_load_w(ms + 8 * i), c);
        for (int q = 0; q < 8; q++) o[q] = r.v[q];
        o += 8;
        j++;
      }
  }
  Scalar r = fr_response<C>(rho_t, t, c);
  for (int q = 0; q < 8; q++) o[q] = r.v[q];
  o += 8;
  if (retr) {
    r = fr_response<C>(rho_e, eps, c);
    for (int q = 0; q < 8; q++) o[q] = r.v[q];
    o += 8;
  }
  for (int i = 0; i < A; i++)
    if (!((hidden_mask >> i) & 1)) {
      for (int q = 0; q < 8; q++) o[q] = ms[8 * i + q];
      o += 8;
    }
  return true;
}

// ------------------------------------------------------------------------------------------------------------
// Setup helpers

// entry (j, d) of the fixed-base table of `base`:  d * 2^(W j) * base, d = 1..2^(W-1) (signed digits); one call fills entries
// [d0, d0+cnt) of window j.  bj = 2^(W j) * base (affine).
template <class F>
ELP_HEAVY void table_fill_chunk(Aff<F>* win, const Aff<F>& bj, int d0, int cnt) {
  Scalar s;
  for (int i = 0; i < 8; i++) s.v[i] = 0;
  s.v[0] = (u32)d0;
  Jac<F> acc;
  jac_mul_var<F>(acc, bj, s);
  for (int d = d0; d < d0 + cnt; d++) {
    Aff<F> e;
    jac_to_aff<F>(e, acc);
    win[d - 1] = e;
    jac_madd<F>(acc, acc, bj);
  }
}
// bj[j] = 2^(W j) * base for all windows (sequential doublings, one lane per base)
template <class F>
ELP_HEAVY void table_window_bases(Aff<F>* bj, const Aff<F>& base, int W, int nwin) {
  Jac<F> acc;
  jac_from_aff(acc, base);
  for (int j = 0; j < nwin; j++) {
    jac_to_aff<F>(bj[j], acc);
    for (int t = 0; t < W; t++) jac_dbl<F>(acc, acc);
  }
}

}  // namespace elp
