// Small C entry points over the C++ protocol classes (for language bindings and for the Python parity tests):
// base64 wire messages in, verdicts / base64 messages out.
#include <string.h>

#include <sstream>

#include "ps-requester.h"
#include "ps-signer.h"
#include "ps-verifier.h"

namespace {
std::string g_err;
template <class F>
int guarded(F f) {
  try {
    return f();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
int copyOut(const std::string& s, char* out, size_t cap) {
  if (s.size() + 1 > cap) {
    g_err = "output buffer too small";
    return -1;
  }
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}
// "value Y value N ..." (Y = hide), the convention of the reference's wasm bindings (wasm-src/el-passo-user.cc:26-45)
std::vector<std::tuple<std::string, bool>> parseAttrs(const std::string& spec) {
  std::vector<std::tuple<std::string, bool>> out;
  std::istringstream is(spec);
  std::string v, f;
  while (is >> v >> f) out.emplace_back(v, f == "Y");
  return out;
}
}  // namespace

extern "C" {

const char* elph_last_error() { return g_err.c_str(); }

int elph_init(int device) {
  return guarded([&] {
    initPairing(BN254, device);
    return 0;
  });
}
// curve: 0 = BN254 (what the reference runs on), 1 = BLS12-381; one curve per process
int elph_init_curve(int curve, int device) {
  return guarded([&] {
    initPairing(curve == 1 ? BLS12_381 : BN254, device);
    return 0;
  });
}

// verifiers created afterwards reject (1, default) or accept like the reference (0) proofs with sig1 == infinity
void elph_set_strict_signature(int strict) { elpSetStrictSignature(strict != 0); }

// 1 = accepted, 0 = rejected, -1 = error
int elph_verify_id_b64(const char* pk_b64, const char* proof_b64, const char* ad, const char* service, int with_retrieval,
                       const char* authority_seed, const char* g_seed, const char* h_seed) {
  return guarded([&] {
    PSVerifier rp(PSPubKey::fromBufferString(PSBuffer::fromBase64(pk_b64)));
    IdProof proof = IdProof::fromBufferString(PSBuffer::fromBase64(proof_b64));
    if (!with_retrieval) return rp.el_passo_verify_id_without_id_retrieval(proof, ad, service) ? 1 : 0;
    G1 apk, g, h;
    hashAndMapToG1(apk, authority_seed);
    hashAndMapToG1(g, g_seed);
    hashAndMapToG1(h, h_seed);
    return rp.el_passo_verify_id(proof, ad, service, apk, g, h) ? 1 : 0;
  });
}

int elph_user_name_b64(const char* proof_b64, char* out, size_t cap) {
  return guarded([&] {
    IdProof proof = IdProof::fromBufferString(PSBuffer::fromBase64(proof_b64));
    return copyOut(PSVerifier::get_user_name_from_signon_request(proof), out, cap);
  });
}

int elph_ps_verify_b64(const char* pk_b64, const char* cred_b64, const char* attrs_spec) {
  return guarded([&] {
    PSVerifier rp(PSPubKey::fromBufferString(PSBuffer::fromBase64(pk_b64)));
    std::vector<std::string> all;
    for (auto& a : parseAttrs(attrs_spec)) all.push_back(std::get<0>(a));
    return rp.verify(PSCredential::fromBufferString(PSBuffer::fromBase64(cred_b64)), all) ? 1 : 0;
  });
}

// Prover with an injected random source (n_rand scalars, 32 bytes each, draw order of the reference).  Returns the length of
// the base64 IdProof written to out.
int elph_prove_id_b64(const char* pk_b64, const char* cred_b64, const char* attrs_spec, const char* ad, const char* service,
                      int with_retrieval, const char* authority_seed, const char* g_seed, const char* h_seed,
                      const uint8_t* rand32, size_t n_rand, char* out, size_t cap) {
  return guarded([&] {
    PSRequester user(PSPubKey::fromBufferString(PSBuffer::fromBase64(pk_b64)));
    std::vector<Fr> rnd(n_rand);
    for (size_t i = 0; i < n_rand; i++) memcpy(rnd[i].b, rand32 + 32 * i, 32);
    user.set_random_source(rnd);
    PSCredential cred = PSCredential::fromBufferString(PSBuffer::fromBase64(cred_b64));
    IdProof proof;
    if (with_retrieval) {
      G1 apk, g, h;
      hashAndMapToG1(apk, authority_seed);
      hashAndMapToG1(g, g_seed);
      hashAndMapToG1(h, h_seed);
      proof = user.el_passo_prove_id(cred, parseAttrs(attrs_spec), ad, service, apk, g, h);
    } else {
      proof = user.el_passo_prove_id_without_id_retrieval(cred, parseAttrs(attrs_spec), ad, service);
    }
    return copyOut(proof.toBufferString().toBase64(), out, cap);
  });
}

// request_id with injected randomness -> base64 PSCredRequest
int elph_request_id_b64(const char* pk_b64, const char* attrs_spec, const char* ad, const uint8_t* rand32, size_t n_rand, char* out,
                        size_t cap) {
  return guarded([&] {
    PSRequester user(PSPubKey::fromBufferString(PSBuffer::fromBase64(pk_b64)));
    std::vector<Fr> rnd(n_rand);
    for (size_t i = 0; i < n_rand; i++) memcpy(rnd[i].b, rand32 + 32 * i, 32);
    user.set_random_source(rnd);
    return copyOut(user.el_passo_request_id(parseAttrs(attrs_spec), ad).toBufferString().toBase64(), out, cap);
  });
}

// PSSigner::key_gen with injected secrets (key_gen_from): generators g, gg are taken from `pk_template_b64`; returns the base64
// PSPubKey (src/ps-signer.cc:29-55).
int elph_key_gen_b64(const char* pk_template_b64, const uint8_t* x32, const uint8_t* ys32, size_t nattr, char* out, size_t cap) {
  return guarded([&] {
    PSPubKey tpl = PSPubKey::fromBufferString(PSBuffer::fromBase64(pk_template_b64));
    PSSigner idp(nattr, tpl.g, tpl.gg);
    Fr x;
    memcpy(x.b, x32, 32);
    std::vector<Fr> ys(nattr);
    for (size_t i = 0; i < nattr; i++) memcpy(ys[i].b, ys32 + 32 * i, 32);
    return copyOut(idp.key_gen_from(x, ys).toBufferString().toBase64(), out, cap);
  });
}

// User-side credential handling with injected randomness (src/ps-requester.cc:19-99,101-113,139-148): el_passo_request_id draws
// t1, rho_0, rho_j from rand32 and remembers t1; the (blinded) credential is then unblinded with it and re-randomised with the
// next injected scalar.  Writes the two base64 PSCredentials; returns 0.
int elph_unblind_randomize_b64(const char* pk_b64, const char* attrs_spec, const char* ad, const uint8_t* rand32, size_t n_rand,
                               const char* blinded_cred_b64, char* out_unblinded, size_t cap1, char* out_randomized, size_t cap2) {
  return guarded([&] {
    PSRequester user(PSPubKey::fromBufferString(PSBuffer::fromBase64(pk_b64)));
    std::vector<Fr> rnd(n_rand);
    for (size_t i = 0; i < n_rand; i++) memcpy(rnd[i].b, rand32 + 32 * i, 32);
    user.set_random_source(rnd);
    (void)user.el_passo_request_id(parseAttrs(attrs_spec), ad);
    PSCredential ub = user.unblind_credential(PSCredential::fromBufferString(PSBuffer::fromBase64(blinded_cred_b64)));
    if (copyOut(ub.toBufferString().toBase64(), out_unblinded, cap1) < 0) return -1;
    PSCredential rz = user.randomize_credential(ub);
    if (copyOut(rz.toBufferString().toBase64(), out_randomized, cap2) < 0) return -1;
    return 0;
  });
}

}  // extern "C"
