// Small C entry points over the C++ protocol classes (for language bindings and for the Python parity tests):
// base64 wire messages in, verdicts / base64 messages out.
#include <string.h>

#include <algorithm>
#include <chrono>
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

// ---- benchmark of the reference-API path (bench.py "host_api"): n el_passo_verify_id proofs as IdProof OBJECTS through
// PSVerifier::el_passo_verify_id_batch, and the same proofs as wire messages through el_passo_verify_id_wire_batch / _wire_packed.
// The proofs arrive as the fixed-stride records the synthetic generator made (sig1 | sig2 | phi | E1 | E2 | k | c | rs[H+2] | m[A-H]); the objects are
// rebuilt from them with the generator's attribute strings "a<i>-<item>" (attributes 0 .. H-1 hidden), so the path under test starts where a caller
// of the reference's API starts: from std::vector<IdProof>.  out[0] = verifier set-up s (tables), out[1] = building the objects s,
// out[2] / out[3] = object path best / median ms, out[4] / out[5] = wire-message path, out[6] / out[7] = packed wire path.
// accepted[0..2] = accepted counts of the three paths; flags = verdicts of the object path.  `ncontexts` contexts on `device` (sharding test) .
static int benchVerifyId(int A, int H, const uint8_t* g, const uint8_t* gg, const uint8_t* XX, const uint8_t* Yi, const uint8_t* YYi,
                         const uint8_t* apk, const uint8_t* g_eg, const uint8_t* h, const char* service, const char* ad, const uint8_t* recs,
                         size_t n, uint64_t first_item, const uint8_t* msgs, const uint32_t* moff, int window_bits, int ncontexts, int device,
                         int reps, double* out, uint64_t* accepted, uint8_t* flags, bool pipelined) {
  return guarded([&] {
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    initPairing(BN254, device >= 0 ? device : 0);
    const size_t S1 = G1::size(), S2 = G2::size();
    PSPubKey pk;
    memcpy(pk.g.b, g, S1);
    memcpy(pk.gg.b, gg, S2);
    memcpy(pk.XX.b, XX, S2);
    pk.Yi.resize(A);
    pk.YYi.resize(A);
    for (int i = 0; i < A; i++) {
      memcpy(pk.Yi[i].b, Yi + S1 * i, S1);
      memcpy(pk.YYi[i].b, YYi + S2 * i, S2);
    }
    G1 Gapk, Gg, Gh;
    memcpy(Gapk.b, apk, S1);
    memcpy(Gg.b, g_eg, S1);
    memcpy(Gh.b, h, S1);
    auto t0 = clk::now();
    // device == -2: one context per VISIBLE GPU (elp_device_count), `ncontexts` of them per GPU -- the multi-GPU dispatch as a node would run it
    std::vector<int> devs;
    if (device == -2) {
      const int nd = elp_device_count();
      if (nd <= 0) throw std::runtime_error("no GPU visible");
      for (int d = 0; d < nd; d++)
        for (int q = 0; q < (ncontexts > 0 ? ncontexts : 1); q++) devs.push_back(d);
    } else {
      devs.assign((size_t)(ncontexts > 0 ? ncontexts : 1), device);
    }
    PSVerifier rp(pk, devs, window_bits);
    // the first call installs the RP parameters (H1(service), authority_pk, g, h and their tables): part of the set-up
    (void)rp.el_passo_verify_id_batch({}, {}, service, Gapk, Gg, Gh);
    out[0] = secs(t0, clk::now());
    t0 = clk::now();
    const size_t rsz = 5 * S1 + S2 + 32 * (size_t)(1 + H + 2 + (A - H));
    std::vector<IdProof> proofs(n);
    elpParallelFor(n, 256, [&](size_t lo, size_t hi) {
      for (size_t j = lo; j < hi; j++) {
        const uint8_t* r = recs + j * rsz;
        IdProof& p = proofs[j];
        memcpy(p.sig1.b, r, S1); r += S1;
        memcpy(p.sig2.b, r, S1); r += S1;
        memcpy(p.phi.b, r, S1); r += S1;
        p.E1.emplace();
        p.E2.emplace();
        memcpy(p.E1->b, r, S1); r += S1;
        memcpy(p.E2->b, r, S1); r += S1;
        memcpy(p.k.b, r, S2); r += S2;
        memcpy(p.c.b, r, 32); r += 32;
        p.rs.resize(H + 2);
        for (int q = 0; q < H + 2; q++) {
          memcpy(p.rs[q].b, r, 32);
          r += 32;
        }
        p.attributes.resize(A);
        for (int i = H; i < A; i++) p.attributes[i] = "a" + std::to_string(i) + "-" + std::to_string(first_item + j);
      }
    });
    std::vector<std::string> ads(n, ad);
    out[1] = secs(t0, clk::now());
    auto timeit = [&](auto&& fn, double& best, double& med) {
      std::vector<double> ts;
      for (int it = 0; it < reps + 1; it++) {
        auto a = clk::now();
        fn();
        if (it) ts.push_back(secs(a, clk::now()) * 1e3);     // the first call is the warm-up
      }
      std::sort(ts.begin(), ts.end());
      best = ts.front();
      med = ts[ts.size() / 2];
    };
    std::vector<bool> verdicts;
    timeit([&] { verdicts = rp.el_passo_verify_id_batch(proofs, ads, service, Gapk, Gg, Gh); }, out[2], out[3]);
    accepted[0] = 0;
    for (size_t j = 0; j < n; j++) {
      flags[j] = verdicts[j] ? 1 : 0;
      accepted[0] += flags[j];
    }
    if (pipelined) {
      // out[8] = ms per batch of a steady stream of such batches through el_passo_verify_id_submit / _collect (two in flight), out[9] = the same for ONE batch
      // submitted and collected at once (the pipelined entry points without pipelining)
      std::vector<bool> v;
      auto stream_of = [&](int nb) {
        auto a = clk::now();
        size_t prev = rp.el_passo_verify_id_submit(proofs, ads, service, Gapk, Gg, Gh);
        for (int it = 1; it < nb; it++) {
          size_t t = rp.el_passo_verify_id_submit(proofs, ads, service, Gapk, Gg, Gh);
          v = rp.el_passo_verify_id_collect(prev);
          prev = t;
        }
        v = rp.el_passo_verify_id_collect(prev);
        return secs(a, clk::now()) * 1e3 / nb;
      };
      (void)stream_of(2);                                   // warm-up: both slots' staging and device buffers
      // a submit that fails half-way (the LAST context refuses its shard after the others were queued) must leave the verifier usable: no ticket, the slot
      // free, the shards that were queued drained -- then two batches in flight work as before (round-4 advisor finding)
      for (int round = 0; round < 2; round++) {
        rp.set_option(ELP_OPT_FAULT_INJECT, 1, (int)rp.contexts() - 1);
        bool threw = false;
        try {
          (void)rp.el_passo_verify_id_submit(proofs, ads, service, Gapk, Gg, Gh);
        } catch (const std::exception&) {
          threw = true;
        }
        if (!threw) throw std::runtime_error("the injected submit failure did not surface");
      }
      {
        const size_t t1 = rp.el_passo_verify_id_submit(proofs, ads, service, Gapk, Gg, Gh);
        const size_t t2 = rp.el_passo_verify_id_submit(proofs, ads, service, Gapk, Gg, Gh);
        const std::vector<bool> v1 = rp.el_passo_verify_id_collect(t1), v2 = rp.el_passo_verify_id_collect(t2);
        for (size_t j = 0; j < n; j++)
          if (v1[j] != verdicts[j] || v2[j] != verdicts[j]) throw std::runtime_error("verdicts after a failed submit differ on item " + std::to_string(j));
      }
      out[8] = stream_of(reps > 1 ? 2 * reps : 4);
      out[9] = stream_of(1);
      for (size_t j = 0; j < n; j++)
        if (v[j] != verdicts[j]) throw std::runtime_error("pipelined path and batch path disagree on item " + std::to_string(j));
    }
    accepted[1] = accepted[2] = 0;
    out[4] = out[5] = out[6] = out[7] = 0;
    if (msgs && moff) {
      std::vector<PSBuffer> wire(n);
      for (size_t j = 0; j < n; j++) wire[j].assign(msgs + moff[j], msgs + moff[j + 1]);
      std::vector<bool> v2;
      timeit([&] { v2 = rp.el_passo_verify_id_wire_batch(wire, ads, service, &Gapk, &Gg, &Gh); }, out[4], out[5]);
      for (size_t j = 0; j < n; j++) accepted[1] += v2[j] ? 1 : 0;
      std::vector<uint8_t> f3(n);
      uint64_t a3 = 0;
      timeit([&] { a3 = rp.el_passo_verify_id_wire_packed(msgs, moff, n, ad, service, f3.data(), &Gapk, &Gg, &Gh); }, out[6], out[7]);
      accepted[2] = a3;
      for (size_t j = 0; j < n; j++)
        if (f3[j] != flags[j] || v2[j] != verdicts[j]) throw std::runtime_error("wire path and object path disagree on item " + std::to_string(j));
    }
    return 0;
  });
}
int elph_bench_verify_id(int A, int H, const uint8_t* g, const uint8_t* gg, const uint8_t* XX, const uint8_t* Yi, const uint8_t* YYi,
                         const uint8_t* apk, const uint8_t* g_eg, const uint8_t* h, const char* service, const char* ad, const uint8_t* recs,
                         size_t n, uint64_t first_item, const uint8_t* msgs, const uint32_t* moff, int window_bits, int ncontexts, int device,
                         int reps, double* out, uint64_t* accepted, uint8_t* flags) {
  return benchVerifyId(A, H, g, gg, XX, Yi, YYi, apk, g_eg, h, service, ad, recs, n, first_item, msgs, moff, window_bits, ncontexts, device, reps, out, accepted, flags, false);
}
// the same + the pipelined object path: `out` has 10 entries (out[8], out[9]: see benchVerifyId)
int elph_bench_verify_id_pipelined(int A, int H, const uint8_t* g, const uint8_t* gg, const uint8_t* XX, const uint8_t* Yi, const uint8_t* YYi,
                                   const uint8_t* apk, const uint8_t* g_eg, const uint8_t* h, const char* service, const char* ad, const uint8_t* recs,
                                   size_t n, uint64_t first_item, const uint8_t* msgs, const uint32_t* moff, int window_bits, int ncontexts, int device,
                                   int reps, double* out, uint64_t* accepted, uint8_t* flags) {
  return benchVerifyId(A, H, g, gg, XX, Yi, YYi, apk, g_eg, h, service, ad, recs, n, first_item, msgs, moff, window_bits, ncontexts, device, reps, out, accepted, flags, true);
}

// process defaults of the contexts the protocol classes create (elpSetDefaults)
void elph_set_defaults(int device, int window_bits) { elpSetDefaults(device, window_bits); }
// the host layer's one-shot SHA-256 (Fr::setHashOf of every revealed attribute goes through it): force_portable = 1 selects the portable rounds, 0 the x86 SHA
// extensions where the CPU has them -- tests compare both with hashlib
void elph_sha256(const uint8_t* msg, size_t n, uint8_t* out32, int force_portable) { cybozu::sha256OneShot(msg, n, out32, force_portable); }
int elph_sha256_has_hardware() { return cybozu::sha256HasHardware() ? 1 : 0; }

}  // extern "C"
