// User side.  Reference behaviour: src/ps-requester.cc:19-99 (request_id), :101-113 (unblind), :115-137 (verify),
// :139-148 (randomize), :150-310 (prove_id), :312-432 (prove_id without id retrieval).
// Every product  prod base_i^{e_i}  over key bases is ONE fixed-base MSM launch; the draw order of random scalars follows
// the reference so that an injected random source reproduces a reference transcript.
#include "ps-requester.h"

#include <cybozu/sha2.hpp>

#include <string.h>

PSRequester::PSRequester(const PSPubKey& pk) : m_pk(pk), m_key(std::make_shared<ElpKey>(pk, -1, elpDefaultSideWindowBits())) {}

size_t PSRequester::maxAllowedAttrNum() const { return m_pk.Yi.size(); }

void PSRequester::set_random_source(const std::vector<Fr>& values) {
  m_rand = values;
  m_rand_pos = 0;
}
Fr PSRequester::draw() const {
  if (m_rand_pos < m_rand.size()) return m_rand[m_rand_pos++];
  Fr x;
  x.setByCSPRNG();
  return x;
}

static Fr challengeOf(cybozu::Sha256& engine, const std::string& ad) {
  Fr c;
  c.setHashOf(engine.digest(ad));   // SHA-256 applied twice (src/ps-requester.cc:73-74)
  return c;
}
static Fr response(const Fr& rho, const Fr& secret, const Fr& c) {   // rho - secret * c
  Fr t, r;
  Fr::mul(t, secret, c);
  Fr::sub(r, rho, t);
  return r;
}

PSCredRequest PSRequester::el_passo_request_id(const std::vector<std::tuple<std::string, bool>> attributes, const std::string& ad) {
  if (attributes.size() != m_pk.Yi.size()) throw std::runtime_error("attribute size does not match");
  PSCredRequest req;
  m_t1 = draw();
  Fr rho0 = draw();
  std::vector<int32_t> ids{m_key->idG()};
  std::vector<Fr> eA{m_t1}, eV{rho0}, hashes, rhos;
  for (size_t i = 0; i < attributes.size(); i++) {
    if (!std::get<1>(attributes[i])) continue;
    Fr m;
    m.setHashOf(std::get<0>(attributes[i]));
    Fr rho = draw();
    hashes.push_back(m);
    rhos.push_back(rho);
    ids.push_back(m_key->idY(i));
    eA.push_back(m);
    eV.push_back(rho);
  }
  req.A = m_key->msmG1(ids, eA);            // A = g^t prod Y_i^{m_i}
  G1 V = m_key->msmG1(ids, eV);             // V = g^rho0 prod Y_i^{rho_i}
  cybozu::Sha256 engine;
  engine.update(req.A.serializeToHexStr());
  engine.update(V.serializeToHexStr());
  req.c = challengeOf(engine, ad);
  req.rs.push_back(response(rho0, m_t1, req.c));
  for (size_t j = 0; j < hashes.size(); j++) req.rs.push_back(response(rhos[j], hashes[j], req.c));
  for (const auto& a : attributes) req.attributes.push_back(std::get<1>(a) ? std::string() : std::get<0>(a));
  return req;
}

PSCredential PSRequester::unblind_credential(const PSCredential& sig) const {
  PSCredential out;
  out.sig1 = sig.sig1;
  G1 t;
  G1::mul(t, sig.sig1, m_t1);
  G1::sub(out.sig2, sig.sig2, t);           // sig2 / sig1^t
  return out;
}

bool PSRequester::verify(const PSCredential& sig, const std::vector<std::string>& all_attributes) const {
  if (all_attributes.size() > m_key->attrs()) return false;
  std::vector<uint8_t> rec;
  rec.insert(rec.end(), sig.sig1.b, sig.sig1.b + G1::size());
  rec.insert(rec.end(), sig.sig2.b, sig.sig2.b + G1::size());
  for (const std::string& a : all_attributes) {
    Fr m;
    m.setHashOf(a);
    rec.insert(rec.end(), m.b, m.b + 32);
  }
  uint8_t flag = 0;
  uint64_t acc = 0;
  elpCheck(m_key->ctx(), elp_ps_verify_batch(m_key->ctx(), 1, rec.data(), (int)all_attributes.size(), &flag, &acc), "elp_ps_verify_batch");
  return flag != 0;
}

PSCredential PSRequester::randomize_credential(const PSCredential& sig) const {
  Fr t = draw();
  const size_t S1 = G1::size();
  uint8_t pts[192], ks[64], out[192];
  memcpy(pts, sig.sig1.b, S1);
  memcpy(pts + S1, sig.sig2.b, S1);
  memcpy(ks, t.b, 32);
  memcpy(ks + 32, t.b, 32);
  elpCheck(m_key->ctx(), elp_g1_mul(m_key->ctx(), 2, pts, ks, out), "elp_g1_mul");
  PSCredential r;
  memcpy(r.sig1.b, out, S1);
  memcpy(r.sig2.b, out + S1, S1);
  return r;
}

IdProof PSRequester::proveImpl(const PSCredential& sig, const std::vector<std::tuple<std::string, bool>>& attributes,
                               const std::string& ad, const std::string& service, const G1* apk, const G1* g, const G1* h) const {
  if (attributes.size() != m_pk.Yi.size()) throw std::runtime_error("attribute size does not match");
  const bool retr = apk != nullptr;
  m_key->useRp(service, apk, g, h);
  IdProof proof;
  // randomised signature (sig1^r, (sig2 * sig1^t)^r)
  Fr t = draw(), r = draw();
  G1 tmp;
  G1::mul(proof.sig1, sig.sig1, r);
  G1::mul(tmp, sig.sig1, t);
  G1::add(tmp, tmp, sig.sig2);
  G1::mul(proof.sig2, tmp, r);
  // ElGamal token E = (g^eps, y^eps h^gamma), gamma = Hr(attribute 1)
  Fr eps, gamma;
  G1 E1, E2;
  if (retr) {
    eps = draw();
    gamma.setHashOf(std::get<0>(attributes[1]));
    E1 = m_key->msmG1({m_key->idGeg()}, {eps});
    E2 = m_key->msmG1({m_key->idApk(), m_key->idH()}, {eps, gamma});
  }
  // phi = H1(service)^s, s = Hr(attribute 0)
  Fr s;
  s.setHashOf(std::get<0>(attributes[0]));
  proof.phi = m_key->msmG1({m_key->idHs()}, {s});
  // k = XX prod YY_j^{m_j} gg^t ; V_k = XX prod YY_j^{rho_j} gg^{rho_t}
  std::vector<int32_t> ids2{m_key->idXX()};
  std::vector<Fr> ek{Fr::one()}, eVk{Fr::one()}, hashes, rhos;
  for (size_t i = 0; i < attributes.size(); i++) {
    if (!std::get<1>(attributes[i])) continue;
    Fr m;
    m.setHashOf(std::get<0>(attributes[i]));
    hashes.push_back(m);
    ids2.push_back(m_key->idYY(i));
    ek.push_back(m);
  }
  for (size_t j = 0; j < hashes.size(); j++) {
    Fr rho = draw();
    rhos.push_back(rho);
    eVk.push_back(rho);
  }
  ids2.push_back(m_key->idGG());
  ek.push_back(t);
  Fr rho_t = draw();
  eVk.push_back(rho_t);
  proof.k = m_key->msmG2(ids2, ek);
  G2 Vk = m_key->msmG2(ids2, eVk);
  if (rhos.empty()) throw std::runtime_error("attribute 0 must be hidden");
  G1 Vphi = m_key->msmG1({m_key->idHs()}, {rhos[0]});
  Fr rho_e;
  G1 VE1, VE2;
  if (retr) {
    if (rhos.size() < 2) throw std::runtime_error("attributes 0 and 1 must be hidden for id retrieval");
    rho_e = draw();
    VE1 = m_key->msmG1({m_key->idGeg()}, {rho_e});
    VE2 = m_key->msmG1({m_key->idApk(), m_key->idH()}, {rho_e, rhos[1]});
  }
  cybozu::Sha256 engine;
  engine.update(proof.k.serializeToHexStr());
  engine.update(proof.phi.serializeToHexStr());
  if (retr) {
    engine.update(E1.serializeToHexStr());
    engine.update(E2.serializeToHexStr());
  }
  engine.update(Vk.serializeToHexStr());
  engine.update(Vphi.serializeToHexStr());
  if (retr) {
    engine.update(VE1.serializeToHexStr());
    engine.update(VE2.serializeToHexStr());
  }
  proof.c = challengeOf(engine, ad);
  for (size_t j = 0; j < hashes.size(); j++) proof.rs.push_back(response(rhos[j], hashes[j], proof.c));
  proof.rs.push_back(response(rho_t, t, proof.c));
  if (retr) proof.rs.push_back(response(rho_e, eps, proof.c));
  for (const auto& a : attributes) proof.attributes.push_back(std::get<1>(a) ? std::string() : std::get<0>(a));
  if (retr) {
    proof.E1 = E1;
    proof.E2 = E2;
  }
  return proof;
}

IdProof PSRequester::el_passo_prove_id(const PSCredential& sig, const std::vector<std::tuple<std::string, bool>> attributes,
                                       const std::string& associated_data, const std::string& service_name, const G1& authority_pk,
                                       const G1& g, const G1& h) const {
  return proveImpl(sig, attributes, associated_data, service_name, &authority_pk, &g, &h);
}
IdProof PSRequester::el_passo_prove_id_without_id_retrieval(const PSCredential& sig,
                                                            const std::vector<std::tuple<std::string, bool>> attributes,
                                                            const std::string& associated_data, const std::string& service_name) const {
  return proveImpl(sig, attributes, associated_data, service_name, nullptr, nullptr, nullptr);
}

std::vector<IdProof> PSRequester::el_passo_prove_id_batch(const std::vector<PSCredential>& sigs,
                                                          const std::vector<std::vector<std::tuple<std::string, bool>>>& attributes,
                                                          const std::vector<std::string>& ads, const std::string& service,
                                                          const G1* apk, const G1* g, const G1* h) const {
  const size_t n = sigs.size(), A = m_pk.Yi.size();
  if (attributes.size() != n || ads.size() != n) throw std::runtime_error("batch sizes do not match");
  std::vector<IdProof> out(n);
  if (n == 0) return out;
  const bool retr = apk != nullptr;
  m_key->useRp(service, apk, g, h);
  uint64_t mask = 0;
  for (size_t i = 0; i < attributes[0].size() && i < 64; i++)
    if (std::get<1>(attributes[0][i])) mask |= (uint64_t)1 << i;
  const size_t H = (size_t)__builtin_popcountll(mask);
  if (!(mask & 1)) throw std::runtime_error("attribute 0 must be hidden");
  if (retr && !(mask & 2)) throw std::runtime_error("attributes 0 and 1 must be hidden for id retrieval");
  std::vector<uint8_t> recs, adbuf;
  std::vector<uint32_t> adoff(1, 0);
  auto put = [](std::vector<uint8_t>& v, const uint8_t* p, size_t len) { v.insert(v.end(), p, p + len); };
  for (size_t u = 0; u < n; u++) {
    if (attributes[u].size() != A) throw std::runtime_error("attribute size does not match");
    for (size_t i = 0; i < A; i++)
      if (std::get<1>(attributes[u][i]) != (((mask >> i) & 1) != 0)) throw std::runtime_error("hidden pattern differs inside the batch");
    put(recs, sigs[u].sig1.b, G1::size());
    put(recs, sigs[u].sig2.b, G1::size());
    for (size_t i = 0; i < A; i++) {
      Fr m;
      m.setHashOf(std::get<0>(attributes[u][i]));
      put(recs, m.b, 32);
    }
    const size_t ndraw = 2 + (retr ? 1 : 0) + H + 1 + (retr ? 1 : 0);
    for (size_t j = 0; j < ndraw; j++) {
      Fr x = draw();
      put(recs, x.b, 32);
    }
    put(adbuf, (const uint8_t*)ads[u].data(), ads[u].size());
    adoff.push_back((uint32_t)adbuf.size());
  }
  if (adbuf.empty()) adbuf.push_back(0);
  const size_t osz = elp_verify_id_record_size(curveId(), (int)A, (int)H, retr ? 1 : 0);
  std::vector<uint8_t> proofs(n * osz), flags(n);
  uint64_t produced = 0;
  elpCheck(m_key->ctx(),
           elp_prove_id_batch(m_key->ctx(), n, recs.data(), mask, retr ? 1 : 0, adbuf.data(), adoff.data(), 0, proofs.data(), flags.data(),
                              &produced),
           "elp_prove_id_batch");
  for (size_t u = 0; u < n; u++) {
    if (!flags[u]) throw std::runtime_error("credential is not a pair of curve points");
    const uint8_t* p = proofs.data() + u * osz;
    IdProof& pr = out[u];
    auto take = [&p](uint8_t* dst, size_t len) { memcpy(dst, p, len); p += len; };
    take(pr.sig1.b, G1::size());
    take(pr.sig2.b, G1::size());
    take(pr.phi.b, G1::size());
    if (retr) {
      G1 e1, e2;
      take(e1.b, G1::size());
      take(e2.b, G1::size());
      pr.E1 = e1;
      pr.E2 = e2;
    }
    take(pr.k.b, G2::size());
    take(pr.c.b, 32);
    pr.rs.resize(H + (retr ? 2 : 1));
    for (Fr& r : pr.rs) take(r.b, 32);
    for (const auto& a : attributes[u]) pr.attributes.push_back(std::get<1>(a) ? std::string() : std::get<0>(a));
  }
  return out;
}
