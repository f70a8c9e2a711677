// IdP side.  Reference behaviour: src/ps-signer.cc:8-27 (constructors), :29-55 (key_gen), :63-72 (el_passo_provide_id),
// :74-110 (Schnorr NIZK check), :112-130 (sign_hybrid), :132-146 (sign_commitment).
#include "ps-signer.h"

#include <string.h>

#include <map>

PSSigner::PSSigner(size_t attribute_num) : m_attribute_num(attribute_num) {
  // random generators: hash a random scalar's hex string to each group (src/ps-signer.cc:13-17)
  Fr seed;
  seed.setByCSPRNG();
  hashAndMapToG1(m_pk.g, seed.serializeToHexStr());
  seed.setByCSPRNG();
  hashAndMapToG2(m_pk.gg, seed.serializeToHexStr());
}
PSSigner::PSSigner(size_t attribute_num, const G1& g, const G2& gg) : m_attribute_num(attribute_num) {
  m_pk.g = g;
  m_pk.gg = gg;
}

PSPubKey PSSigner::key_gen_from(const Fr& x, const std::vector<Fr>& ys) {
  if (ys.size() != m_attribute_num) throw std::runtime_error("attribute size does not match");
  // one batched variable-base launch per group: {x, y_1..y_A} times the generator
  const size_t n = 1 + m_attribute_num;
  const size_t S1 = G1::size(), S2 = G2::size();
  std::vector<uint8_t> p1(S1 * n), p2(S2 * n), ks(32 * n), o1(S1 * n), o2(S2 * n);
  for (size_t i = 0; i < n; i++) {
    memcpy(&p1[S1 * i], m_pk.g.b, S1);
    memcpy(&p2[S2 * i], m_pk.gg.b, S2);
    memcpy(&ks[32 * i], i == 0 ? x.b : ys[i - 1].b, 32);
  }
  elp_ctx* ctx = defaultContext();
  elpCheck(ctx, elp_g1_mul(ctx, n, p1.data(), ks.data(), o1.data()), "elp_g1_mul");
  elpCheck(ctx, elp_g2_mul(ctx, n, p2.data(), ks.data(), o2.data()), "elp_g2_mul");
  memcpy(m_sk_X.b, &o1[0], S1);
  memcpy(m_pk.XX.b, &o2[0], S2);
  m_pk.Yi.assign(m_attribute_num, G1());
  m_pk.YYi.assign(m_attribute_num, G2());
  for (size_t i = 0; i < m_attribute_num; i++) {
    memcpy(m_pk.Yi[i].b, &o1[S1 * (i + 1)], S1);
    memcpy(m_pk.YYi[i].b, &o2[S2 * (i + 1)], S2);
  }
  installKey();
  return m_pk;
}

PSPubKey PSSigner::key_gen() {
  Fr x;
  x.setByCSPRNG();
  std::vector<Fr> ys(m_attribute_num);
  for (Fr& y : ys) y.setByCSPRNG();
  return key_gen_from(x, ys);
}

void PSSigner::installKey() {
  m_key = std::make_shared<ElpKey>(m_pk, -1, elpDefaultSideWindowBits());
  m_key->useSignerSecret(m_sk_X);
}

PSPubKey PSSigner::get_pub_key() const { return m_pk; }

std::vector<bool> PSSigner::el_passo_provide_id_batch(const std::vector<PSCredRequest>& reqs, const std::vector<std::string>& ads,
                                                      std::vector<PSCredential>& sigs, const std::vector<Fr>* nonces) const {
  if (!m_key) throw std::runtime_error("key_gen() has not been called");
  if (ads.size() != reqs.size() || (nonces && nonces->size() != reqs.size())) throw std::runtime_error("batch size mismatch");
  std::vector<bool> out(reqs.size(), false);
  sigs.resize(reqs.size());
  const size_t A = m_key->attrs();
  std::map<uint64_t, std::vector<size_t>> groups;
  for (size_t i = 0; i < reqs.size(); i++) {
    const PSCredRequest& q = reqs[i];
    if (q.attributes.size() != A) continue;
    uint64_t mask = elpHiddenMask(q.attributes);
    if (q.rs.size() != (size_t)__builtin_popcountll(mask) + 1) continue;
    groups[mask].push_back(i);
  }
  for (auto& [mask, idx] : groups) {
    std::vector<uint8_t> recs, adbuf;
    std::vector<uint32_t> adoff(1, 0);
    for (size_t i : idx) {
      const PSCredRequest& q = reqs[i];
      recs.insert(recs.end(), q.A.b, q.A.b + G1::size());
      recs.insert(recs.end(), q.c.b, q.c.b + 32);
      for (const Fr& r : q.rs) recs.insert(recs.end(), r.b, r.b + 32);
      for (const std::string& a : q.attributes)
        if (!a.empty()) {
          Fr m;
          m.setHashOf(a);                                  // src/ps-signer.cc:125
          recs.insert(recs.end(), m.b, m.b + 32);
        }
      Fr u;
      if (nonces)
        u = (*nonces)[i];
      else
        u.setByCSPRNG();                                   // src/ps-signer.cc:135-136
      recs.insert(recs.end(), u.b, u.b + 32);
      adbuf.insert(adbuf.end(), ads[i].begin(), ads[i].end());
      adoff.push_back((uint32_t)adbuf.size());
    }
    if (adbuf.empty()) adbuf.push_back(0);
    const size_t S1 = G1::size();
    std::vector<uint8_t> flags(idx.size()), out_sigs(2 * S1 * idx.size());
    uint64_t acc = 0;
    elpCheck(m_key->ctx(),
             elp_provide_id_batch(m_key->ctx(), idx.size(), recs.data(), mask, adbuf.data(), adoff.data(), 0, out_sigs.data(), flags.data(), &acc),
             "elp_provide_id_batch");
    for (size_t j = 0; j < idx.size(); j++) {
      if (!flags[j]) continue;
      out[idx[j]] = true;
      memcpy(sigs[idx[j]].sig1.b, &out_sigs[2 * S1 * j], S1);
      memcpy(sigs[idx[j]].sig2.b, &out_sigs[2 * S1 * j + S1], S1);
    }
  }
  return out;
}

bool PSSigner::el_passo_provide_id(const PSCredRequest& request, const std::string& associated_data, PSCredential& sig) const {
  std::vector<PSCredential> sigs;
  bool ok = el_passo_provide_id_batch({request}, {associated_data}, sigs)[0];
  if (ok) sig = sigs[0];        // on failure sig is left untouched (src/ps-signer.cc:67-70)
  return ok;
}

PSCredential PSSigner::sign_commitment(const G1& commitment) const {
  if (!m_key) throw std::runtime_error("key_gen() has not been called");
  Fr u;
  u.setByCSPRNG();
  PSCredential sig;
  sig.sig1 = m_key->msmG1({m_key->idG()}, {u});            // g^u
  G1 t;
  G1::add(t, m_sk_X, commitment);
  G1::mul(sig.sig2, t, u);                                 // (X * commitment)^u
  return sig;
}

PSCredential PSSigner::sign_hybrid(const G1& commitment, const std::vector<std::string>& attributes) const {
  if (!m_key) throw std::runtime_error("key_gen() has not been called");
  if (attributes.size() == 1) return sign_commitment(commitment);     // reference quirk, src/ps-signer.cc:115-117
  std::vector<int32_t> ids;
  std::vector<Fr> ms;
  for (size_t i = 0; i < attributes.size() && i < m_key->attrs(); i++) {
    if (attributes[i].empty()) continue;
    Fr m;
    m.setHashOf(attributes[i]);
    ids.push_back(m_key->idY(i));
    ms.push_back(m);
  }
  G1 full = commitment;
  if (!ids.empty()) G1::add(full, commitment, m_key->msmG1(ids, ms));
  return sign_commitment(full);
}
