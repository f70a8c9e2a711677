// PSSigner (IdP): same public interface as the reference's src/ps-signer.h:11-96; issuance runs on the GPU
// (elp_provide_id_batch), plus a batch entry point and nonce injection for reproducible tests.
#ifndef ELP_HOST_PS_SIGNER_H_
#define ELP_HOST_PS_SIGNER_H_

#include <memory>

#include "elp_key.h"
#include "ps-encoding.h"

class PSSigner {
 public:
  PSSigner(size_t attribute_num);
  PSSigner(size_t attribute_num, const G1& g, const G2& gg);

  PSPubKey key_gen();
  PSPubKey get_pub_key() const;

  bool el_passo_provide_id(const PSCredRequest& request, const std::string& associated_data, PSCredential& sig) const;
  PSCredential sign_commitment(const G1& commitment) const;
  PSCredential sign_hybrid(const G1& commitment, const std::vector<std::string>& attributes) const;

  // ---- new: batched issuance.  sigs[i] is written only where the returned flag is true.  `nonces` (optional) replaces the
  // CSPRNG draw of sign_commitment (src/ps-signer.cc:135-136) so that outputs are reproducible.
  std::vector<bool> el_passo_provide_id_batch(const std::vector<PSCredRequest>& requests, const std::vector<std::string>& associated_data,
                                              std::vector<PSCredential>& sigs, const std::vector<Fr>* nonces = nullptr) const;
  // deterministic key generation from caller-supplied secrets (x, y_i) -- the RNG seam for parity tests
  PSPubKey key_gen_from(const Fr& x, const std::vector<Fr>& ys);

 private:
  void installKey();
  size_t m_attribute_num;
  G1 m_sk_X;
  PSPubKey m_pk;
  std::shared_ptr<ElpKey> m_key;
};

#endif  // ELP_HOST_PS_SIGNER_H_
