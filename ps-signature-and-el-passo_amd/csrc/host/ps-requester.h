// PSRequester (user): same public interface as the reference's src/ps-requester.h:11-128.  Group operations run on the GPU
// as fixed-base multi-scalar multiplications over the key tables plus a few variable-base multiplications.
#ifndef ELP_HOST_PS_REQUESTER_H_
#define ELP_HOST_PS_REQUESTER_H_

#include <memory>
#include <tuple>

#include "elp_key.h"
#include "ps-encoding.h"

class PSRequester {
 public:
  PSRequester(const PSPubKey& pk);
  size_t maxAllowedAttrNum() const;

  PSCredRequest el_passo_request_id(const std::vector<std::tuple<std::string, bool>> attributes, const std::string& associated_data);
  PSCredential unblind_credential(const PSCredential& sig) const;
  bool verify(const PSCredential& sig, const std::vector<std::string>& all_attributes) const;
  PSCredential randomize_credential(const PSCredential& sig) const;

  IdProof el_passo_prove_id(const PSCredential& sig, const std::vector<std::tuple<std::string, bool>> attributes,
                            const std::string& associated_data, const std::string& service_name, const G1& authority_pk,
                            const G1& g, const G1& h) const;
  IdProof el_passo_prove_id_without_id_retrieval(const PSCredential& sig, const std::vector<std::tuple<std::string, bool>> attributes,
                                                 const std::string& associated_data, const std::string& service_name) const;

  // ---- added: the proofs of many users against one RP in one launch (elp_prove_id_batch).  All items share the hidden pattern of
  // `attributes[0]`; randomness is drawn per item in the reference's order (t, r, [eps], rho_j, rho_t, [rho_eps]).
  std::vector<IdProof> el_passo_prove_id_batch(const std::vector<PSCredential>& sigs,
                                               const std::vector<std::vector<std::tuple<std::string, bool>>>& attributes,
                                               const std::vector<std::string>& associated_data, const std::string& service_name,
                                               const G1* authority_pk, const G1* g, const G1* h) const;

  // ---- RNG seam for reproducible tests: when set, random scalars are taken from this list (in draw order) instead of the CSPRNG
  void set_random_source(const std::vector<Fr>& values);

 private:
  IdProof proveImpl(const PSCredential& sig, const std::vector<std::tuple<std::string, bool>>& attributes, const std::string& ad,
                    const std::string& service, const G1* authority_pk, const G1* g, const G1* h) const;
  Fr draw() const;
  PSPubKey m_pk;
  std::shared_ptr<ElpKey> m_key;
  Fr m_t1;   // blinding of the last request (needed by unblind_credential)
  mutable std::vector<Fr> m_rand;
  mutable size_t m_rand_pos = 0;
};

#endif  // ELP_HOST_PS_REQUESTER_H_
