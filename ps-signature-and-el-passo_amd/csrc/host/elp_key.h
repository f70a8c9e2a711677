// Per-object GPU key context shared by PSSigner / PSRequester / PSVerifier: owns one elp_ctx holding the public key's
// fixed-base tables, and (re)installs RP parameters / the signer secret on demand.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "ps-encoding.h"

class ElpKey {
 public:
  explicit ElpKey(const PSPubKey& pk, int device = 0, int window_bits = 0);
  ~ElpKey();
  ElpKey(const ElpKey&) = delete;
  ElpKey& operator=(const ElpKey&) = delete;

  elp_ctx* ctx() const { return ctx_; }
  size_t attrs() const { return nattr_; }
  // base ids of include/elpasso.h
  int idG() const { return 0; }
  int idY(size_t i) const { return 1 + (int)i; }
  int idHs() const { return (int)nattr_ + 1; }
  int idGeg() const { return (int)nattr_ + 2; }
  int idApk() const { return (int)nattr_ + 3; }
  int idH() const { return (int)nattr_ + 4; }
  int idGG() const { return 0; }
  int idXX() const { return 1; }
  int idYY(size_t i) const { return 2 + (int)i; }

  // make sure H1(service), authority_pk, g, h are installed (cached by value)
  void useRp(const std::string& service, const G1* authority_pk, const G1* g, const G1* h);
  void useSignerSecret(const G1& X);

  // sum_t scalars[t] * base[ids[t]] for ONE item
  G1 msmG1(const std::vector<int32_t>& ids, const std::vector<Fr>& scalars) const;
  G2 msmG2(const std::vector<int32_t>& ids, const std::vector<Fr>& scalars) const;

 private:
  elp_ctx* ctx_ = nullptr;
  size_t nattr_ = 0;
  bool rp_set_ = false, sk_set_ = false;
  std::string rp_service_;
  G1 rp_apk_, rp_g_, rp_h_, sk_X_;
};

// Process-wide default of ELP_OPT_STRICT_SIGNATURE for contexts created afterwards (library default: strict, i.e. proofs with
// sig1 == infinity are rejected; false = the reference's behaviour, which accepts sig1 = sig2 = infinity, see include/elpasso.h).
void elpSetStrictSignature(bool strict);

// hidden-attribute mask of a message's attribute list ("" = hidden)
uint64_t elpHiddenMask(const std::vector<std::string>& attributes);
