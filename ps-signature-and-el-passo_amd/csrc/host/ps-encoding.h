// Value types and wire codec of the protocol layer.  Public surface identical to the reference's
// src/ps-encoding.h:12-220 (PSEncodingType, PSBuffer, PSCredential, PSPubKey, PSCredRequest, IdProof), so callers of
// the reference compile against this header unchanged; the implementation (ps-encoding.cc) is new.
#ifndef ELP_HOST_PS_ENCODING_H_
#define ELP_HOST_PS_ENCODING_H_

#include <mcl/bls12_381.hpp>

#include <optional>
#include <string>
#include <vector>

using namespace mcl::bls12;

enum class PSEncodingType : uint8_t { G1 = 1, G2 = 2, Fr = 3, G1List = 4, G2List = 5, FrList = 6, StrList = 7 };

// Byte buffer with T-L-V append/parse helpers and base64 transport encoding.
// Layout: element = type(1) | var-length | bytes ; list = type(1) | var-count | (var-length | bytes)* ;
// var = 1 byte below 253, else 0xFD hi lo.
class PSBuffer : public std::vector<uint8_t> {
 public:
  static PSBuffer fromBase64(const std::string& base64Str);
  std::string toBase64();

  void appendType(PSEncodingType type);
  size_t parseType(size_t offset, PSEncodingType& type) const;
  void appendVar(size_t var);
  size_t parseVar(size_t offset, size_t& var) const;

  void appendG1Element(const G1& g, bool withType = true);
  size_t parseG1Element(size_t offset, G1& g, bool withType = true) const;
  void appendG2Element(const G2& g, bool withType = true);
  size_t parseG2Element(size_t offset, G2& g, bool withType = true) const;
  void appendFrElement(const Fr& f, bool withType = true);
  size_t parseFrElement(size_t offset, Fr& f, bool withType = true) const;

  void appendG1List(const std::vector<G1>& gs);
  size_t parseG1List(size_t offset, std::vector<G1>& gs) const;
  void appendG2List(const std::vector<G2>& gs);
  size_t parseG2List(size_t offset, std::vector<G2>& gs) const;
  void appendFrList(const std::vector<Fr>& fs);
  size_t parseFrList(size_t offset, std::vector<Fr>& fs) const;
  void appendStrList(const std::vector<std::string>& strs);
  size_t parseStrList(size_t offset, std::vector<std::string>& strs) const;
};

class PSCredential {
 public:
  G1 sig1, sig2;
  PSBuffer toBufferString();
  static PSCredential fromBufferString(const PSBuffer& buf);
};

class PSPubKey {
 public:
  G1 g;
  G2 gg;
  G2 XX;
  std::vector<G1> Yi;
  std::vector<G2> YYi;
  PSBuffer toBufferString();
  static PSPubKey fromBufferString(const PSBuffer& buf);
};

class PSCredRequest {
 public:
  G1 A;
  Fr c;
  std::vector<Fr> rs;
  std::vector<std::string> attributes;   // "" marks a committed (hidden) attribute
  PSBuffer toBufferString();
  static PSCredRequest fromBufferString(const PSBuffer& buf);
};

class IdProof {
 public:
  G1 sig1, sig2;
  G2 k;
  G1 phi;
  Fr c;
  std::vector<Fr> rs;
  std::vector<std::string> attributes;
  std::optional<G1> E1, E2;
  PSBuffer toBufferString();
  static IdProof fromBufferString(const PSBuffer& buf);
};

#endif  // ELP_HOST_PS_ENCODING_H_
