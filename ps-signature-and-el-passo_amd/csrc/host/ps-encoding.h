// Wire codec and message types of the protocol layer, GPU-backed build.
//
// The public names and signatures are those of the reference (src/ps-encoding.h:12-220: PSEncodingType, PSBuffer, PSCredential,
// PSPubKey, PSCredRequest, IdProof) so that code written against the reference compiles unchanged; everything behind them
// (ps-encoding.cc) is a new implementation.  Group elements are the GPU-backed mcl::bls12 stand-ins of elp_mcl_compat.h.
#ifndef ELP_HOST_PS_ENCODING_H_
#define ELP_HOST_PS_ENCODING_H_

#include <mcl/bls12_381.hpp>

#include <optional>
#include <string>
#include <vector>

using namespace mcl::bls12;

class PSBuffer;

// ---- protocol messages -------------------------------------------------------------------------------------------------

// IdP public key: generators g (G1), gg (G2), XX = gg^x and per-attribute Y_i = g^y_i, YY_i = gg^y_i.
// Wire: G1 g | G2 gg | G2 XX | G1List Yi | G2List YYi.
class PSPubKey {
 public:
  G1 g;
  G2 gg, XX;
  std::vector<G1> Yi;
  std::vector<G2> YYi;

  static PSPubKey fromBufferString(const PSBuffer& wire);
  PSBuffer toBufferString();
};

// PS signature (sigma_1, sigma_2); the credential of EL PASSO.  Wire: G1 sig1 | G1 sig2.
class PSCredential {
 public:
  G1 sig1, sig2;

  static PSCredential fromBufferString(const PSBuffer& wire);
  PSBuffer toBufferString();
};

// Credential request: commitment A to the hidden attributes + Schnorr proof (c, rs); `attributes` holds the plaintext
// attributes with "" at the positions of committed ones.  Wire: G1 A | Fr c | FrList rs | StrList attributes.
class PSCredRequest {
 public:
  G1 A;
  Fr c;
  std::vector<Fr> rs;
  std::vector<std::string> attributes;

  static PSCredRequest fromBufferString(const PSBuffer& wire);
  PSBuffer toBufferString();
};

// Sign-on proof: randomised signature, k, phi = H1(service)^s, NIZK (c, rs), plaintext attributes ("" = hidden) and the
// optional ElGamal identity-retrieval token (E1, E2).
// Wire: G1 sig1 | G1 sig2 | G2 k | G1 phi | Fr c | FrList rs | StrList attributes [| G1 E1 | G1 E2].
class IdProof {
 public:
  G1 sig1, sig2;
  G2 k;
  G1 phi;
  Fr c;
  std::vector<Fr> rs;
  std::vector<std::string> attributes;
  std::optional<G1> E1, E2;

  static IdProof fromBufferString(const PSBuffer& wire);
  PSBuffer toBufferString();
};

// ---- byte buffer with the T-L-V primitives -------------------------------------------------------------------------------
// element = type(1) | var-length | bytes ;  list = type(1) | var-count | (var-length | bytes)* ;
// var     = one byte below 253, otherwise 0xFD followed by the big-endian 16-bit value.

enum class PSEncodingType : uint8_t { G1 = 1, G2 = 2, Fr = 3, G1List = 4, G2List = 5, FrList = 6, StrList = 7 };

class PSBuffer : public std::vector<uint8_t> {
 public:
  // transport encoding (standard alphabet, '=' padding)
  std::string toBase64();
  static PSBuffer fromBase64(const std::string& base64Str);

  // parse* return the number of bytes consumed at `offset` (0 on a type mismatch); append* grow the buffer
  size_t parseType(size_t offset, PSEncodingType& type) const;
  size_t parseVar(size_t offset, size_t& var) const;
  void appendType(PSEncodingType type);
  void appendVar(size_t var);

  size_t parseFrElement(size_t offset, Fr& f, bool withType = true) const;
  size_t parseG1Element(size_t offset, G1& g, bool withType = true) const;
  size_t parseG2Element(size_t offset, G2& g, bool withType = true) const;
  void appendFrElement(const Fr& f, bool withType = true);
  void appendG1Element(const G1& g, bool withType = true);
  void appendG2Element(const G2& g, bool withType = true);

  size_t parseFrList(size_t offset, std::vector<Fr>& fs) const;
  size_t parseG1List(size_t offset, std::vector<G1>& gs) const;
  size_t parseG2List(size_t offset, std::vector<G2>& gs) const;
  size_t parseStrList(size_t offset, std::vector<std::string>& strs) const;
  void appendFrList(const std::vector<Fr>& fs);
  void appendG1List(const std::vector<G1>& gs);
  void appendG2List(const std::vector<G2>& gs);
  void appendStrList(const std::vector<std::string>& strs);
};

#endif  // ELP_HOST_PS_ENCODING_H_
