// Host-side value types with the slice of herumi/mcl's C++ API that the reference's protocol layer uses
// (SURVEY.md section 8b "lower surface"): namespace mcl::bls12 { Fr, G1, G2, GT, initPairing, pairing,
// hashAndMapToG1/G2 } and cybozu::Sha256.  Every group / pairing operation is executed on the GPU through the C-ABI of
// include/elpasso.h (one-item batches); only scalar (Fr) arithmetic, SHA-256 and byte shuffling run on the host.
// There is no CPU implementation of the curve arithmetic in this layer.
#pragma once
#include <stddef.h>
#include <stdint.h>

// mcl/bls12_381.hpp brings these in transitively and the reference's sources rely on that (std::get at src/ps-requester.cc:54,
// std::cout in test/*.cc, std::vector / std::optional in src/ps-encoding.h)
#include <iostream>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../../include/elpasso.h"

namespace mcl {
namespace bls12 {

// Process-wide default context (created by initPairing(), used by the free functions and the static G1/G2 operations).
elp_ctx* defaultContext();
// Replaces mcl's initPairing(const mcl::CurveParam& = mcl::BN254) (reference: test/ps-tests.cc:142, called without an argument: the reference
// runs on mcl's default curve BN254 although it includes the bls12_381 header).  BLS12_381 selects the north-star curve for the whole
// process: same classes, 48-byte coordinates; its conventions that only mcl could pin (hash-to-curve) are this project's own.
enum CurveParam { BN254 = 0, BLS12_381 = 1 };
void initPairing(CurveParam curve = BN254, int device = 0);
int curveId();            // ELP_CURVE_* of the process (after initPairing)
int defaultDevice();      // the GPU ordinal initPairing() was given
size_t fieldBytes();      // F: 32 (BN254) or 48 (BLS12-381)

struct Fr {
  uint8_t b[32];  // canonical little-endian integer < r
  Fr() { clear(); }
  void clear() {
    for (int i = 0; i < 32; i++) b[i] = 0;
  }
  void setByCSPRNG();
  void setHashOf(const std::string& msg);          // mcl Fr::setHashOf (SHA-256, masked)
  void setInt(uint64_t v);
  static Fr one();
  static void mul(Fr& z, const Fr& x, const Fr& y);
  static void sub(Fr& z, const Fr& x, const Fr& y);
  static void add(Fr& z, const Fr& x, const Fr& y);
  bool isZero() const;
  bool operator==(const Fr& o) const;
  bool operator!=(const Fr& o) const { return !(*this == o); }
  size_t serialize(void* buf, size_t maxSize) const;
  size_t deserialize(const void* buf, size_t size);   // returns 0 (and leaves *this unchanged) if size != 32 or value >= r
  std::string serializeToHexStr() const;
};

struct G1 {
  uint8_t b[96];  // affine x | y (F bytes each, canonical little-endian), zero padded; all-zero = infinity
  G1() { clear(); }
  void clear() {
    for (int i = 0; i < 96; i++) b[i] = 0;
  }
  static size_t size() { return 2 * fieldBytes(); }        // bytes of the affine form handed to the C-ABI
  bool isZero() const;
  bool operator==(const G1& o) const;
  bool operator!=(const G1& o) const { return !(*this == o); }
  static void mul(G1& z, const G1& x, const Fr& k);
  static void add(G1& z, const G1& x, const G1& y);
  static void sub(G1& z, const G1& x, const G1& y);
  static void neg(G1& z, const G1& x);
  size_t serialize(void* buf, size_t maxSize) const;     // F-byte mcl wire form
  size_t deserialize(const void* buf, size_t size);       // point decompression on the GPU; 0 on failure
  std::string serializeToHexStr() const;
  std::string getStr() const;                             // "1 <x> <y>" decimal, "0" for infinity
};

struct G2 {
  uint8_t b[192];  // x.a | x.b | y.a | y.b (F bytes each), zero padded
  G2() { clear(); }
  void clear() {
    for (int i = 0; i < 192; i++) b[i] = 0;
  }
  static size_t size() { return 4 * fieldBytes(); }
  bool isZero() const;
  bool operator==(const G2& o) const;
  bool operator!=(const G2& o) const { return !(*this == o); }
  static void mul(G2& z, const G2& x, const Fr& k);
  static void add(G2& z, const G2& x, const G2& y);
  static void sub(G2& z, const G2& x, const G2& y);
  static void neg(G2& z, const G2& x);
  size_t serialize(void* buf, size_t maxSize) const;     // 2F-byte mcl wire form
  size_t deserialize(const void* buf, size_t size);
  std::string serializeToHexStr() const;
};

struct GT {
  uint8_t b[576];  // 12 F bytes used
  GT() {
    for (int i = 0; i < 576; i++) b[i] = 0;
  }
  bool operator==(const GT& o) const;
  bool operator!=(const GT& o) const { return !(*this == o); }
};

void pairing(GT& e, const G1& P, const G2& Q);
void hashAndMapToG1(G1& P, const std::string& msg);
// hashAndMapToG1: mcl's map on either curve (csrc/elp/encode.h), evaluated on the GPU.
// mcl's hashAndMapToG2 is not reproduced: the reference only uses it to pick a generator gg from a random seed (src/ps-signer.cc:17), which travels inside the
// public key, so no observable depends on it; this one is a deterministic try-and-increment + cofactor clearing, evaluated on the GPU.
void hashAndMapToG2(G2& P, const std::string& msg);

// helpers shared by the protocol classes
void elpCheck(elp_ctx* ctx, int rc, const char* what);
std::string toHex(const uint8_t* p, size_t n);

}  // namespace bls12
}  // namespace mcl

namespace cybozu {
// one-shot SHA-256 (x86 SHA extensions when the CPU has them; force_portable = 1 selects the portable rounds, for the test that compares the two)
void sha256OneShot(const uint8_t* msg, size_t n, uint8_t out[32], int force_portable = 0);
bool sha256HasHardware();
// cybozu::Sha256 as the reference uses it: update(str)*, digest(str) -> 32 raw bytes (src/ps-verifier.cc:111-121)
class Sha256 {
 public:
  Sha256();
  void update(const std::string& s);
  std::string digest(const std::string& s);

 private:
  uint32_t h_[8];
  uint8_t buf_[64];
  uint64_t len_;
  void block(const uint8_t* p);
  void feed(const uint8_t* p, size_t n);
};
}  // namespace cybozu
