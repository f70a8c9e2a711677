// SHA-256 (FIPS 180-4), streaming, host/device.  Replaces cybozu::Sha256 (update/digest) and the hash inside
// Fr::setHashOf / Fp::setHashOf (src/ps-verifier.cc:111-122, src/ps-signer.cc:96-101, src/ps-requester.cc:70-74).
#pragma once
#include "common.h"

namespace elp {

struct Sha256 {
  u32 h[8];
  u32 w[16];   // current block, big-endian words
  u32 fill;    // bytes in block
  u64 total;   // total bytes
};

ELP_HD inline u32 sha_k(int i) {
  constexpr u32 K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
      0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
      0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
      0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
      0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
      0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  return K[i];
}
ELP_INL u32 rotr32(u32 x, int n) { return (x >> n) | (x << (32 - n)); }

ELP_HD inline void sha256_init(Sha256& s) {
  s.h[0] = 0x6a09e667; s.h[1] = 0xbb67ae85; s.h[2] = 0x3c6ef372; s.h[3] = 0xa54ff53a;
  s.h[4] = 0x510e527f; s.h[5] = 0x9b05688c; s.h[6] = 0x1f83d9ab; s.h[7] = 0x5be0cd19;
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.fill = 0;
  s.total = 0;
}
ELP_HD __attribute__((noinline)) inline void sha256_block(Sha256& s) {
  u32 w[16];
  for (int i = 0; i < 16; i++) w[i] = s.w[i];
  u32 a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3], e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
  for (int i = 0; i < 64; i++) {
    u32 wi;
    if (i < 16) {
      wi = w[i];
    } else {
      u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      u32 s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
      u32 s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
      wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
      w[i & 15] = wi;
    }
    u32 S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
    u32 ch = (e & f) ^ (~e & g);
    u32 t1 = h + S1 + ch + sha_k(i) + wi;
    u32 S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
    u32 mj = (a & b) ^ (a & c) ^ (b & c);
    u32 t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d; s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += h;
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.fill = 0;
}
ELP_HD inline void sha256_put(Sha256& s, uint8_t byte) {
  s.w[s.fill >> 2] |= (u32)byte << (24 - 8 * (s.fill & 3));
  s.fill++;
  s.total++;
  if (s.fill == 64) sha256_block(s);
}
ELP_HD inline void sha256_update(Sha256& s, const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; i++) sha256_put(s, p[i]);
}
ELP_HD inline void sha256_final(Sha256& s, uint8_t out[32]) {
  u64 bits = s.total * 8;
  sha256_put(s, 0x80);
  while (s.fill != 56) sha256_put(s, 0);
  s.w[14] = (u32)(bits >> 32);
  s.w[15] = (u32)bits;
  sha256_block(s);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(s.h[i] >> 24);
    out[4 * i + 1] = (uint8_t)(s.h[i] >> 16);
    out[4 * i + 2] = (uint8_t)(s.h[i] >> 8);
    out[4 * i + 3] = (uint8_t)s.h[i];
  }
}
// feed the lowercase-hex rendering of a byte string (mcl serializeToHexStr)
ELP_HD inline void sha256_update_hex(Sha256& s, const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; i++) {
    uint8_t hi = p[i] >> 4, lo = p[i] & 15;
    sha256_put(s, (uint8_t)(hi < 10 ? '0' + hi : 'a' + hi - 10));
    sha256_put(s, (uint8_t)(lo < 10 ? '0' + lo : 'a' + lo - 10));
  }
}

}  // namespace elp
