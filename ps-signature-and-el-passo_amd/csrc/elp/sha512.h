// SHA-512 (FIPS 180-4), streaming, host/device.  The hash inside mcl's Fp::setHashOf when the field is wider than 256 bits, i.e. the first
// step of hashAndMapToG1 on BLS12-381 (src/ps-verifier.cc:94,186, src/ps-requester.cc:185,336).  Off the hot path: evaluated once per
// elp_set_rp / elp_hash_to_g1 item.
#pragma once
#include "common.h"

namespace elp {

struct Sha512 {
  u64 h[8];
  u64 w[16];   // current block, big-endian words
  u32 fill;    // bytes in block
  u64 total;   // total bytes (messages below 2^61 bytes)
};

ELP_HD inline u64 sha512_k(int i) {
  constexpr u64 K[80] = {
      0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull, 0x59f111f1b605d019ull,
      0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull, 0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull,
      0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull, 0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull,
      0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull, 0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull,
      0x983e5152ee66dfabull, 0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
      0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull, 0x53380d139d95b3dfull,
      0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull, 0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull,
      0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull, 0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull,
      0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull, 0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull,
      0x5b9cca4f7763e373ull, 0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
      0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull, 0xd186b8c721c0c207ull,
      0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull, 0x113f9804bef90daeull, 0x1b710b35131c471bull,
      0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull, 0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull,
      0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};
  return K[i];
}
ELP_INL u64 rotr64(u64 x, int n) { return (x >> n) | (x << (64 - n)); }

ELP_HD inline void sha512_init(Sha512& s) {
  s.h[0] = 0x6a09e667f3bcc908ull; s.h[1] = 0xbb67ae8584caa73bull; s.h[2] = 0x3c6ef372fe94f82bull; s.h[3] = 0xa54ff53a5f1d36f1ull;
  s.h[4] = 0x510e527fade682d1ull; s.h[5] = 0x9b05688c2b3e6c1full; s.h[6] = 0x1f83d9abfb41bd6bull; s.h[7] = 0x5be0cd19137e2179ull;
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.fill = 0;
  s.total = 0;
}
ELP_HD __attribute__((noinline)) inline void sha512_block(Sha512& s) {
  u64 w[16];
  for (int i = 0; i < 16; i++) w[i] = s.w[i];
  u64 a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3], e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
  for (int i = 0; i < 80; i++) {
    u64 wi;
    if (i < 16) {
      wi = w[i];
    } else {
      u64 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      u64 s0 = rotr64(w15, 1) ^ rotr64(w15, 8) ^ (w15 >> 7);
      u64 s1 = rotr64(w2, 19) ^ rotr64(w2, 61) ^ (w2 >> 6);
      wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
      w[i & 15] = wi;
    }
    u64 S1 = rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41);
    u64 ch = (e & f) ^ (~e & g);
    u64 t1 = h + S1 + ch + sha512_k(i) + wi;
    u64 S0 = rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39);
    u64 mj = (a & b) ^ (a & c) ^ (b & c);
    u64 t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d; s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += h;
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.fill = 0;
}
ELP_HD inline void sha512_put(Sha512& s, uint8_t byte) {
  s.w[s.fill >> 3] |= (u64)byte << (56 - 8 * (s.fill & 7));
  s.fill++;
  s.total++;
  if (s.fill == 128) sha512_block(s);
}
ELP_HD inline void sha512_update(Sha512& s, const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; i++) sha512_put(s, p[i]);
}
ELP_HD inline void sha512_final(Sha512& s, uint8_t out[64]) {
  u64 bits = s.total * 8;
  sha512_put(s, 0x80);
  while (s.fill != 112) sha512_put(s, 0);
  s.w[14] = 0;           // upper half of the 128-bit length
  s.w[15] = bits;
  sha512_block(s);
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(s.h[i] >> (56 - 8 * j));
}

}  // namespace elp
