// mcl-compatible encodings and hashes (conventions pinned by tests/golden/bn254_*.json):
//   Fr/Fp : FBYTES little-endian;  G1: x LE, top bit of last byte = y odd, all-zero = infinity;
//   G2: x.a LE || x.b LE, top bit of last byte = y.a odd;  serializeToHexStr = lowercase hex of those bytes;
//   Fr::setHashOf(m) = LE(SHA-256(m)) masked to bitlen(r) bits, top bit cleared if still >= r.
// Replaces G1/G2::serialize/deserialize/serializeToHexStr, Fr::setHashOf, hashAndMapToG1
// (src/ps-verifier.cc:25,94,112-122,224; src/ps-encoding.cc:167,192,199,224,231,256).
#pragma once
#include "curve.h"
#include "sha256.h"
#include "sha512.h"

namespace elp {

// ---- Fr helpers (plain integers mod r, 8 limbs, no Montgomery form: only add/sub/compare are needed on device)
template <class C>
ELP_INL bool scalar_geq_r(const Scalar& a) {
  u64 br = 0;
  for (int i = 0; i < 8; i++) {
    u64 t = (u64)a.v[i] - C::rmod(i) - br;
    br = (t >> 32) & 1;
  }
  return br == 0;
}
template <class C>
ELP_INL Scalar scalar_sub_mod_r(const Scalar& a, const Scalar& b) {  // a, b < r
  Scalar r;
  u64 br = 0;
  for (int i = 0; i < 8; i++) {
    u64 t = (u64)a.v[i] - b.v[i] - br;
    r.v[i] = (u32)t;
    br = (t >> 32) & 1;
  }
  if (br) {
    u64 c = 0;
    for (int i = 0; i < 8; i++) {
      u64 t = (u64)r.v[i] + C::rmod(i) + c;
      r.v[i] = (u32)t;
      c = t >> 32;
    }
  }
  return r;
}
ELP_INL bool scalar_eq(const Scalar& a, const Scalar& b) {
  u32 t = 0;
  for (int i = 0; i < 8; i++) t |= a.v[i] ^ b.v[i];
  return t == 0;
}
ELP_INL Scalar scalar_load_le(const uint8_t* b) {
  Scalar r;
  for (int i = 0; i < 8; i++)
    r.v[i] = (u32)b[4 * i] | ((u32)b[4 * i + 1] << 8) | ((u32)b[4 * i + 2] << 16) | ((u32)b[4 * i + 3] << 24);
  return r;
}
// mask a 32-byte digest (LE integer) the way mcl's setHashOf/setArrayMask does for a modulus of `bits` bits
template <class C>
ELP_INL Scalar scalar_from_digest(const uint8_t d[32]) {
  Scalar s = scalar_load_le(d);
  const int bits = C::RBITS;
  if (bits < 256) s.v[7] &= (u32)((1ull << (bits - 224)) - 1);
  if (scalar_geq_r<C>(s)) s.v[7] &= (u32)((1ull << (bits - 1 - 224)) - 1);
  return s;
}

// ---- point serialisation (inputs affine, Montgomery form)
template <class C>
ELP_HEAVY void g1_serialize(uint8_t* out, const Aff<F1<C>>& p) {
  if (aff_is_inf(p)) {
    for (int i = 0; i < C::FBYTES; i++) out[i] = 0;
    return;
  }
  StdFp<C> x = fp_to_std<C>(p.x), y = fp_to_std<C>(p.y);
  std_store_le<C>(out, x);
  if (y.w[0] & 1) out[C::FBYTES - 1] |= 0x80;
}
template <class C>
ELP_HEAVY void g2_serialize(uint8_t* out, const Aff<F2<C>>& p) {
  if (aff_is_inf(p)) {
    for (int i = 0; i < 2 * C::FBYTES; i++) out[i] = 0;
    return;
  }
  if constexpr (is_paired<C>()) {
    // each lane canonicalises its own components; the words of the partner's x and the parity of y.a come over by lane exchange, so
    // both lanes of the pair emit the same bytes
    const bool odd = pair_odd();
    StdFp<C> xs = fp_to_std<C>(p.x.c), ys = fp_to_std<C>(p.y.c), xo;
    for (int i = 0; i < C::N; i++) xo.w[i] = (u32)pair_swap_i32((int32_t)xs.w[i]);
    const u32 yo = (u32)pair_swap_i32((int32_t)ys.w[0]);
    std_store_le<C>(out, odd ? xo : xs);
    std_store_le<C>(out + C::FBYTES, odd ? xs : xo);
    if ((odd ? yo : ys.w[0]) & 1) out[2 * C::FBYTES - 1] |= 0x80;
  } else {
    StdFp<C> xa = fp_to_std<C>(p.x.c0), xb = fp_to_std<C>(p.x.c1), ya = fp_to_std<C>(p.y.c0);
    std_store_le<C>(out, xa);
    std_store_le<C>(out + C::FBYTES, xb);
    if (ya.w[0] & 1) out[2 * C::FBYTES - 1] |= 0x80;
  }
}
// the same wire bytes straight from canonical affine words (x | y for G1, x.a | x.b | y.a | y.b for G2): no field arithmetic.  Used by
// the paired kernels for the transcript parts that are inputs (k, phi, E1, E2), which every lane can read for itself.
template <class C>
ELP_INL void g1_serialize_std(uint8_t* out, const u32* w) {
  u32 any = 0;
  for (int i = 0; i < C::N; i++) {
    const u32 x = w[i];
    any |= x | w[C::N + i];
    out[4 * i] = (uint8_t)x;
    out[4 * i + 1] = (uint8_t)(x >> 8);
    out[4 * i + 2] = (uint8_t)(x >> 16);
    out[4 * i + 3] = (uint8_t)(x >> 24);
  }
  if (any && (w[C::N] & 1)) out[C::FBYTES - 1] |= 0x80;
}
template <class C>
ELP_INL void g2_serialize_std(uint8_t* out, const u32* w) {
  u32 any = 0;
  for (int i = 0; i < 2 * C::N; i++) {
    const u32 x = w[i];
    any |= x | w[2 * C::N + i];
    out[4 * i] = (uint8_t)x;
    out[4 * i + 1] = (uint8_t)(x >> 8);
    out[4 * i + 2] = (uint8_t)(x >> 16);
    out[4 * i + 3] = (uint8_t)(x >> 24);
  }
  if (any && (w[2 * C::N] & 1)) out[2 * C::FBYTES - 1] |= 0x80;
}
// Decompression; returns false for x >= p or x not on the curve.  (mcl ignores such failures at
// src/ps-encoding.cc:192,224; we surface them and let the caller reject the item.)
template <class C>
ELP_HEAVY bool g1_deserialize(Aff<F1<C>>& p, const uint8_t* in) {
  uint8_t tmp[C::FBYTES];
  u32 any = 0;
  for (int i = 0; i < C::FBYTES; i++) {
    tmp[i] = in[i];
    any |= in[i];
  }
  if (!any) {
    aff_set_inf(p);
    return true;
  }
  bool odd = (tmp[C::FBYTES - 1] & 0x80) != 0;
  tmp[C::FBYTES - 1] &= 0x7f;
  StdFp<C> xs = std_load_le<C>(tmp);
  if (!std_in_range<C>(xs)) return false;
  Fp<C> x = fp_from_std<C>(xs);
  Fp<C> b;
  ELP_LOAD_FP(b, C::curve_b(i_));
  Fp<C> rhs = fp_add(fp_mul<C>(fp_sqr<C>(x), x), b);
  Fp<C> y;
  if (!fp_sqrt<C>(y, rhs)) return false;
  StdFp<C> ys = fp_to_std<C>(y);
  if (((ys.w[0] & 1) != 0) != odd) y = fp_neg(y);
  p.x = x;
  p.y = y;
  return true;
}
// `canon_flag` (optional) receives the flag bit of the point's CANONICAL encoding: mcl's decoder (and this one) accepts a set flag on a
// point whose y.a is 0 (negation leaves y.a = 0), while re-serialising such a point gives flag 0 -- the reference hashes the re-serialised
// bytes into the transcript (src/ps-verifier.cc:113), so a caller that hashes the message's own bytes must canonicalise the flag.
template <class C>
ELP_HEAVY bool g2_deserialize(Aff<F2<C>>& p, const uint8_t* in, bool* canon_flag = nullptr) {
  if (canon_flag) *canon_flag = false;
  uint8_t tmp[2 * C::FBYTES];
  u32 any = 0;
  for (int i = 0; i < 2 * C::FBYTES; i++) {
    tmp[i] = in[i];
    any |= in[i];
  }
  if (!any) {
    aff_set_inf(p);
    return true;
  }
  bool odd = (tmp[2 * C::FBYTES - 1] & 0x80) != 0;
  tmp[2 * C::FBYTES - 1] &= 0x7f;
  StdFp<C> xa = std_load_le<C>(tmp), xb = std_load_le<C>(tmp + C::FBYTES);
  if (!std_in_range<C>(xa) || !std_in_range<C>(xb)) return false;
  Fp2<C> x = fp2_scatter<C>(fp_from_std<C>(xa), fp_from_std<C>(xb));
  Fp2<C> rhs = fp2_add(fp2_mulv<C>(fp2_sqrv<C>(x), x), F2<C>::curve_b());
  Fp2<C> y;
  if (!fp2_sqrt<C>(y, rhs)) return false;
  StdFp<C> ya = fp_to_std<C>(fp2_gather<C>(y).c0);
  if (((ya.w[0] & 1) != 0) != odd) y = fp2_neg(y);
  if (canon_flag) *canon_flag = odd && !std_is_zero<C>(ya);     // y.a == 0: both roots have an even (zero) y.a
  p.x = x;
  p.y = y;
  return true;
}

// ---- hashAndMapToG1 (src/ps-verifier.cc:94,186, src/ps-requester.cc:185,336), as mcl evaluates it on either curve:
//   t = Fp::setHashOf(msg)   SHA-256 (BN254) / SHA-512 (BLS12-381: mcl hashes with SHA-512 when the modulus is wider than 256 bits); the first
//                            FBYTES bytes as a little-endian integer, masked to bitlen(p) bits, one more bit cleared if still >= p
//   P = SvdW(t)              Shallue-van de Woestijne map for y^2 = x^3 + b (mcl MapTo::calcBN; mcl's default, non-ETH mode uses it for
//                            BLS12-381 too): c1 = sqrt(-3) = (-3)^((p+1)/4), c2 = (c1 - 1)/2, y negated iff Legendre(t) = -1
//   BLS12-381: P <- [(z-1)^2/3] P   (cofactor clearing; the cofactor of BN254's E(Fp) is 1)
// Pinned on BOTH curves by proofs made by the reference's wasm (tests/golden/{bn254,bls12_381}_oracle_flows.json "hash_to_g1": 32 service
// names, all six branch / sign cases; the BLS12-381 run of the wasm: oracle/wasm_curve.js).
template <class C>
ELP_HEAVY void map_to_g1_svdw(Aff<F1<C>>& out, const Fp<C>& t) {  // t != 0, Montgomery form
  Fp<C> c1, c2, b, one = fp_one<C>();
  ELP_LOAD_FP(c1, C::svdw_c1(i_));
  ELP_LOAD_FP(c2, C::svdw_c2(i_));
  ELP_LOAD_FP(b, C::curve_b(i_));
  bool neg = fp_legendre<C>(t) < 0;
  Fp<C> w = fp_add(fp_add(fp_sqr<C>(t), b), one);
  w = fp_mul<C>(fp_mul<C>(c1, t), fp_inv<C>(w));
  Fp<C> x, y;
  for (int i = 0; i < 3; i++) {
    if (i == 0)
      x = fp_sub(c2, fp_mul<C>(t, w));
    else if (i == 1)
      x = fp_sub(fp_neg(x), one);
    else
      x = fp_add(one, fp_inv<C>(fp_sqr<C>(w)));
    Fp<C> rhs = fp_add(fp_mul<C>(fp_sqr<C>(x), x), b);
    if (fp_sqrt<C>(y, rhs)) break;
  }
  if (neg) y = fp_neg(y);
  out.x = x;
  out.y = y;
}
// digest -> canonical integer below p, mcl setHashOf style: take FBYTES little-endian bytes, mask to bitlen(p) bits, clear one
// more bit if still >= p.
template <class C>
ELP_INL StdFp<C> std_from_hash_bytes(const uint8_t* d) {
  StdFp<C> t = std_load_le<C>(d);
  const int bits = C::PBITS;
  if (bits < 32 * C::N) t.w[C::N - 1] &= (u32)((1ull << (bits - 32 * (C::N - 1))) - 1);
  if (!std_in_range<C>(t)) t.w[C::N - 1] &= (u32)((1ull << (bits - 1 - 32 * (C::N - 1))) - 1);
  return t;
}
template <class C>
ELP_HEAVY void hash_and_map_to_g1(Aff<F1<C>>& out, const uint8_t* msg, size_t len) {
  if constexpr (C::PBITS <= 256) {
    Sha256 s;
    sha256_init(s);
    sha256_update(s, msg, len);
    uint8_t d[32];
    sha256_final(s, d);
    map_to_g1_svdw<C>(out, fp_from_std<C>(std_from_hash_bytes<C>(d)));   // BN254: FBYTES == 32
  } else {
    Sha512 s;
    sha512_init(s);
    sha512_update(s, msg, len);
    uint8_t d[64];
    sha512_final(s, d);
    Aff<F1<C>> p;
    map_to_g1_svdw<C>(p, fp_from_std<C>(std_from_hash_bytes<C>(d)));     // BLS12-381: the first 48 of the 64 digest bytes
    Scalar h;
    for (int i = 0; i < 8; i++) h.v[i] = C::g1_cofactor(i);
    Jac<F1<C>> j;
    jac_mul_var<F1<C>>(j, p, h, 32);                                     // plain 4-bit windows over the 126-bit cofactor: p is not in G1 yet
    jac_to_aff<F1<C>>(out, j);
  }
}

}  // namespace elp
