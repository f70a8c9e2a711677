#include "elp_mcl_compat.h"

#include <string.h>

#include <random>

namespace mcl {
namespace bls12 {

typedef unsigned __int128 u128;
static elp_ctx* g_ctx = nullptr;
static int g_device = 0;

// per-curve constants, little-endian 64-bit limbs: group order r (4 limbs) and base-field prime p (6 limbs, zero-extended)
struct CurveConst {
  int id;
  size_t F;          // field bytes
  int rbits;         // bit length of r
  int pbits;         // bit length of p
  uint64_t r[4];
  uint64_t p[6];
};
static const CurveConst kBN254 = {ELP_CURVE_BN254, 32, 254, 254,
                                  {0xa10000000000000dull, 0xff9f800000000010ull, 0xba344d8000000007ull, 0x2523648240000001ull},
                                  {0xa700000000000013ull, 0x6121000000000013ull, 0xba344d8000000008ull, 0x2523648240000001ull, 0, 0}};
static const CurveConst kBLS12_381 = {ELP_CURVE_BLS12_381, 48, 255, 381,
                                      {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull},
                                      {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull,
                                       0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull}};
static const CurveConst* g_cv = &kBN254;
#define R_ (g_cv->r)

void elpCheck(elp_ctx* ctx, int rc, const char* what) {
  if (rc != ELP_OK) throw std::runtime_error(std::string(what) + " failed: " + (ctx ? elp_last_error(ctx) : "no context") +
                                             " (code " + std::to_string(rc) + ")");
}
elp_ctx* defaultContext() {
  if (!g_ctx) throw std::runtime_error("initPairing() has not been called");
  return g_ctx;
}
int curveId() { return g_cv->id; }
int defaultDevice() { return g_device; }
size_t fieldBytes() { return g_cv->F; }
void initPairing(CurveParam curve, int device) {
  const CurveConst* want = curve == BLS12_381 ? &kBLS12_381 : &kBN254;
  if (g_ctx) {
    if (want != g_cv) throw std::runtime_error("initPairing: the process already runs on the other curve");
    return;
  }
  g_cv = want;
  g_device = device;
  int rc = elp_init(g_cv->id, device, &g_ctx);
  if (rc != ELP_OK) throw std::runtime_error("elp_init failed (" + std::to_string(rc) + "): a GPU is required, there is no CPU fallback");
}
std::string toHex(const uint8_t* p, size_t n) {
  static const char hx[] = "0123456789abcdef";
  std::string s(2 * n, '0');
  for (size_t i = 0; i < n; i++) {
    s[2 * i] = hx[p[i] >> 4];
    s[2 * i + 1] = hx[p[i] & 15];
  }
  return s;
}

// ---- 256-bit helpers (scalars)
static void ld(uint64_t v[4], const uint8_t* b) { memcpy(v, b, 32); }
static void st(uint8_t* b, const uint64_t v[4]) { memcpy(b, v, 32); }
static int cmp4(const uint64_t* a, const uint64_t* m) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > m[i]) return 1;
    if (a[i] < m[i]) return -1;
  }
  return 0;
}
static uint64_t sub4(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 t = (u128)a[i] - b[i] - br;
    r[i] = (uint64_t)t;
    br = (t >> 64) & 1;
  }
  return (uint64_t)br;
}
static uint64_t add4(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a[i] + b[i];
    r[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
static void maskScalar(uint64_t v[4]) {   // mcl setArrayMask: keep bitlen(r) bits, drop one more if still >= r
  const int top = g_cv->rbits - 192;      // bits kept in the top limb (62 or 63)
  v[3] &= (1ull << top) - 1;
  if (cmp4(v, R_) >= 0) v[3] &= (1ull << (top - 1)) - 1;
}

// ---- Fr
void Fr::setInt(uint64_t v) {
  clear();
  memcpy(b, &v, 8);
}
Fr Fr::one() {
  Fr x;
  x.setInt(1);
  return x;
}
void Fr::setByCSPRNG() {
  static std::random_device rd;
  uint64_t v[4];
  do {
    for (int i = 0; i < 4; i++) v[i] = ((uint64_t)rd() << 32) | rd();
    v[3] &= (1ull << (g_cv->rbits - 192)) - 1;
  } while (cmp4(v, R_) >= 0);
  st(b, v);
}
void Fr::setHashOf(const std::string& msg) {
  // one-shot SHA-256 without the streaming object (no per-byte feeding, no heap string for the digest): this runs once per revealed attribute of every proof
  // that PSVerifier::el_passo_verify_id_batch packs (src/ps-verifier.cc:224) -- 262 144 times for the headline batch
  uint8_t d[32];
  cybozu::sha256OneShot((const uint8_t*)msg.data(), msg.size(), d);
  uint64_t v[4];
  memcpy(v, d, 32);
  maskScalar(v);
  st(b, v);
}
void Fr::add(Fr& z, const Fr& x, const Fr& y) {
  uint64_t a[4], c[4], r[4];
  ld(a, x.b);
  ld(c, y.b);
  add4(r, a, c);  // < 2r < 2^256: no carry out
  if (cmp4(r, R_) >= 0) sub4(r, r, R_);
  st(z.b, r);
}
void Fr::sub(Fr& z, const Fr& x, const Fr& y) {
  uint64_t a[4], c[4], r[4];
  ld(a, x.b);
  ld(c, y.b);
  if (sub4(r, a, c)) add4(r, r, R_);
  st(z.b, r);
}
void Fr::mul(Fr& z, const Fr& x, const Fr& y) {
  uint64_t a[4], c[4], t[8] = {0};
  ld(a, x.b);
  ld(c, y.b);
  for (int i = 0; i < 4; i++) {
    u128 cy = 0;
    for (int j = 0; j < 4; j++) {
      cy += (u128)a[j] * c[i] + t[i + j];
      t[i + j] = (uint64_t)cy;
      cy >>= 64;
    }
    t[i + 4] = (uint64_t)cy;
  }
  // binary long division of the 512-bit product by r (host-side scalars only; not a hot path)
  uint64_t rem[4] = {0, 0, 0, 0};
  for (int bit = 511; bit >= 0; bit--) {
    uint64_t top = rem[3] >> 63;
    for (int i = 3; i > 0; i--) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
    rem[0] = (rem[0] << 1) | ((t[bit >> 6] >> (bit & 63)) & 1);
    if (top || cmp4(rem, R_) >= 0) sub4(rem, rem, R_);
  }
  st(z.b, rem);
}
bool Fr::isZero() const {
  uint8_t t = 0;
  for (int i = 0; i < 32; i++) t |= b[i];
  return t == 0;
}
bool Fr::operator==(const Fr& o) const { return memcmp(b, o.b, 32) == 0; }
size_t Fr::serialize(void* buf, size_t maxSize) const {
  if (maxSize < 32) return 0;
  memcpy(buf, b, 32);
  return 32;
}
size_t Fr::deserialize(const void* buf, size_t size) {
  if (size != 32) return 0;
  uint64_t v[4];
  memcpy(v, buf, 32);
  if (cmp4(v, R_) >= 0) return 0;
  st(b, v);
  return 32;
}
std::string Fr::serializeToHexStr() const { return toHex(b, 32); }

// ---- G1 / G2 (all group arithmetic on the GPU; coordinates are F = fieldBytes() bytes each)
static bool allZero(const uint8_t* p, size_t n) {
  uint8_t t = 0;
  for (size_t i = 0; i < n; i++) t |= p[i];
  return t == 0;
}
static void negCoord(uint8_t* out, const uint8_t* in) {  // p - y (y != 0), or 0; F-byte coordinates
  const size_t F = g_cv->F, L = F / 8;
  uint64_t y[6] = {0, 0, 0, 0, 0, 0}, r[6];
  memcpy(y, in, F);
  if (allZero(in, F)) {
    memcpy(out, in, F);
    return;
  }
  u128 br = 0;
  for (size_t i = 0; i < L; i++) {
    u128 t = (u128)g_cv->p[i] - y[i] - br;
    r[i] = (uint64_t)t;
    br = (t >> 64) & 1;
  }
  memcpy(out, r, F);
}
bool G1::isZero() const { return allZero(b, size()); }
bool G1::operator==(const G1& o) const { return memcmp(b, o.b, size()) == 0; }
void G1::mul(G1& z, const G1& x, const Fr& k) {
  G1 r;
  elpCheck(defaultContext(), elp_g1_mul(defaultContext(), 1, x.b, k.b, r.b), "elp_g1_mul");
  z = r;
}
void G1::add(G1& z, const G1& x, const G1& y) {
  G1 r;
  elpCheck(defaultContext(), elp_g1_add(defaultContext(), 1, x.b, y.b, r.b), "elp_g1_add");
  z = r;
}
void G1::neg(G1& z, const G1& x) {
  G1 r = x;
  negCoord(r.b + g_cv->F, x.b + g_cv->F);
  z = r;
}
void G1::sub(G1& z, const G1& x, const G1& y) {
  G1 n;
  neg(n, y);
  add(z, x, n);
}
size_t G1::serialize(void* buf, size_t maxSize) const {
  const size_t F = g_cv->F;
  if (maxSize < F) return 0;
  uint8_t* o = (uint8_t*)buf;
  memcpy(o, b, F);
  if (b[F] & 1) o[F - 1] |= 0x80;
  return F;
}
size_t G1::deserialize(const void* buf, size_t size) {
  if (size != g_cv->F) return 0;
  uint8_t ok = 0;
  G1 r;
  elpCheck(defaultContext(), elp_g1_decompress(defaultContext(), 1, (const uint8_t*)buf, r.b, &ok), "elp_g1_decompress");
  if (!ok) return 0;
  *this = r;
  return size;
}
std::string G1::serializeToHexStr() const {
  uint8_t w[48];
  size_t n = serialize(w, sizeof w);
  return toHex(w, n);
}
static std::string toDecimal(const uint8_t* le, size_t nbytes) {
  uint64_t v[6] = {0, 0, 0, 0, 0, 0};
  memcpy(v, le, nbytes);
  const int L = (int)(nbytes / 8);
  auto nz = [&]() {
    uint64_t t = 0;
    for (int i = 0; i < L; i++) t |= v[i];
    return t != 0;
  };
  std::string out;
  const uint64_t base = 1000000000000000000ull;  // 10^18
  while (nz()) {
    u128 rem = 0;
    for (int i = L - 1; i >= 0; i--) {
      u128 cur = (rem << 64) | v[i];
      v[i] = (uint64_t)(cur / base);
      rem = cur % base;
    }
    std::string chunk = std::to_string((uint64_t)rem);
    if (nz()) chunk = std::string(18 - chunk.size(), '0') + chunk;
    out = chunk + out;
  }
  return out.empty() ? "0" : out;
}
std::string G1::getStr() const {
  if (isZero()) return "0";
  const size_t F = g_cv->F;
  return "1 " + toDecimal(b, F) + " " + toDecimal(b + F, F);
}

bool G2::isZero() const { return allZero(b, size()); }
bool G2::operator==(const G2& o) const { return memcmp(b, o.b, size()) == 0; }
void G2::mul(G2& z, const G2& x, const Fr& k) {
  G2 r;
  elpCheck(defaultContext(), elp_g2_mul(defaultContext(), 1, x.b, k.b, r.b), "elp_g2_mul");
  z = r;
}
void G2::add(G2& z, const G2& x, const G2& y) {
  G2 r;
  elpCheck(defaultContext(), elp_g2_add(defaultContext(), 1, x.b, y.b, r.b), "elp_g2_add");
  z = r;
}
void G2::neg(G2& z, const G2& x) {
  const size_t F = g_cv->F;
  G2 r = x;
  negCoord(r.b + 2 * F, x.b + 2 * F);
  negCoord(r.b + 3 * F, x.b + 3 * F);
  z = r;
}
void G2::sub(G2& z, const G2& x, const G2& y) {
  G2 n;
  neg(n, y);
  add(z, x, n);
}
size_t G2::serialize(void* buf, size_t maxSize) const {
  const size_t F = g_cv->F;
  if (maxSize < 2 * F) return 0;
  uint8_t* o = (uint8_t*)buf;
  memcpy(o, b, 2 * F);
  if (b[2 * F] & 1) o[2 * F - 1] |= 0x80;
  return 2 * F;
}
size_t G2::deserialize(const void* buf, size_t size) {
  if (size != 2 * g_cv->F) return 0;
  uint8_t ok = 0;
  G2 r;
  elpCheck(defaultContext(), elp_g2_decompress(defaultContext(), 1, (const uint8_t*)buf, r.b, &ok), "elp_g2_decompress");
  if (!ok) return 0;
  *this = r;
  return size;
}
std::string G2::serializeToHexStr() const {
  uint8_t w[96];
  size_t n = serialize(w, sizeof w);
  return toHex(w, n);
}
bool GT::operator==(const GT& o) const { return memcmp(b, o.b, 12 * g_cv->F) == 0; }

void pairing(GT& e, const G1& P, const G2& Q) {
  elpCheck(defaultContext(), elp_pairing(defaultContext(), 1, P.b, Q.b, e.b), "elp_pairing");
}
void hashAndMapToG1(G1& P, const std::string& msg) {
  uint32_t off[2] = {0, (uint32_t)msg.size()};
  elpCheck(defaultContext(), elp_hash_to_g1(defaultContext(), 1, (const uint8_t*)msg.data(), off, P.b), "elp_hash_to_g1");
}
void hashAndMapToG2(G2& P, const std::string& msg) {
  if (g_cv->id == ELP_CURVE_BLS12_381) {
    // stand-in (only ever used to pick a generator, src/ps-signer.cc:17): [Fr::setHashOf(msg)] times the standard G2 generator
    static const char* gen[4] = {
        "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8",
        "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e",
        "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801",
        "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be"};
    G2 g;
    for (int c = 0; c < 4; c++)
      for (int i = 0; i < 48; i++) {   // big-endian hex -> little-endian bytes
        auto hv = [](char ch) { return (uint8_t)(ch <= '9' ? ch - '0' : ch - 'a' + 10); };
        g.b[48 * c + i] = (uint8_t)((hv(gen[c][2 * (47 - i)]) << 4) | hv(gen[c][2 * (47 - i) + 1]));
      }
    Fr k;
    k.setHashOf(msg);
    if (k.isZero()) k.setInt(1);
    G2::mul(P, g, k);
    return;
  }
  // BN254: x = (Fp::setHashOf(msg || ctr || 0), Fp::setHashOf(msg || ctr || 1)); first x on the twist wins; then clear the cofactor 2p - r
  for (uint32_t ctr = 0;; ctr++) {
    uint8_t wire[64];
    for (int half = 0; half < 2; half++) {
      cybozu::Sha256 h;
      std::string d = h.digest(msg + std::string(1, (char)ctr) + std::string(1, (char)half));
      uint64_t v[4];
      memcpy(v, d.data(), 32);
      v[3] &= (1ull << 62) - 1;
      if (cmp4(v, g_cv->p) >= 0) v[3] &= (1ull << 61) - 1;
      memcpy(wire + 32 * half, v, 32);
    }
    uint8_t ok = 0;
    G2 q;
    elpCheck(defaultContext(), elp_g2_decompress(defaultContext(), 1, wire, q.b, &ok), "elp_g2_decompress");
    if (!ok || q.isZero()) continue;
    uint64_t cof[4], two_p[4];
    add4(two_p, g_cv->p, g_cv->p);
    sub4(cof, two_p, R_);
    uint8_t k[32];
    st(k, cof);
    G2 r;
    elpCheck(defaultContext(), elp_g2_mul(defaultContext(), 1, q.b, k, r.b), "elp_g2_mul");
    if (r.isZero()) continue;
    P = r;
    return;
  }
}

}  // namespace bls12
}  // namespace mcl

// ---- SHA-256 (host; transcript and attribute hashing of single items)
namespace cybozu {
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
Sha256::Sha256() : len_(0) {
  static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(h_, iv, sizeof iv);
}
void Sha256::block(const uint8_t* p) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
  for (int i = 16; i < 64; i++)
    w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
  uint32_t a = h_[0], b = h_[1], c = h_[2], d = h_[3], e = h_[4], f = h_[5], g = h_[6], h = h_[7];
  for (int i = 0; i < 64; i++) {
    uint32_t t1 = h + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
    uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h_[0] += a; h_[1] += b; h_[2] += c; h_[3] += d; h_[4] += e; h_[5] += f; h_[6] += g; h_[7] += h;
}
// ---- one-shot SHA-256 of a byte string: whole blocks straight from the message, the padded tail from a stack buffer; the compression function uses the x86 SHA
// extensions where the CPU has them (checked once at run time), else the portable rounds above.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("sha,sse4.1,ssse3"))) static void compressShaNi(uint32_t state[8], const uint8_t* data, size_t nblocks) {
  const __m128i MASK = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
  __m128i TMP = _mm_loadu_si128((const __m128i*)&state[0]);
  __m128i STATE1 = _mm_loadu_si128((const __m128i*)&state[4]);
  TMP = _mm_shuffle_epi32(TMP, 0xB1);             // CDAB
  STATE1 = _mm_shuffle_epi32(STATE1, 0x1B);       // EFGH
  __m128i STATE0 = _mm_alignr_epi8(TMP, STATE1, 8);   // ABEF
  STATE1 = _mm_blend_epi16(STATE1, TMP, 0xF0);        // CDGH
  while (nblocks--) {
    const __m128i ABEF_SAVE = STATE0, CDGH_SAVE = STATE1;
    __m128i M[4];
    for (int i = 0; i < 4; i++) M[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(data + 16 * i)), MASK);
    for (int r = 0; r < 16; r++) {                // 16 groups of four rounds; the message schedule advances one vector per group
      __m128i MSG = _mm_add_epi32(M[r & 3], _mm_loadu_si128((const __m128i*)&K256[4 * r]));
      STATE1 = _mm_sha256rnds2_epu32(STATE1, STATE0, MSG);
      MSG = _mm_shuffle_epi32(MSG, 0x0E);
      STATE0 = _mm_sha256rnds2_epu32(STATE0, STATE1, MSG);
      if (r < 12) {                               // W[4(r+4) .. 4(r+4)+3] from the four vectors in flight
        __m128i X = _mm_sha256msg1_epu32(M[r & 3], M[(r + 1) & 3]);
        X = _mm_add_epi32(X, _mm_alignr_epi8(M[(r + 3) & 3], M[(r + 2) & 3], 4));
        M[r & 3] = _mm_sha256msg2_epu32(X, M[(r + 3) & 3]);
      }
    }
    STATE0 = _mm_add_epi32(STATE0, ABEF_SAVE);
    STATE1 = _mm_add_epi32(STATE1, CDGH_SAVE);
    data += 64;
  }
  TMP = _mm_shuffle_epi32(STATE0, 0x1B);          // FEBA
  STATE1 = _mm_shuffle_epi32(STATE1, 0xB1);       // DCHG
  STATE0 = _mm_blend_epi16(TMP, STATE1, 0xF0);    // DCBA
  STATE1 = _mm_alignr_epi8(STATE1, TMP, 8);       // HGFE
  _mm_storeu_si128((__m128i*)&state[0], STATE0);
  _mm_storeu_si128((__m128i*)&state[4], STATE1);
}
static bool haveShaNi() {
  static const bool v = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");
  return v;
}
#else
static bool haveShaNi() { return false; }
static void compressShaNi(uint32_t*, const uint8_t*, size_t) {}
#endif
static void compressPortable(uint32_t st[8], const uint8_t* p, size_t nblocks) {
  for (; nblocks--; p += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++)
      w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; i++) {
      uint32_t t1 = h + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
      uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
  }
}
void sha256OneShot(const uint8_t* msg, size_t n, uint8_t out[32], int force_portable) {
  uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  const bool ni = !force_portable && haveShaNi();
  const size_t whole = n / 64;
  if (whole) (ni ? compressShaNi : compressPortable)(st, msg, whole);
  uint8_t tail[128];
  const size_t rem = n - 64 * whole;
  memset(tail, 0, sizeof tail);
  if (rem) memcpy(tail, msg + 64 * whole, rem);
  tail[rem] = 0x80;
  const size_t tb = rem < 56 ? 1 : 2;
  const uint64_t bits = (uint64_t)n * 8;
  for (int i = 0; i < 8; i++) tail[64 * tb - 1 - i] = (uint8_t)(bits >> (8 * i));
  (ni ? compressShaNi : compressPortable)(st, tail, tb);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(st[i] >> 24);
    out[4 * i + 1] = (uint8_t)(st[i] >> 16);
    out[4 * i + 2] = (uint8_t)(st[i] >> 8);
    out[4 * i + 3] = (uint8_t)st[i];
  }
}
bool sha256HasHardware() { return haveShaNi(); }
void Sha256::feed(const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; i++) {
    buf_[len_++ & 63] = p[i];
    if ((len_ & 63) == 0) block(buf_);
  }
}
void Sha256::update(const std::string& s) { feed((const uint8_t*)s.data(), s.size()); }
std::string Sha256::digest(const std::string& s) {
  update(s);
  uint64_t bits = len_ * 8;
  uint8_t pad = 0x80, z = 0;
  feed(&pad, 1);
  while ((len_ & 63) != 56) feed(&z, 1);
  uint8_t lb[8];
  for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
  feed(lb, 8);
  std::string out(32, '\0');
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (char)(h_[i] >> 24);
    out[4 * i + 1] = (char)(h_[i] >> 16);
    out[4 * i + 2] = (char)(h_[i] >> 8);
    out[4 * i + 3] = (char)h_[i];
  }
  return out;
}
}  // namespace cybozu
