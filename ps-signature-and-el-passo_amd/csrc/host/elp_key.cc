#include "elp_key.h"

#include <string.h>

static bool g_strict_signature = true;
void elpSetStrictSignature(bool strict) { g_strict_signature = strict; }

static int g_default_device = -1;      // -1: the device initPairing() selected
static int g_default_window = 16;     // PSVerifier: 16-bit windows: 1.2 GiB of tables for an 8-attribute BN254 key, 17.3 ms per 65 536 proofs (W = 20: 15.5 GiB, 16.9 ms: opt-in)
static bool g_window_chosen = false;  // elpSetDefaults was given a width: it then applies to every class
void elpSetDefaults(int device, int window_bits) {
  g_default_device = device;
  g_window_chosen = window_bits >= 0;
  g_default_window = window_bits < 0 ? 16 : window_bits;      // 0 = the C-ABI's own default (8)
}
int elpDefaultDevice() { return g_default_device; }
int elpDefaultWindowBits() { return g_default_window; }
int elpDefaultSideWindowBits() { return g_window_chosen ? g_default_window : 0; }

ElpPinned::~ElpPinned() {
  if (p_) elp_host_free(ctx_, p_);
}
uint8_t* ElpPinned::get(elp_ctx* ctx, size_t bytes) {
  if (bytes <= cap_ && ctx == ctx_) return p_;
  if (p_) elp_host_free(ctx_, p_);
  p_ = nullptr;
  cap_ = 0;
  ctx_ = ctx;
  void* q = nullptr;
  const size_t want = bytes + bytes / 4 + 4096;
  elpCheck(ctx, elp_host_alloc(ctx, want, &q), "elp_host_alloc");
  p_ = (uint8_t*)q;
  cap_ = want;
  return p_;
}

void elpParallelFor(size_t n, size_t grain, const std::function<void(size_t, size_t)>& fn, unsigned threads) {
  if (n == 0) return;
  unsigned hw = threads ? threads : std::thread::hardware_concurrency();
  if (hw == 0) hw = 4;
  if (hw > 32) hw = 32;
  size_t parts = (n + grain - 1) / (grain ? grain : 1);
  if (parts > hw) parts = hw;
  if (parts <= 1) {
    fn(0, n);
    return;
  }
  std::vector<std::exception_ptr> errs(parts);
  std::vector<std::thread> th;
  auto run = [&](size_t t) {
    try {
      fn(n * t / parts, n * (t + 1) / parts);
    } catch (...) {
      errs[t] = std::current_exception();
    }
  };
  for (size_t t = 1; t < parts; t++) th.emplace_back(run, t);
  run(0);
  for (auto& t : th) t.join();
  for (auto& e : errs)
    if (e) std::rethrow_exception(e);
}

ElpShardSet::ElpShardSet(const PSPubKey& pk, const std::vector<int>& devices, int window_bits) {
  if (devices.empty()) throw std::runtime_error("ElpShardSet: no devices");
  keys_.resize(devices.size());
  std::vector<std::exception_ptr> errs(devices.size());
  std::vector<std::thread> th;
  auto make = [&](size_t r) {
    try {
      keys_[r] = std::make_shared<ElpKey>(pk, devices[r], window_bits);
    } catch (...) {
      errs[r] = std::current_exception();
    }
  };
  for (size_t r = 1; r < devices.size(); r++) th.emplace_back(make, r);     // the table builds of the devices overlap
  make(0);
  for (auto& t : th) t.join();
  for (auto& e : errs)
    if (e) std::rethrow_exception(e);
}

ElpKey::ElpKey(const PSPubKey& pk, int device, int window_bits) {
  if (pk.Yi.size() != pk.YYi.size() || pk.Yi.empty() || pk.Yi.size() > 62) throw std::runtime_error("ElpKey: bad public key shape");
  nattr_ = pk.Yi.size();
  if (device < 0) device = g_default_device >= 0 ? g_default_device : mcl::bls12::defaultDevice();
  if (window_bits < 0) window_bits = g_default_window;
  elpCheck(nullptr, elp_init(curveId(), device, &ctx_), "elp_init");      // the curve initPairing() selected for the process
  elp_set_option(ctx_, ELP_OPT_STRICT_SIGNATURE, g_strict_signature ? 1 : 0);
  const size_t S1 = G1::size(), S2 = G2::size();
  std::vector<uint8_t> yi(S1 * nattr_), yyi(S2 * nattr_);
  for (size_t i = 0; i < nattr_; i++) {
    memcpy(&yi[S1 * i], pk.Yi[i].b, S1);
    memcpy(&yyi[S2 * i], pk.YYi[i].b, S2);
  }
  int rc = elp_set_pubkey(ctx_, (int)nattr_, pk.g.b, pk.gg.b, pk.XX.b, yi.data(), yyi.data(), window_bits);
  if (rc != ELP_OK) {
    std::string msg = elp_last_error(ctx_);
    elp_destroy(ctx_);
    ctx_ = nullptr;
    throw std::runtime_error("elp_set_pubkey failed: " + msg);
  }
}
ElpKey::~ElpKey() {
  if (ctx_) elp_destroy(ctx_);
}
void ElpKey::useRp(const std::string& service, const G1* apk, const G1* g, const G1* h) {
  G1 zero;
  const G1& a = apk ? *apk : zero;
  const G1& gg = g ? *g : zero;
  const G1& hh = h ? *h : zero;
  if (rp_set_ && service == rp_service_ && a == rp_apk_ && gg == rp_g_ && hh == rp_h_) return;
  elpCheck(ctx_, elp_set_rp(ctx_, (const uint8_t*)service.data(), service.size(), apk ? a.b : nullptr, g ? gg.b : nullptr, h ? hh.b : nullptr),
           "elp_set_rp");
  rp_set_ = true;
  rp_service_ = service;
  rp_apk_ = a;
  rp_g_ = gg;
  rp_h_ = hh;
}
void ElpKey::useSignerSecret(const G1& X) {
  if (sk_set_ && X == sk_X_) return;
  elpCheck(ctx_, elp_set_signer_secret(ctx_, X.b), "elp_set_signer_secret");
  sk_set_ = true;
  sk_X_ = X;
}
G1 ElpKey::msmG1(const std::vector<int32_t>& ids, const std::vector<Fr>& scalars) const {
  std::vector<uint8_t> ks(32 * scalars.size());
  for (size_t i = 0; i < scalars.size(); i++) memcpy(&ks[32 * i], scalars[i].b, 32);
  G1 out;
  elpCheck(ctx_, elp_g1_msm_fixed(ctx_, 1, (int)ids.size(), ids.data(), ks.data(), out.b), "elp_g1_msm_fixed");
  return out;
}
G2 ElpKey::msmG2(const std::vector<int32_t>& ids, const std::vector<Fr>& scalars) const {
  std::vector<uint8_t> ks(32 * scalars.size());
  for (size_t i = 0; i < scalars.size(); i++) memcpy(&ks[32 * i], scalars[i].b, 32);
  G2 out;
  elpCheck(ctx_, elp_g2_msm_fixed(ctx_, 1, (int)ids.size(), ids.data(), ks.data(), out.b), "elp_g2_msm_fixed");
  return out;
}
uint64_t elpHiddenMask(const std::vector<std::string>& attributes) {
  uint64_t m = 0;
  for (size_t i = 0; i < attributes.size() && i < 64; i++)
    if (attributes[i].empty()) m |= 1ull << i;
  return m;
}
