// Per-object GPU key context shared by PSSigner / PSRequester / PSVerifier: owns one elp_ctx holding the public key's
// fixed-base tables, and (re)installs RP parameters / the signer secret on demand.
#pragma once
#include <exception>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "ps-encoding.h"

class ElpKey {
 public:
  // device / window_bits < 0: the process defaults (elpSetDefaults; initially the device initPairing() selected and the library's window width)
  explicit ElpKey(const PSPubKey& pk, int device = -1, int window_bits = -1);
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

// Process-wide defaults for the contexts that PSSigner / PSRequester / PSVerifier create from now on: GPU ordinal and width of the fixed-base
// window tables.  Default since round 5: 16 bits -- 1.2 GiB of signed-digit tables for an 8-attribute BN254 key (0.3 s to build), 17.3 ms per 65 536 verifications;
// 20 bits buys 2.6 % (16.9 ms) for 15.5 GiB per key and is an explicit opt-in (a relying party with several IdP keys cannot afford it per key); 0 = the C-ABI's
// own default (8 bits, 80 MB).  The reference's constructors (src/ps-verifier.h:18) carry no such parameters, so a drop-in caller chooses them here once.
// The 16-bit default is the VERIFIER's (PSVerifier: the class that verifies batches).  PSSigner and PSRequester objects made by the reference's constructors
// keep the C-ABI's own width (8 bits, ~80 MB per key) unless a width was chosen here explicitly: a drop-in caller that only signs or proves should not pay
// gigabytes of G2 tables per object.  elpSetDefaults(dev, -1) restores exactly this state.
void elpSetDefaults(int device, int window_bits);
int elpDefaultDevice();
int elpDefaultWindowBits();
int elpDefaultSideWindowBits();      // PSSigner / PSRequester

// Several contexts of the same public key -- one per entry of `devices` (an ordinal may repeat: several contexts on one GPU) -- each with its own
// host thread and HIP stream while a batch is in flight.  Batches are cut into contiguous shards, shard r of N taking items
// [r n / N, (r + 1) n / N); nothing is exchanged between shards, the per-shard counts are summed by the caller.  Set-up (tables) runs in parallel.
class ElpShardSet {
 public:
  ElpShardSet(const PSPubKey& pk, const std::vector<int>& devices, int window_bits = -1);
  size_t size() const { return keys_.size(); }
  ElpKey& key(size_t r) const { return *keys_[r]; }
  static void range(size_t n, size_t r, size_t N, size_t& first, size_t& count) {
    first = n * r / N;
    count = n * (r + 1) / N - first;
  }
  // fn(shard, first, count) on one thread per shard (the calling thread takes shard 0); the first exception is rethrown here
  template <class Fn>
  void forEachShard(size_t n, Fn fn) const;

 private:
  std::vector<std::shared_ptr<ElpKey>> keys_;
};

// Grow-only page-locked staging buffer (elp_host_alloc): the packed records / messages of a batch are written here by the packing threads and
// copied to the GPU by DMA; reused from call to call, so a steady stream of batches pays neither page faults nor runtime staging.
class ElpPinned {
 public:
  ElpPinned() = default;
  ~ElpPinned();
  ElpPinned(const ElpPinned&) = delete;
  ElpPinned& operator=(const ElpPinned&) = delete;
  uint8_t* get(elp_ctx* ctx, size_t bytes);      // at least `bytes` bytes (contents undefined)

 private:
  elp_ctx* ctx_ = nullptr;
  uint8_t* p_ = nullptr;
  size_t cap_ = 0;
};

// fn(lo, hi) over [0, n) on up to `threads` host threads (0 = as many as the machine allows, at most 32); small n runs inline
void elpParallelFor(size_t n, size_t grain, const std::function<void(size_t, size_t)>& fn, unsigned threads = 0);

// Process-wide default of ELP_OPT_STRICT_SIGNATURE for contexts created afterwards (library default: strict, i.e. proofs with
// sig1 == infinity are rejected; false = the reference's behaviour, which accepts sig1 = sig2 = infinity, see include/elpasso.h).
void elpSetStrictSignature(bool strict);

// hidden-attribute mask of a message's attribute list ("" = hidden)
uint64_t elpHiddenMask(const std::vector<std::string>& attributes);

template <class Fn>
void ElpShardSet::forEachShard(size_t n, Fn fn) const {
  const size_t N = keys_.size();
  std::vector<std::exception_ptr> errs(N);
  std::vector<std::thread> th;
  auto run = [&](size_t r) {
    size_t first, count;
    range(n, r, N, first, count);
    try {
      fn(r, first, count);
    } catch (...) {
      errs[r] = std::current_exception();
    }
  };
  for (size_t r = 1; r < N; r++) th.emplace_back(run, r);
  run(0);
  for (auto& t : th) t.join();
  for (auto& e : errs)
    if (e) std::rethrow_exception(e);
}
