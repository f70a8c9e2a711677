// RP side.  Reference behaviour: src/ps-verifier.cc:13-35 (verify), :37-138 (el_passo_verify_id),
// :140-212 (..._without_id_retrieval), :231-235 (get_user_name_from_signon_request).
// Here each call packs fixed-stride records and runs the fused GPU kernel; proofs are grouped by their hidden-attribute
// pattern so that one launch covers every proof with the same pattern.
#include "ps-verifier.h"

#include <atomic>
#include <thread>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <map>

namespace {
void put(std::vector<uint8_t>& v, const uint8_t* p, size_t n) { v.insert(v.end(), p, p + n); }
}  // namespace

PSVerifier::PSVerifier(const PSPubKey& pk) : PSVerifier(pk, std::vector<int>(1, -1), -1) {}
PSVerifier::PSVerifier(const PSPubKey& pk, const std::vector<int>& devices, int window_bits)
    : m_pk(pk), m_set(std::make_shared<ElpShardSet>(pk, devices, window_bits)), m_stage(std::make_shared<Stage>()) {
  m_key = std::shared_ptr<ElpKey>(m_set, &m_set->key(0));
}
void PSVerifier::set_option(int option, int value, int shard) const {
  std::lock_guard<std::mutex> lock(m_stage->mu);
  for (size_t r = 0; r < m_set->size(); r++)
    if (shard < 0 || (size_t)shard == r) elpCheck(m_set->key(r).ctx(), elp_set_option(m_set->key(r).ctx(), option, value), "elp_set_option");
}
void PSVerifier::useRpAll(const std::string& service, const G1* apk, const G1* g, const G1* h) const {
  m_set->forEachShard(m_set->size(), [&](size_t r, size_t, size_t) { m_set->key(r).useRp(service, apk, g, h); });
}

std::vector<bool> PSVerifier::verify_batch(const std::vector<PSCredential>& sigs,
                                           const std::vector<std::vector<std::string>>& attrs) const {
  std::vector<bool> out(sigs.size(), false);
  std::lock_guard<std::mutex> lock(m_stage->mu);            // an elp_ctx serves one call at a time (include/elpasso.h)
  // group by attribute count (the record stride depends on it)
  std::map<size_t, std::vector<size_t>> groups;
  for (size_t i = 0; i < sigs.size(); i++)
    if (attrs[i].size() <= m_key->attrs()) groups[attrs[i].size()].push_back(i);
  for (auto& [na, idx] : groups) {
    std::vector<uint8_t> recs;
    for (size_t i : idx) {
      put(recs, sigs[i].sig1.b, G1::size());
      put(recs, sigs[i].sig2.b, G1::size());
      for (const std::string& a : attrs[i]) {
        Fr m;
        m.setHashOf(a);
        put(recs, m.b, 32);
      }
    }
    std::vector<uint8_t> flags(idx.size());
    uint64_t acc = 0;
    elpCheck(m_key->ctx(), elp_ps_verify_batch(m_key->ctx(), idx.size(), recs.data(), (int)na, flags.data(), &acc), "elp_ps_verify_batch");
    for (size_t j = 0; j < idx.size(); j++) out[idx[j]] = flags[j] != 0;
  }
  return out;
}

bool PSVerifier::verify(const PSCredential& sig, const std::vector<std::string>& all_attributes) const {
  return verify_batch({sig}, {all_attributes})[0];
}

namespace {
// one record (csrc/elp/pipeline.h verify_id record) from an IdProof; the revealed attributes are hashed here (src/ps-verifier.cc:224)
inline void packRecord(uint8_t* w, const IdProof& p, bool retr, size_t S1, size_t S2) {
  auto put = [&](const uint8_t* src, size_t len) {
    memcpy(w, src, len);
    w += len;
  };
  put(p.sig1.b, S1);
  put(p.sig2.b, S1);
  put(p.phi.b, S1);
  if (retr) {
    put(p.E1->b, S1);
    put(p.E2->b, S1);
  }
  put(p.k.b, S2);
  put(p.c.b, 32);
  for (const Fr& r : p.rs) put(r.b, 32);
  for (const std::string& a : p.attributes)
    if (!a.empty()) {
      Fr m;
      m.setHashOf(a);
      put(m.b, 32);
    }
}
}  // namespace
// The synchronous batch call on ONE context when every proof hides the same attributes (the usual case): host threads pack contiguous ranges of records chunk by
// chunk -- validating each proof against the pattern of the first one on the way -- and hand every packed chunk to elp_verify_id_batch_stage at once, so that the
// records cross PCIe while the rest is still being packed and hashed; the kernel is queued behind the last part (round 6: packing + hashing used to run to the
// end before the first byte was copied).  Returns false -- nothing queued that outlives the call -- when a proof deviates; the caller then takes the grouped path.
bool PSVerifier::verifyIdStaged(const std::vector<IdProof>& proofs, const std::vector<std::string>& ads, bool retr, std::vector<bool>& out) const {
  const size_t n = proofs.size(), A = m_key->attrs();
  if (m_set->size() != 1 || n < 4096 || m_stage->slot[0].busy) return false;
  const IdProof& p0 = proofs[0];
  if (p0.attributes.size() != A) return false;
  const uint64_t mask = elpHiddenMask(p0.attributes);
  const size_t H = (size_t)__builtin_popcountll(mask);
  if (H < (retr ? 2u : 1u)) return false;
  const size_t S1 = G1::size(), S2 = G2::size();
  const size_t rsz = elp_verify_id_record_size(curveId(), (int)A, (int)H, retr ? 1 : 0);
  elp_ctx* ctx = m_key->ctx();
  std::vector<uint32_t> adoff(n + 1, 0);
  for (size_t j = 0; j < n; j++) adoff[j + 1] = adoff[j] + (uint32_t)ads[j].size();
  uint8_t* const recs = m_stage->recs.get(ctx, n * rsz);
  uint8_t* const adbuf = m_stage->ads.get(ctx, adoff[n] ? adoff[n] : 1);
  uint8_t* const flags = m_stage->flags.get(ctx, n);
  const int slot = 0;
  elpCheck(ctx, elp_verify_id_batch_stage(ctx, slot, n, rsz, 0, 0, nullptr), "elp_verify_id_batch_stage");      // sizes the device buffer before the workers start
  std::atomic<bool> deviates{false};
  // Workers take blocks of BLOCK records in batch order from a shared counter; the calling thread hands every PART (a run of blocks) to the library as soon as its
  // last block is packed -- one C-ABI call at a time, from one thread, while the workers go on packing -- so that only the last part's copy is left when packing ends.
  static const size_t PARTS = [] {
    const char* e = getenv("ELP_STAGE_PARTS");
    const long v = e ? atol(e) : 0;
    return (size_t)(v >= 1 && v <= 256 ? v : 8);       // measured on the MI355X box (65 536 proofs): 8 parts 19.8 ms, 16: 20.5-21.5, 32: 21.3
  }();
  const size_t BLOCK = 256;
  const size_t nblocks = (n + BLOCK - 1) / BLOCK;
  const size_t parts = PARTS < nblocks ? PARTS : nblocks;
  std::vector<std::atomic<int>> left(parts);
  auto part_of = [&](size_t blk) { return blk * parts / nblocks; };
  {
    std::vector<int> cnt(parts, 0);
    for (size_t bl = 0; bl < nblocks; bl++) cnt[part_of(bl)]++;
    for (size_t q = 0; q < parts; q++) left[q].store(cnt[q]);
  }
  std::atomic<size_t> next{0};
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 8;
  unsigned nthreads = hw > 48 ? 48 : hw;                              // packing + SHA-256 of the revealed attributes: ~1.8 ms on 16 threads for 65 536 proofs
  if (const char* e = getenv("ELP_STAGE_THREADS")) nthreads = (unsigned)(atoi(e) > 0 ? atoi(e) : (int)nthreads);
  if (nthreads > nblocks) nthreads = (unsigned)nblocks;
  const auto t_begin = std::chrono::steady_clock::now();
  auto worker = [&]() {
    for (;;) {
      const size_t blk = next.fetch_add(1);
      if (blk >= nblocks) return;
      const size_t c0 = blk * BLOCK, c1 = c0 + BLOCK < n ? c0 + BLOCK : n;
      if (!deviates.load(std::memory_order_relaxed)) {
        for (size_t j = c0; j < c1; j++) {
          const IdProof& p = proofs[j];
          if ((retr && (!p.E1.has_value() || !p.E2.has_value())) || p.attributes.size() != A || p.rs.size() != H + (retr ? 2 : 1) || elpHiddenMask(p.attributes) != mask) {
            deviates.store(true);
            break;
          }
          packRecord(recs + j * rsz, p, retr, S1, S2);
          const std::string& ad = ads[j];
          if (!ad.empty()) memcpy(adbuf + adoff[j], ad.data(), ad.size());
        }
      }
      left[part_of(blk)].fetch_sub(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nthreads; t++) pool.emplace_back(worker);
  std::exception_ptr stage_err;
  size_t b0 = 0;
  for (size_t q = 0; q < parts; q++) {
    while (left[q].load(std::memory_order_acquire) > 0) std::this_thread::yield();
    size_t b1 = b0;                                                   // blocks of part q: [b0, b1)
    while (b1 < nblocks && part_of(b1) == q) b1++;
    const size_t c0 = b0 * BLOCK, c1 = b1 * BLOCK < n ? b1 * BLOCK : n;
    b0 = b1;
    if (stage_err || deviates.load()) continue;
    try {
      elpCheck(ctx, elp_verify_id_batch_stage(ctx, slot, n, rsz, c0, c1 - c0, recs + c0 * rsz), "elp_verify_id_batch_stage");
    } catch (...) {
      stage_err = std::current_exception();
    }
  }
  for (auto& th : pool) th.join();
  if (stage_err || deviates.load()) {
    // drop what was staged: a submit that does not find exactly n records resets the slot's staging (and fails, on purpose)
    (void)elp_verify_id_batch_submit(ctx, slot, n + 1, nullptr, mask, retr ? 1 : 0, adbuf, adoff.data(), 0, flags);
    if (stage_err) std::rethrow_exception(stage_err);
    return false;
  }
  const auto t_packed = std::chrono::steady_clock::now();
  elpCheck(ctx, elp_verify_id_batch_submit(ctx, slot, n, nullptr, mask, retr ? 1 : 0, adbuf, adoff.data(), 0, flags), "elp_verify_id_batch_submit");
  uint64_t acc = 0;
  elpCheck(ctx, elp_verify_id_batch_wait(ctx, slot, &acc), "elp_verify_id_batch_wait");
  const auto t_done = std::chrono::steady_clock::now();
  out.assign(n, false);
  for (size_t j = 0; j < n; j++) out[j] = flags[j] != 0;
  if (getenv("ELP_HOST_TIMING"))
    fprintf(stderr, "PSVerifier (staged): %zu proofs  pack+hash+stage %.2f ms  submit+wait %.2f ms  verdicts out %.2f ms\n", n,
            std::chrono::duration<double, std::milli>(t_packed - t_begin).count(), std::chrono::duration<double, std::milli>(t_done - t_packed).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_done).count());
  return true;
}

std::vector<bool> PSVerifier::verifyIdImpl(const std::vector<IdProof>& proofs, const std::vector<std::string>& ads, bool retr) const {
  std::vector<bool> out(proofs.size(), false);
  if (!proofs.empty() && verifyIdStaged(proofs, ads, retr, out)) return out;
  out.assign(proofs.size(), false);
  const size_t A = m_key->attrs();
  std::map<uint64_t, std::vector<size_t>> groups;
  for (size_t i = 0; i < proofs.size(); i++) {
    const IdProof& p = proofs[i];
    if (retr && (!p.E1.has_value() || !p.E2.has_value())) continue;      // src/ps-verifier.cc:68-70
    if (p.attributes.size() != A) continue;                               // the reference indexes YYi[i] unchecked (UB); reject
    uint64_t mask = elpHiddenMask(p.attributes);
    size_t H = (size_t)__builtin_popcountll(mask);
    if (p.rs.size() != H + (retr ? 2 : 1) || H < (retr ? 2u : 1u)) continue;
    groups[mask].push_back(i);
  }
  const size_t S1 = G1::size(), S2 = G2::size();
  // the caller holds m_stage->mu (taken BEFORE useRpAll: the relying-party parameters installed in the contexts belong to this call until its shards are done)
  for (auto& [mask, idx] : groups) {
    // fixed-stride records of the group, packed by several host threads (the record of item j starts at j * rsz; the revealed attributes
    // are hashed here, src/ps-verifier.cc:224); the per-item associated data goes into one blob with offsets
    const auto t_group = std::chrono::steady_clock::now();
    const size_t n = idx.size();
    const size_t H = (size_t)__builtin_popcountll(mask);
    const size_t rsz = elp_verify_id_record_size(curveId(), (int)A, (int)H, retr ? 1 : 0);
    std::vector<uint32_t> adoff(n + 1, 0);
    for (size_t j = 0; j < n; j++) adoff[j + 1] = adoff[j] + (uint32_t)ads[idx[j]].size();
    uint8_t* const recs = m_stage->recs.get(m_key->ctx(), n * rsz);
    uint8_t* const adbuf = m_stage->ads.get(m_key->ctx(), adoff[n] ? adoff[n] : 1);
    uint8_t* const flags = m_stage->flags.get(m_key->ctx(), n);
    elpParallelFor(n, 512, [&](size_t lo, size_t hi) {
      for (size_t j = lo; j < hi; j++) {
        const IdProof& p = proofs[idx[j]];
        uint8_t* w = recs + j * rsz;
        auto put = [&](const uint8_t* src, size_t len) {
          memcpy(w, src, len);
          w += len;
        };
        put(p.sig1.b, S1);
        put(p.sig2.b, S1);
        put(p.phi.b, S1);
        if (retr) {
          put(p.E1->b, S1);
          put(p.E2->b, S1);
        }
        put(p.k.b, S2);
        put(p.c.b, 32);
        for (const Fr& r : p.rs) put(r.b, 32);
        for (const std::string& a : p.attributes)
          if (!a.empty()) {
            Fr m;
            m.setHashOf(a);
            put(m.b, 32);
          }
        const std::string& ad = ads[idx[j]];
        if (!ad.empty()) memcpy(adbuf + adoff[j], ad.data(), ad.size());
      }
    });
    const auto t_packed = std::chrono::steady_clock::now();
    // contiguous shards, one context (host thread + HIP stream, possibly another GPU) each; per-shard offsets are rebased to the shard
    m_set->forEachShard(n, [&](size_t r, size_t first, size_t count) {
      if (count == 0) return;
      std::vector<uint32_t> off(count + 1);
      for (size_t j = 0; j <= count; j++) off[j] = adoff[first + j] - adoff[first];
      uint64_t acc = 0;
      elp_ctx* ctx = m_set->key(r).ctx();
      elpCheck(ctx, elp_verify_id_batch(ctx, count, recs + first * rsz, mask, retr ? 1 : 0, adbuf + adoff[first], off.data(), 0,
                                        flags + first, &acc),
               "elp_verify_id_batch");
    });
    if (getenv("ELP_HOST_TIMING")) {
      const auto t_done = std::chrono::steady_clock::now();
      fprintf(stderr, "PSVerifier: %zu proofs  pack+hash %.2f ms  verify (copies included) %.2f ms\n", n,
              std::chrono::duration<double, std::milli>(t_packed - t_group).count(), std::chrono::duration<double, std::milli>(t_done - t_packed).count());
    }
    for (size_t j = 0; j < n; j++) out[idx[j]] = flags[j] != 0;
  }
  return out;
}

std::vector<bool> PSVerifier::el_passo_verify_id_batch(const std::vector<IdProof>& proofs, const std::vector<std::string>& ads,
                                                       const std::string& service_name, const G1& authority_pk, const G1& g,
                                                       const G1& h) const {
  if (ads.size() != proofs.size()) throw std::runtime_error("associated data count does not match");
  std::lock_guard<std::mutex> lock(m_stage->mu);
  useRpAll(service_name, &authority_pk, &g, &h);
  return verifyIdImpl(proofs, ads, true);
}
// ---- pipelined form (include/elpasso.h elp_verify_id_batch_submit / _wait)
size_t PSVerifier::el_passo_verify_id_submit(const std::vector<IdProof>& proofs, const std::vector<std::string>& ads, const std::string& service_name,
                                             const G1& authority_pk, const G1& g, const G1& h) const {
  if (ads.size() != proofs.size()) throw std::runtime_error("associated data count does not match");
  std::lock_guard<std::mutex> lock(m_stage->mu);
  Stage& st = *m_stage;
  int si = !st.slot[0].busy ? 0 : (!st.slot[1].busy ? 1 : -1);
  if (si < 0) throw std::runtime_error("el_passo_verify_id_submit: two batches are in flight, collect one first");
  Stage::Slot& sl = st.slot[si];
  const std::string sig = service_name + "|" + authority_pk.serializeToHexStr() + g.serializeToHexStr() + h.serializeToHexStr();
  const bool others = st.slot[1 - si].busy;
  if (others && sig != st.rp_sig) throw std::runtime_error("el_passo_verify_id_submit: relying-party parameters differ from the batch in flight");
  // Nothing of the slot or of the stage is committed before the batch is queued (or verified) successfully: a submit that throws -- a transient HIP
  // error, staging that cannot grow -- hands out no ticket, so it must leave the slot free and the signature of the batches in flight as it was
  // (round-4 advisor finding: two failed submits used to block the verifier for good).
  useRpAll(service_name, &authority_pk, &g, &h);        // a no-op when the parameters are the installed ones (ElpKey::useRp compares by value)
  // overlapped path: every proof well-formed with one hidden pattern; the batch is cut into one contiguous shard per context
  const size_t A = m_key->attrs(), n = proofs.size(), N = m_set->size();
  bool uniform = n > 0;
  uint64_t mask = n ? elpHiddenMask(proofs[0].attributes) : 0;
  const size_t H = (size_t)__builtin_popcountll(mask);
  for (size_t i = 0; uniform && i < n; i++) {
    const IdProof& p = proofs[i];
    uniform = p.E1.has_value() && p.E2.has_value() && p.attributes.size() == A && p.rs.size() == H + 2 && H >= 2 && elpHiddenMask(p.attributes) == mask;
  }
  std::vector<bool> ready;
  bool sync_done = false;
  if (!uniform) {
    ready = verifyIdImpl(proofs, ads, true);
    sync_done = true;
  } else {
    const size_t S1 = G1::size(), S2 = G2::size();
    const size_t rsz = elp_verify_id_record_size(curveId(), (int)A, (int)H, 1);
    if (sl.shard.size() != N) sl.shard = std::vector<Stage::ShardBuf>(N);
    // page-locked staging of every shard from its own context (grown before anything is queued)
    std::vector<uint8_t*> recs(N), adbuf(N), flags(N);
    std::vector<uint32_t*> adoff(N);
    std::vector<size_t> first(N), count(N);
    for (size_t r = 0; r < N; r++) {
      ElpShardSet::range(n, r, N, first[r], count[r]);
      elp_ctx* ctx = m_set->key(r).ctx();
      const size_t c = count[r];
      size_t adbytes = 0;
      for (size_t j = 0; j < c; j++) adbytes += ads[first[r] + j].size();
      recs[r] = sl.shard[r].recs.get(ctx, (c ? c : 1) * rsz);
      adoff[r] = (uint32_t*)sl.shard[r].offs.get(ctx, (c + 1) * 4);
      adbuf[r] = sl.shard[r].ads.get(ctx, adbytes ? adbytes : 1);
      flags[r] = sl.shard[r].flags.get(ctx, c ? c : 1);
      adoff[r][0] = 0;
      for (size_t j = 0; j < c; j++) adoff[r][j + 1] = adoff[r][j] + (uint32_t)ads[first[r] + j].size();
    }
    // shard by shard: pack on all host threads, queue copies + kernel on the shard's context (asynchronous), go on packing the next shard while the
    // first ones already verify
    std::vector<char> queued(N, 0);
    try {
      for (size_t r = 0; r < N; r++) {
        if (count[r] == 0) continue;
        elpParallelFor(count[r], 512, [&](size_t lo, size_t hi) {
          for (size_t j = lo; j < hi; j++) {
            packRecord(recs[r] + j * rsz, proofs[first[r] + j], true, S1, S2);
            const std::string& ad = ads[first[r] + j];
            if (!ad.empty()) memcpy(adbuf[r] + adoff[r][j], ad.data(), ad.size());
          }
        });
        elp_ctx* ctx = m_set->key(r).ctx();
        elpCheck(ctx, elp_verify_id_batch_submit(ctx, si, count[r], recs[r], mask, 1, adbuf[r], adoff[r], 0, flags[r]), "elp_verify_id_batch_submit");
        queued[r] = 1;
      }
    } catch (...) {
      for (size_t r = 0; r < N; r++)          // drain what was queued: the slot of those contexts must be idle again
        if (queued[r]) (void)elp_verify_id_batch_wait(m_set->key(r).ctx(), si, nullptr);
      throw;
    }
    sl.first = first;
    sl.count = count;
  }
  st.rp_sig = sig;
  sl.ready = std::move(ready);
  sl.sync_done = sync_done;
  sl.n = n;
  sl.ticket = st.next_ticket++;
  sl.busy = true;
  return sl.ticket;
}
std::vector<bool> PSVerifier::el_passo_verify_id_collect(size_t ticket) const {
  std::lock_guard<std::mutex> lock(m_stage->mu);
  Stage& st = *m_stage;
  int si = st.slot[0].busy && st.slot[0].ticket == ticket ? 0 : (st.slot[1].busy && st.slot[1].ticket == ticket ? 1 : -1);
  if (si < 0) throw std::runtime_error("el_passo_verify_id_collect: unknown ticket");
  Stage::Slot& sl = st.slot[si];
  sl.busy = false;
  if (sl.sync_done) return std::move(sl.ready);
  std::vector<bool> out(sl.n);
  std::exception_ptr err;
  for (size_t r = 0; r < sl.shard.size(); r++) {          // every shard is waited for, whatever the others returned: the slot is idle afterwards
    if (sl.count[r] == 0) continue;
    elp_ctx* ctx = m_set->key(r).ctx();
    uint64_t acc = 0;
    try {
      elpCheck(ctx, elp_verify_id_batch_wait(ctx, si, &acc), "elp_verify_id_batch_wait");
    } catch (...) {
      if (!err) err = std::current_exception();
      continue;
    }
    const uint8_t* flags = sl.shard[r].flags.get(ctx, sl.count[r]);        // the same block: get() only grows
    for (size_t j = 0; j < sl.count[r]; j++) out[sl.first[r] + j] = flags[j] != 0;
  }
  if (err) std::rethrow_exception(err);
  return out;
}

std::vector<bool> PSVerifier::el_passo_verify_id_without_id_retrieval_batch(const std::vector<IdProof>& proofs,
                                                                            const std::vector<std::string>& ads,
                                                                            const std::string& service_name) const {
  if (ads.size() != proofs.size()) throw std::runtime_error("associated data count does not match");
  std::lock_guard<std::mutex> lock(m_stage->mu);
  useRpAll(service_name, nullptr, nullptr, nullptr);
  return verifyIdImpl(proofs, ads, false);
}
std::vector<bool> PSVerifier::el_passo_verify_id_wire_batch(const std::vector<PSBuffer>& messages, const std::vector<std::string>& ads,
                                                            const std::string& service_name, const G1* authority_pk, const G1* g,
                                                            const G1* h) const {
  if (ads.size() != messages.size()) throw std::runtime_error("associated data count does not match");
  const bool retr = authority_pk != nullptr;
  // one batch call at a time per verifier, from the installation of the relying-party parameters to the last shard: two threads with different service
  // names / RP keys would otherwise race on ElpKey::rp_* and on elp_set_rp of the shared contexts and verify against each other's parameters
  std::lock_guard<std::mutex> lock(m_stage->mu);
  useRpAll(service_name, authority_pk, g, h);
  const size_t n = messages.size();
  std::vector<uint32_t> moff(n + 1, 0), adoff(n + 1, 0);
  for (size_t i = 0; i < n; i++) {
    moff[i + 1] = moff[i] + (uint32_t)messages[i].size();
    adoff[i + 1] = adoff[i] + (uint32_t)ads[i].size();
  }
  uint8_t* const buf = m_stage->recs.get(m_key->ctx(), moff[n] ? moff[n] : 1);
  uint8_t* const adbuf = m_stage->ads.get(m_key->ctx(), adoff[n] ? adoff[n] : 1);
  uint8_t* const flags = m_stage->flags.get(m_key->ctx(), n ? n : 1);
  elpParallelFor(n, 1024, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      if (!messages[i].empty()) memcpy(buf + moff[i], messages[i].data(), messages[i].size());
      if (!ads[i].empty()) memcpy(adbuf + adoff[i], ads[i].data(), ads[i].size());
    }
  });
  m_set->forEachShard(n, [&](size_t r, size_t first, size_t count) {
    if (count == 0) return;
    std::vector<uint32_t> mo(count + 1), ao(count + 1);
    for (size_t j = 0; j <= count; j++) {
      mo[j] = moff[first + j] - moff[first];
      ao[j] = adoff[first + j] - adoff[first];
    }
    uint64_t acc = 0;
    elp_ctx* ctx = m_set->key(r).ctx();
    elpCheck(ctx, elp_verify_id_wire_batch(ctx, count, buf + moff[first], mo.data(), retr ? 1 : 0, adbuf + adoff[first], ao.data(), 0,
                                           flags + first, &acc),
             "elp_verify_id_wire_batch");
  });
  std::vector<bool> out(n);
  for (size_t i = 0; i < n; i++) out[i] = flags[i] != 0;
  return out;
}
uint64_t PSVerifier::el_passo_verify_id_wire_packed(const uint8_t* messages, const uint32_t* offsets, size_t n, const std::string& ad,
                                                    const std::string& service_name, uint8_t* flags, const G1* authority_pk, const G1* g,
                                                    const G1* h) const {
  const bool retr = authority_pk != nullptr;
  std::lock_guard<std::mutex> lock(m_stage->mu);            // as above: RP parameters and contexts belong to this call until its shards are done
  useRpAll(service_name, authority_pk, g, h);
  std::vector<uint64_t> acc(m_set->size(), 0);
  const uint8_t zero = 0;
  m_set->forEachShard(n, [&](size_t r, size_t first, size_t count) {
    if (count == 0) return;
    std::vector<uint32_t> mo(count + 1);
    for (size_t j = 0; j <= count; j++) mo[j] = offsets[first + j] - offsets[first];
    elp_ctx* ctx = m_set->key(r).ctx();
    elpCheck(ctx, elp_verify_id_wire_batch(ctx, count, messages + offsets[first], mo.data(), retr ? 1 : 0, ad.empty() ? &zero : (const uint8_t*)ad.data(),
                                           nullptr, ad.size(), flags + first, &acc[r]),
             "elp_verify_id_wire_batch");
  });
  uint64_t total = 0;
  for (uint64_t a : acc) total += a;         // the only reduction across shards: the accepted counts
  return total;
}

bool PSVerifier::el_passo_verify_id(const IdProof& proof, const std::string& associated_data, const std::string& service_name,
                                    const G1& authority_pk, const G1& g, const G1& h) const {
  return el_passo_verify_id_batch({proof}, {associated_data}, service_name, authority_pk, g, h)[0];
}
bool PSVerifier::el_passo_verify_id_without_id_retrieval(const IdProof& proof, const std::string& associated_data,
                                                         const std::string& service_name) const {
  return el_passo_verify_id_without_id_retrieval_batch({proof}, {associated_data}, service_name)[0];
}
std::string PSVerifier::get_user_name_from_signon_request(const IdProof& proof) { return proof.phi.getStr(); }
