// PSVerifier: same public interface as the reference's src/ps-verifier.h:11-71, evaluated on the GPU
// (fused kernels behind elp_verify_id_batch / elp_ps_verify_batch), plus batch entry points.
#ifndef ELP_HOST_PS_VERIFIER_H_
#define ELP_HOST_PS_VERIFIER_H_

#include <memory>
#include <mutex>

#include "elp_key.h"
#include "ps-encoding.h"

class PSVerifier {
 public:
  PSVerifier(const PSPubKey& pk);                       // the reference's constructor (src/ps-verifier.h:18): device and table width from elpSetDefaults
  // one context per entry of `devices` (ordinals may repeat); batches are cut into contiguous shards, one host thread + one HIP stream per context
  // (SURVEY.md section 8e); window_bits < 0 = the process default
  PSVerifier(const PSPubKey& pk, const std::vector<int>& devices, int window_bits = -1);
  size_t contexts() const { return m_set->size(); }
  // elp_set_option (include/elpasso.h) on context `shard`, or on every context of this verifier (shard < 0)
  void set_option(int option, int value, int shard = -1) const;

  bool verify(const PSCredential& sig, const std::vector<std::string>& all_attributes) const;

  bool el_passo_verify_id(const IdProof& proof, const std::string& associated_data, const std::string& service_name,
                          const G1& authority_pk, const G1& g, const G1& h) const;

  bool el_passo_verify_id_without_id_retrieval(const IdProof& proof, const std::string& associated_data,
                                               const std::string& service_name) const;

  static std::string get_user_name_from_signon_request(const IdProof& proof);

  // ---- batch entry points (new): one verdict per proof, each with its own associated data
  std::vector<bool> el_passo_verify_id_batch(const std::vector<IdProof>& proofs, const std::vector<std::string>& associated_data,
                                             const std::string& service_name, const G1& authority_pk, const G1& g,
                                             const G1& h) const;
  std::vector<bool> el_passo_verify_id_without_id_retrieval_batch(const std::vector<IdProof>& proofs,
                                                                  const std::vector<std::string>& associated_data,
                                                                  const std::string& service_name) const;
  // wire messages (IdProof::toBufferString bytes) verified without host-side decoding: T-L-V parsing, point decompression and
  // attribute hashing run on the GPU (elp_verify_id_wire_batch); the hidden pattern may differ per message
  std::vector<bool> el_passo_verify_id_wire_batch(const std::vector<PSBuffer>& messages, const std::vector<std::string>& associated_data,
                                                  const std::string& service_name, const G1* authority_pk = nullptr,
                                                  const G1* g = nullptr, const G1* h = nullptr) const;
  // the same on messages that are already contiguous (message i = bytes [offsets[i], offsets[i+1]) of `messages`): no per-message copy on the host.
  // `flags` receives one byte per message; returns the number of accepted messages.  One associated-data string for the whole batch.
  uint64_t el_passo_verify_id_wire_packed(const uint8_t* messages, const uint32_t* offsets, size_t n, const std::string& associated_data,
                                          const std::string& service_name, uint8_t* flags, const G1* authority_pk = nullptr, const G1* g = nullptr,
                                          const G1* h = nullptr) const;
  std::vector<bool> verify_batch(const std::vector<PSCredential>& sigs, const std::vector<std::vector<std::string>>& all_attributes) const;

  // ---- pipelined form of el_passo_verify_id_batch (round 4; several contexts since round 5): submit() packs the batch into page-locked staging, queues
  // copies + kernel and returns a ticket; collect(ticket) waits for it and returns the verdicts.  Up to TWO batches in flight: while the GPUs verify one,
  // the host packs the next and its records travel over PCIe -- a steady stream of batches costs the kernel time alone (elp_verify_id_batch_submit / _wait).
  // A verifier with several contexts (PSVerifier(pk, devices, W): the one-process form of the multi-GPU split) cuts every batch into one contiguous shard
  // per context, each with its own staging, copy stream and slot pair; shard r is queued as soon as it is packed.  Condition for the overlapped path:
  // every proof of the batch well-formed with the same hidden pattern; anything else is verified synchronously inside submit() and merely handed out by
  // collect().  The relying-party parameters must be the same for all batches in flight (a change throws).  A submit() that throws leaves no trace: no
  // ticket, the slot free.  Tickets are collected in submission order.
  size_t el_passo_verify_id_submit(const std::vector<IdProof>& proofs, const std::vector<std::string>& associated_data, const std::string& service_name,
                                   const G1& authority_pk, const G1& g, const G1& h) const;
  std::vector<bool> el_passo_verify_id_collect(size_t ticket) const;

 private:
  std::vector<bool> verifyIdImpl(const std::vector<IdProof>& proofs, const std::vector<std::string>& ads, bool retrieval) const;
  bool verifyIdStaged(const std::vector<IdProof>& proofs, const std::vector<std::string>& ads, bool retr, std::vector<bool>& out) const;
  void useRpAll(const std::string& service, const G1* authority_pk, const G1* g, const G1* h) const;
  PSPubKey m_pk;
  std::shared_ptr<ElpShardSet> m_set;    // the contexts of this verifier (one unless constructed with a device list)
  std::shared_ptr<ElpKey> m_key;         // = context 0
  // page-locked staging of the batch entry points (records / messages, associated data, verdicts); one batch call at a time per verifier
  struct Stage {
    std::mutex mu;
    ElpPinned recs, ads, flags;
    // the two slots of the pipelined form
    struct ShardBuf {
      ElpPinned recs, ads, offs, flags;   // staging of one shard, page-locked through the shard's own context
    };
    struct Slot {
      std::vector<ShardBuf> shard;       // one per context
      std::vector<size_t> first, count;  // the shards of the batch in flight
      bool busy = false;
      size_t ticket = 0, n = 0;
      std::vector<bool> ready;         // verdicts of a batch that took the synchronous path
      bool sync_done = false;
    } slot[2];
    size_t next_ticket = 1;
    std::string rp_sig;                // relying-party parameters of the batches in flight
  };
  std::shared_ptr<Stage> m_stage;
};

#endif  // ELP_HOST_PS_VERIFIER_H_
