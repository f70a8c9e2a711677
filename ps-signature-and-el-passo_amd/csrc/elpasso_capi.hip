// C-ABI entry points of include/elpasso.h: curve dispatch onto the per-curve instantiations.
#include "elpasso_impl.h"

extern template int elp_verify_id_batch_aggregated_dev_t<BN254>(elp_ctx* c, void* stream_, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, const uint8_t* seed32, void* d_flags, void* d_accepted);
extern template int elp_verify_id_batch_aggregated_dev_t<BLS12_381>(elp_ctx* c, void* stream_, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, const uint8_t* seed32, void* d_flags, void* d_accepted);
extern template int msm_impl_t<BN254, 1>(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out);
extern template int msm_impl_t<BN254, 2>(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out);
extern template int msm_impl_t<BLS12_381, 1>(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out);
extern template int msm_impl_t<BLS12_381, 2>(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out);
extern template int msm_dev_t<BN254, 1>(elp_ctx* c, void* stream_, size_t n, const void* d_points, const void* d_scalars, void* d_workspace, void* d_out);
extern template int msm_dev_t<BN254, 2>(elp_ctx* c, void* stream_, size_t n, const void* d_points, const void* d_scalars, void* d_workspace, void* d_out);
extern template int msm_dev_t<BLS12_381, 1>(elp_ctx* c, void* stream_, size_t n, const void* d_points, const void* d_scalars, void* d_workspace, void* d_out);
extern template int msm_dev_t<BLS12_381, 2>(elp_ctx* c, void* stream_, size_t n, const void* d_points, const void* d_scalars, void* d_workspace, void* d_out);
extern template int elp_set_pubkey_t<BN254>(elp_ctx* c, int nattr, const uint8_t* g, const uint8_t* gg, const uint8_t* XX, const uint8_t* Yi, const uint8_t* YYi, int window_bits);
extern template int elp_set_rp_t<BN254>(elp_ctx* c, const uint8_t* service_name, size_t service_len, const uint8_t* authority_pk, const uint8_t* g, const uint8_t* h);
extern template int elp_set_signer_secret_t<BN254>(elp_ctx* c, const uint8_t* X);
extern template int decompress_impl_t<BN254, 1>(elp_ctx* c, size_t n, const uint8_t* wire, uint8_t* out, uint8_t* ok);
extern template int decompress_impl_t<BN254, 2>(elp_ctx* c, size_t n, const uint8_t* wire, uint8_t* out, uint8_t* ok);
extern template int mul_impl_t<BN254, 1>(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out);
extern template int mul_impl_t<BN254, 2>(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out);
extern template int add_impl_t<BN254, 1>(elp_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out);
extern template int add_impl_t<BN254, 2>(elp_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out);
extern template int msm_fixed_impl_t<BN254, 1>(elp_ctx* c, size_t n, int nterms, const int32_t* ids, const uint8_t* ks, uint8_t* out);
extern template int msm_fixed_impl_t<BN254, 2>(elp_ctx* c, size_t n, int nterms, const int32_t* ids, const uint8_t* ks, uint8_t* out);
extern template int elp_hash_to_g1_t<BN254>(elp_ctx* c, size_t n, const uint8_t* msgs, const uint32_t* off, uint8_t* out);
extern template int elp_pairing_t<BN254>(elp_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt);
extern template int elp_pairing_check_t<BN254>(elp_ctx* c, size_t n, int npairs, const uint8_t* g1, const uint8_t* g2, uint8_t* ok);
extern template int elp_verify_id_batch_dev_t<BN254>(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted);
extern template int elp_verify_id_wire_batch_dev_t<BN254>(elp_ctx* c, void* stream, size_t n, const void* d_msgs, const void* d_msg_off, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted);
extern template int elp_ps_verify_batch_dev_t<BN254>(elp_ctx* c, void* stream, size_t n, const void* d_records, int nattr, void* d_flags, void* d_accepted);
extern template int elp_prove_id_batch_dev_t<BN254>(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_proofs, void* d_flags, void* d_accepted);
extern template int elp_request_id_batch_dev_t<BN254>(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_requests);
extern template int elp_provide_id_batch_dev_t<BN254>(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_sigs, void* d_flags, void* d_accepted);
extern template int elp_provide_id_batch_t<BN254>(elp_ctx* c, size_t n, const uint8_t* records, uint64_t mask, const uint8_t* ad, const uint32_t* ad_off, size_t ad_len, uint8_t* sigs, uint8_t* flags, uint64_t* accepted);
extern template int elp_bench_op_t<BN254>(elp_ctx* c, int op, size_t lanes, int iters, float* ms);
extern template int elp_bench_fp_mul_t<BN254>(elp_ctx* c, size_t lanes, int iters, float* ms);
extern template int elp_set_pubkey_t<BLS12_381>(elp_ctx* c, int nattr, const uint8_t* g, const uint8_t* gg, const uint8_t* XX, const uint8_t* Yi, const uint8_t* YYi, int window_bits);
extern template int elp_set_rp_t<BLS12_381>(elp_ctx* c, const uint8_t* service_name, size_t service_len, const uint8_t* authority_pk, const uint8_t* g, const uint8_t* h);
extern template int elp_set_signer_secret_t<BLS12_381>(elp_ctx* c, const uint8_t* X);
extern template int decompress_impl_t<BLS12_381, 1>(elp_ctx* c, size_t n, const uint8_t* wire, uint8_t* out, uint8_t* ok);
extern template int decompress_impl_t<BLS12_381, 2>(elp_ctx* c, size_t n, const uint8_t* wire, uint8_t* out, uint8_t* ok);
extern template int mul_impl_t<BLS12_381, 1>(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out);
extern template int mul_impl_t<BLS12_381, 2>(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out);
extern template int add_impl_t<BLS12_381, 1>(elp_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out);
extern template int add_impl_t<BLS12_381, 2>(elp_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out);
extern template int msm_fixed_impl_t<BLS12_381, 1>(elp_ctx* c, size_t n, int nterms, const int32_t* ids, const uint8_t* ks, uint8_t* out);
extern template int msm_fixed_impl_t<BLS12_381, 2>(elp_ctx* c, size_t n, int nterms, const int32_t* ids, const uint8_t* ks, uint8_t* out);
extern template int elp_hash_to_g1_t<BLS12_381>(elp_ctx* c, size_t n, const uint8_t* msgs, const uint32_t* off, uint8_t* out);
extern template int elp_pairing_t<BLS12_381>(elp_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt);
extern template int elp_pairing_check_t<BLS12_381>(elp_ctx* c, size_t n, int npairs, const uint8_t* g1, const uint8_t* g2, uint8_t* ok);
extern template int elp_verify_id_batch_dev_t<BLS12_381>(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted);
extern template int elp_verify_id_wire_batch_dev_t<BLS12_381>(elp_ctx* c, void* stream, size_t n, const void* d_msgs, const void* d_msg_off, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted);
extern template int elp_ps_verify_batch_dev_t<BLS12_381>(elp_ctx* c, void* stream, size_t n, const void* d_records, int nattr, void* d_flags, void* d_accepted);
extern template int elp_prove_id_batch_dev_t<BLS12_381>(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_proofs, void* d_flags, void* d_accepted);
extern template int elp_request_id_batch_dev_t<BLS12_381>(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_requests);
extern template int elp_provide_id_batch_dev_t<BLS12_381>(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_sigs, void* d_flags, void* d_accepted);
extern template int elp_provide_id_batch_t<BLS12_381>(elp_ctx* c, size_t n, const uint8_t* records, uint64_t mask, const uint8_t* ad, const uint32_t* ad_off, size_t ad_len, uint8_t* sigs, uint8_t* flags, uint64_t* accepted);
extern template int elp_bench_op_t<BLS12_381>(elp_ctx* c, int op, size_t lanes, int iters, float* ms);
extern template int elp_bench_fp_mul_t<BLS12_381>(elp_ctx* c, size_t lanes, int iters, float* ms);


// ELP_OPT_PAIR16 default (elp_init): on -- 4 096 PS verifications 2.77 -> 2.02 ms, 64: 1.88 -> 1.73 ms (profiles/r06_pair16.md)
#ifndef ELP_PAIR16_DEFAULT
#define ELP_PAIR16_DEFAULT 1
#endif

// (definitions below get C linkage from their declarations in include/elpasso.h)

const char* elp_version(void) { return "elpasso-hip 0.1 (gfx950)"; }

int elp_field_bytes(int curve) { return curve == ELP_CURVE_BN254 ? 32 : curve == ELP_CURVE_BLS12_381 ? 48 : 0; }

int elp_device_count(void) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return ndev;
}

int elp_init(int curve, int device, elp_ctx** out) {
  if (!out) return ELP_ERR_ARG;
  *out = nullptr;
  if (curve != ELP_CURVE_BN254 && curve != ELP_CURVE_BLS12_381) return ELP_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ELP_ERR_NODEVICE;
  if (device < 0 || device >= ndev) return ELP_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return ELP_ERR_NODEVICE;
  elp_ctx* c = new elp_ctx();
  c->curve = curve;
  c->device = device;
  if (const char* e = getenv("ELP_LAYOUT")) c->paired = !strcmp(e, "plain") ? 0 : !strcmp(e, "paired") ? 1 : 2;     // A/B runs
  if (const char* e = getenv("ELP_SPLIT")) c->split = atoi(e);
  if (const char* e = getenv("ELP_PHASE_MIX")) c->phase_mix = atoi(e);                                                      // A/B runs: KEY_PHASE_MIX
  if (const char* e = getenv("ELP_SMALL_DENSE_FROM")) c->small_dense_from = (size_t)atol(e);                               // A/B runs: k_vid_small2 above this many items
  if (const char* e = getenv("ELP_PAIR4_TWO_LAUNCHES")) c->mid_two_launches = atoi(e);                                      // A/B runs: the mid-size path as two launches
  if (const char* e = getenv("ELP_AGG_PAIRED")) c->agg_paired = atoi(e) != 0;                                            // A/B runs: aggregated main kernel on lane pairs (BLS12-381)
  if (const char* e = getenv("ELP_AGG_TWO")) c->agg_two = atoi(e) < 0 ? 0 : (atoi(e) > 2 ? 2 : atoi(e));      // A/B runs: ELP_OPT_AGG_TWO_PER_LANE from the environment
  if (const char* e = getenv("ELP_SMALL_ONE_MAX")) c->small_one_max = (size_t)atol(e);                                    // A/B runs: one launch (k_vid_small) up to this many items
  if (const char* e = getenv("ELP_OVERLAP")) c->overlap = atoi(e) != 0;                                                   // A/B runs: second-stream overlap inside a call
  if (const char* e = getenv("ELP_STAGE")) c->stage_records = atoi(e) != 0;                                              // A/B runs: coalesced record loads
  if (const char* e = getenv("ELP_COOP")) c->coop = atoi(e) != 0;                                                      // A/B runs: cooperative pairing for small batches                                            // A/B runs: one fused kernel per verification
  c->pair16 = ELP_PAIR16_DEFAULT;
  if (curve == ELP_CURVE_BLS12_381)        // measured (profiles/r06_pair16.md): the interpreter serves up to 2 048 items in one round of 4.8 ms, the rows take 5.0-5.25 ms for any size up to 4 096
    c->pair16_min = 2049;                  // (the closing step of aggregated verification runs on a row on both curves: 2.21 against 2.63 ms on this one)
  if (const char* e = getenv("ELP_PAIR16_MIN")) c->pair16_min = (size_t)atol(e);                                       // A/B runs
  if (const char* e = getenv("ELP_PAIR16_TAIL")) c->pair16_tail = atoi(e) != 0;                                        // A/B runs: the closing step of aggregated verification on one row
  if (const char* e = getenv("ELP_PAIR16")) c->pair16 = atoi(e) != 0;                                                  // A/B runs: the row-of-16 pairing check for small PS batches
  if (const char* e = getenv("ELP_VTAB")) c->use_vtab = strcmp(e, "0") != 0;                                         // A/B runs: tables of multiples in private memory
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->simds = 4 * prop.multiProcessorCount;
  if (hipStreamCreate(&c->stream) != hipSuccess) {
    delete c;
    return ELP_ERR_HIP;
  }
  {
    std::lock_guard<std::mutex> g(dev_cache().mu);
    dev_cache().live_ctx++;
  }
  *out = c;
  return ELP_OK;
}

void elp_destroy(elp_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  free_key(c);
  for (int i = 0; i < elp_ctx::NPIPE; i++)
    if (c->pstream[i]) (void)hipStreamDestroy(c->pstream[i]);
  bool last;
  {
    std::lock_guard<std::mutex> g(dev_cache().mu);
    last = --dev_cache().live_ctx == 0;
  }
  if (last) dev_cache().trim();
  if (c->agg_ws) (void)hipFree(c->agg_ws);
  for (auto& w : c->vtab_ws)
    if (w.p) (void)hipFree(w.p);
  if (c->agg_ok) (void)hipFree(c->agg_ok);
  for (auto& w : c->agg_parked) {
    if (w.ws) (void)hipFree(w.ws);
    if (w.ok) (void)hipFree(w.ok);
  }
  for (auto& a : c->aslot) {
    for (void* q : {a.drec, a.dad, a.doff, a.dfl, a.dcnt})
      if (q) (void)hipFree(q);
    if (a.copied) (void)hipEventDestroy(a.copied);
    if (a.done) (void)hipEventDestroy(a.done);
    if (a.h_cnt) (void)hipHostFree(a.h_cnt);
  }
  if (c->coop_consts) (void)hipFree(c->coop_consts);
  for (auto& e : c->wire_ws)
    if (e.p) (void)hipFree(e.p);
  if (c->wire_mask_host) (void)hipHostFree(c->wire_mask_host);
  if (c->jstream) (void)hipStreamDestroy(c->jstream);
  for (int i = 0; i < 4; i++)
    if (c->jev[i]) (void)hipEventDestroy(c->jev[i]);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* elp_last_error(const elp_ctx* c) { return c ? c->err.c_str() : "null context"; }
int elp_set_option(elp_ctx* c, int option, int value) {
  if (!c) return ELP_ERR_ARG;
  switch (option) {
    case ELP_OPT_STRICT_SIGNATURE: c->strict_sig = value ? 1 : 0; return ELP_OK;
    case ELP_OPT_PAIRED_LAYOUT:
      if (value < 0 || value > 2) return ELP_ERR_ARG;
      c->paired = value;
      return ELP_OK;
    case ELP_OPT_TABLE_WORKSPACE: c->use_vtab = value ? 1 : 0; return ELP_OK;
    case ELP_OPT_COOP_PAIRING:
      if (value < 0) return ELP_ERR_ARG;
      c->coop = value ? 1 : 0;
      c->coop_max = value > 1 ? (size_t)value : 4096;     // 1 = the documented default limits again (also after an earlier larger value)
      c->vid_coop_max = value > 1 ? (size_t)value : 0;
      return ELP_OK;
    case ELP_OPT_SUBGROUP_CHECK: c->subgroup_check = value ? 1 : 0; return ELP_OK;
    case ELP_OPT_COALESCED_RECORDS: c->stage_records = value ? 1 : 0; return ELP_OK;
    case ELP_OPT_STREAM_OVERLAP: c->overlap = value ? 1 : 0; return ELP_OK;
    case ELP_OPT_FAULT_INJECT: c->fail_submits = value > 0 ? value : 0; return ELP_OK;
    case ELP_OPT_WIRE_DECODE: c->wire_decode = value ? 1 : 0; return ELP_OK;
    case ELP_OPT_AGG_TWO_PER_LANE:
      if (value < 0 || value > 2) return ELP_ERR_ARG;
      c->agg_two = value;
      return ELP_OK;
    case ELP_OPT_PAIR4:
      if (value < 0 || value > 2) return ELP_ERR_ARG;
      c->pair4 = value;
      return ELP_OK;
    case ELP_OPT_PAIR16:
      if (value < 0) return ELP_ERR_ARG;
      c->pair16 = value ? 1 : 0;
      c->pair16_max = value > 1 ? (size_t)value : 4096;
      return ELP_OK;
    case ELP_OPT_SPLIT_PHASES:
      if (value < 0 || value > 3) return ELP_ERR_ARG;
      c->split = value;
      return ELP_OK;
    default: return ELP_ERR_ARG;
  }
}
int elp_set_pubkey(elp_ctx* c, int nattr, const uint8_t* g, const uint8_t* gg, const uint8_t* XX, const uint8_t* Yi,
                   const uint8_t* YYi, int window_bits) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_set_pubkey_t<BN254>(c, nattr, g, gg, XX, Yi, YYi, window_bits) : elp_set_pubkey_t<BLS12_381>(c, nattr, g, gg, XX, Yi, YYi, window_bits);
}
int elp_set_rp(elp_ctx* c, const uint8_t* service_name, size_t service_len, const uint8_t* authority_pk, const uint8_t* g,
               const uint8_t* h) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_set_rp_t<BN254>(c, service_name, service_len, authority_pk, g, h) : elp_set_rp_t<BLS12_381>(c, service_name, service_len, authority_pk, g, h);
}
int elp_set_signer_secret(elp_ctx* c, const uint8_t* X) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_set_signer_secret_t<BN254>(c, X) : elp_set_signer_secret_t<BLS12_381>(c, X);
}
int elp_g1_decompress(elp_ctx* c, size_t n, const uint8_t* wire, uint8_t* out, uint8_t* ok) {
  return decompress_impl<1>(c, n, wire, out, ok);
}
int elp_g2_decompress(elp_ctx* c, size_t n, const uint8_t* wire, uint8_t* out, uint8_t* ok) {
  return decompress_impl<2>(c, n, wire, out, ok);
}
int elp_g1_mul(elp_ctx* c, size_t n, const uint8_t* p, const uint8_t* k, uint8_t* o) { return mul_impl<1>(c, n, p, k, o); }
int elp_g2_mul(elp_ctx* c, size_t n, const uint8_t* p, const uint8_t* k, uint8_t* o) { return mul_impl<2>(c, n, p, k, o); }
int elp_g1_add(elp_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* o) { return add_impl<1>(c, n, a, b, o); }
int elp_g2_add(elp_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* o) { return add_impl<2>(c, n, a, b, o); }
int elp_g1_msm_fixed(elp_ctx* c, size_t n, int nt, const int32_t* ids, const uint8_t* ks, uint8_t* o) {
  return msm_fixed_impl<1>(c, n, nt, ids, ks, o);
}
int elp_g2_msm_fixed(elp_ctx* c, size_t n, int nt, const int32_t* ids, const uint8_t* ks, uint8_t* o) {
  return msm_fixed_impl<2>(c, n, nt, ids, ks, o);
}
int elp_hash_to_g1(elp_ctx* c, size_t n, const uint8_t* msgs, const uint32_t* off, uint8_t* out) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_hash_to_g1_t<BN254>(c, n, msgs, off, out) : elp_hash_to_g1_t<BLS12_381>(c, n, msgs, off, out);
}
int elp_pairing(elp_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_pairing_t<BN254>(c, n, g1, g2, gt) : elp_pairing_t<BLS12_381>(c, n, g1, g2, gt);
}
int elp_pairing_check(elp_ctx* c, size_t n, int npairs, const uint8_t* g1, const uint8_t* g2, uint8_t* ok) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_pairing_check_t<BN254>(c, n, npairs, g1, g2, ok) : elp_pairing_check_t<BLS12_381>(c, n, npairs, g1, g2, ok);
}
size_t elp_verify_id_record_size(int curve, int A, int H, int retr) {
  if (curve == ELP_CURVE_BLS12_381) return 4 * (size_t)verify_id_record_words<BLS12_381>(A, H, retr != 0);
  if (curve != ELP_CURVE_BN254) return 0;
  return 4 * (size_t)verify_id_record_words<BN254>(A, H, retr != 0);
}
size_t elp_prove_id_record_size(int curve, int A, int H, int retr) {
  if (curve == ELP_CURVE_BLS12_381) return 4 * (size_t)prove_id_record_words<BLS12_381>(A, H, retr != 0);
  if (curve != ELP_CURVE_BN254) return 0;
  return 4 * (size_t)prove_id_record_words<BN254>(A, H, retr != 0);
}
size_t elp_request_id_record_size(int curve, int A, int H) {
  if (curve != ELP_CURVE_BLS12_381 && curve != ELP_CURVE_BN254) return 0;
  return 4 * (size_t)request_id_record_words<BN254>(A, H);
}
size_t elp_request_id_out_size(int curve, int H) {
  if (curve == ELP_CURVE_BLS12_381) return 4 * (size_t)request_id_out_words<BLS12_381>(H);
  if (curve != ELP_CURVE_BN254) return 0;
  return 4 * (size_t)request_id_out_words<BN254>(H);
}
size_t elp_ps_verify_record_size(int curve, int A) {
  if (curve == ELP_CURVE_BLS12_381) return 4 * (size_t)(4 * BLS12_381::N + 8 * A);
  if (curve != ELP_CURVE_BN254) return 0;
  return 4 * (size_t)(4 * BN254::N + 8 * A);
}
size_t elp_provide_id_record_size(int curve, int A, int H) {
  if (curve == ELP_CURVE_BLS12_381) return 4 * (size_t)provide_id_record_words<BLS12_381>(A, H);
  if (curve != ELP_CURVE_BN254) return 0;
  return 4 * (size_t)provide_id_record_words<BN254>(A, H);
}
int elp_verify_id_batch_dev(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad,
                            const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_verify_id_batch_dev_t<BN254>(c, stream, n, d_records, mask, retr, d_ad, d_ad_off, ad_len, d_flags, d_accepted) : elp_verify_id_batch_dev_t<BLS12_381>(c, stream, n, d_records, mask, retr, d_ad, d_ad_off, ad_len, d_flags, d_accepted);
}
int elp_verify_id_wire_batch_dev(elp_ctx* c, void* stream, size_t n, const void* d_msgs, const void* d_msg_off, int retr,
                                 const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254
             ? elp_verify_id_wire_batch_dev_t<BN254>(c, stream, n, d_msgs, d_msg_off, retr, d_ad, d_ad_off, ad_len, d_flags, d_accepted)
             : elp_verify_id_wire_batch_dev_t<BLS12_381>(c, stream, n, d_msgs, d_msg_off, retr, d_ad, d_ad_off, ad_len, d_flags, d_accepted);
}
int elp_ps_verify_batch_dev(elp_ctx* c, void* stream, size_t n, const void* d_records, int nattr, void* d_flags, void* d_accepted) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_ps_verify_batch_dev_t<BN254>(c, stream, n, d_records, nattr, d_flags, d_accepted) : elp_ps_verify_batch_dev_t<BLS12_381>(c, stream, n, d_records, nattr, d_flags, d_accepted);
}
int elp_provide_id_batch_dev(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, const void* d_ad,
                             const void* d_ad_off, size_t ad_len, void* d_sigs, void* d_flags, void* d_accepted) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_provide_id_batch_dev_t<BN254>(c, stream, n, d_records, mask, d_ad, d_ad_off, ad_len, d_sigs, d_flags, d_accepted) : elp_provide_id_batch_dev_t<BLS12_381>(c, stream, n, d_records, mask, d_ad, d_ad_off, ad_len, d_sigs, d_flags, d_accepted);
}

// Host buffers in, flags out (the path PSVerifier::el_passo_verify_id_batch takes).  Device blocks persist across calls (DevBlockCache).  The
// batch is processed in rounds of 64 x SIMDs items: the records of round r + 1 are copied on a copy stream while the kernel of round r runs on
// the context stream (the kernels themselves are NOT spread over several streams: each dispatch needs ~1 GB of private memory, and concurrent
// dispatches of that size serialise on its allocation -- measured 35 ms against 23 ms for one round of 65 536 items).
int elp_verify_id_batch(elp_ctx* c, size_t n, const uint8_t* records, uint64_t mask, int retr, const uint8_t* ad,
                        const uint32_t* ad_off, size_t ad_len, uint8_t* flags, uint64_t* accepted) {
  int rc = check_fused(c, mask, need_rp(retr));
  if (rc) return sync_fail(rc);
  if (accepted) *accepted = 0;
  if (n == 0) return ELP_OK;
  if (!records || !flags || (!ad && (ad_off ? ad_off[n] : ad_len))) return ELP_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));                 // (ELP_OPT_FAULT_INJECT concerns elp_verify_id_batch_submit only)
  const size_t rsz = elp_verify_id_record_size(c->curve, c->A, popcount_mask(mask, c->A), retr);
  if (!c->pstream[0]) HIPCHK(c, hipStreamCreateWithFlags(&c->pstream[0], hipStreamNonBlocking));
  DevBuf drec, dad, doff, dfl, dcnt;
  const void *pad, *poff;
  HIPCHK(c, drec.alloc(n * rsz));
  HIPCHK(c, dfl.alloc(n));
  HIPCHK(c, dcnt.alloc(8));
  HIPCHK(c, hipMemsetAsync(dcnt.p, 0, 8, c->stream));
  rc = stage_ad(c, n, ad, ad_off, ad_len, dad, doff, &pad, &poff);
  if (rc) return sync_fail(rc);
  const size_t round = (size_t)64 * c->simds;
  std::vector<hipEvent_t> ev;
  int status = ELP_OK;
  for (size_t lo = 0; lo < n && status == ELP_OK; lo += round) {
    const size_t cnt = n - lo < round ? n - lo : round;
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      status = ELP_ERR_HIP;
      break;
    }
    ev.push_back(e);
    // pageable source: the runtime stages it through its own pinned buffers; the call returns once the round is staged
    if (hipMemcpyAsync((uint8_t*)drec.p + lo * rsz, records + lo * rsz, cnt * rsz, hipMemcpyHostToDevice, c->pstream[0]) != hipSuccess ||
        hipEventRecord(e, c->pstream[0]) != hipSuccess || hipStreamWaitEvent(c->stream, e, 0) != hipSuccess) {
      status = ELP_ERR_HIP;
      break;
    }
    status = elp_verify_id_batch_dev(c, c->stream, cnt, (uint8_t*)drec.p + lo * rsz, mask, retr, pad, poff ? (const uint32_t*)poff + lo : nullptr,
                                     ad_len, (uint8_t*)dfl.p + lo, dcnt.p);
  }
  uint64_t cntv = 0;
  if (status == ELP_OK && (hipMemcpyAsync(flags, dfl.p, n, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                           hipMemcpyAsync(&cntv, dcnt.p, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess))
    status = ELP_ERR_HIP;
  hipError_t e1 = hipStreamSynchronize(c->pstream[0]), e2 = hipStreamSynchronize(c->stream);
  for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  if (status == ELP_OK && (e1 != hipSuccess || e2 != hipSuccess)) status = ELP_ERR_HIP;
  if (status == ELP_ERR_HIP && c->err.empty()) c->err = "HIP error in elp_verify_id_batch";
  if (status != ELP_OK) return status;
  if (accepted) *accepted = cntv;
  return ELP_OK;
}

// ---- the pipelined form: submit / wait over two slots with persistent device buffers
static int grow_dev(elp_ctx* c, void** p, size_t* cap, size_t need) {
  if (*cap >= need && *p) return ELP_OK;
  if (*p) HIPCHK(c, hipFree(*p));        // waits for work that may still use it (a slot is idle when it is resubmitted anyway)
  *p = nullptr;
  *cap = 0;
  const size_t want = need + need / 8 + 256;
  HIPCHK(c, hipMalloc(p, want));
  *cap = want;
  return ELP_OK;
}
// Records of a batch handed over IN PARTS while the caller is still packing the rest (round 6): each call queues the copy of records [first, first + count) into the slot's
// device buffer on the copy stream and returns; elp_verify_id_batch_submit with records == NULL then launches over what was staged.
int elp_verify_id_batch_stage(elp_ctx* c, int slot, size_t n_total, size_t record_size, size_t first, size_t count, const uint8_t* records_part) {
  if (!c || slot < 0 || slot > 1 || n_total == 0 || record_size == 0 || first + count > n_total || (count && !records_part)) return ELP_ERR_ARG;
  elp_ctx::AsyncSlot& s = c->aslot[slot];
  if (s.busy) {
    c->err = "elp_verify_id_batch_stage: the slot has a batch in flight (call elp_verify_id_batch_wait first)";
    return ELP_ERR_STATE;
  }
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->pstream[0]) HIPCHK(c, hipStreamCreateWithFlags(&c->pstream[0], hipStreamNonBlocking));
  if (first == 0 && count == 0) s.staged_bytes = 0;      // the sizing call that opens a batch also forgets parts a failed earlier batch may have left behind
  if (s.rec_cap < n_total * record_size) {
    if (s.staged_bytes) {
      c->err = "elp_verify_id_batch_stage: n_total x record_size grew while parts of the batch were staged";
      return ELP_ERR_ARG;
    }
    int rc = grow_dev(c, &s.drec, &s.rec_cap, n_total * record_size);
    if (rc) return rc;
  }
  if (count) HIPCHK(c, hipMemcpyAsync((uint8_t*)s.drec + first * record_size, records_part, count * record_size, hipMemcpyHostToDevice, c->pstream[0]));
  s.staged_bytes += count * record_size;
  return ELP_OK;
}
int elp_verify_id_batch_submit(elp_ctx* c, int slot, size_t n, const uint8_t* records, uint64_t mask, int retr, const uint8_t* ad, const uint32_t* ad_off,
                               size_t ad_len, uint8_t* flags) {
  int rc = check_fused(c, mask, need_rp(retr));
  if (rc) return rc;
  if (slot < 0 || slot > 1 || n == 0 || !flags || (!ad && (ad_off ? ad_off[n] : ad_len))) return ELP_ERR_ARG;
  elp_ctx::AsyncSlot& s = c->aslot[slot];
  if (!records && s.staged_bytes != n * elp_verify_id_record_size(c->curve, c->A, popcount_mask(mask, c->A), retr)) {
    c->err = "elp_verify_id_batch_submit: records == NULL, but elp_verify_id_batch_stage has not delivered exactly n records";
    s.staged_bytes = 0;
    return ELP_ERR_ARG;
  }
  if (s.busy) {
    c->err = "elp_verify_id_batch_submit: the slot has a batch in flight (call elp_verify_id_batch_wait first)";
    return ELP_ERR_STATE;
  }
  if (c->fail_submits > 0) {                     // ELP_OPT_FAULT_INJECT
    c->fail_submits--;
    c->err = "elp_verify_id_batch_submit: injected failure (ELP_OPT_FAULT_INJECT)";
    return ELP_ERR_STATE;
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t rsz = elp_verify_id_record_size(c->curve, c->A, popcount_mask(mask, c->A), retr);
  if (!c->pstream[0]) HIPCHK(c, hipStreamCreateWithFlags(&c->pstream[0], hipStreamNonBlocking));
  // one-time resources of the slot, each guarded on its own: a failure half-way leaves the others to be created by the next submit
  if (!s.copied) HIPCHK(c, hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
  if (!s.done) HIPCHK(c, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  if (!s.h_cnt) HIPCHK(c, hipHostMalloc((void**)&s.h_cnt, 64, hipHostMallocDefault));
  if (!s.dcnt) {
    size_t dummy = 0;
    if ((rc = grow_dev(c, &s.dcnt, &dummy, 8))) return rc;
  }
  const size_t ad_total = ad_off ? ad_off[n] : ad_len;
  if ((rc = grow_dev(c, &s.drec, &s.rec_cap, n * rsz)) || (rc = grow_dev(c, &s.dfl, &s.fl_cap, n)) || (rc = grow_dev(c, &s.dad, &s.ad_cap, ad_total ? ad_total : 4)) ||
      (ad_off && (rc = grow_dev(c, &s.doff, &s.off_cap, (n + 1) * 4))))
    return rc;
  // copies on the copy stream (they overlap the kernel of the other slot, which runs on the context's stream), then the kernel behind them
  if (records) HIPCHK(c, hipMemcpyAsync(s.drec, records, n * rsz, hipMemcpyHostToDevice, c->pstream[0]));      // NULL: delivered by elp_verify_id_batch_stage
  s.staged_bytes = 0;
  if (ad_total) HIPCHK(c, hipMemcpyAsync(s.dad, ad, ad_total, hipMemcpyHostToDevice, c->pstream[0]));
  if (ad_off) HIPCHK(c, hipMemcpyAsync(s.doff, ad_off, (n + 1) * 4, hipMemcpyHostToDevice, c->pstream[0]));
  HIPCHK(c, hipMemsetAsync(s.dcnt, 0, 8, c->pstream[0]));
  HIPCHK(c, hipEventRecord(s.copied, c->pstream[0]));
  HIPCHK(c, hipStreamWaitEvent(c->stream, s.copied, 0));
  rc = elp_verify_id_batch_dev(c, c->stream, n, s.drec, mask, retr, s.dad, ad_off ? s.doff : nullptr, ad_len, s.dfl, s.dcnt);
  if (rc) return sync_fail(rc);
  HIPCHK(c, hipMemcpyAsync(flags, s.dfl, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(s.h_cnt, s.dcnt, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipEventRecord(s.done, c->stream));
  s.busy = true;
  return ELP_OK;
}
int elp_verify_id_batch_wait(elp_ctx* c, int slot, uint64_t* accepted) {
  if (!c || slot < 0 || slot > 1) return ELP_ERR_ARG;
  elp_ctx::AsyncSlot& s = c->aslot[slot];
  if (!s.busy) {
    c->err = "elp_verify_id_batch_wait: nothing was submitted to the slot";
    return ELP_ERR_STATE;
  }
  HIPCHK(c, hipSetDevice(c->device));
  s.busy = false;
  HIPCHK(c, hipEventSynchronize(s.done));
  if (accepted) *accepted = *s.h_cnt;
  return ELP_OK;
}

int elp_prove_id_batch_dev(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad,
                           const void* d_ad_off, size_t ad_len, void* d_proofs, void* d_flags, void* d_accepted) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254
             ? elp_prove_id_batch_dev_t<BN254>(c, stream, n, d_records, mask, retr, d_ad, d_ad_off, ad_len, d_proofs, d_flags, d_accepted)
             : elp_prove_id_batch_dev_t<BLS12_381>(c, stream, n, d_records, mask, retr, d_ad, d_ad_off, ad_len, d_proofs, d_flags, d_accepted);
}
int elp_request_id_batch_dev(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, const void* d_ad,
                             const void* d_ad_off, size_t ad_len, void* d_requests) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_request_id_batch_dev_t<BN254>(c, stream, n, d_records, mask, d_ad, d_ad_off, ad_len, d_requests)
                                     : elp_request_id_batch_dev_t<BLS12_381>(c, stream, n, d_records, mask, d_ad, d_ad_off, ad_len, d_requests);
}

// user side, host buffers
int elp_prove_id_batch(elp_ctx* c, size_t n, const uint8_t* records, uint64_t mask, int retr, const uint8_t* ad, const uint32_t* ad_off,
                       size_t ad_len, uint8_t* proofs, uint8_t* flags, uint64_t* produced) {
  int rc = check_fused(c, mask, need_rp(retr));
  if (rc) return sync_fail(rc);
  if (produced) *produced = 0;
  if (n == 0) return ELP_OK;
  if (!records || !flags || !proofs || (!ad && (ad_off ? ad_off[n] : ad_len))) return ELP_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const int H = popcount_mask(mask, c->A);
  const size_t rsz = elp_prove_id_record_size(c->curve, c->A, H, retr), osz = elp_verify_id_record_size(c->curve, c->A, H, retr);
  DevBuf drec, dad, doff, dfl, dcnt, dout;
  const void *pad, *poff;
  HIPCHK(c, drec.alloc(n * rsz));
  HIPCHK(c, dfl.alloc(n));
  HIPCHK(c, dcnt.alloc(8));
  HIPCHK(c, dout.alloc(n * osz));
  HIPCHK(c, hipMemcpyAsync(drec.p, records, n * rsz, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(dcnt.p, 0, 8, c->stream));
  rc = stage_ad(c, n, ad, ad_off, ad_len, dad, doff, &pad, &poff);
  if (rc) return sync_fail(rc);
  rc = elp_prove_id_batch_dev(c, c->stream, n, drec.p, mask, retr, pad, poff, ad_len, dout.p, dfl.p, dcnt.p);
  if (rc) return sync_fail(rc);
  uint64_t cnt = 0;
  HIPCHK(c, hipMemcpyAsync(flags, dfl.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(proofs, dout.p, n * osz, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&cnt, dcnt.p, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (produced) *produced = cnt;
  return ELP_OK;
}
int elp_request_id_batch(elp_ctx* c, size_t n, const uint8_t* records, uint64_t mask, const uint8_t* ad, const uint32_t* ad_off,
                         size_t ad_len, uint8_t* requests) {
  int rc = check_fused(c, mask);
  if (rc) return sync_fail(rc);
  if (n == 0) return ELP_OK;
  if (!records || !requests || (!ad && (ad_off ? ad_off[n] : ad_len))) return ELP_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const int H = popcount_mask(mask, c->A);
  const size_t rsz = elp_request_id_record_size(c->curve, c->A, H), osz = elp_request_id_out_size(c->curve, H);
  DevBuf drec, dad, doff, dout;
  const void *pad, *poff;
  HIPCHK(c, drec.alloc(n * rsz));
  HIPCHK(c, dout.alloc(n * osz));
  HIPCHK(c, hipMemcpyAsync(drec.p, records, n * rsz, hipMemcpyHostToDevice, c->stream));
  rc = stage_ad(c, n, ad, ad_off, ad_len, dad, doff, &pad, &poff);
  if (rc) return sync_fail(rc);
  rc = elp_request_id_batch_dev(c, c->stream, n, drec.p, mask, pad, poff, ad_len, dout.p);
  if (rc) return sync_fail(rc);
  HIPCHK(c, hipMemcpyAsync(requests, dout.p, n * osz, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ELP_OK;
}

int elp_verify_id_wire_batch(elp_ctx* c, size_t n, const uint8_t* msgs, const uint32_t* msg_off, int retr, const uint8_t* ad,
                             const uint32_t* ad_off, size_t ad_len, uint8_t* flags, uint64_t* accepted) {
  int rc = check_fused(c, 0, need_rp(retr));
  if (rc) return sync_fail(rc);
  if (accepted) *accepted = 0;
  if (n == 0) return ELP_OK;
  if (!msgs || !msg_off || !flags || (!ad && (ad_off ? ad_off[n] : ad_len))) return ELP_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf dmsg, dmoff, dad, doff, dfl, dcnt;
  const void *pad, *poff;
  HIPCHK(c, dmsg.alloc(msg_off[n]));
  HIPCHK(c, dmoff.alloc((n + 1) * 4));
  HIPCHK(c, dfl.alloc(n));
  HIPCHK(c, dcnt.alloc(8));
  if (msg_off[n]) HIPCHK(c, hipMemcpyAsync(dmsg.p, msgs, msg_off[n], hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dmoff.p, msg_off, (n + 1) * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(dcnt.p, 0, 8, c->stream));
  rc = stage_ad(c, n, ad, ad_off, ad_len, dad, doff, &pad, &poff);
  if (rc) return sync_fail(rc);
  rc = elp_verify_id_wire_batch_dev(c, c->stream, n, dmsg.p, dmoff.p, retr, pad, poff, ad_len, dfl.p, dcnt.p);
  if (rc) return sync_fail(rc);
  uint64_t cnt = 0;
  HIPCHK(c, hipMemcpyAsync(flags, dfl.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&cnt, dcnt.p, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (accepted) *accepted = cnt;
  return ELP_OK;
}

int elp_ps_verify_batch(elp_ctx* c, size_t n, const uint8_t* records, int nattr, uint8_t* flags, uint64_t* accepted) {
  int rc = check_fused(c, 0);
  if (rc) return sync_fail(rc);
  if (accepted) *accepted = 0;
  if (n == 0) return ELP_OK;
  if (!records || !flags) return ELP_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t rsz = elp_ps_verify_record_size(c->curve, nattr);
  DevBuf drec, dfl, dcnt;
  HIPCHK(c, drec.alloc(n * rsz));
  HIPCHK(c, dfl.alloc(n));
  HIPCHK(c, dcnt.alloc(8));
  HIPCHK(c, hipMemcpyAsync(drec.p, records, n * rsz, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(dcnt.p, 0, 8, c->stream));
  rc = elp_ps_verify_batch_dev(c, c->stream, n, drec.p, nattr, dfl.p, dcnt.p);
  if (rc) return sync_fail(rc);
  uint64_t cnt = 0;
  HIPCHK(c, hipMemcpyAsync(flags, dfl.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&cnt, dcnt.p, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (accepted) *accepted = cnt;
  return ELP_OK;
}
int elp_provide_id_batch(elp_ctx* c, size_t n, const uint8_t* records, uint64_t mask, const uint8_t* ad, const uint32_t* ad_off,
                         size_t ad_len, uint8_t* sigs, uint8_t* flags, uint64_t* accepted) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_provide_id_batch_t<BN254>(c, n, records, mask, ad, ad_off, ad_len, sigs, flags, accepted) : elp_provide_id_batch_t<BLS12_381>(c, n, records, mask, ad, ad_off, ad_len, sigs, flags, accepted);
}

size_t elp_key_table_bytes(const elp_ctx* c) {
  if (!c || !c->have_pk) return 0;
  const size_t f = c->curve == ELP_CURVE_BN254 ? sizeof(Fp<BN254>) : sizeof(Fp<BLS12_381>);      // one Montgomery-form field element (9 x 29-bit / 14 x 28-bit limbs)
  const size_t entries = (size_t)c->per * c->nwin;
  return entries * ((size_t)(c->A + 6) * 2 * f + (size_t)(c->A + 2) * 4 * f);
}
int elp_host_alloc(elp_ctx* c, size_t bytes, void** out) {
  if (!c || !out) return ELP_ERR_ARG;
  *out = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return ELP_OK;
}
void elp_host_free(elp_ctx* c, void* p) {
  if (!c || !p) return;
  (void)hipSetDevice(c->device);
  (void)hipHostFree(p);
}
int elp_time_verify_id_dev(elp_ctx* c, void* stream, int reps, size_t n, const void* d_records, uint64_t mask, int retr,
                           const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted, float* avg_ms) {
  if (!c || reps < 1 || !avg_ms) return ELP_ERR_ARG;
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  HIPCHK(c, hipEventRecord(e0, (hipStream_t)stream));
  for (int r = 0; r < reps; r++) {
    int rc = elp_verify_id_batch_dev(c, stream, n, d_records, mask, retr, d_ad, d_ad_off, ad_len, d_flags, d_accepted);
    if (rc) return sync_fail(rc);
  }
  HIPCHK(c, hipEventRecord(e1, (hipStream_t)stream));
  HIPCHK(c, hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = ms / reps;
  return ELP_OK;
}
int elp_bench_op(elp_ctx* c, int op, size_t lanes, int iters, float* ms) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_bench_op_t<BN254>(c, op, lanes, iters, ms) : elp_bench_op_t<BLS12_381>(c, op, lanes, iters, ms);
}
int elp_bench_fp_mul(elp_ctx* c, size_t lanes, int iters, float* ms) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? elp_bench_fp_mul_t<BN254>(c, lanes, iters, ms) : elp_bench_fp_mul_t<BLS12_381>(c, lanes, iters, ms);
}

int elp_g1_msm(elp_ctx* c, size_t n, const uint8_t* points, const uint8_t* scalars, uint8_t* out) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? msm_impl_t<BN254, 1>(c, n, points, scalars, out) : msm_impl_t<BLS12_381, 1>(c, n, points, scalars, out);
}
int elp_g2_msm(elp_ctx* c, size_t n, const uint8_t* points, const uint8_t* scalars, uint8_t* out) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? msm_impl_t<BN254, 2>(c, n, points, scalars, out) : msm_impl_t<BLS12_381, 2>(c, n, points, scalars, out);
}
size_t elp_msm_workspace_bytes(int curve, int group, size_t n) {
  if (group != 1 && group != 2) return 0;
  if (curve == ELP_CURVE_BLS12_381) return group == 1 ? msm_ws_bytes<BLS12_381, 1>(n) : msm_ws_bytes<BLS12_381, 2>(n);
  return group == 1 ? msm_ws_bytes<BN254, 1>(n) : msm_ws_bytes<BN254, 2>(n);
}
int elp_g1_msm_dev(elp_ctx* c, void* stream, size_t n, const void* d_points, const void* d_scalars, void* d_workspace, void* d_out) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? msm_dev_t<BN254, 1>(c, stream, n, d_points, d_scalars, d_workspace, d_out)
                                     : msm_dev_t<BLS12_381, 1>(c, stream, n, d_points, d_scalars, d_workspace, d_out);
}
int elp_g2_msm_dev(elp_ctx* c, void* stream, size_t n, const void* d_points, const void* d_scalars, void* d_workspace, void* d_out) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? msm_dev_t<BN254, 2>(c, stream, n, d_points, d_scalars, d_workspace, d_out)
                                     : msm_dev_t<BLS12_381, 2>(c, stream, n, d_points, d_scalars, d_workspace, d_out);
}

int elp_verify_id_batch_aggregated_dev(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad,
                                       const void* d_ad_off, size_t ad_len, const uint8_t* seed32, void* d_flags, void* d_accepted) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254
             ? elp_verify_id_batch_aggregated_dev_t<BN254>(c, stream, n, d_records, mask, retr, d_ad, d_ad_off, ad_len, seed32, d_flags, d_accepted)
             : elp_verify_id_batch_aggregated_dev_t<BLS12_381>(c, stream, n, d_records, mask, retr, d_ad, d_ad_off, ad_len, seed32, d_flags,
                                                               d_accepted);
}

int elp_verify_id_batch_aggregated(elp_ctx* c, size_t n, const uint8_t* records, uint64_t mask, int retr, const uint8_t* ad,
                                   const uint32_t* ad_off, size_t ad_len, const uint8_t* seed32, uint8_t* flags, uint64_t* accepted,
                                   int* batch_equation_held) {
  int rc = check_fused(c, mask, need_rp(retr));
  if (rc) return sync_fail(rc);
  if (accepted) *accepted = 0;
  if (batch_equation_held) *batch_equation_held = 1;
  if (n == 0) return ELP_OK;
  if (!records || !flags || (!ad && (ad_off ? ad_off[n] : ad_len))) return ELP_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t rsz = elp_verify_id_record_size(c->curve, c->A, popcount_mask(mask, c->A), retr);
  DevBuf drec, dad, doff, dfl, dcnt;
  const void *pad, *poff;
  HIPCHK(c, drec.alloc(n * rsz));
  HIPCHK(c, dfl.alloc(n));
  HIPCHK(c, dcnt.alloc(8));
  HIPCHK(c, hipMemcpyAsync(drec.p, records, n * rsz, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(dcnt.p, 0, 8, c->stream));
  rc = stage_ad(c, n, ad, ad_off, ad_len, dad, doff, &pad, &poff);
  if (rc) return sync_fail(rc);
  rc = elp_verify_id_batch_aggregated_dev(c, c->stream, n, drec.p, mask, retr, pad, poff, ad_len, seed32, dfl.p, dcnt.p);
  if (rc) return sync_fail(rc);
  uint64_t cnt = 0;
  int held = 0;
  HIPCHK(c, hipMemcpyAsync(flags, dfl.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&cnt, dcnt.p, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&held, c->agg_ok, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (accepted) *accepted = cnt;
  if (batch_equation_held) *batch_equation_held = held;
  return ELP_OK;
}
