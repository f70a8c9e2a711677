// Explicit instantiations of the kernels and host implementations for BN254.
#include "elpasso_impl.h"

template int elp_set_pubkey_t<BN254>(elp_ctx* c, int nattr, const uint8_t* g, const uint8_t* gg, const uint8_t* XX, const uint8_t* Yi, const uint8_t* YYi, int window_bits);
template int elp_set_rp_t<BN254>(elp_ctx* c, const uint8_t* service_name, size_t service_len, const uint8_t* authority_pk, const uint8_t* g, const uint8_t* h);
template int elp_set_signer_secret_t<BN254>(elp_ctx* c, const uint8_t* X);
template int decompress_impl_t<BN254, 1>(elp_ctx* c, size_t n, const uint8_t* wire, uint8_t* out, uint8_t* ok);
template int decompress_impl_t<BN254, 2>(elp_ctx* c, size_t n, const uint8_t* wire, uint8_t* out, uint8_t* ok);
template int mul_impl_t<BN254, 1>(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out);
template int mul_impl_t<BN254, 2>(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out);
template int add_impl_t<BN254, 1>(elp_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out);
template int add_impl_t<BN254, 2>(elp_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out);
template int msm_fixed_impl_t<BN254, 1>(elp_ctx* c, size_t n, int nterms, const int32_t* ids, const uint8_t* ks, uint8_t* out);
template int msm_fixed_impl_t<BN254, 2>(elp_ctx* c, size_t n, int nterms, const int32_t* ids, const uint8_t* ks, uint8_t* out);
template int elp_hash_to_g1_t<BN254>(elp_ctx* c, size_t n, const uint8_t* msgs, const uint32_t* off, uint8_t* out);
template int elp_pairing_t<BN254>(elp_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt);
template int elp_pairing_check_t<BN254>(elp_ctx* c, size_t n, int npairs, const uint8_t* g1, const uint8_t* g2, uint8_t* ok);
template int elp_verify_id_batch_dev_t<BN254>(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted);
template int elp_verify_id_wire_batch_dev_t<BN254>(elp_ctx* c, void* stream, size_t n, const void* d_msgs, const void* d_msg_off, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted);
template int elp_ps_verify_batch_dev_t<BN254>(elp_ctx* c, void* stream, size_t n, const void* d_records, int nattr, void* d_flags, void* d_accepted);
template int elp_provide_id_batch_dev_t<BN254>(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_sigs, void* d_flags, void* d_accepted);
template int elp_provide_id_batch_t<BN254>(elp_ctx* c, size_t n, const uint8_t* records, uint64_t mask, const uint8_t* ad, const uint32_t* ad_off, size_t ad_len, uint8_t* sigs, uint8_t* flags, uint64_t* accepted);
template int elp_bench_op_t<BN254>(elp_ctx* c, int op, size_t lanes, int iters, float* ms);
template int elp_bench_fp_mul_t<BN254>(elp_ctx* c, size_t lanes, int iters, float* ms);
template int msm_impl_t<BN254, 1>(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out);
template int msm_impl_t<BN254, 2>(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out);
template int msm_dev_t<BN254, 1>(elp_ctx* c, void* stream_, size_t n, const void* d_points, const void* d_scalars, void* d_workspace, void* d_out);
template int msm_dev_t<BN254, 2>(elp_ctx* c, void* stream_, size_t n, const void* d_points, const void* d_scalars, void* d_workspace, void* d_out);
template int elp_verify_id_batch_aggregated_dev_t<BN254>(elp_ctx* c, void* stream_, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, const uint8_t* seed32, void* d_flags, void* d_accepted);
template int elp_prove_id_batch_dev_t<BN254>(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_proofs, void* d_flags, void* d_accepted);
template int elp_request_id_batch_dev_t<BN254>(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_requests);
