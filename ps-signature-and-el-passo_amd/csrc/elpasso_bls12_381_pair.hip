// Paired-layout kernels for BLS12-381 (two lanes per item, elp/common.h "Lane pairs"), a translation unit of their own.
#define ELP_PAIR_TU 1
#include "elpasso_impl.h"

template void launch_verify_id_paired<BLS12_381>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted);
template void launch_ps_verify_paired<BLS12_381>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int nattr, void* d_flags, void* d_accepted);
template void launch_verify_id_wire_paired<BLS12_381>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_msgs, const void* d_msg_off, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted);
template void launch_agg_final_paired<BLS12_381>(elp_ctx* c, hipStream_t stream, const void* F, const void* s2_std);
template void launch_verify_id_paired_g1<BLS12_381>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, const u32* g1ws, size_t g1stride, void* d_flags, void* d_accepted, const KeyCtx<Paired<BLS12_381>>& key);
template void launch_verify_id_agg_paired<BLS12_381>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, const AggSeed& seed, uint8_t* nizk_flags, void* deltas, void* sig2s, void* wave_prod);
