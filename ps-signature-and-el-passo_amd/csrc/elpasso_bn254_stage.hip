// el_passo_verify_id with coalesced record loads for BN254 (k_verify_id_staged): a translation unit of its own (one more copy of the fused kernel).
#define ELP_STAGE_TU 1
#include "elpasso_impl.h"

template void launch_verify_id_staged<BN254>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted, const KeyCtx<BN254>& key);
