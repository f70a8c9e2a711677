// NIZK half of small el_passo_verify_id batches on BLS12-381 (k_vid_nizk4: the four jobs of an item on four waves, plain layout): a translation unit of its own.
#define ELP_NIZK_TU 1
#include "elpasso_impl.h"

template void launch_vid_nizk4<BLS12_381>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, u32* kws, size_t kstride, const KeyCtx<BLS12_381>& key, const void* pre, int k_done);
