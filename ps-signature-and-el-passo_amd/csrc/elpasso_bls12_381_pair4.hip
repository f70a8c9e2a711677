// Four-lanes-per-item pairing check for BLS12-381 (elp/pair4.h, elpasso_pair4.h; the curve is pinned by tests/golden/bls12_381_*.json): a translation unit of its own.
#define ELP_PAIR4_TU 1
#include "elpasso_pair4.h"

template void launch_pair4<BLS12_381>(hipStream_t stream, const void* gg_lines, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags, void* d_accepted);
template void launch_vid_mid<BLS12_381>(hipStream_t stream, const KeyCtx<BLS12_381>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, const void* pre);
