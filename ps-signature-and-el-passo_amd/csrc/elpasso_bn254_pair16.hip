// Row-of-16 pairing check for BN254 (elpasso_pair16.h): a translation unit of its own.
#define ELP_PAIR16_TU 1
#include "elpasso_pair16.h"

template void launch_pair16<BN254>(hipStream_t stream, const void* gg_lines, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags, void* d_accepted);
template void launch_agg_final16<BN254>(hipStream_t stream, const void* gg_lines, const void* F, const void* s2_std, int* agg_ok);
template void launch_fp12_reduce16<BN254>(hipStream_t stream, const void* in, size_t n, void* out);
