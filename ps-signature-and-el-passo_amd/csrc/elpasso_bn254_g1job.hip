// G1 job of the two-phase EL PASSO verification for BN254 (k_vid_g1): a translation unit of its own, compiled for 128 registers per lane.
#define ELP_G1JOB_TU 1
#include "elpasso_impl.h"

template void launch_vid_g1<BN254>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, u32* ws, size_t stride, const KeyCtx<BN254>& key);
