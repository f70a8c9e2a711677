// G2 job of the two-phase EL PASSO verification for BN254 (k_vid_g2, elp/pipeline.h "EL PASSO VerifyID as TWO PHASES"): a translation unit of its own,
// so that its device functions are compiled for its register budget (384 per lane: one such wave and one 128-register G1-job wave share a SIMD).
#define ELP_G2JOB_TU 1
#include "elpasso_impl.h"

template void launch_vid_g2<BN254>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, uint8_t* ok_g2, u32* ws, size_t stride, const KeyCtx<BN254>& key);
