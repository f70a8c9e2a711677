// Phase 1 of the two-phase EL PASSO verification for BN254 (k_vid_nizk, elp/pipeline.h "EL PASSO VerifyID as TWO PHASES"): a translation unit of
// its own, so that its device functions are compiled for the 256-register budget of two waves per SIMD.
#define ELP_NIZK_TU 1
#include "elpasso_impl.h"

template void launch_vid_nizk<BN254>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, u32* kws, size_t kstride, const KeyCtx<BN254>& key, const void* pre);
template void launch_vid_nizk4<BN254>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, u32* kws, size_t kstride, const KeyCtx<BN254>& key, const void* pre, int k_done);
