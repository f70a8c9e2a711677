// k_vid_small2<BN254>: the one-launch small-batch kernel of el_passo_verify_id built for TWO waves per SIMD (every kernel and, through the attributor, every device
// function of this translation unit: -DELP_WAVES_PER_EU=2 in build.py), for batches that need more than one round of pairing workgroups.  Carries its own copy
// of the program tables.
#define ELP_COOP_TU 1
#define ELP_COOP_HAVE_BN254 1
#define ELP_COOP_TABLE static __device__
#include <hip/hip_runtime.h>
#include "elp/coop_prog_bn254.h"
#include "elp/coop.h"
#include "elpasso_impl.h"

template void launch_vid_small2<BN254>(hipStream_t stream, const KeyCtx<BN254>& key, const void* d_consts, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, uint8_t* done, const void* pre);
