// Pippenger multi-scalar multiplication for BN254 (k_msm_*: elpasso_impl.h msm_launch), a translation unit of its own: see the note at msm_launch.
#define ELP_MSM_TU 1
#include "elpasso_impl.h"

template int* msm_launch<BN254, 1>(hipStream_t stream, size_t n, const void* d_pts_std, const void* d_ks, void* d_out_std, uint8_t* ws, bool glv_pairs);
template int* msm_launch<BN254, 2>(hipStream_t stream, size_t n, const void* d_pts_std, const void* d_ks, void* d_out_std, uint8_t* ws, bool glv_pairs);
