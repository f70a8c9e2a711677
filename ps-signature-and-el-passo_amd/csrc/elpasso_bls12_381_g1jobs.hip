// G1 jobs of the EL PASSO verification on BLS12-381 (k_vid_g1jobs, ELP_OPT_SPLIT_PHASES = 3): a translation unit of its own -- the kernel's register budget
// (several waves per SIMD) reaches the device functions it shares with nobody else here.
#define ELP_G1JOBS_TU 1
#include "elpasso_impl.h"

template void launch_vid_g1jobs<BLS12_381>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, u32* ws, size_t stride, const KeyCtx<BLS12_381>& key);
