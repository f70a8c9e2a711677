// Kernels of the FOUR-LANES-PER-ITEM pairing check (elp/pair4.h; round 5, VERDICT r4 #1b): included by the translation units elpasso_<curve>_pair4.hip only.
// One item = one DPP quad; 64-thread workgroups = 16 items; built for two waves per SIMD (256 registers, no LDS).
#pragma once
#include "elpasso_impl.h"
#include "elp/pair4.h"

// K of item i from the launch workspace (pipeline.h vid_store_k: word q of the plain-layout Aff<F2<B>> at ws[q * stride + i]) -> this lane's components
template <class C>
ELP_INL void vid_load_k_paired(Aff<F2<C>>& aK, const u32* ws, size_t stride, size_t i) {
  const int o = pair_odd() ? C::NL : 0;
  ELP_UNROLL
  for (int q = 0; q < C::NL; q++) {
    aK.x.c.v[q] = (i32)ws[(size_t)(o + q) * stride + i];
    aK.y.c.v[q] = (i32)ws[(size_t)(2 * C::NL + o + q) * stride + i];
  }
}
// the pairing half of a verification whose K is in the workspace (k_ps_k_coop for PS verification, k_vid_prep for el_passo_verify_id):
// flags[i] = todo[i] && the record's sig1, sig2 decode && e(sig1, K) e(-sig2, gg) == 1      (src/ps-verifier.cc:31-34, 132-137)
template <class C>
__global__ void __launch_bounds__(ELP_BLOCK, 2) k_pair4(const LineMem<C>* gg_lines, const u32* recs, int rec_words, const uint8_t* todo, const u32* kws, size_t kstride,
                                                       uint8_t* flags, unsigned long long* accepted, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  bool ok = false;
  if (i < n) {                       // quad-uniform: the four lanes of an item take every branch together
    if (todo[i]) {
      Aff<F1<C>> sig1, sig2;
      const u32* rec = recs + i * (size_t)rec_words;
      if (g1_load<C>(sig1, rec) && g1_load<C>(sig2, rec + 2 * C::N)) {
        Aff<F2<C>> aK;
        vid_load_k_paired<C>(aK, kws, kstride, i);
        ok = ps_pairing_check4<C>(gg_lines, sig1, sig2, aK);
      }
    }
    if ((threadIdx.x & 3) == 0) flags[i] = ok ? 1 : 0;
  }
  const unsigned long long b = __ballot(ok && (threadIdx.x & 3) == 0);
  if ((threadIdx.x & 63) == 0 && b != 0 && accepted) atomicAdd(accepted, (unsigned long long)__popcll(b));
}
template <class B>
void launch_pair4(hipStream_t stream, const void* gg_lines, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags,
                  void* d_accepted) {
  hipLaunchKernelGGL((k_pair4<Paired<B>>), dim3((unsigned)((n * 4 + ELP_BLOCK - 1) / ELP_BLOCK)), dim3(ELP_BLOCK), 0, stream, (const LineMem<Paired<B>>*)gg_lines,
                     (const u32*)d_records, words, todo, kws, kstride, d_flags, (unsigned long long*)d_accepted, n);
}
