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
// `first_lane`: index of the workgroup's first lane among the launch's pairing lanes (item = lane / 4).
template <class C>
__device__ __forceinline__ void pair4_body(const LineMem<C>* gg_lines, const u32* recs, int rec_words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* flags,
                                           unsigned long long* accepted, size_t n, size_t first_lane) {
  const size_t i = (first_lane + threadIdx.x) >> 2;
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
  if (accepted) {
    const unsigned long long b = __ballot(ok && (threadIdx.x & 3) == 0);
    if ((threadIdx.x & 63) == 0 && b != 0) atomicAdd(accepted, (unsigned long long)__popcll(b));
  }
}
// 256-thread workgroups: the four waves of a workgroup go to the four SIMDs of one compute unit, so a batch of up to 16 384 items (256 workgroups) runs ONE wave per
// SIMD.  With one-wave workgroups the dispatcher doubled waves up on some SIMDs from ~600 waves on while others stayed idle: 12 288 items took 5.0 ms against 3.1 ms
// for 8 192 (profiles/r05_four_lane.md, kernel trace).
#define ELP_PAIR4_BLOCK 256
template <class C>
__global__ void __launch_bounds__(ELP_PAIR4_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_pair4(const LineMem<C>* gg_lines, const u32* recs, int rec_words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* flags, unsigned long long* accepted, size_t n) {
  pair4_body<C>(gg_lines, recs, rec_words, todo, kws, kstride, flags, accepted, n, (size_t)blockIdx.x * blockDim.x);
}
template <class B>
void launch_pair4(hipStream_t stream, const void* gg_lines, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags,
                  void* d_accepted) {
  hipLaunchKernelGGL((k_pair4<Paired<B>>), dim3((unsigned)((n * 4 + ELP_PAIR4_BLOCK - 1) / ELP_PAIR4_BLOCK)), dim3(ELP_PAIR4_BLOCK), 0, stream,
                     (const LineMem<Paired<B>>*)gg_lines, (const u32*)d_records, words, todo, kws, kstride, d_flags, (unsigned long long*)d_accepted, n);
}
// Mid-size el_passo_verify_id batches in ONE launch (like k_vid_small for the interpreter): workgroups [0, nb_nizk) run the NIZK half of 64 items each on four job
// waves (vid_nizk4_body), the workgroups after them the pairing check of 64 items each on four lanes per item.  The two halves are independent once K, the fixed-base
// sums and the table of multiples of k exist (k_vid_prep); built for two waves per SIMD, so that a pairing wave sits beside a job wave on every SIMD instead of
// after it.
template <class B>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_vid_mid(KeyCtx<B> key, const u32* recs, int rec_words, u64 mask, int retr, const uint8_t* ad, const u32* ad_off, u32 ad_len, uint8_t* nizk_ok, const uint8_t* kvalid,
          const u32* kws, size_t kstride, uint8_t* pair_ok, size_t n, const Jac<F2<B>>* pre, unsigned nb_nizk) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[Nizk4Lds<B>::BYTES];
  if (blockIdx.x < nb_nizk) {
    vid_nizk4_body<B>(key, recs, rec_words, mask, retr, ad, ad_off, ad_len, nizk_ok, (u32*)nullptr, kstride, n, pre, 1, blockIdx.x, (u32*)smem,
                      (VidShared<B>*)(smem + Nizk4Lds<B>::HOT_BYTES));
  } else {
    pair4_body<Paired<B>>((const LineMem<Paired<B>>*)key.gg_lines, recs, rec_words, kvalid, kws, kstride, pair_ok, nullptr, n, (size_t)(blockIdx.x - nb_nizk) * 256);
  }
}
template <class B>
void launch_vid_mid(hipStream_t stream, const KeyCtx<B>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off,
                    size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, const void* pre) {
  const unsigned nb_nizk = grid_for(n);
  hipLaunchKernelGGL((k_vid_mid<B>), dim3(nb_nizk + (unsigned)((n * 4 + 255) / 256)), dim3(256), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr,
                     (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, nizk_ok, kvalid, kws, kstride, pair_ok, n, (const Jac<F2<B>>*)pre, nb_nizk);
}
