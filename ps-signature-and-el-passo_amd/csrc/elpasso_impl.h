// Kernels, templated host-side implementations and the context of the C-ABI (included by the three translation units
// elpasso_capi.hip, elpasso_bn254.hip, elpasso_bls12_381.hip; the per-curve units hold the explicit instantiations so that the
// two curves compile in parallel).
#pragma once
// HIP kernels (gfx950 / MI355X) and the C-ABI of include/elpasso.h.
// One independent item (credential / proof / point) per lane; all arithmetic lives in elp/*.h.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>

#include <mutex>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/elpasso.h"
#include "elp/params_bls12_381.h"
#include "elp/params_bn254.h"
#include "elp/pipeline.h"

// The G2 job of the NIZK half on four lanes per item (vid_job_g2_quad) for the one-launch batches of at most ELP_QUAD_G2_MAX (1 280) items
#ifndef ELP_QUAD_G2
#define ELP_QUAD_G2 1      /* 0: the G2 job of the smallest batches stays on one lane per item (A/B builds) */
#endif
static constexpr bool elp_quad_g2_on = ELP_QUAD_G2 != 0;
static inline long elp_quad_g2_max() {      // largest batch whose G2 jobs run on four lanes (ELP_QUAD_G2_MAX in the environment: A/B runs)
  static long v = -1;
  if (v < 0) {
    const char* e = getenv("ELP_QUAD_G2_MAX");
    v = e ? atol(e) : 1280;      // 0.75 waves per item (NIZK workgroups of 16 items + pairing workgroups of 8): 1 280 items are 960 waves on 1 024 SIMDs; 1 536 need a second round (4.1 instead of 2.4 ms)
    if (v < 0) v = 0;
  }
  return v;
}

using namespace elp;

// ------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------
#define ELP_BLOCK 64
// register budget of every kernel and (through the attributor) of the device functions they share: ELP_WAVES_PER_EU resident waves per SIMD
#ifdef ELP_WAVES_PER_EU
#define ELP_LAUNCH_BOUNDS __launch_bounds__(ELP_BLOCK, ELP_WAVES_PER_EU)
#define ELP_MSM_LAUNCH_BOUNDS __launch_bounds__(ELP_MSM_TPB, ELP_WAVES_PER_EU)
#else
#define ELP_LAUNCH_BOUNDS __launch_bounds__(ELP_BLOCK)
#define ELP_MSM_LAUNCH_BOUNDS __launch_bounds__(ELP_MSM_TPB)
#endif
// gives every lane of the (one-wave) workgroup its LDS hot slot, see elp/common.h
#ifdef ELP_NO_HOT   /* experiments: no LDS hot slot (KeyCtx::hot stays null) */
#define ELP_HOT_SETUP(key) (void)0
#else
#define ELP_HOT_SETUP(key)                                                                   \
  __shared__ __attribute__((aligned(16))) u32 elp_hot_lds[ELP_BLOCK * elp::ELP_HOT_WORDS]; \
  (key).hot = elp_hot_lds + threadIdx.x * elp::ELP_HOT_WORDS
#endif

static inline unsigned grid_for(size_t n) { return (unsigned)((n + ELP_BLOCK - 1) / ELP_BLOCK); }

__device__ __forceinline__ void count_accept(bool ok, unsigned long long* counter) {
  unsigned long long b = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && b != 0 && counter) atomicAdd(counter, (unsigned long long)__popcll(b));
}

template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_verify_id(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr,
                                                         const uint8_t* ad, const u32* ad_off, u32 ad_len, uint8_t* flags,
                                                         unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP(key);
  if (key.vtab) key.vtab += ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)vtab_words<C>();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n) {
    const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
    size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
    ok = verify_id_item<C>(key, recs + i * (size_t)rec_words, mask, retr != 0, a, al);
    flags[i] = ok ? 1 : 0;
  }
  count_accept(ok, accepted);
}

// The same with COALESCED record loads (north_star: "coalesced HBM loads"; ELP_OPT_COALESCED_RECORDS): the workgroup's 64 records are one contiguous
// 64 x rec_words block of the input; all lanes fetch it with 16-byte loads (a wave instruction covers 1 KB of consecutive addresses) into the LDS area
// that later serves as the hot slots, P records at a time with an odd word stride per record, and every lane copies ITS record out of LDS into private
// memory, which the hardware interleaves by lane.  The body then reads the record through that private copy.  (As handed over, lane l reads words at
// recs + l * rec_words: 64 different cache lines per load instruction.)
#define ELP_STAGE_CAP 288      // words of the private copy: records up to 1152 bytes (A = 16, H = 4 with id-retrieval: 265 words)
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_verify_id_staged(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr,
                                                                const uint8_t* ad, const u32* ad_off, u32 ad_len, uint8_t* flags,
                                                                unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP(key);
  if (key.vtab) key.vtab += ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)vtab_words<C>();
  u32 myrec[ELP_STAGE_CAP];
  {
    u32* const tile = elp_hot_lds;
    const int stride = rec_words | 1;                              // odd: the per-lane copy below walks 32 banks without conflicts
    int P = ELP_BLOCK;
    while (P * stride > ELP_BLOCK * elp::ELP_HOT_WORDS) P >>= 1;   // records per pass that fit the area
    const size_t wg0 = (size_t)blockIdx.x * blockDim.x;
    const int lane = (int)threadIdx.x;
    for (int base = 0; base < ELP_BLOCK; base += P) {
      const size_t first = wg0 + base;
      const int avail = first < n ? (int)((n - first) < (size_t)P ? (n - first) : (size_t)P) : 0;
      const int total = avail * rec_words;                         // words of this pass, contiguous from recs + first * rec_words
      const uint4* src = reinterpret_cast<const uint4*>(recs + first * (size_t)rec_words);
      for (int k = lane * 4; k < total; k += ELP_BLOCK * 4) {      // rec_words % 4 == 0 (checked by the launcher): a 16-byte word never straddles two records
        const uint4 v = src[k >> 2];
        const int r = k / rec_words, w = k - r * rec_words;
        u32* d = tile + r * stride + w;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
      __syncthreads();
      if (lane >= base && lane < base + avail) {
        const u32* mine = tile + (lane - base) * stride;
        for (int w = 0; w < rec_words; w++) myrec[w] = mine[w];
      }
      __syncthreads();
    }
  }
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n) {
    const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
    size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
    ok = verify_id_item<C>(key, myrec, mask, retr != 0, a, al);
    flags[i] = ok ? 1 : 0;
  }
  count_accept(ok, accepted);
}

template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_verify_id_wire(KeyCtx<C> key, const uint8_t* msgs, const u32* msg_off, int retr,
                                                              const uint8_t* ad, const u32* ad_off, u32 ad_len, uint8_t* flags,
                                                              unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP(key);
  if (key.vtab) key.vtab += ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)vtab_words<C>();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n) {
    const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
    size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
    ok = verify_id_wire_item<C>(key, msgs + msg_off[i], (size_t)(msg_off[i + 1] - msg_off[i]), retr != 0, a, al);
    flags[i] = ok ? 1 : 0;
  }
  count_accept(ok, accepted);
}

template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_ps_verify(KeyCtx<C> key, const u32* recs, int rec_words, int nattr, uint8_t* flags,
                                                         unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP(key);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n) {
    ok = ps_verify_item<C>(key, recs + i * (size_t)rec_words, nattr);
    flags[i] = ok ? 1 : 0;
  }
  count_accept(ok, accepted);
}

// ---- two-phase verification (elp/pipeline.h "EL PASSO VerifyID as TWO PHASES").
// Phase 1: 128-thread workgroups = two waves over the same 64 items: wave 0 runs the G2 job and the closing hash, wave 1 the G1 job (+ K).  256
// registers per lane, so every SIMD holds two such waves (of different workgroups, usually of different roles) and issues from both; the
// LDS carries a Jacobian-sized hot slot per lane and the 164-byte exchange record per item: 38 KB per workgroup, four workgroups per CU.
#define ELP_NIZK_BLOCK 128
template <class C>
__global__ void __launch_bounds__(ELP_NIZK_BLOCK, 2) k_vid_nizk(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, const uint8_t* ad,
                                                                const u32* ad_off, u32 ad_len, uint8_t* nizk_ok, u32* kws, size_t kstride, size_t n,
                                                                const Jac<F2<C>>* pre) {      // pre: two sums per item from k_vid_fixed_coop, or null
  constexpr int HOTW = (int)(sizeof(Jac<F2<C>>) / 4);      // the only users of the slot in this kernel are the running sums of jac_acc_fixed
  __shared__ __attribute__((aligned(16))) u32 hot_lds[ELP_NIZK_BLOCK * HOTW];
  __shared__ VidShared<C> sh[64];
  const int role = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  key.hot = hot_lds + threadIdx.x * HOTW;
  const size_t i = (size_t)blockIdx.x * 64 + lane;
  if (key.vtab) key.vtab += i * (size_t)vtab_words<C>();     // one slice per item: the G2 job uses its first part, the G1 job the rest
  VidNizkState<C> st;
  st.ok = false;
  if (i < n) {
    Aff<F2<C>> aK;
    vid_nizk_jobs<C>(key, role, recs + i * (size_t)rec_words, mask, retr != 0, sh[lane], st, aK, pre ? pre + 2 * i : nullptr);
    if (role == 1) vid_store_k<C>(kws, kstride, i, aK);
  }
  __syncthreads();
  if (role == 0 && i < n) {
    const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
    const size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
    nizk_ok[i] = vid_nizk_finish<C>(sh[lane], st, retr != 0, a, al) ? 1 : 0;
  }
}
// Phase 1 with four job waves per 64 items (vid_nizk_jobs4) for small batches: G2 job | V_phi + K | V_E1 | V_E2 side by side.  The G2 job runs without a hot
// slot (its fixed-base part arrives precomputed), the G1 jobs keep a Jac<F1>-sized one.
template <class C>
struct Nizk4Lds {
  static constexpr int HOTW = (int)(sizeof(Jac<F1<C>>) / 4);
  static constexpr size_t HOT_BYTES = (size_t)192 * HOTW * 4;
  static constexpr size_t BYTES = HOT_BYTES + 64 * sizeof(VidShared<C>);
};
// The G2 job of an item on FOUR lanes (round 5; batches of at most 16 items, where the job wave of a lone call is the longest thing in the launch): lane j of the quad
// multiplies by dimension j of the GLS decomposition of c (curve.h g2_mul_gls_dim: 64 doublings + 17 additions instead of 64 + 68, the psi-images read from the table
// k_vid_prep built), the four results are added through lane exchanges, lane 0 adds the fixed-base part and serialises V_k.  Every lane of the wave must call (the exchanges).
template <class C>
__device__ __forceinline__ void vid_job_g2_quad(const KeyCtx<C>& key, const Scalar& c, u32* vk, const Jac<F2<C>>* pre, int sub, bool live) {
  typedef F2<C> G2F;
  Jac<G2F> Vk;
  jac_set_inf(Vk);
  if (live) g2_mul_gls_dim<C, WsTabPsi<G2F>>(Vk, WsTabPsi<G2F>{key.vtab, key.vpsi}, c, sub);
  ELP_NOUNROLL
  for (int m = 1; m < 4; m <<= 1) {
    Jac<G2F> o;
    {
      constexpr int NW = (int)(sizeof(Jac<G2F>) / 4);
      const i32* src = reinterpret_cast<const i32*>(&Vk);
      i32* dst = reinterpret_cast<i32*>(&o);
      ELP_UNROLL
      for (int w = 0; w < NW; w++) dst[w] = __shfl_xor(src[w], m);
    }
    if (live && (sub & (2 * m - 1)) == 0) jac_add<G2F>(Vk, Vk, o);
  }
  if (live && sub == 0) {
    Jac<G2F> S = *pre;
    jac_add<G2F>(Vk, Vk, S);
    Aff<G2F> aVk;
    jac_to_aff<G2F>(aVk, Vk);
    uint8_t b[2 * C::FBYTES];
    g2_serialize<C>(b, aVk);
    bytes_to_words(vk, b, 2 * C::FBYTES);
  }
}
// body of k_vid_nizk4 for workgroup `block` (256 lanes); hot_lds / sh: Nizk4Lds<C>::HOT_BYTES and 64 VidShared of LDS
template <class C>
__device__ __forceinline__ void vid_nizk4_body(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, const uint8_t* ad, const u32* ad_off, u32 ad_len,
                                               uint8_t* nizk_ok, u32* kws, size_t kstride, size_t n, const Jac<F2<C>>* pre, int k_done, size_t block, u32* hot_lds,
                                               VidShared<C>* sh, bool quad = false) {      // quad: the G2 job wave gives every item four lanes; the workgroup then serves 16 items (psi tables required)
  constexpr int HOTW = Nizk4Lds<C>::HOTW;
  const int role = (int)(threadIdx.x >> 6), lane0 = (int)(threadIdx.x & 63);
  const bool quad0 = quad && role == 0;
  const int rows = quad ? 16 : 64;                                                     // items of this workgroup
  const int sub = quad0 ? (lane0 & 3) : 0, lane = quad0 ? (lane0 >> 2) : lane0;      // `lane`: the item's row of the workgroup (rows and up: lanes that only keep the wave's memory rows full)
  key.hot = role == 0 ? nullptr : hot_lds + (threadIdx.x - 64) * HOTW;
  // A PARTIALLY FILLED WAVE RUNS ITS IDLE LANES ON A COPY OF THE LAST ITEM (round 4).  Private memory is interleaved by lane: one dword of all 64 lanes is one
  // 256-byte row.  With a single active lane every spill / table store is a 4-byte write into a row nobody else touches -- a partial-line write that the memory
  // side turns into read-modify-write -- and a lone el_passo_verify_id took 3.2-4.9 ms in this kernel, depending on which CU (which scratch addresses) it landed
  // on, against 2.17 ms for the same wave with 64 items (profiles/r04_lone_call.md: per-wave times, same clocks).  Full rows cost nothing extra: the idle lanes
  // share the wave's instruction stream anyway.  Copies use their own workspace slices and LDS rows and publish nothing.
  const size_t slot = block * rows + (lane < rows ? lane : rows - 1);
  const bool real = lane < rows && slot < n;
  const size_t i = slot < n ? slot : n - 1;
  // workspace slices.  One item per lane (64 rows): slice = the lane's slot, written by k_vid_prep (G2 part) and by the G1 jobs (their parts).  Four lanes per G2 job
  // (16 rows): the G2 job READS the table and psi-images of its item's slot; the G1 job lanes -- 16 real ones and 48 that keep the rows full -- write tables of their own
  // and get a slice per lane of the launch (block * 64 + lane0: the launcher provides 4 x the slices)
  const size_t gslot = quad ? i : block * 64 + lane0, tslot = quad ? (role == 0 ? i : block * 64 + lane0) : block * 64 + lane0;
  if (key.vtab) key.vtab += tslot * (size_t)vtab_words<C>();
  if (key.vpsi) key.vpsi += gslot * (size_t)(24 * vtab_entry_words<F2<C>>());
  VidNizkState<C> st;
  st.ok = false;
  {
    Aff<F2<C>> aK;
    vid_nizk_jobs4<C>(key, role, recs + i * (size_t)rec_words, mask, retr != 0, sh[lane], st, aK, pre + 2 * i, k_done != 0, k_done == 1, quad0);
    if (real && role == 1 && !k_done) vid_store_k<C>(kws, kstride, i, aK);
  }
  if (quad) {      // uniform over the workgroup: the wave of role 0 runs the exchanges, the others skip
    if (role == 0) vid_job_g2_quad<C>(key, st.c, sh[lane].vk, pre + 2 * i, sub, st.ok);
  }
  __syncthreads();
  if (role == 0 && real && sub == 0) {
    const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
    const size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
    nizk_ok[i] = vid_nizk_finish<C>(sh[lane], st, retr != 0, a, al) ? 1 : 0;
  }
}
template <class C>
__global__ void __launch_bounds__(256) k_vid_nizk4(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, const uint8_t* ad, const u32* ad_off, u32 ad_len,
                                                  uint8_t* nizk_ok, u32* kws, size_t kstride, size_t n, const Jac<F2<C>>* pre, int k_done) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[Nizk4Lds<C>::BYTES];
  vid_nizk4_body<C>(key, recs, rec_words, mask, retr, ad, ad_off, ad_len, nizk_ok, kws, kstride, n, pre, k_done, blockIdx.x, (u32*)smem,
                    (VidShared<C>*)(smem + Nizk4Lds<C>::HOT_BYTES));
}
// ---- the same phase 1 as two CONCURRENT kernels with asymmetric register budgets (ELP_OPT_SPLIT_PHASES = 2): the G2 job (+ K) on the caller's stream
// with 384 registers per lane (256 + 128 accumulation registers as spill space: its Fp2 mixed additions do not fit 256), the G1 job on a second stream
// with 128 -- one wave of each per SIMD.  The jobs hand their results (serialised commitments, K) to phase 2 through the launch workspace, word w of
// item i at ws[w * stride + i]; phase 2 opens with the transcript hash.
template <class C>
ELP_HD constexpr int vid_ws_words() { return 2 * C::FBYTES / 4 + 3 * C::FBYTES / 4 + vid_k_words<C>(); }   // vk | v1[3] | K
#ifndef ELP_G2JOB_VGPRS
#define ELP_G2JOB_VGPRS 384
#endif
template <class C>
__global__ void __launch_bounds__(ELP_BLOCK) __attribute__((amdgpu_num_vgpr(ELP_G2JOB_VGPRS)))
k_vid_g2(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, uint8_t* ok_g2, u32* ws, size_t stride, size_t n) {
  constexpr int HOTW = (int)(sizeof(Jac<F2<C>>) / 4);
  __shared__ __attribute__((aligned(16))) u32 hot_lds[ELP_BLOCK * HOTW];
  key.hot = hot_lds + threadIdx.x * HOTW;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (key.vtab) key.vtab += i * (size_t)vtab_words<C>();
  Aff<F1<C>> sig1, sig2, phi, E1, E2;
  Aff<F2<C>> kk, aK;
  Scalar c;
  RecordSrc<C> src;
  src.sub_ = (key.flags & KEY_NO_SUBGROUP_CHECK) == 0;
  bool ok = src.open(recs + i * (size_t)rec_words, mask, key.A, retr != 0, sig1, sig2, phi, E1, E2, kk, c);
  if (ok && !sig1_strict_ok<C>(key.flags, sig1)) ok = false;
  ok_g2[i] = ok ? 1 : 0;
  if (!ok) return;
  u32 vk[2 * C::FBYTES / 4];
  vid_job_g2<C, RecordSrc<C>>(key, src, retr != 0, kk, c, vk);
  for (int q = 0; q < 2 * C::FBYTES / 4; q++) ws[(size_t)q * stride + i] = vk[q];
  vid_job_k<C, RecordSrc<C>>(key, src, kk, aK);
  vid_store_k<C>(ws + (size_t)(5 * C::FBYTES / 4) * stride, stride, i, aK);
}
template <class C>
__global__ void __launch_bounds__(ELP_BLOCK, 4) k_vid_g1(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, u32* ws, size_t stride, size_t n) {
  constexpr int HOTW = (int)(sizeof(Jac<F1<C>>) / 4);
  __shared__ __attribute__((aligned(16))) u32 hot_lds[ELP_BLOCK * HOTW];
  key.hot = hot_lds + threadIdx.x * HOTW;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (key.vtab) key.vtab += i * (size_t)vtab_words<C>();
  Aff<F1<C>> sig1, sig2, phi, E1, E2;
  Aff<F2<C>> kk;
  Scalar c;
  RecordSrc<C> src;
  src.sub_ = (key.flags & KEY_NO_SUBGROUP_CHECK) == 0;
  if (!src.open(recs + i * (size_t)rec_words, mask, key.A, retr != 0, sig1, sig2, phi, E1, E2, kk, c)) return;   // k_vid_g2 publishes the verdict on the inputs
  u32 v1[3][C::FBYTES / 4];
  vid_job_g1<C, RecordSrc<C>>(key, src, retr != 0, phi, E1, E2, c, v1);
  for (int t = 0; t < (retr ? 3 : 1); t++)
    for (int q = 0; q < C::FBYTES / 4; q++) ws[(size_t)(2 * C::FBYTES / 4 + t * (C::FBYTES / 4) + q) * stride + i] = v1[t][q];
}
// ---- G1 jobs as a kernel of their own (elp/pipeline.h "G1 JOBS AS A KERNEL OF THEIR OWN"): one lane per job, job-uniform waves -- workgroup b runs job b & 3 of
// items [64 (b >> 2), 64 (b >> 2) + 64) -- in the plain layout, Jacobian G1 points only: compiled for several waves per SIMD in a translation unit of its own.
#ifndef ELP_G1JOBS_WAVES
#define ELP_G1JOBS_WAVES 3
#endif
template <class C>
__global__ void __launch_bounds__(ELP_BLOCK, ELP_G1JOBS_WAVES) k_vid_g1jobs(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, u32* ws, size_t stride, size_t n) {
  constexpr int HOTW = (int)(sizeof(Jac<F1<C>>) / 4);
  __shared__ __attribute__((aligned(16))) u32 hot_lds[ELP_BLOCK * HOTW];
  key.hot = hot_lds + threadIdx.x * HOTW;
  const int job = (int)(blockIdx.x & 3u);
  const size_t i = (size_t)(blockIdx.x >> 2) * ELP_BLOCK + threadIdx.x;
  if (i >= n) return;
  // this lane's 8-entry slice of the launch workspace for the table of multiples of its point (jobs 0..2); the workspace of n plain lanes is large enough
  u32* slot = nullptr;
  if (key.vtab && job < 3) slot = key.vtab + ((size_t)job * n + i) * (size_t)(8 * vtab_entry_words<F1<C>>());
  vid_g1_job<C>(key, job, recs + i * (size_t)rec_words, mask, retr != 0, slot, ws, stride, i);
}
// Phase 2 behind the concurrent jobs: transcript hash (src/ps-verifier.cc:111-130), then the pairing check.
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_vid_pair2(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, const uint8_t* ad, const u32* ad_off, u32 ad_len,
                                              const uint8_t* ok_g2, const u32* ws, size_t stride, uint8_t* flags, unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP(key);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n) {
    if (ok_g2[i]) {
      const u32* rec = recs + i * (size_t)rec_words;
      Aff<F1<C>> sig1, sig2, phi, E1, E2;
      Aff<F2<C>> kk, aK;
      Scalar c;
      RecordSrc<C> src;
      src.open_lite(rec, mask, key.A, retr != 0, c);
      u32 vk[2 * C::FBYTES / 4], v1[3][C::FBYTES / 4];
      for (int q = 0; q < 2 * C::FBYTES / 4; q++) vk[q] = ws[(size_t)q * stride + i];
      for (int t = 0; t < (retr ? 3 : 1); t++)
        for (int q = 0; q < C::FBYTES / 4; q++) v1[t][q] = ws[(size_t)(2 * C::FBYTES / 4 + t * (C::FBYTES / 4) + q) * stride + i];
      const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
      const size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
      if (vid_challenge_ok<C, RecordSrc<C>>(src, retr != 0, vk, v1, c, a, al)) {
        vid_load_k<C>(aK, ws + (size_t)(5 * C::FBYTES / 4) * stride, stride, i);
        ok = vid_pair_item<C>(key, rec, aK);
      }
    }
    flags[i] = ok ? 1 : 0;
  }
  count_accept(ok, accepted);
}

// Phase 2: one lane per item with the whole register file: Miller loop on (sig1, K), (-sig2, gg) and the final exponentiation.
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_vid_pair(KeyCtx<C> key, const u32* recs, int rec_words, const uint8_t* nizk_ok, const u32* kws, size_t kstride,
                                             uint8_t* flags, unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP(key);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n) {
    if (nizk_ok[i]) {
      Aff<F2<C>> aK;
      vid_load_k<C>(aK, kws, kstride, i);
      ok = vid_pair_item<C>(key, recs + i * (size_t)rec_words, aK);
    }
    flags[i] = ok ? 1 : 0;
  }
  count_accept(ok, accepted);
}

// ---- cooperative pairing check (elp/coop.h): 32 lanes per item, Fp2 register file in LDS, level-scheduled program from tools/gen_coop.py.
// Only the translation unit elpasso_<curve>_coop.hip sees the program tables (ELP_COOP_TU); the other units call the launchers below.
#ifdef ELP_COOP_TU
template <class C>
struct CoopTables;
// one specialisation per curve whose generated programs the translation unit includes (elp/coop_prog_<curve>.h, namespace coop_<curve>)
#define ELP_COOP_TABLES(CURVE, NS) \
template <> \
struct CoopTables<CURVE> { \
  static __device__ __forceinline__ CoopProg check() { \
    using namespace elp::NS; \
    return CoopProg{CHECK_PROG, CHECK_CLASS, CHECK_TERMS, CHECK_NSTEPS, {CHECK_OUT[0], CHECK_OUT[1], CHECK_OUT[2], CHECK_OUT[3], CHECK_OUT[4], CHECK_OUT[5]}, CHECK_CHUNK_OFF, 16, CHECK_CHUNK_LINE}; \
  } \
  static __device__ __forceinline__ CoopProg check32() { \
    using namespace elp::NS; \
    return CoopProg{CHECK32_PROG, CHECK32_CLASS, CHECK32_TERMS, CHECK32_NSTEPS, {CHECK32_OUT[0], CHECK32_OUT[1], CHECK32_OUT[2], CHECK32_OUT[3], CHECK32_OUT[4], CHECK32_OUT[5]}, CHECK32_CHUNK_OFF, 32, CHECK32_CHUNK_LINE}; \
  } \
  static __device__ __forceinline__ CoopProg tail() { \
    using namespace elp::NS; \
    return CoopProg{TAIL_PROG, TAIL_CLASS, TAIL_TERMS, TAIL_NSTEPS, {TAIL_OUT[0], TAIL_OUT[1], TAIL_OUT[2], TAIL_OUT[3], TAIL_OUT[4], TAIL_OUT[5]}, TAIL_CHUNK_OFF, 32, TAIL_CHUNK_LINE}; \
  } \
  static_assert(elp::NS::CHECK_NP == 16 && elp::NS::CHECK32_NP == 32 && elp::NS::TAIL_NP == 32, "lane pairs per item of the generated programs"); \
  static constexpr int CHUNK = elp::NS::COOP_CHUNK; \
  template <int NP_> \
  static constexpr int max_chunk_terms() { \
    using namespace elp::NS; \
    return NP_ == 16 ? CHECK_MAX_CHUNK_TERMS : (CHECK32_MAX_CHUNK_TERMS > TAIL_MAX_CHUNK_TERMS ? CHECK32_MAX_CHUNK_TERMS : TAIL_MAX_CHUNK_TERMS); \
  } \
  template <int NP_> \
  static constexpr int max_chunk_lines() { \
    using namespace elp::NS; \
    return NP_ == 16 ? CHECK_MAX_CHUNK_LINES : (CHECK32_MAX_CHUNK_LINES > TAIL_MAX_CHUNK_LINES ? CHECK32_MAX_CHUNK_LINES : TAIL_MAX_CHUNK_LINES); \
  } \
  static_assert(elp::NS::IN_ONE == COOP_IN_ONE, "the interpreter reads its zero from the pinned input ONE"); \
  static constexpr int NREG = elp::NS::COOP_NREG, NCONST = elp::NS::COOP_NCONST; \
  static constexpr int IN_P1 = elp::NS::IN_P1, IN_P2 = elp::NS::IN_P2, IN_QX = elp::NS::IN_QX, IN_QY = elp::NS::IN_QY, \
                       IN_ONE = elp::NS::IN_ONE, IN_F0 = elp::NS::IN_F0; \
  static __device__ __forceinline__ const uint8_t (*kinds())[3] { return elp::NS::CONST_KIND; } \
};
#ifdef ELP_COOP_HAVE_BN254
ELP_COOP_TABLES(BN254, coop_bn254)
#endif
#ifdef ELP_COOP_HAVE_BLS12_381
ELP_COOP_TABLES(BLS12_381, coop_bls12_381)
#endif
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_coop_consts(Fp2<C>* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= CoopTables<C>::NCONST) return;
  out[i].c0 = coop_const<C>(CoopTables<C>::kinds(), i, 0);
  out[i].c1 = coop_const<C>(CoopTables<C>::kinds(), i, 1);
}
// runs program P over the register files of the workgroup's two items; every lane walks all steps (empty slots and inactive items idle)
// The program itself (descriptors and the terms of the linear combinations) is staged through LDS in chunks of CHUNK steps by all 64 lanes with coalesced
// loads: read straight from global memory every step would open with a dependent ~1 us load and every term of a combination with another.
// Workgroup of the cooperative kernels: 128 lanes = 4 items x 16 lane pairs (batches: the items share one staged copy of the program; 4 x 8.6 KB of registers +
// 4.9 KB of program = 39.5 KB per workgroup, four workgroups -- 16 items, two waves per SIMD -- per CU) or 2 items x 32 lane pairs (one item per wave: lone items,
// batches that leave SIMDs idle anyway, the one closing pairing of aggregated verification: a quarter fewer steps).
#define ELP_COOP_BLOCK 128
template <class C, int NP, int BLOCK = ELP_COOP_BLOCK>      // BLOCK: lanes of the workgroup that interpret (k_vid_small: all 256, twice the items per workgroup)
struct CoopLds {
  typedef CoopTables<C> T;
  static constexpr int LANES = 2 * NP;                // lanes of one item
  static constexpr int ITEMS = BLOCK / LANES;
  static constexpr int RW = T::NREG * 2 * C::NL;      // words of one item's register file
  static constexpr int RPAD = 17;                     // the items of a wave run the same program in lockstep: without the offset item 1 would hit the banks of item 0 on every access
  static constexpr int R_WORDS = ITEMS * (RW + RPAD) + 2 * ITEMS;      // + two flag words per item
  static constexpr int MAXT = T::template max_chunk_terms<NP>();
  static constexpr int MAXLW = T::template max_chunk_lines<NP>() * 2 * C::NL;          // words of the line coefficients one chunk reads
  static constexpr int STAGE_WORDS = T::CHUNK * NP * 2 + MAXT + T::NCONST * 2 * C::NL + MAXLW;
};
// orders the LDS traffic of one wave: the lanes of an item sit in one wave and the LDS pipeline serves a wave's accesses in order, so between two steps the
// program needs no workgroup barrier, only that the compiler keeps the order
__device__ __forceinline__ void coop_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <class C, int NP, int BLOCK = ELP_COOP_BLOCK>
__device__ __forceinline__ void coop_run_device(const CoopProg& P, coop_i32* R, coop_u32* stage, bool active, int pair, int comp, const Fp2<C>* consts, const Fp2<C>* lines,
                                                bool member = true) {      // member = false: a lane of a wider workgroup that only keeps the workgroup barriers company (k_vid_small)
  typedef CoopTables<C> T;
  constexpr int MAXT = CoopLds<C, NP, BLOCK>::MAXT, MAXLW = CoopLds<C, NP, BLOCK>::MAXLW;
  constexpr int SW = NP * 2;                            // descriptor words per step (two per lane pair)
  constexpr int CH = T::CHUNK, DW = CH * SW;            // descriptor words per chunk
  constexpr int ND = (DW + BLOCK - 1) / BLOCK;     // descriptor words per lane and chunk
  constexpr int NT = (MAXT + BLOCK - 1) / BLOCK;   // entry words per lane and chunk
  constexpr int NLW = (MAXLW + BLOCK - 1) / BLOCK; // line words per lane and chunk
  static_assert(DW % 2 == 0 && MAXT % 2 == 0, "the entries of a chunk start at a multiple of 8 bytes");
  coop_u32* const sd = stage;
  coop_u32* const stt = stage + DW;
  coop_i32* const cw = (coop_i32*)(stage + DW + MAXT);     // the constants, in the layout of the register file
  coop_i32* const lw = cw + T::NCONST * 2 * C::NL;         // the line coefficients of the current chunk
  {
    const i32* src = reinterpret_cast<const i32*>(consts);
    if (member)
      for (int k = (int)threadIdx.x; k < T::NCONST * 2 * C::NL; k += BLOCK) cw[k] = src[k];
  }
  // chunk c+1 travels from global memory to registers WHILE chunk c executes (all loads of a chunk in flight together, fixed trip counts):
  // fetched on demand, every chunk opened with a chain of dependent global round trips of ~1 us on waves that have nothing else to run -- and so did
  // every step that reads a line of the fixed argument, until the lines travelled with the chunk
  u32 pd[ND];
  u32 pt[NT];
  i32 pl[NLW > 0 ? NLW : 1];
  u32 t0n = 0, t1n = 0, cln = 0;
  auto fetch = [&](int s0) {
    if (!member) return;
    const int ns = P.nsteps - s0 < CH ? P.nsteps - s0 : CH;
    t0n = P.chunk_off[s0 / CH];
    t1n = P.chunk_off[s0 / CH + 1];
    cln = P.chunk_line[s0 / CH];
    const u32* src = P.prog + (size_t)s0 * SW;
    ELP_UNROLL
    for (int j = 0; j < ND; j++) {
      const int k = (int)threadIdx.x + j * BLOCK;
      pd[j] = k < ns * SW ? src[k] : ((k & 1) ? 0u : 0xF0000000u);     // past the end: empty slots
    }
    ELP_UNROLL
    for (int j = 0; j < NT; j++) {
      const u32 k = threadIdx.x + (u32)j * BLOCK;
      pt[j] = k < t1n - t0n ? P.terms[t0n + k] : 0u;
    }
    const i32* lsrc = reinterpret_cast<const i32*>(lines) + (size_t)(cln & 0xFFFFu) * (2 * C::NL);
    const u32 lwords = (cln >> 16) * (u32)(2 * C::NL);
    ELP_UNROLL
    for (int j = 0; j < NLW; j++) {
      const u32 k = threadIdx.x + (u32)j * BLOCK;
      pl[j] = k < lwords ? lsrc[k] : 0;
    }
  };
  fetch(0);
  ELP_NOUNROLL
  for (int s0 = 0; s0 < P.nsteps; s0 += CH) {
    const int ns = P.nsteps - s0 < CH ? P.nsteps - s0 : CH;
    const u32 t0 = t0n;
    const int lbase = (int)(cln & 0xFFFFu);
    __syncthreads();                                   // every wave is done with the previous chunk
    if (member) {
      ELP_UNROLL
      for (int j = 0; j < ND; j++) {
        const int k = (int)threadIdx.x + j * BLOCK;
        if (k < DW) sd[k] = pd[j];
      }
      ELP_UNROLL
      for (int j = 0; j < NT; j++) {
        const u32 k = threadIdx.x + (u32)j * BLOCK;
        if (k < (u32)MAXT) stt[k] = pt[j];
      }
      ELP_UNROLL
      for (int j = 0; j < NLW; j++) {
        const u32 k = threadIdx.x + (u32)j * BLOCK;
        if (k < (u32)MAXLW) lw[k] = pl[j];
      }
    }
    __syncthreads();
    if (s0 + CH < P.nsteps) fetch(s0 + CH);
    // the descriptor of step s + 1 is read while step s runs (it does not depend on the registers): one LDS round trip less on the path of every step
    u32 nd0 = sd[pair * 2], nd1 = sd[pair * 2 + 1];
    ELP_NOUNROLL
    for (int s = 0; s < ns; s++) {
      const u32 d0 = nd0, d1 = nd1;
      if (s + 1 < ns) {
        nd0 = sd[(s + 1) * SW + pair * 2];
        nd1 = sd[(s + 1) * SW + pair * 2 + 1];
      }
      if (active) {
        Fp<C> out;
        const int dst = coop_exec_desc<C>(d0, d1, (const coop_u16*)stt - 2 * (size_t)t0, comp, R, cw, lw, lbase, out);
        if (dst >= 0) coop_st<C>(R, dst, comp, out);     // the register allocation never lets a step write a register that the same step reads
      }
      coop_wave_sync();
    }
  }
  __syncthreads();
}
// is the Fp12 value in registers P.out[0..5] equal to 1?  (lane pair j < 6 tests coefficient j; result through LDS word `flagw`)
template <class C>
__device__ __forceinline__ bool coop_is_one(const CoopProg& P, coop_i32* R, bool active, int pair, int comp, coop_i32* flagw, bool member = true) {
  if (member && pair == 0 && comp == 0) *flagw = 1;
  __syncthreads();
  if (active && pair < 6) {
    Fp<C> v = coop_ld<C>(R, P.out[pair], comp);
    if (pair == 0 && comp == 0) v = fp_sub_lazy(v, fp_one<C>());
    if (!fp_is_zero<C>(v)) *flagw = 0;
  }
  __syncthreads();
  return member && *flagw != 0;
}
// items [0, n): sig1 | sig2 at the head of the record, K in the workspace (vid_store_k layout); todo[i] != 0 selects the items to check.  An item whose sig1,
// sig2 or K is the point at infinity (or fails validation) is left to the per-lane kernel: done[i] stays 0.  Otherwise flags[i] = verdict, done[i] = 1.
template <class C, int NP, int BLOCK = ELP_COOP_BLOCK>
__device__ __forceinline__ void pair_coop_body(const KeyCtx<C>& key, const Fp2<C>* consts, const u32* recs, int rec_words, const uint8_t* todo, const u32* kws,
                                               size_t kstride, uint8_t* flags, uint8_t* done, unsigned long long* accepted, size_t n, size_t block, i32* Rall,
                                               u32* stage) {      // Rall / stage: CoopLds<C, NP>::R_WORDS / STAGE_WORDS words of LDS; lanes 0..127 of the workgroup
  typedef CoopTables<C> T;
  typedef CoopLds<C, NP, BLOCK> L;
  // The interpreter is laid out for BLOCK lanes.  In a wider workgroup the lanes above take part in every workgroup barrier and in
  // nothing else -- `member` false, no LDS / global access: leaving early instead would rely on the barrier not counting terminated waves, which the HIP
  // programming model does not promise.
  const bool member = threadIdx.x < BLOCK;
  const int slot = member ? (int)(threadIdx.x / L::LANES) : 0, pair = (int)((threadIdx.x % L::LANES) >> 1), comp = (int)(threadIdx.x & 1);
  coop_i32* R = (coop_i32*)Rall + slot * (L::RW + L::RPAD);
  coop_i32* flagw = (coop_i32*)Rall + L::ITEMS * (L::RW + L::RPAD) + slot;     // [slot]: is-one flag, [ITEMS + slot]: "inputs usable"
  const size_t i = block * L::ITEMS + slot;
  if (member && pair == 0 && comp == 0) {
    int usable = 0;
    if (i < n && todo[i]) {
      const u32* rec = recs + i * (size_t)rec_words;
      Aff<F1<C>> s1, s2;
      Aff<F2<C>> K;
      if (g1_load<C>(s1, rec) && g1_load<C>(s2, rec + 2 * C::N)) {
        vid_load_k<C>(K, kws, kstride, i);
        if (!aff_is_inf(s1) && !aff_is_inf(s2) && !aff_is_inf(K)) {
          usable = 1;
          coop_st<C>(R, T::IN_P1, 0, s1.x);
          coop_st<C>(R, T::IN_P1, 1, s1.y);
          coop_st<C>(R, T::IN_P2, 0, s2.x);
          coop_st<C>(R, T::IN_P2, 1, fp_neg(s2.y));
          coop_st<C>(R, T::IN_QX, 0, K.x.c0);
          coop_st<C>(R, T::IN_QX, 1, K.x.c1);
          coop_st<C>(R, T::IN_QY, 0, K.y.c0);
          coop_st<C>(R, T::IN_QY, 1, K.y.c1);
          coop_st<C>(R, T::IN_ONE, 0, fp_one<C>());
          coop_st<C>(R, T::IN_ONE, 1, fp_zero<C>());
        }
      }
    }
    flagw[L::ITEMS] = usable;
  }
  __syncthreads();
  const bool active = member && flagw[L::ITEMS] != 0;
  const CoopProg P = NP == 16 ? T::check() : T::check32();
  coop_run_device<C, NP, BLOCK>(P, R, (coop_u32*)stage, active, pair, comp, consts, reinterpret_cast<const Fp2<C>*>(key.gg_lines), member);
  const bool one = coop_is_one<C>(P, R, active, pair, comp, flagw, member);
  if (pair == 0 && comp == 0 && active) {
    flags[i] = one ? 1 : 0;
    done[i] = 1;
    if (one && accepted) atomicAdd(accepted, 1ull);
  }
}
template <class C, int NP>
__global__ void __launch_bounds__(ELP_COOP_BLOCK) k_pair_coop(KeyCtx<C> key, const Fp2<C>* consts, const u32* recs, int rec_words, const uint8_t* todo, const u32* kws,
                                                            size_t kstride, uint8_t* flags, uint8_t* done, unsigned long long* accepted, size_t n) {
  typedef CoopLds<C, NP> L;
  __shared__ __attribute__((aligned(16))) i32 Rall[L::R_WORDS];
  __shared__ __attribute__((aligned(16))) u32 stage[L::STAGE_WORDS];
  pair_coop_body<C, NP>(key, consts, recs, rec_words, todo, kws, kstride, flags, done, accepted, n, blockIdx.x, Rall, stage);
}
#ifdef ELP_DBG_SMALL
__device__ unsigned long long elp_dbg_small[32];
#endif
// Small batches of el_passo_verify_id: the NIZK half (vid_nizk4_body, workgroups [0, nb_nizk)) and the pairing check (pair_coop_body, the workgroups after them)
// of the SAME launch -- the two are independent once K, the fixed-base sums and the table of multiples of k exist, and one launch lets the chip
// run them side by side without a second stream.
#define ELP_VID_SMALL_PARAMS                                                                                                                                     \
  KeyCtx<C> key, const Fp2<C>*consts, const u32 *recs, int rec_words, u64 mask, int retr, const uint8_t *ad, const u32 *ad_off, u32 ad_len, uint8_t *nizk_ok,          \
      const uint8_t *kvalid, const u32 *kws, size_t kstride, uint8_t *pair_ok, uint8_t *done, size_t n, const Jac<F2<C>>*pre, unsigned nb_nizk
template <class C, int NP>
__device__ __forceinline__ void vid_small_body(ELP_VID_SMALL_PARAMS, unsigned char* smem) {
  typedef CoopLds<C, NP, 256> L;              // all four waves of a pairing workgroup interpret: 8 (4) items per workgroup
  if (blockIdx.x < nb_nizk) {
    vid_nizk4_body<C>(key, recs, rec_words, mask, retr, ad, ad_off, ad_len, nizk_ok, (u32*)nullptr, kstride, n, pre, 1, blockIdx.x, (u32*)smem,
                      (VidShared<C>*)(smem + Nizk4Lds<C>::HOT_BYTES), (key.flags & KEY_QUAD_G2) != 0 && key.vtab != nullptr && key.vpsi != nullptr);
  } else {
    // the workgroup carries the NIZK half's register budget, so all four waves interpret: a compute unit then holds half the items of
    // k_pair_coop's sixteen on half its waves, each wave alone on its SIMD (round 4; before, two of the four waves only kept the barriers company)
    KeyCtx<C> k2 = key;
    k2.vtab = nullptr;
    pair_coop_body<C, NP, 256>(k2, consts, recs, rec_words, kvalid, kws, kstride, pair_ok, done, nullptr, n, blockIdx.x - nb_nizk, (i32*)smem,
                               (u32*)(smem + (((size_t)L::R_WORDS * 4 + 15) & ~(size_t)15)));
  }
}
template <class C, int NP>
struct VidSmallLds {
  typedef CoopLds<C, NP, 256> L;
  static constexpr size_t PAIR_BYTES = (size_t)(L::R_WORDS + L::STAGE_WORDS) * 4 + 16;
  static constexpr size_t BYTES = PAIR_BYTES > Nizk4Lds<C>::BYTES ? PAIR_BYTES : Nizk4Lds<C>::BYTES;
};
template <class C, int NP>
__global__ void __launch_bounds__(256) k_vid_small(ELP_VID_SMALL_PARAMS) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[VidSmallLds<C, NP>::BYTES];
#ifdef ELP_DBG_SMALL      /* measurement builds only (tools/probes/lone_call_probe.py): per-wave wall-clock times and placement of the first two workgroups */
  const unsigned long long dbg_t0 = wall_clock64();
  struct DbgAtExit {
    unsigned long long t0;
    __device__ ~DbgAtExit() {
      if ((threadIdx.x & 63) == 0 && blockIdx.x < 2) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* d = elp_dbg_small + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        d[0] = t0;
        d[1] = wall_clock64();
        d[2] = hw;
        d[3] = xcc;
      }
    }
  } dbg_at_exit{dbg_t0};
#endif
  vid_small_body<C, NP>(key, consts, recs, rec_words, mask, retr, ad, ad_off, ad_len, nizk_ok, kvalid, kws, kstride, pair_ok, done, n, pre, nb_nizk, smem);
}
// The same launch for batches that need more than one round of pairing workgroups: built for TWO waves per SIMD (256 registers: the NIZK half spills more and
// takes longer, which a batch of thousands hides), so that a compute unit holds two pairing workgroups -- the sixteen items of k_pair_coop -- beside the NIZK
// workgroups instead of after them.
template <class C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_vid_small2(ELP_VID_SMALL_PARAMS) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[VidSmallLds<C, 16>::BYTES];
  vid_small_body<C, 16>(key, consts, recs, rec_words, mask, retr, ad, ad_off, ad_len, nizk_ok, kvalid, kws, kstride, pair_ok, done, n, pre, nb_nizk, smem);
}
// the items k_pair_coop left alone (points at infinity): the ordinary per-lane check
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_pair_rest(KeyCtx<C> key, const u32* recs, int rec_words, const uint8_t* todo, const uint8_t* done, const u32* kws, size_t kstride,
                                              uint8_t* flags, unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP(key);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n && !done[i]) {
    if (todo[i]) {
      Aff<F2<C>> aK;
      vid_load_k<C>(aK, kws, kstride, i);
      ok = vid_pair_item<C>(key, recs + i * (size_t)rec_words, aK);
    }
    flags[i] = ok ? 1 : 0;
  }
  count_accept(ok, accepted);
}
// K of a plain PS verification (src/ps-verifier.cc:20-29: XX prod YY_i^{m_i}), affine, into the workspace; todo[i] = the record decodes and sig1 != infinity (:16-18).
// Fixed-base sums on ELP_PSK_LANES lanes: the nterms x nwin table entries of  sum_t k_t B_t  are dealt round-robin to the lanes (each recomputes the signed digit
// of its window from the scalar: the carry chain is a few integer operations per window), partial sums are folded through lane shuffles with complete Jacobian
// additions.  `term(t, base, k)` names term t.  All 64 lanes of the wave must call (the shuffles); lane 0 of every group of 8 returns the sum.
#ifndef ELP_PSK_LANES
#define ELP_PSK_LANES 8       /* 16 everywhere was measured in round 4: a lone el_passo_verify_id 2.41 instead of 2.49 ms, but 4 096 of them 4.93 instead of 4.76 and config 2 unchanged (2.96 ms): launch_vid_prep takes 16 for the smallest batches only */
#endif
template <class C, int J = ELP_PSK_LANES, class TermFn>
__device__ __forceinline__ void coop_fixed_sum_g2(Jac<F2<C>>& K, const KeyCtx<C>& key, int sub, int nterms, bool live, TermFn term) {
  typedef F2<C> G;
  if (live) {
    // the mixed addition inlined once (the running sum stays in registers: as a call it moved 216 bytes per lane through private memory on every step -- the
    // kernel ran at 8 cycles per instruction with under one wave resident per SIMD, profiles/r05_four_lane.md) and the table entry of the NEXT step fetched before
    // the current addition is computed: its address depends on the scalars only
    const int W = key.W, nwin = key.nwin, total = nterms * nwin;
    auto entry_of = [&](int t, bool& have) {
      Aff<G> e;
      aff_set_inf(e);
      have = false;
      if (t < total) {
        const int a = t / nwin, j = t - a * nwin;
        int base;
        Scalar kraw;
        term(a, base, kraw);
        const Scalar k = scalar_mod_r<C>(kraw);
        int carry = 0, d = 0;
        for (int jj = 0; jj <= j; jj++) d = fixed_base_digit(k, jj, W, carry);
        if (d != 0) {
          e = aff_from_mem<G>(key.t2[((size_t)base * nwin + j) * key.per + ((d < 0 ? -d : d) - 1)]);
          if (d < 0) e.y = G::neg(e.y);
          have = true;
        }
      }
      return e;
    };
    bool have = false;
    Aff<G> e = entry_of(sub, have);
    ELP_NOUNROLL
    for (int t = sub; t < total; t += J) {
      bool have_next = false;
      const Aff<G> en = entry_of(t + J, have_next);
      if (have) jac_madd_inl<G>(K, K, e);
      e = en;
      have = have_next;
    }
  }
  ELP_NOUNROLL
  for (int m = 1; m < J; m <<= 1) {
    Jac<G> o;
    {
      constexpr int NW = (int)(sizeof(Jac<G>) / 4);
      const i32* src = reinterpret_cast<const i32*>(&K);
      i32* dst = reinterpret_cast<i32*>(&o);
      ELP_UNROLL
      for (int w = 0; w < NW; w++) dst[w] = __shfl_xor(src[w], m);
    }
    if (live && (sub & (2 * m - 1)) == 0) jac_add<G>(K, K, o);
  }
}
// K of a plain PS verification on ELP_PSK_LANES lanes per item: 48 sequential mixed additions (A = 3, W = 16) become 6 + 3 full ones;
// the one inversion of the affine result stays.
template <class C, int J>
__device__ __forceinline__ void ps_k_coop_body(const KeyCtx<C>& key, const u32* recs, int rec_words, int nattr, uint8_t* todo, u32* kws, size_t kstride, size_t n) {
  typedef F2<C> G;
  const int sub = (int)(threadIdx.x & (J - 1));
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / J;
  bool ok = false;
  Jac<G> K;
  jac_set_inf(K);
  const u32* rec = recs + (i < n ? i : 0) * (size_t)rec_words;
  if (i < n) {
    Aff<F1<C>> s1, s2;
    ok = g1_load<C>(s1, rec) && g1_load<C>(s2, rec + 2 * C::N) && sig1_admissible<C>(key.flags, s1);       // every lane of the item decides the same (src/ps-verifier.cc:16-18, in the order-r component)
    if (ok && sub == 0) jac_from_aff(K, aff_from_mem<G>(key.b2[G2_BASE_XX]));
  }
  coop_fixed_sum_g2<C, J>(K, key, sub, nattr, ok, [&](int a, int& base, Scalar& k) {
    base = G2_BASE_YY0 + a;
    k = scalar_load_w(rec + 4 * C::N + 8 * a);
  });
  if (i < n && sub == 0) {
    todo[i] = ok ? 1 : 0;
    if (ok) {
      Aff<G> aK;
      jac_to_aff<G>(aK, K);
      vid_store_k<C>(kws, kstride, i, aK);
    }
  }
}
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_ps_k_coop(KeyCtx<C> key, const u32* recs, int rec_words, int nattr, uint8_t* todo, u32* kws, size_t kstride, size_t n) {
  ps_k_coop_body<C, ELP_PSK_LANES>(key, recs, rec_words, nattr, todo, kws, kstride, n);
}
// The same for batches that are work, not latency (above 4 096 items): four lanes per sum and 256-thread workgroups (the four waves of a workgroup go to the four SIMDs
// of one compute unit; one-wave workgroups were seen to double up on SIMDs in k_pair4, profiles/r05_four_lane.md -- for these kernels the measured gain came from
// the inlined mixed addition of coop_fixed_sum_g2, not from the workgroup shape: 1.83 -> 1.79 -> 1.51 ms for the sums and tables of 16 384 items).
#define ELP_WIDE_BLOCK 256
template <class C>
__global__ void __launch_bounds__(ELP_WIDE_BLOCK) k_ps_k_coop_wide(KeyCtx<C> key, const u32* recs, int rec_words, int nattr, uint8_t* todo, u32* kws, size_t kstride, size_t n) {
  ps_k_coop_body<C, 4>(key, recs, rec_words, nattr, todo, kws, kstride, n);
}
// The fixed-base halves of el_passo_verify_id's G2 work for small batches, 2 x ELP_PSK_LANES lanes per item (src/ps-verifier.cc:76-88,220-227): the lower half computes
//     out[2 i]     = sum_{hidden} rs_j YY_i + r_t gg + (1 - c) XX          (V_k without its [c]k term)
// and the upper half
//     out[2 i + 1] = sum_{revealed} m_i YY_i                               (K without k)
// from the record's scalars alone (any 256-bit value is a valid scalar: nothing to validate here; k_vid_nizk validates the points and ignores the sums of an
// invalid record).  Jacobian results: the NIZK jobs add them with complete additions.
template <class C, int J = ELP_PSK_LANES>      // J lanes per sum: 8 for batches, 16 for the smallest (launch_vid_prep)
__device__ __forceinline__ void vid_fixed_coop_body(const KeyCtx<C>& key, const u32* recs, int rec_words, u64 mask, int retr, Jac<F2<C>>* out, u32* kws, size_t kstride,
                                                    uint8_t* kvalid, size_t n, size_t block) {
  typedef F2<C> G;
  const int sub = (int)(threadIdx.x & (J - 1)), which = (int)((threadIdx.x / J) & 1);
  const size_t i = (block * blockDim.x + threadIdx.x) / (2 * J);
  const bool live = i < n;
  PairedRecordSrc<C> src;
  src.init(recs + (live ? i : 0) * (size_t)rec_words, mask, key.A, retr != 0);
  int H = 0;
  for (int a = 0; a < key.A; a++) H += (int)((mask >> a) & 1);
  const int nterms = which == 0 ? H + 2 : key.A - H;
  Jac<G> S;
  jac_set_inf(S);
  coop_fixed_sum_g2<C, J>(S, key, sub, nterms, live, [&](int t, int& base, Scalar& k) {
    if (which == 0) {
      if (t < H) {                                  // t-th hidden attribute: response rs_t on YY_a
        int a = 0, seen = 0;
        for (; a < key.A; a++)
          if ((mask >> a) & 1) {
            if (seen == t) break;
            seen++;
          }
        base = G2_BASE_YY0 + a;
        k = src.rs(t);
      } else if (t == H) {
        base = G2_BASE_GG;
        k = src.rs(retr ? src.nrs() - 2 : src.nrs() - 1);
      } else {
        base = G2_BASE_XX;
        k = scalar_one_minus<C>(scalar_load_w(src.w_k_ + 4 * C::N));
      }
    } else {                                        // t-th revealed attribute: its hash on YY_a
      int a = 0, seen = 0;
      for (; a < key.A; a++)
        if (!((mask >> a) & 1)) {
          if (seen == t) break;
          seen++;
        }
      base = G2_BASE_YY0 + a;
      k = scalar_load_w(src.w_ms_ + 8 * t);
    }
  });
  if (live && sub == 0) {
    out[2 * i + which] = S;
    if (which == 1) {
      // K = k + that sum, affine, straight into the workspace (vid_store_k layout): the pairing check of the item can start while its NIZK half is still
      // being verified (k_pair_coop on a second stream).  kvalid[i] = the record's k decodes (on the twist); everything else is the NIZK kernel's business.
      Aff<G> kk, aK;
      const bool okk = g2_load<C>(kk, src.w_k_);
      aff_set_inf(aK);
      if (okk) {
        jac_madd<G>(S, S, kk);
        jac_to_aff<G>(aK, S);
      }
      vid_store_k<C>(kws, kstride, i, aK);
      kvalid[i] = okk ? 1 : 0;
    }
  }
}
// The table of multiples 1k .. 8k of an item's commitment k (affine, one inversion), into the G2 part of the item's workspace slice, where the GLS multiplication
// of k_vid_nizk4 reads it: one lane per item, beside k_vid_fixed_coop on the second stream.  A record whose k does not decode leaves its slice alone (the NIZK
// kernel rejects the item before it would read it).
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_vid_fixed_coop(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, Jac<F2<C>>* out, u32* kws, size_t kstride,
                                                   uint8_t* kvalid, size_t n) {
  vid_fixed_coop_body<C>(key, recs, rec_words, mask, retr, out, kws, kstride, kvalid, n, blockIdx.x);
}
template <class C, bool QUAD = false>      // QUAD: four lanes per item -- all four build the multiples, lane 0 stores them, lane j = 1, 2, 3 makes and stores their psi^j images
__device__ __forceinline__ void vid_ktab_body(const KeyCtx<C>& key, const u32* recs, int rec_words, int retr, size_t n, size_t block) {
  typedef F2<C> G;
  // idle lanes of the last wave work on a copy of the last item and fill their OWN slices (full rows of private memory, and the table a copy lane of
  // vid_nizk4_body reads): see the comment there
  const size_t lin = block * blockDim.x + threadIdx.x;
  const size_t slot = QUAD ? lin >> 2 : lin;
  const int sub = QUAD ? (int)(lin & 3) : 0;
  if (n == 0 || !key.vtab) return;
  if (slot >= ((n + ELP_BLOCK - 1) / ELP_BLOCK) * (size_t)ELP_BLOCK) return;      // the workspace has one slice per lane of the 64-item waves: workgroups wider than a wave (k_vid_prep_wide) end beyond it
  const size_t i = slot < n ? slot : n - 1;
  u32* const w = key.vtab + slot * (size_t)vtab_words<C>();
  u32* const wp = key.vpsi ? key.vpsi + slot * (size_t)(24 * vtab_entry_words<G>()) : nullptr;     // psi^j of every multiple beside the table (WsTabPsi)
  Aff<G> kk;
  if (!g2_load<C>(kk, recs + i * (size_t)rec_words + (retr ? 5 : 3) * 2 * C::N)) return;
  if (aff_is_inf(kk)) {                     // k = O: every multiple is O (the multiplication then contributes nothing, as with a table built in place)
    if (sub == 0)
      for (int q = 0; q < 8; q++) vtab_store<G>(w, q, kk);
    if (wp)
      for (int jj = 1; jj < 4; jj++)
        if (!QUAD || sub == jj)
          for (int q = 0; q < 8; q++) vtab_store<G>(wp, (jj - 1) * 8 + q, kk);
    return;
  }
  Jac<G> jk[8];
  jac_multiples8<G>(jk, kk);
  Fp2<C> z2[7], zi2[7];
  for (int q = 1; q < 8; q++) z2[q - 1] = jk[q].Z;
  batch_zinv<C, 0, 7>((Fp<C>*)0, (const Fp<C>*)0, zi2, z2);
  for (int q = 0; q < 8; q++) {
    Aff<G> a = kk;
    if (q) jac_to_aff_with_zinv<G>(a, jk[q], zi2[q - 1]);
    if (sub == 0) vtab_store<G>(w, q, a);
    if (wp) {
      if (QUAD) {
        if (sub != 0) {
          g2_psi_aff<C>(a, sub);
          vtab_store<G>(wp, (sub - 1) * 8 + q, a);
        }
      } else {
        for (int jj = 1; jj < 4; jj++) {
          Aff<G> t = a;
          g2_psi_aff<C>(t, jj);
          vtab_store<G>(wp, (jj - 1) * 8 + q, t);
        }
      }
    }
  }
}
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_vid_ktab(KeyCtx<C> key, const u32* recs, int rec_words, int retr, size_t n) {
  vid_ktab_body<C>(key, recs, rec_words, retr, n, blockIdx.x);
}
// The two of them as workgroup ranges of ONE launch (round 4: they are independent, and on a lone call each is a single workgroup on a chip of 256 CUs --
// 0.33 ms and 0.21 ms one after the other, 0.33 ms side by side): workgroups [0, nb_fixed) sum, the rest build tables.
template <class C, int J, bool QUAD = false>
__global__ void ELP_LAUNCH_BOUNDS k_vid_prep(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, Jac<F2<C>>* out, u32* kws, size_t kstride, uint8_t* kvalid,
                                             size_t n, unsigned nb_fixed) {
  if (blockIdx.x < nb_fixed)
    vid_fixed_coop_body<C, J>(key, recs, rec_words, mask, retr, out, kws, kstride, kvalid, n, blockIdx.x);
  else
    vid_ktab_body<C, QUAD>(key, recs, rec_words, retr, n, blockIdx.x - nb_fixed);
}
template <class C, int J>
__global__ void __launch_bounds__(ELP_WIDE_BLOCK) k_vid_prep_wide(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, Jac<F2<C>>* out, u32* kws, size_t kstride,
                                                                  uint8_t* kvalid, size_t n, unsigned nb_fixed) {
  if (blockIdx.x < nb_fixed)
    vid_fixed_coop_body<C, J>(key, recs, rec_words, mask, retr, out, kws, kstride, kvalid, n, blockIdx.x);
  else
    vid_ktab_body<C, false>(key, recs, rec_words, retr, n, blockIdx.x - nb_fixed);
}
// verdict of a small-batch el_passo_verify_id = its NIZK half (k_vid_nizk4) AND its pairing check (k_pair_coop / k_pair_rest), which ran side by side
template <class C>      // (a template only so that every translation unit that instantiates it gets its own copy)
__global__ void ELP_LAUNCH_BOUNDS k_vid_combine(const uint8_t* nizk_ok, const uint8_t* pair_ok, uint8_t* flags, unsigned long long* accepted, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n) {
    ok = nizk_ok[i] != 0 && pair_ok[i] != 0;
    flags[i] = ok ? 1 : 0;
  }
  count_accept(ok, accepted);
}
// closing step of aggregated verification on one wave (32 lane pairs): [F f_gg(-S2)]^e == 1
template <class C>
__global__ void __launch_bounds__(ELP_COOP_BLOCK) k_agg_final_coop(KeyCtx<C> key, const Fp2<C>* consts, const Fp12<C>* F, const u32* s2_std, int* agg_ok) {
  typedef CoopTables<C> T;
  typedef CoopLds<C, 32> L;
  __shared__ __attribute__((aligned(16))) i32 Rall[L::R_WORDS];
  __shared__ __attribute__((aligned(16))) u32 stage[L::STAGE_WORDS];
  const int slot = (int)(threadIdx.x / L::LANES), pair = (int)((threadIdx.x % L::LANES) >> 1), comp = (int)(threadIdx.x & 1);
  coop_i32* R = (coop_i32*)Rall + slot * (L::RW + L::RPAD);
  coop_i32* flagw = (coop_i32*)Rall + L::ITEMS * (L::RW + L::RPAD) + slot;
  if (pair == 0 && comp == 0) {
    int state = 0;                                  // 0: run the program, 1: verdict is "false" (bad point)
    if (slot == 0) {
      Aff<F1<C>> s2;
      if (!g1_load<C>(s2, s2_std)) state = 1;
      // S2 = infinity: f_gg(O) = 1 -- the line evaluations degenerate; let a y-coordinate of 0 stand in (lines evaluated at (0, 0) leave only their
      // constant coefficient, an element of a proper subfield): handled by the ordinary kernel instead, see launch code
      coop_st<C>(R, T::IN_P2, 0, s2.x);
      coop_st<C>(R, T::IN_P2, 1, fp_neg(s2.y));
      coop_st<C>(R, T::IN_ONE, 0, fp_one<C>());
      coop_st<C>(R, T::IN_ONE, 1, fp_zero<C>());
      const Fp2<C>* e[6] = {&F->c0.c0, &F->c0.c1, &F->c0.c2, &F->c1.c0, &F->c1.c1, &F->c1.c2};
      for (int j = 0; j < 6; j++) {
        coop_st<C>(R, T::IN_F0 + j, 0, e[j]->c0);
        coop_st<C>(R, T::IN_F0 + j, 1, e[j]->c1);
      }
    }
    flagw[L::ITEMS] = (slot == 0 && state == 0) ? 1 : 0;
  }
  __syncthreads();
  const bool active = flagw[L::ITEMS] != 0;
  const CoopProg P = T::tail();
  coop_run_device<C, 32>(P, R, (coop_u32*)stage, active, pair, comp, consts, reinterpret_cast<const Fp2<C>*>(key.gg_lines));
  const bool one = coop_is_one<C>(P, R, active, pair, comp, flagw);
  if (threadIdx.x == 0) *agg_ok = (active && one) ? 1 : 0;
}
template <class B>
void launch_coop_consts(hipStream_t stream, void* d_consts) {
  hipLaunchKernelGGL((k_coop_consts<B>), dim3(1), dim3(ELP_BLOCK), 0, stream, (Fp2<B>*)d_consts);
}
#define ELP_REST_BY_CALLER ((hipStream_t)(intptr_t)-1)
template <class B>
void launch_pair_coop(hipStream_t stream, const KeyCtx<B>& key, const void* d_consts, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride,
                      uint8_t* d_flags, uint8_t* done, void* d_accepted, hipStream_t rest_stream) {      // rest_stream: where the per-lane kernel for the left-over items runs (null: `stream`)
  // up to 512 items (half the SIMDs: the other half is free for the kernel that runs beside this one): one item per wave, 32 lane pairs, 982 steps; above: two
  // items per wave, 16 lane pairs each, 1293 steps
  if (n <= 512)
    hipLaunchKernelGGL((k_pair_coop<B, 32>), dim3((unsigned)((n + 1) / 2)), dim3(ELP_COOP_BLOCK), 0, stream, key, (const Fp2<B>*)d_consts, (const u32*)d_records, words, todo, kws, kstride,
                       d_flags, done, (unsigned long long*)d_accepted, n);
  else
    hipLaunchKernelGGL((k_pair_coop<B, 16>), dim3((unsigned)((n + 3) / 4)), dim3(ELP_COOP_BLOCK), 0, stream, key, (const Fp2<B>*)d_consts, (const u32*)d_records, words, todo, kws, kstride,
                       d_flags, done, (unsigned long long*)d_accepted, n);
  // k_pair_rest carries the per-lane pairing's private frame (13.8 KB per lane): kernels with a large frame stay on ONE hardware queue per process -- scratch blocks
  // are per queue, and a second queue that needs one makes the runtime reclaim the first queue's (seconds per event; DESIGN.md section 5, profiles/r04_scratch_stall.md)
  if (rest_stream == ELP_REST_BY_CALLER) return;      // the caller queues launch_pair_rest itself (on its own stream, behind an event of `stream`)
  hipLaunchKernelGGL((k_pair_rest<B>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, rest_stream ? rest_stream : stream, key, (const u32*)d_records, words, todo, (const uint8_t*)done, kws,
                     kstride, d_flags, (unsigned long long*)d_accepted, n);
}
template <class B>
void launch_pair_rest(hipStream_t stream, const KeyCtx<B>& key, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags,
                      uint8_t* done, void* d_accepted) {
  hipLaunchKernelGGL((k_pair_rest<B>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, todo, (const uint8_t*)done, kws, kstride, d_flags,
                     (unsigned long long*)d_accepted, n);
}
template <class B>
void launch_ps_k(hipStream_t stream, const KeyCtx<B>& key, size_t n, const void* d_records, int words, int nattr, uint8_t* todo, u32* kws, size_t kstride) {
  if (n > 4096) {      // four lanes per sum, 256-thread workgroups (k_ps_k_coop_wide)
    hipLaunchKernelGGL((k_ps_k_coop_wide<B>), dim3((unsigned)((n * 4 + ELP_WIDE_BLOCK - 1) / ELP_WIDE_BLOCK)), dim3(ELP_WIDE_BLOCK), 0, stream, key, (const u32*)d_records, words,
                       nattr, todo, kws, kstride, n);
    return;
  }
  hipLaunchKernelGGL((k_ps_k_coop<B>), dim3(grid_for(n * ELP_PSK_LANES)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, nattr, todo, kws, kstride, n);
}
template <class B>
void launch_vid_fixed_coop(hipStream_t stream, const KeyCtx<B>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, void* pre, u32* kws, size_t kstride,
                           uint8_t* kvalid) {
  hipLaunchKernelGGL((k_vid_fixed_coop<B>), dim3(grid_for(n * 2 * ELP_PSK_LANES)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr,
                     (Jac<F2<B>>*)pre, kws, kstride, kvalid, n);
}
template <class B>
void launch_vid_prep(hipStream_t stream, const KeyCtx<B>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, void* pre, u32* kws, size_t kstride,
                     uint8_t* kvalid) {
  // 16 lanes per fixed-base sum where the call is latency (a lone call 2.41 instead of 2.49 ms, 64 items 2.54 / 2.61), 8 where it is work (4 096 items: 4.76 / 4.93)
  // with the psi-images wanted (key.vpsi: batches whose NIZK workgroup is the critical path) the table of an item is built on four lanes: a lone call's k_vid_prep 0.36 -> 0.30 ms;
  // 4 * grid_for(n) table workgroups, so that the slots up to the end of the last 64-item wave are filled like in the one-lane form (the copy lanes of
  // vid_nizk4_body read their own slices: round-4 advisor finding -- with grid_for(4 n) they read workspace this call had not written)
  if (n <= 512) {
    const unsigned nbf = grid_for(n * 2 * 16);
    if (key.vpsi)
      hipLaunchKernelGGL((k_vid_prep<B, 16, true>), dim3(nbf + 4 * grid_for(n)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr, (Jac<F2<B>>*)pre,
                         kws, kstride, kvalid, n, nbf);
    else
      hipLaunchKernelGGL((k_vid_prep<B, 16>), dim3(nbf + grid_for(n)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr, (Jac<F2<B>>*)pre, kws,
                         kstride, kvalid, n, nbf);
    return;
  }
  if (key.vpsi) {
    const unsigned nbf8 = grid_for(n * 2 * ELP_PSK_LANES);
    hipLaunchKernelGGL((k_vid_prep<B, ELP_PSK_LANES, true>), dim3(nbf8 + 4 * grid_for(n)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr,
                       (Jac<F2<B>>*)pre, kws, kstride, kvalid, n, nbf8);
    return;
  }
  if (n > 4096) {      // beyond one wave per SIMD of eight-lane sums the lane count is work: four lanes per sum (20 mixed + 2 complete additions per lane instead of 10 + 3)
    // ... in 256-thread workgroups (k_vid_prep_wide; 16 384 items: 1.83 -> see profiles/r05_four_lane.md): one-wave workgroups leave SIMDs idle at these launch sizes
    const unsigned nbf4 = (unsigned)((n * 2 * 4 + ELP_WIDE_BLOCK - 1) / ELP_WIDE_BLOCK);
    const unsigned nbt = (unsigned)(((size_t)grid_for(n) * ELP_BLOCK + ELP_WIDE_BLOCK - 1) / ELP_WIDE_BLOCK);      // table slots up to the end of the last 64-item wave
    hipLaunchKernelGGL((k_vid_prep_wide<B, 4>), dim3(nbf4 + nbt), dim3(ELP_WIDE_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr, (Jac<F2<B>>*)pre, kws,
                       kstride, kvalid, n, nbf4);
    return;
  }
  if (n > 2048) {      // eight lanes per sum, but in 256-thread workgroups: 3 072 items are 816 one-wave workgroups, more than the dispatcher spreads evenly
    const unsigned nbf8 = (unsigned)((n * 2 * ELP_PSK_LANES + ELP_WIDE_BLOCK - 1) / ELP_WIDE_BLOCK);
    const unsigned nbt = (unsigned)(((size_t)grid_for(n) * ELP_BLOCK + ELP_WIDE_BLOCK - 1) / ELP_WIDE_BLOCK);
    hipLaunchKernelGGL((k_vid_prep_wide<B, ELP_PSK_LANES>), dim3(nbf8 + nbt), dim3(ELP_WIDE_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr,
                       (Jac<F2<B>>*)pre, kws, kstride, kvalid, n, nbf8);
    return;
  }
  const unsigned nbf = grid_for(n * 2 * ELP_PSK_LANES);
  hipLaunchKernelGGL((k_vid_prep<B, ELP_PSK_LANES>), dim3(nbf + grid_for(n)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr, (Jac<F2<B>>*)pre,
                     kws, kstride, kvalid, n, nbf);
}
template <class B>
void launch_vid_small(hipStream_t stream, const KeyCtx<B>& key, const void* d_consts, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad,
                      const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, uint8_t* done,
                      const void* pre) {
  const bool quad = (key.flags & KEY_QUAD_G2) != 0 && key.vtab != nullptr && key.vpsi != nullptr;      // the caller sized the table workspace for it
  const unsigned nb_nizk = quad ? (unsigned)((n + 15) / 16) : grid_for(n);
  if (n <= 512)
    hipLaunchKernelGGL((k_vid_small<B, 32>), dim3(nb_nizk + (unsigned)((n + 3) / 4)), dim3(256), 0, stream, key, (const Fp2<B>*)d_consts, (const u32*)d_records, words, (u64)mask,
                       retr, (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, nizk_ok, kvalid, kws, kstride, pair_ok, done, n, (const Jac<F2<B>>*)pre, nb_nizk);
  else
    hipLaunchKernelGGL((k_vid_small<B, 16>), dim3(nb_nizk + (unsigned)((n + 7) / 8)), dim3(256), 0, stream, key, (const Fp2<B>*)d_consts, (const u32*)d_records, words, (u64)mask,
                       retr, (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, nizk_ok, kvalid, kws, kstride, pair_ok, done, n, (const Jac<F2<B>>*)pre, nb_nizk);
#ifdef ELP_DBG_SMALL
  if (getenv("ELP_DBG_SMALL")) {
    unsigned long long h[32];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(elp_dbg_small), sizeof h);
    unsigned long long base = h[0];
    fprintf(stderr, "k_vid_small n=%zu:", n);
    for (int w = 0; w < 8; w++) {
      const unsigned hw = (unsigned)h[w * 4 + 2];
      fprintf(stderr, " %c%d[x%llu c%02u m%u] %+.2f..%.2f ms |", w < 4 ? 'N' : 'P', w & 3, h[w * 4 + 3] & 15, (hw >> 8) & 15, (hw >> 4) & 3, ((double)h[w * 4] - (double)base) / 1e5,
              ((double)h[w * 4 + 1] - (double)base) / 1e5);
    }
    fprintf(stderr, "\n");
  }
#endif
  KeyCtx<B> k2 = key;
  k2.vtab = nullptr;
  hipLaunchKernelGGL((k_pair_rest<B>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, stream, k2, (const u32*)d_records, words, kvalid, (const uint8_t*)done, kws, kstride, pair_ok,
                     (unsigned long long*)nullptr, n);
}
// k_vid_small2 only (the caller queues k_pair_rest through launch_pair_rest): instantiated in elpasso_<curve>_small2.hip, the translation unit built for two waves per SIMD
template <class B>
void launch_vid_small2(hipStream_t stream, const KeyCtx<B>& key, const void* d_consts, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad,
                       const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, uint8_t* done,
                       const void* pre) {
  const unsigned nb_nizk = grid_for(n);
  hipLaunchKernelGGL((k_vid_small2<B>), dim3(nb_nizk + (unsigned)((n + 7) / 8)), dim3(256), 0, stream, key, (const Fp2<B>*)d_consts, (const u32*)d_records, words, (u64)mask,
                     retr, (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, nizk_ok, kvalid, kws, kstride, pair_ok, done, n, (const Jac<F2<B>>*)pre, nb_nizk);
}
template <class B>
void launch_vid_ktab(hipStream_t stream, const KeyCtx<B>& key, size_t n, const void* d_records, int words, int retr) {
  hipLaunchKernelGGL((k_vid_ktab<B>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, retr, n);
}
template <class B>
void launch_vid_combine(hipStream_t stream, size_t n, const uint8_t* nizk_ok, const uint8_t* pair_ok, void* d_flags, void* d_accepted) {
  hipLaunchKernelGGL((k_vid_combine<B>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, stream, nizk_ok, pair_ok, (uint8_t*)d_flags, (unsigned long long*)d_accepted, n);
}
template <class B>
void launch_agg_final_coop(hipStream_t stream, const KeyCtx<B>& key, const void* d_consts, const void* F, const void* s2_std, int* agg_ok) {
  hipLaunchKernelGGL((k_agg_final_coop<B>), dim3(1), dim3(ELP_COOP_BLOCK), 0, stream, key, (const Fp2<B>*)d_consts, (const Fp12<B>*)F, (const u32*)s2_std, agg_ok);
}
#else
template <class B>
void launch_coop_consts(hipStream_t stream, void* d_consts);
template <class B>
void launch_pair_coop(hipStream_t stream, const KeyCtx<B>& key, const void* d_consts, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride,
                      uint8_t* d_flags, uint8_t* done, void* d_accepted, hipStream_t rest_stream);
#define ELP_REST_BY_CALLER ((hipStream_t)(intptr_t)-1)
template <class B>
void launch_pair_rest(hipStream_t stream, const KeyCtx<B>& key, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags,
                      uint8_t* done, void* d_accepted);
template <class B>
void launch_ps_k(hipStream_t stream, const KeyCtx<B>& key, size_t n, const void* d_records, int words, int nattr, uint8_t* todo, u32* kws, size_t kstride);
template <class B>
void launch_agg_final_coop(hipStream_t stream, const KeyCtx<B>& key, const void* d_consts, const void* F, const void* s2_std, int* agg_ok);
template <class B>
void launch_vid_fixed_coop(hipStream_t stream, const KeyCtx<B>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, void* pre, u32* kws, size_t kstride,
                           uint8_t* kvalid);
template <class B>
void launch_vid_prep(hipStream_t stream, const KeyCtx<B>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, void* pre, u32* kws, size_t kstride,
                     uint8_t* kvalid);
template <class B>
void launch_vid_combine(hipStream_t stream, size_t n, const uint8_t* nizk_ok, const uint8_t* pair_ok, void* d_flags, void* d_accepted);
template <class B>
void launch_vid_ktab(hipStream_t stream, const KeyCtx<B>& key, size_t n, const void* d_records, int words, int retr);
template <class B>
void launch_vid_small(hipStream_t stream, const KeyCtx<B>& key, const void* d_consts, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad,
                      const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, uint8_t* done,
                      const void* pre);
template <class B>
void launch_vid_small2(hipStream_t stream, const KeyCtx<B>& key, const void* d_consts, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad,
                       const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, uint8_t* done,
                       const void* pre);
#endif
// which curves have the cooperative kernels (their own translation unit, elpasso_<curve>_coop.hip)
template <class B>
struct CoopBuild {
  static constexpr bool value = false;
};
template <>
struct CoopBuild<BN254> {
  static constexpr bool value = true;
};
// small batches of el_passo_verify_id (k_vid_prep -> k_vid_small / k_vid_nizk4 + k_pair_coop -> k_vid_combine): curves that have k_vid_nizk4 built for them too
template <class B>
struct SmallBuild {
  static constexpr bool value = false;
};
template <>
struct SmallBuild<BN254> {
  static constexpr bool value = true;
};
template <>
struct SmallBuild<BLS12_381> {
  static constexpr bool value = true;
};
template <>
struct CoopBuild<BLS12_381> {      // round 4: the pairing check (PS verification of small batches, the tail of aggregated verification); pinned like everything on this curve
  static constexpr bool value = true;
};
// k_vid_small2 (one-launch small batches at two waves per SIMD): where two pairing workgroups fit a compute unit's LDS
template <class B>
struct Small2Build {
  static constexpr bool value = false;
};
template <>
struct Small2Build<BN254> {
  static constexpr bool value = true;
};
#ifndef ELP_COOP_TU
extern template void launch_vid_small2<BN254>(hipStream_t stream, const KeyCtx<BN254>& key, const void* d_consts, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, uint8_t* done, const void* pre);
extern template void launch_coop_consts<BN254>(hipStream_t stream, void* d_consts);
extern template void launch_vid_prep<BN254>(hipStream_t stream, const KeyCtx<BN254>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, void* pre, u32* kws, size_t kstride, uint8_t* kvalid);
extern template void launch_pair_rest<BN254>(hipStream_t stream, const KeyCtx<BN254>& key, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags, uint8_t* done, void* d_accepted);
extern template void launch_pair_coop<BN254>(hipStream_t stream, const KeyCtx<BN254>& key, const void* d_consts, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags, uint8_t* done, void* d_accepted, hipStream_t rest_stream);
extern template void launch_ps_k<BN254>(hipStream_t stream, const KeyCtx<BN254>& key, size_t n, const void* d_records, int words, int nattr, uint8_t* todo, u32* kws, size_t kstride);
extern template void launch_agg_final_coop<BN254>(hipStream_t stream, const KeyCtx<BN254>& key, const void* d_consts, const void* F, const void* s2_std, int* agg_ok);
extern template void launch_vid_fixed_coop<BN254>(hipStream_t stream, const KeyCtx<BN254>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, void* pre, u32* kws, size_t kstride, uint8_t* kvalid);
extern template void launch_vid_combine<BN254>(hipStream_t stream, size_t n, const uint8_t* nizk_ok, const uint8_t* pair_ok, void* d_flags, void* d_accepted);
extern template void launch_vid_ktab<BN254>(hipStream_t stream, const KeyCtx<BN254>& key, size_t n, const void* d_records, int words, int retr);
extern template void launch_vid_small<BN254>(hipStream_t stream, const KeyCtx<BN254>& key, const void* d_consts, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, uint8_t* done, const void* pre);
extern template void launch_coop_consts<BLS12_381>(hipStream_t stream, void* d_consts);
extern template void launch_pair_rest<BLS12_381>(hipStream_t stream, const KeyCtx<BLS12_381>& key, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags, uint8_t* done, void* d_accepted);
extern template void launch_pair_coop<BLS12_381>(hipStream_t stream, const KeyCtx<BLS12_381>& key, const void* d_consts, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags, uint8_t* done, void* d_accepted, hipStream_t rest_stream);
extern template void launch_ps_k<BLS12_381>(hipStream_t stream, const KeyCtx<BLS12_381>& key, size_t n, const void* d_records, int words, int nattr, uint8_t* todo, u32* kws, size_t kstride);
extern template void launch_agg_final_coop<BLS12_381>(hipStream_t stream, const KeyCtx<BLS12_381>& key, const void* d_consts, const void* F, const void* s2_std, int* agg_ok);
extern template void launch_vid_prep<BLS12_381>(hipStream_t stream, const KeyCtx<BLS12_381>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, void* pre, u32* kws, size_t kstride, uint8_t* kvalid);
extern template void launch_vid_fixed_coop<BLS12_381>(hipStream_t stream, const KeyCtx<BLS12_381>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, void* pre, u32* kws, size_t kstride, uint8_t* kvalid);
extern template void launch_vid_combine<BLS12_381>(hipStream_t stream, size_t n, const uint8_t* nizk_ok, const uint8_t* pair_ok, void* d_flags, void* d_accepted);
extern template void launch_vid_ktab<BLS12_381>(hipStream_t stream, const KeyCtx<BLS12_381>& key, size_t n, const void* d_records, int words, int retr);
extern template void launch_vid_small<BLS12_381>(hipStream_t stream, const KeyCtx<BLS12_381>& key, const void* d_consts, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, uint8_t* done, const void* pre);
#endif

// ---- paired layout (elp/common.h "Lane pairs"): two lanes per item, 64-thread workgroups = 32 items, 256 registers per lane and two
// waves per SIMD; the LDS hot slot is half as large per lane (8 workgroups x 13.5 KB per CU).
#ifndef ELP_PAIR_WAVES
#define ELP_PAIR_WAVES 2
#endif
#define ELP_PAIR_LAUNCH_BOUNDS __launch_bounds__(ELP_BLOCK, ELP_PAIR_WAVES)
#define ELP_HOT_SETUP_PAIRED(key)                                                                    \
  __shared__ __attribute__((aligned(16))) u32 elp_hot_lds[ELP_BLOCK * elp::ELP_HOT_WORDS_PAIRED];  \
  (key).hot = elp_hot_lds + threadIdx.x * elp::ELP_HOT_WORDS_PAIRED
__device__ __forceinline__ void count_accept_paired(bool ok, unsigned long long* counter) {
  unsigned long long b = __ballot(ok && (threadIdx.x & 1) == 0);
  if ((threadIdx.x & 63) == 0 && b != 0 && counter) atomicAdd(counter, (unsigned long long)__popcll(b));
}
template <class C>
__global__ void ELP_PAIR_LAUNCH_BOUNDS k_verify_id_paired(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, const uint8_t* ad,
                                                          const u32* ad_off, u32 ad_len, uint8_t* flags, unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP_PAIRED(key);
  if (key.vtab) key.vtab += ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)vtab_words<C>();
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  bool ok = false;
  if (i < n) {
    const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
    size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
    ok = verify_id_item_paired<C>(key, recs + i * (size_t)rec_words, mask, retr != 0, a, al, (key.flags & KEY_PHASE_MIX) && blockIdx.x >= gridDim.x / 2);
    if ((threadIdx.x & 1) == 0) flags[i] = ok ? 1 : 0;
  }
  count_accept_paired(ok, accepted);
}
// the paired kernel behind k_vid_g1jobs: everything over Fp2 of a verification; commitments and verdicts of the G1 jobs from the workspace
template <class C>
__global__ void ELP_PAIR_LAUNCH_BOUNDS k_verify_id_paired_g1(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, const uint8_t* ad,
                                                             const u32* ad_off, u32 ad_len, const u32* g1ws, size_t g1stride, uint8_t* flags,
                                                             unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP_PAIRED(key);
  if (key.vtab) key.vtab += ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)vtab_words<C>();
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  bool ok = false;
  if (i < n) {
    const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
    size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
    G1JobsOut pre{g1ws, g1stride, i, C::FBYTES / 4};
    ok = verify_id_item_paired_g1done<C>(key, recs + i * (size_t)rec_words, mask, retr != 0, a, al, pre);
    if ((threadIdx.x & 1) == 0) flags[i] = ok ? 1 : 0;
  }
  count_accept_paired(ok, accepted);
}
template <class C>
__global__ void ELP_PAIR_LAUNCH_BOUNDS k_verify_id_wire_paired(KeyCtx<C> key, const uint8_t* msgs, const u32* msg_off, int retr, const uint8_t* ad,
                                                               const u32* ad_off, u32 ad_len, uint8_t* flags, unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP_PAIRED(key);
  if (key.vtab) key.vtab += ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)vtab_words<C>();
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  bool ok = false;
  if (i < n) {
    const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
    size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
    ok = verify_id_wire_item_paired<C>(key, msgs + msg_off[i], (size_t)(msg_off[i + 1] - msg_off[i]), retr != 0, a, al);
    if ((threadIdx.x & 1) == 0) flags[i] = ok ? 1 : 0;
  }
  count_accept_paired(ok, accepted);
}
template <class C>
__global__ void ELP_PAIR_LAUNCH_BOUNDS k_ps_verify_paired(KeyCtx<C> key, const u32* recs, int rec_words, int nattr, uint8_t* flags,
                                                          unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP_PAIRED(key);
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  bool ok = false;
  if (i < n) {
    ok = ps_verify_item<C>(key, recs + i * (size_t)rec_words, nattr);
    if ((threadIdx.x & 1) == 0) flags[i] = ok ? 1 : 0;
  }
  count_accept_paired(ok, accepted);
}

// Closing step of aggregated verification on ONE lane pair (the plain-layout k_agg_final runs it on one lane): F * f(-S2, gg), final exponentiation,
// comparison with 1.  The pair halves the latency of this serial tail (7.4 -> ~4.5 ms on BN254), which every aggregated batch pays once.
template <class C>
__global__ void ELP_PAIR_LAUNCH_BOUNDS k_agg_final_paired(KeyCtx<C> key, const Fp12<typename PairInfo<C>::Base>* F, const u32* s2_std, int* agg_ok) {
  if (blockIdx.x != 0 || threadIdx.x >= 2) return;
  Aff<F1<C>> s2, ns2;
  bool ok = g1_load<C>(s2, s2_std);
  aff_neg(ns2, s2);
  if (aff_is_inf(s2)) aff_set_inf(ns2);
  Fp12<C> f, Fm;
  fp12_from_mem<C>(Fm, F[0]);
  const LineMem<C>* lines[1] = {key.gg_lines};
  miller_loop<C, 0, 1>(f, &ns2, (const Aff<F2<C>>*)0, &ns2, lines);
  fp12_mul<C>(f, f, Fm);
  const bool one = final_exp_is_one<C>(f);
  if (threadIdx.x == 0) *agg_ok = (ok && one) ? 1 : 0;
}

template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_provide_id(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, const uint8_t* ad,
                                                          const u32* ad_off, u32 ad_len, u32* sigs, uint8_t* flags,
                                                          unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP(key);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n) {
    const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
    size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
    ok = provide_id_item<C>(key, recs + i * (size_t)rec_words, mask, a, al, sigs + i * (size_t)(4 * C::N));
    flags[i] = ok ? 1 : 0;
  }
  count_accept(ok, accepted);
}

template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_prove_id(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, const uint8_t* ad,
                                                        const u32* ad_off, u32 ad_len, u32* out, int out_words, uint8_t* flags,
                                                        unsigned long long* accepted, size_t n) {
  ELP_HOT_SETUP(key);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n) {
    const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
    size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
    ok = prove_id_item<C>(key, recs + i * (size_t)rec_words, mask, retr != 0, a, al, out + i * (size_t)out_words);
    flags[i] = ok ? 1 : 0;
  }
  count_accept(ok, accepted);
}

template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_request_id(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, const uint8_t* ad,
                                                          const u32* ad_off, u32 ad_len, u32* out, int out_words, size_t n) {
  ELP_HOT_SETUP(key);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
  size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
  request_id_item<C>(key, recs + i * (size_t)rec_words, mask, a, al, out + i * (size_t)out_words);
}

template <class C, int G>  // G = 1: G1, 2: G2
__global__ void ELP_LAUNCH_BOUNDS k_decompress(const uint8_t* wire, u32* out, uint8_t* okf, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (G == 1) {
    Aff<F1<C>> p;
    bool ok = g1_deserialize<C>(p, wire + i * C::FBYTES);
    if (!ok) aff_set_inf(p);
    g1_store<C>(out + i * 2 * C::N, p);
    okf[i] = ok;
  } else {
    Aff<F2<C>> p;
    bool ok = g2_deserialize<C>(p, wire + i * 2 * C::FBYTES);
    if (!ok) aff_set_inf(p);
    g2_store<C>(out + i * 4 * C::N, p);
    okf[i] = ok;
  }
}

template <class C, int G>
__global__ void ELP_LAUNCH_BOUNDS k_mul(const u32* pts, const u32* ks, u32* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (G == 1) {
    Aff<F1<C>> p, r;
    Jac<F1<C>> j;
    if (!g1_load<C>(p, pts + i * 2 * C::N)) aff_set_inf(p);
    jac_mul_var<F1<C>>(j, p, scalar_load_w(ks + i * 8));
    jac_to_aff<F1<C>>(r, j);
    g1_store<C>(out + i * 2 * C::N, r);
  } else {
    Aff<F2<C>> p, r;
    Jac<F2<C>> j;
    if (!g2_load<C>(p, pts + i * 4 * C::N)) aff_set_inf(p);
    jac_mul_var<F2<C>>(j, p, scalar_load_w(ks + i * 8));
    jac_to_aff<F2<C>>(r, j);
    g2_store<C>(out + i * 4 * C::N, r);
  }
}

template <class C, int G>
__global__ void ELP_LAUNCH_BOUNDS k_add(const u32* a, const u32* b, u32* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (G == 1) {
    Aff<F1<C>> p, q, r;
    Jac<F1<C>> j;
    if (!g1_load<C>(p, a + i * 2 * C::N)) aff_set_inf(p);
    if (!g1_load<C>(q, b + i * 2 * C::N)) aff_set_inf(q);
    jac_from_aff(j, p);
    jac_madd<F1<C>>(j, j, q);
    jac_to_aff<F1<C>>(r, j);
    g1_store<C>(out + i * 2 * C::N, r);
  } else {
    Aff<F2<C>> p, q, r;
    Jac<F2<C>> j;
    if (!g2_load<C>(p, a + i * 4 * C::N)) aff_set_inf(p);
    if (!g2_load<C>(q, b + i * 4 * C::N)) aff_set_inf(q);
    jac_from_aff(j, p);
    jac_madd<F2<C>>(j, j, q);
    jac_to_aff<F2<C>>(r, j);
    g2_store<C>(out + i * 4 * C::N, r);
  }
}

template <class C, int G>
__global__ void ELP_LAUNCH_BOUNDS k_msm_fixed(KeyCtx<C> key, int nterms, const int* base_ids, const u32* ks, u32* out,
                                                         size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (G == 1) {
    Jac<F1<C>> acc;
    jac_set_inf(acc);
    for (int t = 0; t < nterms; t++) acc_fixed_g1<C>(acc, key, base_ids[t], scalar_load_w(ks + (i * nterms + t) * 8));
    Aff<F1<C>> r;
    jac_to_aff<F1<C>>(r, acc);
    g1_store<C>(out + i * 2 * C::N, r);
  } else {
    Jac<F2<C>> acc;
    jac_set_inf(acc);
    for (int t = 0; t < nterms; t++) acc_fixed_g2<C>(acc, key, base_ids[t], scalar_load_w(ks + (i * nterms + t) * 8));
    Aff<F2<C>> r;
    jac_to_aff<F2<C>>(r, acc);
    g2_store<C>(out + i * 4 * C::N, r);
  }
}

template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_hash_to_g1(const uint8_t* msgs, const u32* off, u32* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F1<C>> p;
  hash_and_map_to_g1<C>(p, msgs + off[i], off[i + 1] - off[i]);
  g1_store<C>(out + i * 2 * C::N, p);
}

template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_pairing(const u32* g1, const u32* g2, u32* gt, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F1<C>> p;
  Aff<F2<C>> q;
  if (!g1_load<C>(p, g1 + i * 2 * C::N)) aff_set_inf(p);
  if (!g2_load<C>(q, g2 + i * 4 * C::N)) aff_set_inf(q);
  Fp12<C> f, g;
  const LineCoef<C>* no_lines[1] = {reinterpret_cast<const LineCoef<C>*>(g2)};   // never read, see miller_loop
  miller_loop<C, 1, 0>(f, &p, &q, &p, no_lines);
  final_exp<C>(g, f);
  gt_store<C>(gt + i * 12 * C::N, g);
}

template <class C, int NP>
__global__ void ELP_LAUNCH_BOUNDS k_pairing_check(const u32* g1, const u32* g2, uint8_t* okf, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F1<C>> p[NP];
  Aff<F2<C>> q[NP];
  bool ok = true;
  for (int j = 0; j < NP; j++) {
    ok &= g1_load<C>(p[j], g1 + (i * NP + j) * 2 * C::N);
    ok &= g2_load<C>(q[j], g2 + (i * NP + j) * 4 * C::N);
  }
  if (ok) {
    Fp12<C> f;
    const LineCoef<C>* no_lines[1] = {reinterpret_cast<const LineCoef<C>*>(g2)};   // never read, see miller_loop
    miller_loop<C, NP, 0>(f, p, q, p, no_lines);
    ok = final_exp_is_one<C>(f);
  }
  okf[i] = ok;
}

// ------------------------------------------------------------------------------------------------------------
// Single-output multi-scalar multiplication  sum_i k_i * P_i  (Pippenger bucket method, 8-bit windows).
// The reference has no equivalent call (mcl's mulVec is unused); it is the general MSM operator of this library and the
// building block of aggregated verification.
//   k_msm_prepare : std affine points -> Montgomery affine (validated), one lane per point
//   k_msm_buckets : one 256-thread workgroup per (window, slice of <= 8192 points): LDS histogram of the window's digits
//                   (byte w of each scalar), shfl-based exclusive scan, LDS counting sort of the slice's indices by digit, then
//                   lane b sums the points of bucket b (no elliptic-curve atomics, no divergence inside the additions)
//   k_msm_combine : (more than eight slices per window) the slices' bucket sums combined by 32 lanes per bucket
//   k_msm_reduce  : per window: slices combined, then sum_b b*B_b as a suffix scan + tree reduction through LDS
//   k_msm_final   : Horner over the non-empty windows (of 32), normalisation, one std affine point out
#define ELP_MSM_SLICE 8192
// Slices per window: at least ceil(n / ELP_MSM_SLICE) (the LDS sort buffer), more -- down to 512 points per slice -- while the launch would otherwise leave compute
// units idle: nwin x S workgroups should reach four per compute unit (65 536 points, 16 windows: 8 -> 64 slices; profiles/r06_ubench_msm_buckets.log)
static inline int msm_slices(size_t n, int nwin) {
  static const size_t target = [] {
    const char* e = getenv("ELP_MSM_WORKGROUPS");           // (measurements)
    const long v = e ? atol(e) : 0;
    return (size_t)(v > 0 ? v : 1024);
  }();
  size_t s = (n + ELP_MSM_SLICE - 1) / ELP_MSM_SLICE;
  const size_t want = (target + (size_t)nwin - 1) / (size_t)nwin, most = (n + 511) / 512;
  if (s < want) s = want < most ? want : most;
  return (int)(s ? s : 1);
}
#define ELP_MSM_TPB 256
// G1 sums over full-width scalars (round 6): every scalar is split as k = +-k1 +- k2 lam (mod r) with |k1|, |k2| < 2^136 (curve.h lattice_split, the split of g1_mul_glv),
// the windows 0 .. 16 hold the bytes of |k1|, the windows 17 .. 33 those of |k2|, a sign byte tells the bucket kernel to add -P; the closing Horner chains then have 128
// doublings each on two lanes instead of 248 on one (k_msm_final 1.30 -> k_msm_final_glv<17>; profiles/r06_aggregated.md).  Bytes per point of the split scalars:
#define ELP_MSM_GLV_HW 17
#define ELP_MSM_VPARTS 4           /* the window of a half's top byte is worked on in this many parts (a power of two) */
#define ELP_MSM_NWMAX (2 * ELP_MSM_GLV_HW + 2 * (ELP_MSM_VPARTS - 1))      /* windows of the widest launch shape: 2 x ELP_MSM_GLV_HW and the virtual windows below */
// The halves have 126 - 128 bits, so the digits of their top byte (windows HW - 2 and 2 HW - 2) take few of the 256 values -- 49 on BN254 --: those buckets are five times
// as long as the others' and their workgroups finished last (0.53 ms for 65 536 points where uniformly distributed digits take 0.27).  Each of the two windows is therefore
// worked on in ELP_MSM_VPARTS parts: the real window takes the points whose index is 0 modulo VPARTS, VPARTS - 1 virtual windows behind the real ones (2 HW ...) the other
// residues; k_msm_final_glv adds the parts.
#define ELP_MSM_GLV_STRIDE 36      /* bytes per point in the workspace: 34 digit bytes -- stored WINDOW-MAJOR, digit w of point i at [w * n + i], so that a workgroup's pass over its slice
                                      reads consecutive bytes (point-major, every digit byte cost a cache line: 0.53 -> see profiles/r06_aggregated.md) --, one sign byte at [34 * n + i], pad */
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_msm_split_scalars(const u32* ks, uint8_t* out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 m[2][5];
  bool neg[2];
  lattice_split<2, 5, 5>(scalar_mod_r<C>(scalar_load_w(ks + i * 8)), m, neg, Glv1Lat<C>());
  for (int h = 0; h < 2; h++)
    for (int b = 0; b < ELP_MSM_GLV_HW; b++) out[(size_t)(h * ELP_MSM_GLV_HW + b) * n + i] = (uint8_t)(m[h][b >> 2] >> (8 * (b & 3)));
  out[(size_t)(2 * ELP_MSM_GLV_HW) * n + i] = (uint8_t)((neg[0] ? 1 : 0) | (neg[1] ? 2 : 0));
}

template <class C, int G>
__global__ void ELP_LAUNCH_BOUNDS k_msm_prepare(const u32* pts, void* out, int* bad, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (G == 1) {
    Aff<F1<C>> p;
    if (!g1_load<C>(p, pts + i * 2 * C::N)) {
      aff_set_inf(p);
      atomicAdd(bad, 1);
    }
    ((Aff<F1<C>>*)out)[i] = p;
  } else {
    Aff<F2<C>> p;
    if (!g2_load<C>(p, pts + i * 4 * C::N)) {
      aff_set_inf(p);
      atomicAdd(bad, 1);
    }
    ((Aff<F2<C>>*)out)[i] = p;
  }
}

// G1: four waves per SIMD (at most 128 registers, the mixed addition inlined): a launch of 1 024 workgroups is then resident at once.  Left to itself the
// compiler took 164 registers -- three waves per SIMD, a quarter of the workgroups in a second round -- and the phase ran 2.5 x longer (0.68 against 0.27 ms for
// 65 536 points x 32 windows; tools/ubench_msm.hip had the tighter allocation by accident of its other kernels).
template <class F>
__global__ void __launch_bounds__(ELP_MSM_TPB, (F::IS_EXT ? 1 : 4)) k_msm_buckets(const Aff<F>* pts, const uint8_t* scalars, size_t n, int S,
                                                             Jac<F>* partial, int kstride, int sign_off, int half_w) {      // plain scalars: 32, -1, 0; split ones (window-major): 0, 34, 17 -- see k_msm_split_scalars
  __shared__ unsigned cnt[256];
  __shared__ unsigned start[256];
  __shared__ unsigned cnt0[256];
  __shared__ unsigned char perm[256];
  __shared__ unsigned wave_tot[4];
  __shared__ unsigned short idx[ELP_MSM_SLICE];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int w = blockIdx.x / S, s = blockIdx.x % S;
  const size_t lo = n * (size_t)s / S, hi = n * (size_t)(s + 1) / S;
  const int M = (int)(hi - lo);
  // split scalars (sign_off >= 0): the windows of the halves' top bytes take the even points only, the virtual windows behind the real ones their odd points
  int wr = w, par = -1;                                     // the real window whose digits this workgroup reads; the parity of the points it takes (-1: all)
  if (sign_off >= 0) {
    if (w >= 2 * half_w) {
      const int v = w - 2 * half_w;                         // half v / (VPARTS - 1), part 1 + v % (VPARTS - 1)
      wr = (v / (ELP_MSM_VPARTS - 1) + 1) * half_w - 2;
      par = 1 + v % (ELP_MSM_VPARTS - 1);
    } else if (w == half_w - 2 || w == 2 * half_w - 2) {
      par = 0;
    }
  }
  // digit w of point i: point-major scalars (kstride bytes per point) or window-major digits (kstride == 0: digit w of all points, then digit w + 1 ...)
  const uint8_t* const digits = kstride ? scalars + wr : scalars + (size_t)wr * n;
  const size_t dstep = kstride ? (size_t)kstride : 1;
  cnt[tid] = 0;
  __syncthreads();
  for (int j = tid; j < M; j += ELP_MSM_TPB) {
    unsigned d = digits[(size_t)(lo + j) * dstep];
    if (par >= 0 && (int)((lo + j) & (ELP_MSM_VPARTS - 1)) != par) d = 0;
    if (d != 0 && !aff_is_inf(pts[lo + j])) atomicAdd(&cnt[d], 1u);
  }
  __syncthreads();
  // exclusive prefix sum over the 256 bucket counts: wave-level inclusive scan with shuffles, wave totals through LDS
  unsigned c0 = cnt[tid], x = c0;
  for (int d = 1; d < 64; d <<= 1) {
    unsigned y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  if (lane == 63) wave_tot[wv] = x;
  cnt0[tid] = c0;
  __syncthreads();
  unsigned base = 0;
  for (int k = 0; k < wv; k++) base += wave_tot[k];
  const unsigned my_start = base + x - c0;
  start[tid] = my_start;
  // The buckets are dealt to the lanes IN THE ORDER OF THEIR SIZES (lane t takes the bucket of rank t, largest first): the 64 buckets of a wave are then of similar
  // length -- a wave runs as long as its longest bucket, and with the buckets in digit order every wave waited for a Poisson tail (25 points where the mean is 8) --
  // and the waves of the small buckets retire early.  tools/ubench_msm.hip: 2^20 points 4.16 -> 3.29 ms, 65 536 points x 16 windows 0.25 -> 0.17 ms.
  {
    unsigned rank = 0;
    for (int k = 0; k < 256; k++) {
      const unsigned ck = cnt0[k];
      rank += (ck > c0 || (ck == c0 && k < tid)) ? 1u : 0u;
    }
    perm[rank] = (unsigned char)tid;
  }
  __syncthreads();
  cnt[tid] = my_start;   // becomes the scatter cursor
  __syncthreads();
  const int sign_shift = (sign_off >= 0 && wr >= half_w) ? 1 : 0;
  for (int j = tid; j < M; j += ELP_MSM_TPB) {
    unsigned d = digits[(size_t)(lo + j) * dstep];
    if (par >= 0 && (int)((lo + j) & (ELP_MSM_VPARTS - 1)) != par) d = 0;
    if (d != 0 && !aff_is_inf(pts[lo + j])) {
      const unsigned sg = sign_off >= 0 ? (((unsigned)scalars[(size_t)sign_off * n + lo + j] >> sign_shift) & 1u) : 0u;
      idx[atomicAdd(&cnt[d], 1u)] = (unsigned short)((unsigned)j | (sg << 15));      // a slice has at most 8 192 points: bit 15 carries the sign of this half's sub-scalar
    }
  }
  __syncthreads();
  const int b = perm[tid];
  Jac<F> acc;
  jac_set_inf(acc);
  const unsigned t0 = start[b], t1 = t0 + cnt0[b];
  for (unsigned t = t0; t < t1; t++) {
    const unsigned e = idx[t];
    Aff<F> p = pts[lo + (e & 0x7FFFu)];
    if (e & 0x8000u) aff_neg(p, p);
    if constexpr (F::IS_EXT) jac_madd<F>(acc, acc, p);
    else jac_madd_inl_t<F, true>(acc, acc, p);      // inlined with its exceptional doubling: a call would bring the callee's own register allocation (134) with it
  }
  partial[(size_t)blockIdx.x * 256 + b] = acc;
}

// More than eight slices per window: their bucket sums are combined by 32 lanes per bucket (a tree of five additions through LDS) instead of one lane walking all of
// them in k_msm_reduce (32 slices: 31 additions in sequence there, 0.41 ms against 0.21 with this kernel in front; profiles/r06_aggregated.md).  A workgroup takes eight
// neighbouring buckets of one window (thread = 8 * lane-of-the-bucket + bucket); the sum lands IN PLACE in slice 0 -- a bucket's slots are touched by its 32 lanes only.
template <class F>
__global__ void ELP_MSM_LAUNCH_BOUNDS k_msm_combine(Jac<F>* partial, int S) {
  __shared__ Jac<F> sh[256];
  const int tid = threadIdx.x, j = tid >> 3, w = blockIdx.x >> 5, b = ((blockIdx.x & 31) << 3) | (tid & 7);
  Jac<F> acc;
  jac_set_inf(acc);
  if (j < S) acc = partial[((size_t)w * S + j) * 256 + b];
  for (int s = j + 32; s < S; s += 32) jac_add<F>(acc, acc, partial[((size_t)w * S + s) * 256 + b]);
  for (int d = 16; d >= 1; d >>= 1) {
    sh[tid] = acc;
    __syncthreads();
    if (j < d) jac_add<F>(acc, acc, sh[tid + 8 * d]);
    __syncthreads();
  }
  if (j == 0) partial[((size_t)w * S) * 256 + b] = acc;
}

// partial: slice s of window w at [(w * stride + s) * 256, + 256)
template <class F>
__global__ void ELP_MSM_LAUNCH_BOUNDS k_msm_reduce(const Jac<F>* partial, int S, int stride, Jac<F>* win) {
  __shared__ Jac<F> sh[256];
  const int tid = threadIdx.x, w = blockIdx.x;
  Jac<F> acc = partial[((size_t)w * stride) * 256 + tid];
  for (int s = 1; s < S; s++) jac_add<F>(acc, acc, partial[((size_t)w * stride + s) * 256 + tid]);
  // suffix scan: acc_b = sum_{j >= b} B_j
  for (int d = 1; d < 256; d <<= 1) {
    sh[tid] = acc;
    __syncthreads();
    if (tid + d < 256) jac_add<F>(acc, acc, sh[tid + d]);
    __syncthreads();
  }
  if (tid == 0) jac_set_inf(acc);          // sum_b b*B_b = sum_{b>=1} S_b
  for (int d = 128; d >= 1; d >>= 1) {
    sh[tid] = acc;
    __syncthreads();
    if (tid < d) jac_add<F>(acc, acc, sh[tid + d]);
    __syncthreads();
  }
  if (tid == 0) win[w] = acc;
}

// bucket sums per (window, slice) -> [slices combined] -> one sum per window
template <class F>
static void msm_windows(hipStream_t stream, int nwin, int S, const Aff<F>* aff, const uint8_t* ks, size_t n, Jac<F>* part, Jac<F>* win, bool split = false) {
  hipLaunchKernelGGL((k_msm_buckets<F>), dim3(nwin * S), dim3(ELP_MSM_TPB), 0, stream, aff, ks, n, S, part, split ? 0 : 32, split ? 2 * ELP_MSM_GLV_HW : -1,
                     split ? ELP_MSM_GLV_HW : 0);
  if (S > 8) hipLaunchKernelGGL((k_msm_combine<F>), dim3(nwin * 32), dim3(ELP_MSM_TPB), 0, stream, part, S);
  hipLaunchKernelGGL((k_msm_reduce<F>), dim3(nwin), dim3(ELP_MSM_TPB), 0, stream, (const Jac<F>*)part, S > 8 ? 1 : S, S, win);
}

// Closing step for the multipliers of aggregated verification, d_i = a_i + b_i lam with a_i in scalar bytes 0-7 and b_i in bytes 8-15 (pipeline.h
// verify_id_agg_item): windows 0-7 hold the bucket sums of sum a_i P_i =: A, windows 8-15 those of sum b_i P_i =: B; the result is A + phi(B).  The two Horner
// chains (56 doublings each) run on two lanes.
template <class C, int HW = 8, bool VIRT = false>      // HW windows per half: 8 for the 64-bit pairs of aggregated verification, ELP_MSM_GLV_HW for split full-width scalars (VIRT: with the virtual windows)
__global__ void ELP_LAUNCH_BOUNDS k_msm_final_glv(const void* win_, u32* out) {
  typedef F1<C> F;
  __shared__ Jac<F> half[2];
  if (blockIdx.x != 0 || threadIdx.x >= 2) return;
  const Jac<F>* win = (const Jac<F>*)win_ + HW * threadIdx.x;
  Jac<F> r = win[HW - 1];
  for (int w = HW - 2; w >= 0; w--) {
    for (int k = 0; k < 8; k++) jac_dbl<F>(r, r);
    jac_add<F>(r, r, win[w]);
    if (VIRT && w == HW - 2) {                                // the other parts of the top byte's window
      for (int v = 0; v < ELP_MSM_VPARTS - 1; v++) jac_add<F>(r, r, ((const Jac<F>*)win_)[2 * HW + (ELP_MSM_VPARTS - 1) * threadIdx.x + v]);
    }
  }
  if (threadIdx.x == 1) {
    Fp<C> beta;
    ELP_LOAD_FP(beta, C::glv_beta(i_));
    r.X = fp_mul<C>(r.X, beta);
  }
  half[threadIdx.x] = r;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (threadIdx.x == 0) {
    jac_add<F>(r, half[0], half[1]);
    Aff<F> a;
    jac_to_aff<F>(a, r);
    g1_store<C>(out, a);
  }
}
template <class C, int G>
__global__ void ELP_LAUNCH_BOUNDS k_msm_final(const void* win_, u32* out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  if (G == 1) {
    const Jac<F1<C>>* win = (const Jac<F1<C>>*)win_;
    int top = 31;                           // short scalars (the 128-bit multipliers of aggregated verification) leave the upper windows empty
    while (top > 0 && jac_is_inf(win[top])) top--;
    Jac<F1<C>> r = win[top];
    for (int w = top - 1; w >= 0; w--) {
      for (int k = 0; k < 8; k++) jac_dbl<F1<C>>(r, r);
      jac_add<F1<C>>(r, r, win[w]);
    }
    Aff<F1<C>> a;
    jac_to_aff<F1<C>>(a, r);
    g1_store<C>(out, a);
  } else {
    const Jac<F2<C>>* win = (const Jac<F2<C>>*)win_;
    int top = 31;
    while (top > 0 && jac_is_inf(win[top])) top--;
    Jac<F2<C>> r = win[top];
    for (int w = top - 1; w >= 0; w--) {
      for (int k = 0; k < 8; k++) jac_dbl<F2<C>>(r, r);
      jac_add<F2<C>>(r, r, win[w]);
    }
    Aff<F2<C>> a;
    jac_to_aff<F2<C>>(a, r);
    g2_store<C>(out, a);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Aggregated VerifyID: k_verify_id_agg does the NIZK half and one Miller loop per item and multiplies
// the 64 Miller values of its wave through LDS; k_fp12_reduce shrinks the per-wave products; the Pippenger kernels above give
// sum d_i sig2_i; k_agg_final closes the batch equation with one more Miller loop and ONE final exponentiation; k_agg_finish
// publishes the verdicts, re-verifying every item individually only when the batch equation failed.
struct AggSeed {
  uint8_t b[32];
};

template <class C>
__device__ __forceinline__ void wave_fp12_product(Fp12<C>& f, Fp12<C>* sh) {   // sh: 64 entries in LDS; result valid in lane 0
  const int lane = threadIdx.x & 63;
  for (int d = 32; d >= 1; d >>= 1) {
    sh[lane] = f;
    __syncthreads();
    if (lane < d) fp12_mul<C>(f, f, sh[lane + d]);
    __syncthreads();
  }
}

template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_verify_id_agg(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr,
                                                             const uint8_t* ad, const u32* ad_off, u32 ad_len, AggSeed seed,
                                                             uint8_t* nizk_flags, u32* deltas, u32* sig2s, Fp12<C>* wave_prod, size_t n) {
  __shared__ __attribute__((aligned(16))) Fp12<C> sh[ELP_BLOCK];
  static_assert(sizeof(Fp12<C>) >= (size_t)elp::ELP_HOT_WORDS * 4, "hot slot must fit an Fp12 entry");
  key.hot = reinterpret_cast<u32*>(&sh[threadIdx.x]);   // the lane's entry of the product buffer doubles as its hot slot until the reduction
  if (key.vtab) key.vtab += ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)vtab_words<C>();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fp12<C> f;
  fp12_set_one(f);
  if (i < n) {
    const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
    size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
    bool ok = verify_id_agg_item<C>(key, recs + i * (size_t)rec_words, mask, retr != 0, a, al, seed.b, (u64)i, f, deltas + i * 8,
                                    sig2s + i * (size_t)(2 * C::N));
    nizk_flags[i] = ok ? 1 : 0;
  }
  wave_fp12_product<C>(f, sh);
  if (threadIdx.x == 0) wave_prod[blockIdx.x] = f;
}

// The same with TWO items per lane (items 2 t and 2 t + 1 on lane t): for batches of more than one full round of lanes the two Miller loops of a lane share the
// squarings of one accumulator and multiply their lines pairwise (elp/pairing.h miller_loop_two) -- about 6 % fewer instructions per item.  Half as many waves: n / 128.
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_verify_id_agg2(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr,
                                                              const uint8_t* ad, const u32* ad_off, u32 ad_len, AggSeed seed,
                                                              uint8_t* nizk_flags, u32* deltas, u32* sig2s, Fp12<C>* wave_prod, size_t n) {
  __shared__ __attribute__((aligned(16))) Fp12<C> sh[ELP_BLOCK];
  key.hot = reinterpret_cast<u32*>(&sh[threadIdx.x]);
  if (key.vtab) key.vtab += ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)vtab_words<C>();
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i0 = 2 * t, i1 = 2 * t + 1;
  Fp12<C> f;
  fp12_set_one(f);
  if (i0 < n) {
    const bool two = i1 < n;
    const uint8_t* a0 = ad_off ? ad + ad_off[i0] : ad;
    const size_t al0 = ad_off ? (size_t)(ad_off[i0 + 1] - ad_off[i0]) : (size_t)ad_len;
    const uint8_t* a1 = (ad_off && two) ? ad + ad_off[i1] : ad;
    const size_t al1 = (ad_off && two) ? (size_t)(ad_off[i1 + 1] - ad_off[i1]) : (size_t)ad_len;
    bool ok[2];
    u32 dummy_d[8], dummy_s[2 * C::N];
    verify_id_agg_item2<C>(key, recs + i0 * (size_t)rec_words, two ? recs + i1 * (size_t)rec_words : nullptr, mask, retr != 0, a0, al0, a1, al1, seed.b, (u64)i0, f,
                           deltas + i0 * 8, sig2s + i0 * (size_t)(2 * C::N), two ? deltas + i1 * 8 : dummy_d, two ? sig2s + i1 * (size_t)(2 * C::N) : dummy_s, ok);
    nizk_flags[i0] = ok[0] ? 1 : 0;
    if (two) nizk_flags[i1] = ok[1] ? 1 : 0;
  }
  wave_fp12_product<C>(f, sh);
  if (threadIdx.x == 0) wave_prod[blockIdx.x] = f;
}

// The same on LANE PAIRS (round 6; BLS12-381, whose 14-limb field makes one lane per item the slowest layout): 32 items per wave, pipeline.h
// verify_id_agg_item_paired; the wave's product is taken over the pairs (strides 32 ... 2 keep the halves of a value on their lanes) and written in the plain layout
// the rest of the call works in.
template <class C>
__device__ __forceinline__ void wave_fp12_product_paired(Fp12<C>& f, Fp12<C>* sh) {   // sh: 64 entries in LDS; result valid in lanes 0 and 1
  const int lane = threadIdx.x & 63;
  for (int d = 32; d >= 2; d >>= 1) {
    sh[lane] = f;
    __syncthreads();
    if (lane < d) fp12_mul<C>(f, f, sh[lane + d]);
    __syncthreads();
  }
}
template <class C>
__global__ void ELP_PAIR_LAUNCH_BOUNDS k_verify_id_agg_paired(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr, const uint8_t* ad, const u32* ad_off,
                                                              u32 ad_len, AggSeed seed, uint8_t* nizk_flags, u32* deltas, u32* sig2s,
                                                              Fp12<typename PairInfo<C>::Base>* wave_prod, size_t n) {
  ELP_HOT_SETUP_PAIRED(key);
  __shared__ __attribute__((aligned(16))) Fp12<C> sh[ELP_BLOCK];
  if (key.vtab) key.vtab += ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (size_t)vtab_words<C>();
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  Fp12<C> f;
  fp12_set_one(f);
  if (i < n) {
    const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
    size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
    const bool ok = verify_id_agg_item_paired<C>(key, recs + i * (size_t)rec_words, mask, retr != 0, a, al, seed.b, (u64)i, f, deltas + i * 8,
                                                 sig2s + i * (size_t)(2 * C::N));
    if ((threadIdx.x & 1) == 0) nizk_flags[i] = ok ? 1 : 0;
  }
  wave_fp12_product_paired<C>(f, sh);
  if (threadIdx.x < 2) fp12_to_mem<C>(wave_prod[blockIdx.x], f);
}

template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_fp12_reduce(const Fp12<C>* in, size_t n, Fp12<C>* out) {
  __shared__ Fp12<C> sh[ELP_BLOCK];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fp12<C> f;
  if (i < n)
    f = in[i];
  else
    fp12_set_one(f);
  wave_fp12_product<C>(f, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = f;
}

// F * f(-S2, gg) -> final exponentiation -> *agg_ok = (result == 1).  One lane.
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_agg_final(KeyCtx<C> key, const Fp12<C>* F, const u32* s2_std, int* agg_ok) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Aff<F1<C>> s2, ns2;
  bool ok = g1_load<C>(s2, s2_std);
  aff_neg(ns2, s2);
  if (aff_is_inf(s2)) aff_set_inf(ns2);
  Fp12<C> f;
  const LineCoef<C>* lines[1] = {key.gg_lines};
  miller_loop<C, 0, 1>(f, (const Aff<F1<C>>*)0, (const Aff<F2<C>>*)0, &ns2, lines);
  fp12_mul<C>(f, f, F[0]);
  *agg_ok = (ok && final_exp_is_one<C>(f)) ? 1 : 0;
}

// verdicts: the NIZK flags when the batch equation held, otherwise the exact per-item verification (rare, slow path)
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_agg_finish(KeyCtx<C> key, const u32* recs, int rec_words, u64 mask, int retr,
                                                          const uint8_t* ad, const u32* ad_off, u32 ad_len, const uint8_t* nizk_flags,
                                                          const int* agg_ok, uint8_t* flags, unsigned long long* accepted, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n) {
    ok = nizk_flags[i] != 0;
    if (ok && *agg_ok == 0) {
      const uint8_t* a = ad_off ? ad + ad_off[i] : ad;
      size_t al = ad_off ? (size_t)(ad_off[i + 1] - ad_off[i]) : (size_t)ad_len;
      ok = verify_id_item<C>(key, recs + i * (size_t)rec_words, mask, retr != 0, a, al);
    }
    flags[i] = ok ? 1 : 0;
  }
  count_accept(ok, accepted);
}

// ---- setup kernels
template <class F>
__global__ void ELP_LAUNCH_BOUNDS k_window_bases(const Aff<F>* bases, int nb, int W, int nwin, Aff<F>* bj) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  if (aff_is_inf(bases[b])) {
    for (int j = 0; j < nwin; j++) aff_set_inf(bj[(size_t)b * nwin + j]);
    return;
  }
  table_window_bases<F>(bj + (size_t)b * nwin, bases[b], W, nwin);
}
template <class F>
__global__ void ELP_LAUNCH_BOUNDS k_table_fill(Aff<F>* tbl, const Aff<F>* bj, int nb, int nwin, int per, int chunk) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int nchunk = (per + chunk - 1) / chunk;
  size_t total = (size_t)nb * nwin * nchunk;
  if (t >= total) return;
  int c = (int)(t % nchunk);
  size_t bw = t / nchunk;  // b * nwin + j
  Aff<F>* win = tbl + bw * per;
  int d0 = 1 + c * chunk;
  int cnt = (d0 + chunk - 1 <= per) ? chunk : per - d0 + 1;
  Aff<F> base = bj[bw];
  if (aff_is_inf(base)) {
    for (int d = d0; d < d0 + cnt; d++) aff_set_inf(win[d - 1]);
    return;
  }
  table_fill_chunk<F>(win, base, d0, cnt);
}
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_lines(const Aff<F2<C>>* gg, LineCoef<C>* out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) ml_precompute<C>(out, gg[0]);
}
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_load_bases(const u32* g1w, int n1, const u32* g2w, int n2, Aff<F1<C>>* b1, Aff<F2<C>>* b2, int* bad) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n1) {
    if (!g1_load<C>(b1[t], g1w + (size_t)t * 2 * C::N)) {
      aff_set_inf(b1[t]);
      atomicAdd(bad, 1);
    }
  } else if (t < n1 + n2) {
    int u = t - n1;
    if (!g2_load<C>(b2[u], g2w + (size_t)u * 4 * C::N)) {
      aff_set_inf(b2[u]);
      atomicAdd(bad, 1);
    }
  }
}
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_bench_fp_mul(u32* out, int iters, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp<C> a, b;
  for (int k = 0; k < C::NL; k++) {
    a.v[k] = C::one(k);
    b.v[k] = C::r2(k);
  }
  a.v[0] += (i32)(i & 0xffff);
  b.v[1] -= (i32)(i >> 16);
  for (int it = 0; it < iters; it++) {
    a = fp_mul<C>(a, b);
    b = fp_mul<C>(b, a);
  }
  u32 acc = 0;
  for (int k = 0; k < C::NL; k++) acc ^= (u32)a.v[k] ^ (u32)b.v[k];
  out[i] = acc;
}

// Per-routine micro-benchmark: every lane runs `iters` dependent applications of one device routine on lane-private data.
// op: 0 fp_mul 1 fp_sqr 2 fp2_mul 3 fp2_sqr 4 fp6_mul 5 fp12_mul 6 fp12_sqr 7 fp12_cyc_sqr 8 mul_by_line
//     9 jac_dbl<G1> 10 jac_madd<G1> 11 jac_add<G1> 12 jac_dbl<G2> 13 jac_madd<G2> 14 jac_add<G2> 15 ml_dbl_step 16 ml_add_step
//     17 fp_inv 18 fp_add 19 fp2_add 20 jac_mul_var<G1> 21 jac_mul_var<G2> 22 miller_loop (1 pair) 23 final_exp
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_bench_op(int op, u32* out, int iters, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp<C> a = fp_one<C>(), b;
  ELP_LOAD_FP(b, C::r2(i_));
  a.v[0] += (i32)(i & 0xffff);
  b.v[1] -= (i32)(i >> 16);
  Fp2<C> x, y;
  x.c0 = a; x.c1 = b; y.c0 = b; y.c1 = fp_add(a, b);
  Fp12<C> f, g;
  f.c0.c0 = x; f.c0.c1 = y; f.c0.c2 = x; f.c1.c0 = y; f.c1.c1 = x; f.c1.c2 = y;
  g = f;
  Aff<F1<C>> p1; p1.x = a; p1.y = b;
  Jac<F1<C>> j1; j1.X = b; j1.Y = a; j1.Z = fp_add(a, b);
  Aff<F2<C>> p2; p2.x = x; p2.y = y;
  Jac<F2<C>> j2; j2.X = y; j2.Y = x; j2.Z = fp2_add(x, y);
  G2Proj<C> T; T.X = x; T.Y = y; T.Z = fp2_add(x, y);
  LineCoef<C> l;
  for (int it = 0; it < iters; it++) {
    switch (op) {
      case 0: a = fp_mul<C>(a, b); break;
      case 1: a = fp_sqr<C>(a); break;
      case 2: fp2_mul<C>(x, x, y); break;
      case 3: fp2_sqr<C>(x, x); break;
      case 4: fp6_mul<C>(f.c0, f.c0, g.c1); break;
      case 5: fp12_mul<C>(f, f, g); break;
      case 6: fp12_sqr<C>(f, f); break;
      case 7: fp12_cyc_sqr<C>(f, f); break;
      case 8: fp12_mul_by_line<C>(f, x, y, g.c0.c0); break;
      case 9: jac_dbl<F1<C>>(j1, j1); break;
      case 10: jac_madd<F1<C>>(j1, j1, p1); break;
      case 11: { Jac<F1<C>> t = j1; t.X = b; jac_add<F1<C>>(j1, j1, t); } break;
      case 12: jac_dbl<F2<C>>(j2, j2); break;
      case 13: jac_madd<F2<C>>(j2, j2, p2); break;
      case 14: { Jac<F2<C>> t = j2; t.X = y; jac_add<F2<C>>(j2, j2, t); } break;
      case 15: ml_dbl_step<C>(T, l); T.Z = fp2_add(T.Z, l.c); break;
      case 16: ml_add_step<C>(T, l, x, y); T.Z = fp2_add(T.Z, l.c); break;
      case 17: a = fp_inv<C>(a); break;
      case 20: { Scalar k; for (int q = 0; q < 8; q++) k.v[q] = (u32)a.v[q] * 2654435761u; k.v[7] &= 0x1fffffff; jac_mul_var<F1<C>>(j1, p1, k); p1.x = j1.X; } break;
      case 21: { Scalar k; for (int q = 0; q < 8; q++) k.v[q] = (u32)a.v[q] * 2654435761u; k.v[7] &= 0x1fffffff; jac_mul_var<F2<C>>(j2, p2, k); p2.x = j2.X; } break;
      case 22: { const LineCoef<C>* nl[1] = {reinterpret_cast<const LineCoef<C>*>(out)}; miller_loop<C, 1, 0>(f, &p1, &p2, &p1, nl); p1.x = f.c0.c0.c0; } break;
      case 23: final_exp<C>(f, g); g.c0.c0 = f.c1.c1; break;
      case 18: a = fp_add(a, b); b = fp_sub(b, a); break;
      default: x = fp2_add(x, y); y = fp2_sub(y, x); break;
    }
  }
  u32 acc = 0;
  for (int k = 0; k < C::NL; k++)
    acc ^= (u32)a.v[k] ^ (u32)x.c0.v[k] ^ (u32)f.c0.c0.c0.v[k] ^ (u32)f.c1.c2.c1.v[k] ^ (u32)j1.X.v[k] ^ (u32)j2.Z.c1.v[k] ^ (u32)T.Z.c0.v[k];
  out[i] = acc;
}

// ------------------------------------------------------------------------------------------------------------
// host side of the C-ABI
// ------------------------------------------------------------------------------------------------------------
struct elp_ctx {
  int curve;
  int device;
  hipStream_t stream;
  std::string err;
  // key state (device memory)
  int A = 0, W = 0, nwin = 0, per = 0;
  void* b1 = nullptr;   // Aff<F1>[A+6]
  void* b2 = nullptr;   // Aff<F2>[A+2]
  void* t1 = nullptr;
  void* t2 = nullptr;
  void* lines = nullptr;
  std::vector<uint8_t> h_b1;  // host mirror of the G1 base words (std form) so set_rp / set_signer_secret can rebuild
  bool have_pk = false;
  bool rp_set = false;        // elp_set_rp installed H1(service)
  bool retr_set = false;      // ... and all of g, authority_pk, h (needed by the id-retrieval variants)
  bool sk_set = false;        // elp_set_signer_secret installed X
  int strict_sig = 1;         // ELP_OPT_STRICT_SIGNATURE
  int subgroup_check = 1;     // ELP_OPT_SUBGROUP_CHECK (curves with a G1 cofactor: prover-supplied G1 points must lie in the order-r subgroup)
  int paired = 2;             // ELP_OPT_PAIRED_LAYOUT: 0 = one lane per item, 1 = two lanes per item, 2 = by batch size (layout_split)
  int simds = 1024;           // SIMDs of the device (4 per CU): one resident wave per SIMD is the unit of the layout choice
  // copy stream of the host-buffer pipeline (elp_verify_id_batch): the next round's records travel while the current round is verified
  static constexpr int NPIPE = 1;
  hipStream_t pstream[NPIPE] = {nullptr};
  // workspace of the aggregated verification (grown on demand, reused across calls)
  void* agg_ws = nullptr;
  size_t agg_ws_bytes = 0;
  int* agg_ok = nullptr;      // device flag of the last aggregated batch
  // The three fields above belong to ONE launch stream at a time (agg_stream).  A caller that pipelines aggregated batches over several streams of one context
  // (batch i on stream A, batch i + 1 on stream B: the serial tail of one batch runs beside the per-item kernel of the next) gets a workspace per stream: the
  // entry point parks the current set in agg_parked and takes out (or creates) the set of its stream.
  struct AggWs {
    hipStream_t stream;
    void* ws;
    size_t bytes;
    int* ok;
  };
  hipStream_t agg_stream = nullptr;
  std::vector<AggWs> agg_parked;
  // slots of elp_verify_id_batch_submit / _wait: device buffers that live as long as the context (grown on demand), an event for "this slot's batch is done", the
  // page-locked landing pad of its accepted count
  struct AsyncSlot {
    void *drec = nullptr, *dad = nullptr, *doff = nullptr, *dfl = nullptr, *dcnt = nullptr;
    size_t rec_cap = 0, ad_cap = 0, off_cap = 0, fl_cap = 0;
    hipEvent_t copied = nullptr, done = nullptr;
    uint64_t* h_cnt = nullptr;
    bool busy = false;
    size_t staged_bytes = 0;      // record bytes delivered by elp_verify_id_batch_stage for the next submit of the slot
  };
  AsyncSlot aslot[2];
  int fail_submits = 0;          // ELP_OPT_FAULT_INJECT
  int mid_two_launches = 0;      // experiments (ELP_PAIR4_TWO_LAUNCHES=1): the mid-size path as k_vid_nizk4 then k_pair4 instead of the one launch k_vid_mid
  unsigned long long* wire_mask_host = nullptr;      // page-locked slot for the decoded wire path's 16-byte read-back
  int wire_decode = 1;           // ELP_OPT_WIRE_DECODE: wire batches of up to 16 384 messages are decoded into records and take the small / mid-size record paths
  int agg_paired = 1;         // BLS12-381 with the two-lane layout allowed: the main kernel of aggregated verification on lane pairs (k_verify_id_agg_paired; ELP_AGG_PAIRED=0 for A/B runs)
  int agg_two = 0;            // ELP_OPT_AGG_TWO_PER_LANE: aggregated batches put two items on a lane (0 = never -- the default: 2 % at best, and the kernel's larger frame makes the runtime re-provision scratch --, 1 = where it saves rounds of lanes, 2 = always)
  int pair16 = 0;                // ELP_OPT_PAIR16 (set by elp_init): PS verifications of at most pair16_max items run the pairing check on one 16-lane row per item
  int pair16_tail = 1;           // the closing step of aggregated verification on one row (k_agg_final16; both curves: 1.02 against 1.19 ms on BN254, 2.21 against 2.63 ms on BLS12-381)
  size_t pair16_max = 4096, pair16_min = 4;      // below pair16_min (less than one wave of rows) the interpreter's 32 lane pairs per item are a little faster: 1.60 vs 1.72 ms for a lone item
  int pair4 = 1;                 // ELP_OPT_PAIR4: 0 = off, 1 = by batch size (default), 2 = wherever the path exists
  // per-lane tables of the variable-base multiplications (KeyCtx::vtab): one workspace per stream that launched a verification, grown on demand
  struct VtabWs {
    hipStream_t stream;
    void* p;
    size_t bytes;
  };
  std::vector<VtabWs> vtab_ws;
  std::vector<VtabWs> wire_ws;       // records + verdicts of the decoded wire path, one block per launch stream like vtab_ws (grown on demand)
  int use_vtab = 1;           // ELP_VTAB=0 in the environment keeps the tables in private memory (A/B measurements)
  int split = 0;              // ELP_OPT_SPLIT_PHASES: 1 = el_passo_verify_id as two kernels (k_vid_nizk, k_vid_pair) where the curve has them; 2 = the two jobs
                              // of the first phase as concurrent kernels on two streams (k_vid_g2 || k_vid_g1, then k_vid_pair2)
  int overlap = 0;            // ELP_OPT_STREAM_OVERLAP: independent kernels of one call on the context's second stream (small-batch verify_id, aggregated tail)
  int stage_records = 1;      // ELP_OPT_COALESCED_RECORDS: k_verify_id_staged (records through LDS into a private copy) instead of k_verify_id; measured equal in time
  int coop = 1;               // ELP_OPT_COOP_PAIRING: small batches (<= coop_max items) and the aggregated tail run the pairing check on 32 lanes per item (elp/coop.h)
  int phase_mix = 0;             // KEY_PHASE_MIX (experiment, ELP_PHASE_MIX): half of the two-lane verify launch's workgroups check the pairing before the NIZK half
  size_t small_dense_from = 1792;   // one-launch batches above this many items run k_vid_small2 (built for two waves per SIMD): up to here the pairing and NIZK workgroups
                                    // of k_vid_small fit the chip in one round (n / 8 + n / 64 <= 256); measured 2 048 items 3.80 vs 4.38 ms, 3 072: 3.84 vs 4.28, 4 096: 4.69 vs 5.56 (two launches)
  size_t small_one_max = ~(size_t)0;   // el_passo_verify_id batches up to this many items run k_vid_small / k_vid_small2 (one launch); above, k_vid_nizk4 then k_pair_coop
                                       // (4 096 items: BN254 4.69 vs 5.56 ms, BLS12-381 11.9 vs 14.2 ms): the whole cooperative range by default (ELP_SMALL_ONE_MAX: A/B)
  size_t vid_coop_max = 0;       // upper limit of the cooperative path for el_passo_verify_id; 0 = the measured cross-over against the two-lane kernels' 9.1-9.4 ms round:
                                 // 9 216 items (BN254: 8 192 items 7.3 ms, 9 216: 8.5, 10 240: 9.3), 8 192 (BLS12-381)
  size_t coop_max = 4096;     // 16 items per CU x 256 CUs: one round of the cooperative kernel; measured cross-over against the per-lane kernels between 4096 and 8192 items
  void* coop_consts = nullptr;     // constants table of the cooperative programs (built on first use)
  hipStream_t jstream = nullptr;   // second stream of split = 2 (the G1 job)
  hipEvent_t jev[4] = {nullptr, nullptr, nullptr, nullptr};
};

#define HIPCHK(ctx, expr)                                                                       \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                           \
      (void)hipDeviceSynchronize(); /* queued work may still use DevBufs that go back to the block cache on this exit */ \
      return ELP_ERR_HIP;                                                                       \
    }                                                                                           \
  } while (0)


template <class C>
static KeyCtx<C> make_key(const elp_ctx* c) {   // C may be Paired<B>: the key material is the same plain-layout memory
  KeyCtx<C> k;
  k.A = c->A;
  k.W = c->W;
  k.nwin = c->nwin;
  k.per = c->per;
  k.t1 = (const Aff<typename F1<C>::MemF>*)c->t1;
  k.t2 = (const Aff<typename F2<C>::MemF>*)c->t2;
  k.b1 = (const Aff<typename F1<C>::MemF>*)c->b1;
  k.b2 = (const Aff<typename F2<C>::MemF>*)c->b2;
  k.gg_lines = (const LineMem<C>*)c->lines;
  k.flags = (c->strict_sig ? KEY_STRICT_SIG : 0) | (c->subgroup_check ? 0 : KEY_NO_SUBGROUP_CHECK) | (c->phase_mix ? KEY_PHASE_MIX : 0);
  return k;
}

// make_key + this launch's table workspace: `lanes` slices of vtab_words<C>() words.  No workspace (allocation failure, ELP_VTAB=0) is not an error:
// the kernels then keep the tables in private memory.
template <class C>
static KeyCtx<C> make_key_ws(elp_ctx* c, hipStream_t stream, size_t lanes, size_t extra_bytes = 0, void** extra = nullptr) {
  KeyCtx<C> k = make_key<C>(c);
  if (extra) *extra = nullptr;
  if (!c->use_vtab && !extra_bytes) return k;
  const size_t tab_bytes = c->use_vtab ? ((lanes * (size_t)vtab_words<C>() * 4 + 255) & ~(size_t)255) : 0;
  const size_t need = tab_bytes + extra_bytes;   // `extra`: per-launch state behind the tables (the K's and verdicts of the two-phase kernels)
  elp_ctx::VtabWs* w = nullptr;
  for (auto& e : c->vtab_ws)
    if (e.stream == stream) w = &e;
  if (!w) {
    c->vtab_ws.push_back({stream, nullptr, 0});
    w = &c->vtab_ws.back();
  }
  if (w->bytes < need) {
    if (w->p) (void)hipFree(w->p);        // hipFree waits for the work that may still read it
    w->p = nullptr;
    w->bytes = 0;
    if (hipMalloc(&w->p, need) != hipSuccess) {
      (void)hipGetLastError();
      w->p = nullptr;
      return k;
    }
    w->bytes = need;
  }
  if (c->use_vtab) k.vtab = (u32*)w->p;
  if (extra) *extra = (uint8_t*)w->p + tab_bytes;
  return k;
}

template <class C>
static const void* coop_consts_for(elp_ctx* c, hipStream_t stream) {
  if (!c->coop_consts) {
    if (hipMalloc(&c->coop_consts, 64 * sizeof(Fp2<C>)) != hipSuccess) {
      (void)hipGetLastError();
      c->coop_consts = nullptr;
      return nullptr;
    }
    launch_coop_consts<C>(stream, c->coop_consts);      // later launches on other streams: the table is complete long before (same device, in-order first use)
    (void)hipStreamSynchronize(stream);
  }
  return c->coop_consts;
}

// RAII device buffer for the host-buffer entry points.  Blocks come from (and go back to) a small per-process cache, so a steady stream of
// host-buffer calls does not pay hipMalloc / hipFree (each a device synchronisation) per call; elp_destroy() of the last context empties it.
struct DevBlockCache {
  struct Blk {
    void* p;
    size_t cap;
    int dev;
  };
  std::mutex mu;
  std::vector<Blk> free_blocks;
  size_t cached = 0;
  int live_ctx = 0;
  static constexpr size_t MAX_CACHED = (size_t)2 << 30;
  hipError_t get(size_t n, void** out, size_t* cap) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
      std::lock_guard<std::mutex> g(mu);
      int best = -1;
      for (int i = 0; i < (int)free_blocks.size(); i++)
        if (free_blocks[i].dev == dev && free_blocks[i].cap >= n && free_blocks[i].cap <= 2 * n + 4096 &&
            (best < 0 || free_blocks[i].cap < free_blocks[best].cap))
          best = i;
      if (best >= 0) {
        *out = free_blocks[best].p;
        *cap = free_blocks[best].cap;
        cached -= free_blocks[best].cap;
        free_blocks.erase(free_blocks.begin() + best);
        return hipSuccess;
      }
    }
    *cap = n;
    return hipMalloc(out, n);
  }
  void put(void* p, size_t cap) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    if (live_ctx > 0 && cached + cap <= MAX_CACHED) {
      free_blocks.push_back({p, cap, dev});
      cached += cap;
    } else {
      (void)hipFree(p);
    }
  }
  void trim() {
    std::lock_guard<std::mutex> g(mu);
    for (auto& b : free_blocks) (void)hipFree(b.p);
    free_blocks.clear();
    cached = 0;
  }
};
inline DevBlockCache& dev_cache() {
  static DevBlockCache c;
  return c;
}
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  ~DevBuf() {
    if (p) dev_cache().put(p, cap);
  }
  hipError_t alloc(size_t n) { return dev_cache().get(n ? n : 4, &p, &cap); }
};

static inline void free_key(elp_ctx* c) {
  void** ps[] = {&c->b1, &c->b2, &c->t1, &c->t2, &c->lines};
  for (void** p : ps) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  c->have_pk = c->rp_set = c->retr_set = c->sk_set = false;
}

// (re)build every device-side key structure from host copies of the base points
template <class C>
static int rebuild_key(elp_ctx* c, const uint8_t* g2_bases_std) {
  const int A = c->A, n1 = A + 6, n2 = A + 2;
  const bool g2_changed = g2_bases_std != nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf w1, w2, bad;
  HIPCHK(c, w1.alloc((size_t)n1 * Sizes<C>::G1));
  HIPCHK(c, bad.alloc(4));
  HIPCHK(c, hipMemsetAsync(bad.p, 0, 4, c->stream));
  HIPCHK(c, hipMemcpyAsync(w1.p, c->h_b1.data(), (size_t)n1 * Sizes<C>::G1, hipMemcpyHostToDevice, c->stream));
  if (!c->b1) HIPCHK(c, hipMalloc(&c->b1, sizeof(Aff<F1<C>>) * n1));
  if (g2_changed) {
    HIPCHK(c, w2.alloc((size_t)n2 * Sizes<C>::G2));
    HIPCHK(c, hipMemcpyAsync(w2.p, g2_bases_std, (size_t)n2 * Sizes<C>::G2, hipMemcpyHostToDevice, c->stream));
    if (!c->b2) HIPCHK(c, hipMalloc(&c->b2, sizeof(Aff<F2<C>>) * n2));
  }
  hipLaunchKernelGGL((k_load_bases<C>), dim3(grid_for(n1 + n2)), dim3(ELP_BLOCK), 0, c->stream, (const u32*)w1.p, n1,
                     (const u32*)w2.p, g2_changed ? n2 : 0, (Aff<F1<C>>*)c->b1, (Aff<F2<C>>*)c->b2, (int*)bad.p);
  int hbad = 0;
  HIPCHK(c, hipMemcpyAsync(&hbad, bad.p, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (hbad) {
    c->err = "a key point is not a valid curve point";
    return ELP_ERR_POINT;
  }
  const int nwin = c->nwin, per = c->per, W = c->W;
  const int chunk = per < 32 ? per : 32;
  const int nchunk = (per + chunk - 1) / chunk;
  {
    DevBuf bj;
    HIPCHK(c, bj.alloc(sizeof(Aff<F1<C>>) * (size_t)n1 * nwin));
    if (!c->t1) HIPCHK(c, hipMalloc(&c->t1, sizeof(Aff<F1<C>>) * (size_t)n1 * nwin * per));
    hipLaunchKernelGGL((k_window_bases<F1<C>>), dim3(grid_for(n1)), dim3(ELP_BLOCK), 0, c->stream, (const Aff<F1<C>>*)c->b1, n1, W,
                       nwin, (Aff<F1<C>>*)bj.p);
    hipLaunchKernelGGL((k_table_fill<F1<C>>), dim3(grid_for((size_t)n1 * nwin * nchunk)), dim3(ELP_BLOCK), 0, c->stream,
                       (Aff<F1<C>>*)c->t1, (const Aff<F1<C>>*)bj.p, n1, nwin, per, chunk);
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (g2_changed) {
    DevBuf bj;
    HIPCHK(c, bj.alloc(sizeof(Aff<F2<C>>) * (size_t)n2 * nwin));
    if (!c->t2) HIPCHK(c, hipMalloc(&c->t2, sizeof(Aff<F2<C>>) * (size_t)n2 * nwin * per));
    if (!c->lines) HIPCHK(c, hipMalloc(&c->lines, sizeof(LineCoef<C>) * ml_num_lines<C>()));
    hipLaunchKernelGGL((k_window_bases<F2<C>>), dim3(grid_for(n2)), dim3(ELP_BLOCK), 0, c->stream, (const Aff<F2<C>>*)c->b2, n2, W,
                       nwin, (Aff<F2<C>>*)bj.p);
    hipLaunchKernelGGL((k_table_fill<F2<C>>), dim3(grid_for((size_t)n2 * nwin * nchunk)), dim3(ELP_BLOCK), 0, c->stream,
                       (Aff<F2<C>>*)c->t2, (const Aff<F2<C>>*)bj.p, n2, nwin, per, chunk);
    hipLaunchKernelGGL((k_lines<C>), dim3(1), dim3(64), 0, c->stream, (const Aff<F2<C>>*)c->b2, (LineCoef<C>*)c->lines);
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  HIPCHK(c, hipGetLastError());
  return ELP_OK;
}

template <class C>
int elp_set_pubkey_t(elp_ctx* c, int nattr, const uint8_t* g, const uint8_t* gg, const uint8_t* XX, const uint8_t* Yi,
                   const uint8_t* YYi, int window_bits) {
  if (!c || nattr < 1 || nattr > 62 || !g || !gg || !XX || !Yi || !YYi) return ELP_ERR_ARG;
  if (window_bits == 0) window_bits = 8;
  if (window_bits < 2 || window_bits > 22) return ELP_ERR_ARG;
  {   // the tables must fit the device (signed digits: W = 16: 1.25 GiB for an 8-attribute BN254 key; W = 20: 16 GiB; W = 22: 60 GiB)
    const size_t per = (size_t)fixed_base_entries(window_bits), nwin = (256 + window_bits - 1) / window_bits;
    const size_t need = per * nwin * ((size_t)(nattr + 6) * sizeof(Aff<F1<C>>) + (size_t)(nattr + 2) * sizeof(Aff<F2<C>>));
    HIPCHK(c, hipSetDevice(c->device));      // the guard must look at THIS context's device, and count the tables about to be released
    size_t held = 0;
    if (c->t1) held += (size_t)c->per * c->nwin * (size_t)(c->A + 6) * sizeof(Aff<F1<C>>);
    if (c->t2) held += (size_t)c->per * c->nwin * (size_t)(c->A + 2) * sizeof(Aff<F2<C>>);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need + ((size_t)2 << 30) > free_b + held) {
      c->err = "fixed-base tables of this window width do not fit the device memory";
      return ELP_ERR_ARG;
    }
  }
  HIPCHK(c, hipSetDevice(c->device));
  free_key(c);
  c->A = nattr;
  c->W = window_bits;
  c->nwin = (256 + window_bits - 1) / window_bits;
  c->per = fixed_base_entries(window_bits);
  const size_t G1 = Sizes<C>::G1, G2 = Sizes<C>::G2;
  c->h_b1.assign((size_t)(nattr + 6) * G1, 0);
  memcpy(c->h_b1.data(), g, G1);
  memcpy(c->h_b1.data() + G1, Yi, (size_t)nattr * G1);
  std::vector<uint8_t> h2((size_t)(nattr + 2) * G2);
  memcpy(h2.data(), gg, G2);
  memcpy(h2.data() + G2, XX, G2);
  memcpy(h2.data() + 2 * G2, YYi, (size_t)nattr * G2);
  int rc = rebuild_key<C>(c, h2.data());
  if (rc != ELP_OK) {
    free_key(c);
    return rc;
  }
  c->have_pk = true;
  return ELP_OK;
}

template <class C>
int elp_set_rp_t(elp_ctx* c, const uint8_t* service_name, size_t service_len, const uint8_t* authority_pk, const uint8_t* g,
               const uint8_t* h) {
  if (!c) return ELP_ERR_ARG;
  if (!c->have_pk) {
    c->err = "elp_set_pubkey must be called first";
    return ELP_ERR_STATE;
  }
  const size_t G1 = Sizes<C>::G1;
  const int A = c->A;
  uint8_t hs[Sizes<C>::G1];
  memset(hs, 0, sizeof hs);
  if (service_name) {
    uint32_t off[2] = {0, (uint32_t)service_len};
    int rc = elp_hash_to_g1(c, 1, service_name, off, hs);
    if (rc != ELP_OK) return rc;
  }
  memcpy(c->h_b1.data() + (size_t)(A + 1) * G1, hs, G1);
  const uint8_t* src[3] = {g, authority_pk, h};
  for (int i = 0; i < 3; i++) {
    if (src[i])
      memcpy(c->h_b1.data() + (size_t)(A + 2 + i) * G1, src[i], G1);
    else
      memset(c->h_b1.data() + (size_t)(A + 2 + i) * G1, 0, G1);
  }
  c->rp_set = c->retr_set = false;
  int rc = rebuild_key<C>(c, nullptr);
  if (rc != ELP_OK) {          // tables and bases may be inconsistent now: the context needs a fresh elp_set_pubkey
    free_key(c);
    return rc;
  }
  c->rp_set = service_name != nullptr;
  c->retr_set = c->rp_set && g && authority_pk && h;
  return ELP_OK;
}

template <class C>
int elp_set_signer_secret_t(elp_ctx* c, const uint8_t* X) {
  if (!c || !X) return ELP_ERR_ARG;
  if (!c->have_pk) {
    c->err = "elp_set_pubkey must be called first";
    return ELP_ERR_STATE;
  }
  memcpy(c->h_b1.data() + (size_t)(c->A + 5) * Sizes<C>::G1, X, Sizes<C>::G1);
  c->sk_set = false;
  int rc = rebuild_key<C>(c, nullptr);
  if (rc != ELP_OK) {
    free_key(c);
    return rc;
  }
  c->sk_set = true;
  return ELP_OK;
}

// ---- generic "copy in, launch, copy out" helper for the host-buffer primitives
struct IoSpec {
  const void* src;
  size_t bytes;
};
#define MAX_IO 4
template <class Launch>
static int run_host(elp_ctx* c, const IoSpec* ins, int nin, void* const* outs, const size_t* out_bytes, int nout, Launch launch) {
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf din[MAX_IO], dout[MAX_IO];
  void* pin[MAX_IO] = {0};
  void* pout[MAX_IO] = {0};
  for (int i = 0; i < nin; i++) {
    if (!ins[i].src) continue;
    HIPCHK(c, din[i].alloc(ins[i].bytes));
    HIPCHK(c, hipMemcpyAsync(din[i].p, ins[i].src, ins[i].bytes, hipMemcpyHostToDevice, c->stream));
    pin[i] = din[i].p;
  }
  for (int i = 0; i < nout; i++) {
    HIPCHK(c, dout[i].alloc(out_bytes[i]));
    HIPCHK(c, hipMemsetAsync(dout[i].p, 0, out_bytes[i], c->stream));
    pout[i] = dout[i].p;
  }
  launch(pin, pout);
  HIPCHK(c, hipGetLastError());
  for (int i = 0; i < nout; i++)
    if (outs[i]) HIPCHK(c, hipMemcpyAsync(outs[i], dout[i].p, out_bytes[i], hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ELP_OK;
}

template <class C, int G>
int decompress_impl_t(elp_ctx* c, size_t n, const uint8_t* wire, uint8_t* out, uint8_t* ok) {
  if (!c || (n && (!wire || !out || !ok))) return ELP_ERR_ARG;
  if (n == 0) return ELP_OK;
  IoSpec ins[1] = {{wire, n * (size_t)(G * C::FBYTES)}};
  void* outs[2] = {out, ok};
  size_t ob[2] = {n * (size_t)(G == 1 ? Sizes<C>::G1 : Sizes<C>::G2), n};
  return run_host(c, ins, 1, outs, ob, 2, [&](void** pi, void** po) {
    hipLaunchKernelGGL((k_decompress<C, G>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, c->stream, (const uint8_t*)pi[0], (u32*)po[0],
                       (uint8_t*)po[1], n);
  });
}
template <int G>
static int decompress_impl(elp_ctx* c, size_t n, const uint8_t* wire, uint8_t* out, uint8_t* ok) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? decompress_impl_t<BN254, G>(c, n, wire, out, ok) : decompress_impl_t<BLS12_381, G>(c, n, wire, out, ok);
}

template <class C, int G>
int mul_impl_t(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out) {
  if (!c || (n && (!pts || !ks || !out))) return ELP_ERR_ARG;
  if (n == 0) return ELP_OK;
  const size_t P = G == 1 ? Sizes<C>::G1 : Sizes<C>::G2;
  IoSpec ins[2] = {{pts, n * P}, {ks, n * 32}};
  void* outs[1] = {out};
  size_t ob[1] = {n * P};
  return run_host(c, ins, 2, outs, ob, 1, [&](void** pi, void** po) {
    hipLaunchKernelGGL((k_mul<C, G>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, c->stream, (const u32*)pi[0], (const u32*)pi[1],
                       (u32*)po[0], n);
  });
}
template <int G>
static int mul_impl(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? mul_impl_t<BN254, G>(c, n, pts, ks, out) : mul_impl_t<BLS12_381, G>(c, n, pts, ks, out);
}

template <class C, int G>
int add_impl_t(elp_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  if (!c || (n && (!a || !b || !out))) return ELP_ERR_ARG;
  if (n == 0) return ELP_OK;
  const size_t P = G == 1 ? Sizes<C>::G1 : Sizes<C>::G2;
  IoSpec ins[2] = {{a, n * P}, {b, n * P}};
  void* outs[1] = {out};
  size_t ob[1] = {n * P};
  return run_host(c, ins, 2, outs, ob, 1, [&](void** pi, void** po) {
    hipLaunchKernelGGL((k_add<C, G>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, c->stream, (const u32*)pi[0], (const u32*)pi[1],
                       (u32*)po[0], n);
  });
}
template <int G>
static int add_impl(elp_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? add_impl_t<BN254, G>(c, n, a, b, out) : add_impl_t<BLS12_381, G>(c, n, a, b, out);
}

template <class C, int G>
int msm_fixed_impl_t(elp_ctx* c, size_t n, int nterms, const int32_t* ids, const uint8_t* ks, uint8_t* out) {
  if (!c || nterms < 1 || (n && (!ids || !ks || !out))) return ELP_ERR_ARG;
  if (!c->have_pk) {
    c->err = "elp_set_pubkey must be called first";
    return ELP_ERR_STATE;
  }
  const int nb = G == 1 ? c->A + 6 : c->A + 2;
  for (int t = 0; t < nterms; t++)
    if (ids[t] < 0 || ids[t] >= nb) return ELP_ERR_ARG;
  if (n == 0) return ELP_OK;
  const size_t P = G == 1 ? Sizes<C>::G1 : Sizes<C>::G2;
  IoSpec ins[2] = {{ids, (size_t)nterms * 4}, {ks, n * (size_t)nterms * 32}};
  void* outs[1] = {out};
  size_t ob[1] = {n * P};
  KeyCtx<C> key = make_key<C>(c);
  return run_host(c, ins, 2, outs, ob, 1, [&](void** pi, void** po) {
    hipLaunchKernelGGL((k_msm_fixed<C, G>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, c->stream, key, nterms, (const int*)pi[0],
                       (const u32*)pi[1], (u32*)po[0], n);
  });
}
template <int G>
static int msm_fixed_impl(elp_ctx* c, size_t n, int nterms, const int32_t* ids, const uint8_t* ks, uint8_t* out) {
  if (!c) return ELP_ERR_ARG;
  return c->curve == ELP_CURVE_BN254 ? msm_fixed_impl_t<BN254, G>(c, n, nterms, ids, ks, out) : msm_fixed_impl_t<BLS12_381, G>(c, n, nterms, ids, ks, out);
}

template <class C, int G>
static size_t msm_ws_bytes(size_t n);
// The Pippenger launch sequence lives in a translation unit of its own (elpasso_<curve>_msm.hip, round 6): compiled beside the verification kernels, the group law it
// shares with them (jac_madd, jac_add: real functions) took their register budget -- 164 registers for k_msm_buckets instead of 120, three waves per SIMD instead of
// four -- and the bucket phase ran 2.7 x slower than the same source in tools/ubench_msm.hip.
template <class C, int G>
int* msm_launch(hipStream_t stream, size_t n, const void* d_pts_std, const void* d_ks, void* d_out_std, uint8_t* ws, bool glv_pairs = false);
template <class C, int G>
int msm_impl_t(elp_ctx* c, size_t n, const uint8_t* pts, const uint8_t* ks, uint8_t* out) {
  if (!c || !out || (n && (!pts || !ks))) return ELP_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t P = G == 1 ? Sizes<C>::G1 : Sizes<C>::G2;
  const size_t AFF = G == 1 ? sizeof(Aff<F1<C>>) : sizeof(Aff<F2<C>>);
  const size_t JAC = G == 1 ? sizeof(Jac<F1<C>>) : sizeof(Jac<F2<C>>);
  if (n == 0) {
    memset(out, 0, P);
    return ELP_OK;
  }
  DevBuf dpts, dks, dws, dout;
  HIPCHK(c, dpts.alloc(n * P));
  HIPCHK(c, dks.alloc(n * 32));
  HIPCHK(c, dws.alloc(msm_ws_bytes<C, G>(n)));
  HIPCHK(c, dout.alloc(P));
  HIPCHK(c, hipMemcpyAsync(dpts.p, pts, n * P, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dks.p, ks, n * 32, hipMemcpyHostToDevice, c->stream));
  const int* d_bad = msm_launch<C, G>(c->stream, n, dpts.p, dks.p, dout.p, (uint8_t*)dws.p, false);
  HIPCHK(c, hipGetLastError());
  int hbad = 0;
  HIPCHK(c, hipMemcpyAsync(&hbad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(out, dout.p, P, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (hbad) {
    c->err = "an input point is not a valid curve point";
    return ELP_ERR_POINT;
  }
  return ELP_OK;
}

template <class C>
int elp_hash_to_g1_t(elp_ctx* c, size_t n, const uint8_t* msgs, const uint32_t* off, uint8_t* out) {
  if (!c || (n && (!off || !out))) return ELP_ERR_ARG;
  if (n == 0) return ELP_OK;
  uint8_t dummy[4] = {0};
  IoSpec ins[2] = {{off[n] ? msgs : dummy, off[n] ? (size_t)off[n] : 4}, {off, (n + 1) * 4}};
  void* outs[1] = {out};
  size_t ob[1] = {n * Sizes<C>::G1};
  return run_host(c, ins, 2, outs, ob, 1, [&](void** pi, void** po) {
    hipLaunchKernelGGL((k_hash_to_g1<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, c->stream, (const uint8_t*)pi[0], (const u32*)pi[1],
                       (u32*)po[0], n);
  });
}

template <class C>
int elp_pairing_t(elp_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt) {
  if (!c || (n && (!g1 || !g2 || !gt))) return ELP_ERR_ARG;
  if (n == 0) return ELP_OK;
  IoSpec ins[2] = {{g1, n * Sizes<C>::G1}, {g2, n * Sizes<C>::G2}};
  void* outs[1] = {gt};
  size_t ob[1] = {n * Sizes<C>::GT};
  return run_host(c, ins, 2, outs, ob, 1, [&](void** pi, void** po) {
    hipLaunchKernelGGL((k_pairing<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, c->stream, (const u32*)pi[0], (const u32*)pi[1],
                       (u32*)po[0], n);
  });
}

template <class C>
int elp_pairing_check_t(elp_ctx* c, size_t n, int npairs, const uint8_t* g1, const uint8_t* g2, uint8_t* ok) {
  if (!c || npairs < 1 || npairs > 4 || (n && (!g1 || !g2 || !ok))) return ELP_ERR_ARG;
  if (n == 0) return ELP_OK;
  IoSpec ins[2] = {{g1, n * npairs * Sizes<C>::G1}, {g2, n * npairs * Sizes<C>::G2}};
  void* outs[1] = {ok};
  size_t ob[1] = {n};
  return run_host(c, ins, 2, outs, ob, 1, [&](void** pi, void** po) {
    const u32* a = (const u32*)pi[0];
    const u32* b = (const u32*)pi[1];
    uint8_t* o = (uint8_t*)po[0];
    switch (npairs) {
      case 1: hipLaunchKernelGGL((k_pairing_check<C, 1>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, c->stream, a, b, o, n); break;
      case 2: hipLaunchKernelGGL((k_pairing_check<C, 2>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, c->stream, a, b, o, n); break;
      case 3: hipLaunchKernelGGL((k_pairing_check<C, 3>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, c->stream, a, b, o, n); break;
      default: hipLaunchKernelGGL((k_pairing_check<C, 4>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, c->stream, a, b, o, n); break;
    }
  });
}

// ---- fused batches
static inline int popcount_mask(uint64_t m, int A) {
  int h = 0;
  for (int i = 0; i < A; i++) h += (int)((m >> i) & 1);
  return h;
}

// what a fused entry point needs besides the public key: without these the bases would silently be the point at infinity
enum { NEED_RP = 1, NEED_RETR = 2, NEED_SK = 4 };
static inline int check_fused(elp_ctx* c, uint64_t mask, int need = 0) {
  if (!c) return ELP_ERR_ARG;
  if (!c->have_pk) {
    c->err = "elp_set_pubkey must be called first";
    return ELP_ERR_STATE;
  }
  if ((need & NEED_RP) && !c->rp_set) {
    c->err = "elp_set_rp (service name) must be called first";
    return ELP_ERR_STATE;
  }
  if ((need & NEED_RETR) && !c->retr_set) {
    c->err = "id-retrieval needs elp_set_rp with authority_pk, g and h";
    return ELP_ERR_STATE;
  }
  if ((need & NEED_SK) && !c->sk_set) {
    c->err = "elp_set_signer_secret must be called first";
    return ELP_ERR_STATE;
  }
  if (c->A < 64 && (mask >> c->A) != 0) return ELP_ERR_ARG;
  return ELP_OK;
}
static inline int need_rp(int retr) { return NEED_RP | (retr ? NEED_RETR : 0); }

// Pippenger launch sequence over device buffers (points std affine, 32-byte scalars); `ws` needs msm_ws_bytes() bytes.
template <class C, int G>
static size_t msm_ws_bytes(size_t n) {
  const size_t AFF = G == 1 ? sizeof(Aff<F1<C>>) : sizeof(Aff<F2<C>>);
  const size_t JAC = G == 1 ? sizeof(Jac<F1<C>>) : sizeof(Jac<F2<C>>);
  const size_t S = (size_t)msm_slices(n, 16);      // the largest of the launch shapes (16 windows for the pairs of aggregated verification, 32 / 34 otherwise)
  return ((n * AFF + 255) & ~(size_t)255) + ((ELP_MSM_NWMAX * S * 256 * JAC + 255) & ~(size_t)255) + ((ELP_MSM_NWMAX * JAC + 255) & ~(size_t)255) + 256 +
         (G == 1 ? ((n * ELP_MSM_GLV_STRIDE + 255) & ~(size_t)255) : 0);
}
// returns the device address of the count of invalid input points (an int, zero after a clean run)
template <class C, int G>
int* msm_launch(hipStream_t stream, size_t n, const void* d_pts_std, const void* d_ks, void* d_out_std, uint8_t* ws, bool glv_pairs) {
  const size_t AFF = G == 1 ? sizeof(Aff<F1<C>>) : sizeof(Aff<F2<C>>);
  const size_t JAC = G == 1 ? sizeof(Jac<F1<C>>) : sizeof(Jac<F2<C>>);
  const bool split = G == 1 && !glv_pairs;        // full-width scalars in G1: split in two halves of ELP_MSM_GLV_HW bytes each (k_msm_split_scalars)
  const int NWL = glv_pairs ? 16 : split ? ELP_MSM_NWMAX : 32;                 // windows: the pairs (a, b) of aggregated verification fill 16 bytes; split scalars 2 x 17 + the virtual windows
  const int S = msm_slices(n, NWL);
  uint8_t* aff = ws;
  uint8_t* part = aff + ((n * AFF + 255) & ~(size_t)255);
  uint8_t* win = part + (((size_t)ELP_MSM_NWMAX * S * 256 * JAC + 255) & ~(size_t)255);
  int* bad = (int*)(win + ((ELP_MSM_NWMAX * JAC + 255) & ~(size_t)255));
  uint8_t* ks2 = (uint8_t*)bad + 256;
  (void)hipMemsetAsync(bad, 0, 4, stream);
  hipLaunchKernelGGL((k_msm_prepare<C, G>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, stream, (const u32*)d_pts_std, (void*)aff, bad, n);
  if constexpr (G == 1) {
    typedef F1<C> F;
    if (split) {
      hipLaunchKernelGGL((k_msm_split_scalars<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, stream, (const u32*)d_ks, ks2, n);
      msm_windows<F>(stream, NWL, S, (const Aff<F>*)aff, ks2, n, (Jac<F>*)part, (Jac<F>*)win, true);
      hipLaunchKernelGGL((k_msm_final_glv<C, ELP_MSM_GLV_HW, true>), dim3(1), dim3(ELP_BLOCK), 0, stream, (const void*)win, (u32*)d_out_std);
      return bad;
    }
    msm_windows<F>(stream, NWL, S, (const Aff<F>*)aff, (const uint8_t*)d_ks, n, (Jac<F>*)part, (Jac<F>*)win);
  } else {
    typedef F2<C> F;
    msm_windows<F>(stream, NWL, S, (const Aff<F>*)aff, (const uint8_t*)d_ks, n, (Jac<F>*)part, (Jac<F>*)win);
  }
  if constexpr (G == 1) {
    if (glv_pairs) {
      hipLaunchKernelGGL((k_msm_final_glv<C>), dim3(1), dim3(ELP_BLOCK), 0, stream, (const void*)win, (u32*)d_out_std);
      return bad;
    }
  }
  hipLaunchKernelGGL((k_msm_final<C, G>), dim3(1), dim3(ELP_BLOCK), 0, stream, (const void*)win, (u32*)d_out_std);
  return bad;
}

#ifndef ELP_MSM_TU
extern template int* msm_launch<BN254, 1>(hipStream_t stream, size_t n, const void* d_pts_std, const void* d_ks, void* d_out_std, uint8_t* ws, bool glv_pairs);
extern template int* msm_launch<BN254, 2>(hipStream_t stream, size_t n, const void* d_pts_std, const void* d_ks, void* d_out_std, uint8_t* ws, bool glv_pairs);
extern template int* msm_launch<BLS12_381, 1>(hipStream_t stream, size_t n, const void* d_pts_std, const void* d_ks, void* d_out_std, uint8_t* ws, bool glv_pairs);
extern template int* msm_launch<BLS12_381, 2>(hipStream_t stream, size_t n, const void* d_pts_std, const void* d_ks, void* d_out_std, uint8_t* ws, bool glv_pairs);
#endif
// elp_g1_msm_dev / elp_g2_msm_dev: the same launch sequence over the caller's device buffers and workspace, asynchronous on the caller's stream
template <class C, int G>
int msm_dev_t(elp_ctx* c, void* stream_, size_t n, const void* d_points, const void* d_scalars, void* d_workspace, void* d_out) {
  if (!c || !d_out || (n && (!d_points || !d_scalars || !d_workspace))) return ELP_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) {
    HIPCHK(c, hipMemsetAsync(d_out, 0, G == 1 ? Sizes<C>::G1 : Sizes<C>::G2, stream));
    return ELP_OK;
  }
  msm_launch<C, G>(stream, n, d_points, d_scalars, d_out, (uint8_t*)d_workspace);
  HIPCHK(c, hipGetLastError());
  return ELP_OK;
}

// the pairing check with ONE ITEM PER 16-LANE ROW (round 6; elpasso_pair16.h, translation unit elpasso_bn254_pair16.hip): 12 lanes hold one base-field coefficient each
// of the Fp12 value, every operation is one inner product per lane over operands published in LDS; reads K and `todo` like launch_pair4
template <class B>
struct Pair16Build {
  static constexpr bool value = false;
};
template <>
struct Pair16Build<BN254> {
  static constexpr bool value = true;
};
template <>
struct Pair16Build<BLS12_381> {
  static constexpr bool value = true;
};
template <class B>
void launch_pair16(hipStream_t stream, const void* gg_lines, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags,
                   void* d_accepted);
// ... the product tree of aggregated verification (k_fp12_reduce16: every product one twelve-term step on a row; out[b] = the product of in[32 b .. 32 b + 32)) ...
template <class B>
void launch_fp12_reduce16(hipStream_t stream, const void* in, size_t n, void* out);
// ... and the closing step of aggregated verification on one row (k_agg_final16): the fixed pair's Miller loop, the product with F, the final exponentiation
template <class B>
void launch_agg_final16(hipStream_t stream, const void* gg_lines, const void* F, const void* s2_std, int* agg_ok);
#ifndef ELP_PAIR16_TU
extern template void launch_fp12_reduce16<BN254>(hipStream_t stream, const void* in, size_t n, void* out);
extern template void launch_fp12_reduce16<BLS12_381>(hipStream_t stream, const void* in, size_t n, void* out);
extern template void launch_agg_final16<BN254>(hipStream_t stream, const void* gg_lines, const void* F, const void* s2_std, int* agg_ok);
extern template void launch_agg_final16<BLS12_381>(hipStream_t stream, const void* gg_lines, const void* F, const void* s2_std, int* agg_ok);
extern template void launch_pair16<BLS12_381>(hipStream_t stream, const void* gg_lines, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags, void* d_accepted);
extern template void launch_pair16<BN254>(hipStream_t stream, const void* gg_lines, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags, void* d_accepted);
#endif
// Which curves have the paired kernels in this build (their own translation unit, elpasso_<curve>_pair.hip).
template <class B>
struct PairedBuild {
  static constexpr bool value = false;
};
template <>
struct PairedBuild<BN254> {
  static constexpr bool value = true;
};
template <>
struct PairedBuild<BLS12_381> {
  static constexpr bool value = true;
};
template <class B>
void launch_agg_final_paired(elp_ctx* c, hipStream_t stream, const void* F, const void* s2_std);   // defined with the other paired launchers below
template <class B>
void launch_verify_id_agg_paired(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off,
                                 size_t ad_len, const AggSeed& seed, uint8_t* nizk_flags, void* deltas, void* sig2s, void* wave_prod);

template <class C>
int elp_verify_id_batch_aggregated_dev_t(elp_ctx* c, void* stream_, size_t n, const void* d_records, uint64_t mask, int retr,
                                         const void* d_ad, const void* d_ad_off, size_t ad_len, const uint8_t* seed32, void* d_flags,
                                         void* d_accepted) {
  int rc = check_fused(c, mask, need_rp(retr));
  if (rc) return rc;
  if (n == 0) return ELP_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const int H = popcount_mask(mask, c->A);
  if (H < (retr ? 2 : 1)) return ELP_ERR_ARG;
  const int words = verify_id_record_words<C>(c->A, H, retr != 0);
  // Two items per lane (k_verify_id_agg2, BN254): a round of lanes then takes 1.89 x the time of a one-item round and holds twice the items -- chosen when it needs
  // fewer than 1 / 1.89 of the rounds (65 537 ... 131 072 items on a chip of 1 024 SIMDs: one round instead of two; 131 073 ... 196 608: two instead of three does NOT pay)
  bool two_per_lane = false;
  if constexpr (std::is_same<C, BN254>::value) {
    const size_t round = (size_t)64 * c->simds, r1 = (n + round - 1) / round, r2 = (n + 2 * round - 1) / (2 * round);
    two_per_lane = c->agg_two == 2 || (c->agg_two == 1 && 189 * r2 < 100 * r1);
  }
  bool paired_main = false;                                // BLS12-381: the main kernel on lane pairs (32 items per wave)
  if constexpr (PairedBuild<C>::value && !C::IS_BN) paired_main = c->paired != 0 && c->agg_paired != 0;
  const size_t nw = paired_main ? (size_t)((2 * n + ELP_BLOCK - 1) / ELP_BLOCK) : two_per_lane ? grid_for((n + 1) / 2) : grid_for(n);      // waves = per-wave Miller products
  const size_t nw2 = (nw + 31) / 32;                                                          // the larger of the two product trees' first level (rows: 32 values per workgroup)
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_flags = 0, o_delta = al(n), o_sig2 = o_delta + al(n * 32), o_f1 = o_sig2 + al(n * Sizes<C>::G1),
               o_f2 = o_f1 + al(nw * sizeof(Fp12<C>)), o_f3 = o_f2 + al(nw2 * sizeof(Fp12<C>)), o_s2 = o_f3 + al(sizeof(Fp12<C>)),
               o_msm = o_s2 + al(Sizes<C>::G1), total = o_msm + msm_ws_bytes<C, 1>(n);
  if (c->agg_stream != stream) {          // a workspace per launch stream (see elp_ctx::agg_parked)
    if (c->agg_ws || c->agg_ok) c->agg_parked.push_back({c->agg_stream, c->agg_ws, c->agg_ws_bytes, c->agg_ok});
    c->agg_ws = nullptr;
    c->agg_ws_bytes = 0;
    c->agg_ok = nullptr;
    c->agg_stream = stream;
    for (size_t q = 0; q < c->agg_parked.size(); q++)
      if (c->agg_parked[q].stream == stream) {
        c->agg_ws = c->agg_parked[q].ws;
        c->agg_ws_bytes = c->agg_parked[q].bytes;
        c->agg_ok = c->agg_parked[q].ok;
        c->agg_parked.erase(c->agg_parked.begin() + (long)q);
        break;
      }
  }
  if (c->agg_ws_bytes < total) {
    HIPCHK(c, hipStreamSynchronize(stream));
    if (c->agg_ws) (void)hipFree(c->agg_ws);
    c->agg_ws = nullptr;
    c->agg_ws_bytes = 0;
    HIPCHK(c, hipMalloc(&c->agg_ws, total));
    c->agg_ws_bytes = total;
  }
  if (!c->agg_ok) HIPCHK(c, hipMalloc((void**)&c->agg_ok, 256));
  uint8_t* ws = (uint8_t*)c->agg_ws;
  // The multipliers must be unpredictable to every prover once the batch is fixed: seed32 == NULL (recommended) draws the seed from
  // the OS CSPRNG here; a caller-supplied seed (tests, reproducible runs) must be 32 fresh random bytes per batch.
  AggSeed seed;
  if (seed32) {
    memcpy(seed.b, seed32, 32);
  } else if (getrandom(seed.b, 32, 0) != 32) {
    c->err = "getrandom failed";
    return ELP_ERR_STATE;
  }
  KeyCtx<C> key = make_key_ws<C>(c, stream, (size_t)nw * ELP_BLOCK);
  if (paired_main) {
    if constexpr (PairedBuild<C>::value && !C::IS_BN)
      launch_verify_id_agg_paired<C>(c, stream, n, d_records, words, mask, retr, d_ad, d_ad_off, ad_len, seed, ws + o_flags, ws + o_delta, ws + o_sig2, ws + o_f1);
  } else if (two_per_lane)
    hipLaunchKernelGGL((k_verify_id_agg2<C>), dim3(nw), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr,
                       (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, seed, ws + o_flags, (u32*)(ws + o_delta), (u32*)(ws + o_sig2),
                       (Fp12<C>*)(ws + o_f1), n);
  else
    hipLaunchKernelGGL((k_verify_id_agg<C>), dim3(nw), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr,
                       (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, seed, ws + o_flags, (u32*)(ws + o_delta), (u32*)(ws + o_sig2),
                       (Fp12<C>*)(ws + o_f1), n);
  key.vtab = nullptr;   // the kernels below do not advance the pointer to their lane (the rare per-item fallback keeps its tables in private memory)
  // The two halves of the tail are independent -- the product of the per-wave Miller values (nw -> nw2 -> 1) and S2 = sum d_i sig2_i (Pippenger) -- and
  // neither fills the chip: with ELP_OPT_STREAM_OVERLAP the product runs on the context's second stream beside the sum.
  hipStream_t rstream = stream;                       // where the product runs: the second stream with ELP_OPT_STREAM_OVERLAP, else in line
  if (c->overlap) {
    if (!c->jstream) {
      HIPCHK(c, hipStreamCreateWithFlags(&c->jstream, hipStreamNonBlocking));
      HIPCHK(c, hipEventCreateWithFlags(&c->jev[0], hipEventDisableTiming));
      HIPCHK(c, hipEventCreateWithFlags(&c->jev[1], hipEventDisableTiming));
    }
    rstream = c->jstream;
    HIPCHK(c, hipEventRecord(c->jev[0], stream));
    HIPCHK(c, hipStreamWaitEvent(c->jstream, c->jev[0], 0));
  }
  bool tree16 = false;                                    // 32 values per wave, every product one step on a 16-lane row (round 6), or 64 per wave, one lane each
  if constexpr (Pair16Build<C>::value) tree16 = c->pair16 != 0;
  size_t left = nw;
  uint8_t *cur = ws + o_f1, *nxt = ws + o_f2;             // ping-pong (o_f1 is free again after the first level)
  do {
    const size_t nl = tree16 ? (left + 31) / 32 : grid_for(left);
    if constexpr (Pair16Build<C>::value) {
      if (tree16) launch_fp12_reduce16<C>(rstream, cur, left, nxt);
    }
    if (!tree16) hipLaunchKernelGGL((k_fp12_reduce<C>), dim3(nl), dim3(ELP_BLOCK), 0, rstream, (const Fp12<C>*)cur, left, (Fp12<C>*)nxt);
    uint8_t* t = cur;
    cur = nxt;
    nxt = t;
    left = nl;
  } while (left > 1);
  const Fp12<C>* F = (const Fp12<C>*)cur;
  if (c->overlap) HIPCHK(c, hipEventRecord(c->jev[1], c->jstream));
  // S2 = sum d_i sig2_i
  msm_launch<C, 1>(stream, n, ws + o_sig2, ws + o_delta, ws + o_s2, ws + o_msm, true);
  if (c->overlap) HIPCHK(c, hipStreamWaitEvent(stream, c->jev[1], 0));
  bool tail_done = false;
  if constexpr (Pair16Build<C>::value) {
    if (c->pair16 && c->pair16_tail) {                        // the serial tail on one 16-lane row (round 6): 472 steps of one inner product per lane
      launch_agg_final16<C>(stream, key.gg_lines, F, ws + o_s2, c->agg_ok);
      tail_done = true;
    }
  }
  if constexpr (CoopBuild<C>::value) {
    if (!tail_done && c->coop) {                              // the serial tail on 32 lanes (level-scheduled program): ~4x faster than a lane pair
      const void* consts = coop_consts_for<C>(c, stream);
      if (consts) {
        launch_agg_final_coop<C>(stream, key, consts, F, ws + o_s2, c->agg_ok);
        tail_done = true;
      }
    }
  }
  if constexpr (PairedBuild<C>::value) {
    if (tail_done) {
    } else
    if (c->paired != 0) {                                     // the serial tail on a lane pair: about 0.6x the latency of one lane
      launch_agg_final_paired<C>(c, stream, F, ws + o_s2);
      tail_done = true;
    }
  }
  if (!tail_done) hipLaunchKernelGGL((k_agg_final<C>), dim3(1), dim3(ELP_BLOCK), 0, stream, key, F, (const u32*)(ws + o_s2), c->agg_ok);
  hipLaunchKernelGGL((k_agg_finish<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr,
                     (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, (const uint8_t*)(ws + o_flags), (const int*)c->agg_ok,
                     (uint8_t*)d_flags, (unsigned long long*)d_accepted, n);
  HIPCHK(c, hipGetLastError());
  return ELP_OK;
}

static inline unsigned grid_for_paired(size_t n) { return (unsigned)((2 * n + ELP_BLOCK - 1) / ELP_BLOCK); }
// How many of n items the one-lane-per-item kernel takes (the rest goes to the two-lanes-per-item kernel; today the answer is all or none).
// Measured on MI355X (tools/probes/scale_probe.py, profiles/r02_layout_scale_*.log): both kernels are bound by vector-instruction issue.  The
// plain kernel runs one wave per SIMD and needs 64 x SIMDs items per round to fill the chip (18.0 ms per round of 65 536 at A = 8, W = 20); the paired
// kernel halves the latency of an item (a wave holds 32 items: 10.3 ms for 32 768 items) but issues ~17 % more instructions per item, so at two waves per
// SIMD it only ties (18.1 ms per 65 536).  Policy "by batch size": when the last round of the batch is at most half full (n mod 64 x SIMDs in
// (0, 32 x SIMDs]) the whole batch goes to the paired kernel in ONE launch (98 304 items: 26.8 ms against 34.1 ms plain and 28.3 ms for a plain
// round followed by a paired remainder), otherwise to the plain kernel.  BLS12-381 always takes the paired kernel.
static inline size_t layout_split(const elp_ctx* c, size_t n) {
  if (c->paired == 0) return n;
  if (c->paired == 1) return 0;
  if (c->curve == ELP_CURVE_BLS12_381) return 0;     // 14-limb field: the paired kernel is faster at every batch size (profiles/r02_layout_scale.log)
  const size_t round = (size_t)64 * c->simds;
  const size_t rem = n % round;
  return (rem != 0 && rem <= round / 2) ? 0 : n;
}
template <class B>
void launch_verify_id_paired(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr,
                             const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted) {
  hipLaunchKernelGGL((k_verify_id_paired<Paired<B>>), dim3(grid_for_paired(n)), dim3(ELP_BLOCK), 0, stream, make_key_ws<Paired<B>>(c, stream, (size_t)grid_for_paired(n) * ELP_BLOCK),
                     (const u32*)d_records, words, (u64)mask, retr, (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len,
                     (uint8_t*)d_flags, (unsigned long long*)d_accepted, n);
}
// the two launches of ELP_OPT_SPLIT_PHASES = 3: G1 jobs (plain layout, own translation unit), then the paired kernel over their output
template <class B>
void launch_vid_g1jobs(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, u32* ws, size_t stride, const KeyCtx<B>& key) {
  hipLaunchKernelGGL((k_vid_g1jobs<B>), dim3(4 * grid_for(n)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr, ws, stride, n);
}
template <class B>
void launch_verify_id_paired_g1(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad,
                                const void* d_ad_off, size_t ad_len, const u32* g1ws, size_t g1stride, void* d_flags, void* d_accepted, const KeyCtx<Paired<B>>& key) {
  (void)c;
  hipLaunchKernelGGL((k_verify_id_paired_g1<Paired<B>>), dim3(grid_for_paired(n)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr,
                     (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, g1ws, g1stride, (uint8_t*)d_flags, (unsigned long long*)d_accepted, n);
}
template <class B>
void launch_verify_id_wire_paired(elp_ctx* c, hipStream_t stream, size_t n, const void* d_msgs, const void* d_msg_off, int retr, const void* d_ad,
                                  const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted) {
  hipLaunchKernelGGL((k_verify_id_wire_paired<Paired<B>>), dim3(grid_for_paired(n)), dim3(ELP_BLOCK), 0, stream, make_key_ws<Paired<B>>(c, stream, (size_t)grid_for_paired(n) * ELP_BLOCK),
                     (const uint8_t*)d_msgs, (const u32*)d_msg_off, retr, (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, (uint8_t*)d_flags,
                     (unsigned long long*)d_accepted, n);
}
template <class B>
void launch_ps_verify_paired(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int nattr, void* d_flags, void* d_accepted) {
  hipLaunchKernelGGL((k_ps_verify_paired<Paired<B>>), dim3(grid_for_paired(n)), dim3(ELP_BLOCK), 0, stream, make_key<Paired<B>>(c),
                     (const u32*)d_records, 4 * B::N + 8 * nattr, nattr, (uint8_t*)d_flags, (unsigned long long*)d_accepted, n);
}
template <class B>
void launch_verify_id_agg_paired(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off,
                                 size_t ad_len, const AggSeed& seed, uint8_t* nizk_flags, void* deltas, void* sig2s, void* wave_prod) {
  hipLaunchKernelGGL((k_verify_id_agg_paired<Paired<B>>), dim3(grid_for_paired(n)), dim3(ELP_BLOCK), 0, stream,
                     make_key_ws<Paired<B>>(c, stream, (size_t)grid_for_paired(n) * ELP_BLOCK), (const u32*)d_records, words, (u64)mask, retr, (const uint8_t*)d_ad,
                     (const u32*)d_ad_off, (u32)ad_len, seed, nizk_flags, (u32*)deltas, (u32*)sig2s, (Fp12<B>*)wave_prod, n);
}
template <class B>
void launch_agg_final_paired(elp_ctx* c, hipStream_t stream, const void* F, const void* s2_std) {
  hipLaunchKernelGGL((k_agg_final_paired<Paired<B>>), dim3(1), dim3(ELP_BLOCK), 0, stream, make_key<Paired<B>>(c), (const Fp12<B>*)F, (const u32*)s2_std,
                     c->agg_ok);
}
// phase 1 of the two-phase verification lives in a translation unit of its own (elpasso_<curve>_nizk.hip): its device functions are compiled
// for 256 registers (two waves per SIMD) without constraining the kernels of the plain-layout unit, which own the whole register file
template <class B>
struct SplitBuild {
  static constexpr bool value = false;
};
template <>
struct SplitBuild<BN254> {
  static constexpr bool value = true;
};
template <class B>
void launch_vid_nizk(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad,
                     const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, u32* kws, size_t kstride, const KeyCtx<B>& key, const void* pre) {
  hipLaunchKernelGGL((k_vid_nizk<B>), dim3(grid_for(n)), dim3(ELP_NIZK_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr,
                     (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, nizk_ok, kws, kstride, n, (const Jac<F2<B>>*)pre);
}
template <class B>
void launch_vid_nizk4(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad,
                      const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, u32* kws, size_t kstride, const KeyCtx<B>& key, const void* pre, int k_done) {
  hipLaunchKernelGGL((k_vid_nizk4<B>), dim3(grid_for(n)), dim3(256), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr,
                     (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, nizk_ok, kws, kstride, n, (const Jac<F2<B>>*)pre, k_done);
}
template <class B>
void launch_vid_g2(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, uint8_t* ok_g2, u32* ws, size_t stride,
                   const KeyCtx<B>& key) {
  hipLaunchKernelGGL((k_vid_g2<B>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr, ok_g2, ws, stride, n);
}
template <class B>
void launch_vid_g1(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, u32* ws, size_t stride, const KeyCtx<B>& key) {
  hipLaunchKernelGGL((k_vid_g1<B>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr, ws, stride, n);
}
template <class B>
void launch_verify_id_staged(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off,
                             size_t ad_len, void* d_flags, void* d_accepted, const KeyCtx<B>& key) {
  hipLaunchKernelGGL((k_verify_id_staged<B>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, stream, key, (const u32*)d_records, words, (u64)mask, retr,
                     (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, (uint8_t*)d_flags, (unsigned long long*)d_accepted, n);
}
#ifndef ELP_STAGE_TU
extern template void launch_verify_id_staged<BN254>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted, const KeyCtx<BN254>& key);
#endif
#ifndef ELP_G2JOB_TU
extern template void launch_vid_g2<BN254>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, uint8_t* ok_g2, u32* ws, size_t stride, const KeyCtx<BN254>& key);
#endif
#ifndef ELP_G1JOB_TU
extern template void launch_vid_g1<BN254>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, u32* ws, size_t stride, const KeyCtx<BN254>& key);
#endif
#ifndef ELP_NIZK_TU
extern template void launch_vid_nizk4<BN254>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, u32* kws, size_t kstride, const KeyCtx<BN254>& key, const void* pre, int k_done);
extern template void launch_vid_nizk4<BLS12_381>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, u32* kws, size_t kstride, const KeyCtx<BLS12_381>& key, const void* pre, int k_done);
extern template void launch_vid_nizk<BN254>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, u32* kws, size_t kstride, const KeyCtx<BN254>& key, const void* pre);
#endif
#ifndef ELP_G1JOBS_TU
extern template void launch_vid_g1jobs<BN254>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, u32* ws, size_t stride, const KeyCtx<BN254>& key);
extern template void launch_vid_g1jobs<BLS12_381>(hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, u32* ws, size_t stride, const KeyCtx<BLS12_381>& key);
#endif
#ifndef ELP_PAIR_TU
extern template void launch_verify_id_paired<BN254>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted);
extern template void launch_ps_verify_paired<BN254>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int nattr, void* d_flags, void* d_accepted);
extern template void launch_verify_id_wire_paired<BN254>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_msgs, const void* d_msg_off, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted);
extern template void launch_verify_id_paired<BLS12_381>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted);
extern template void launch_ps_verify_paired<BLS12_381>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int nattr, void* d_flags, void* d_accepted);
extern template void launch_verify_id_wire_paired<BLS12_381>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_msgs, const void* d_msg_off, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted);
extern template void launch_verify_id_paired_g1<BN254>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, const u32* g1ws, size_t g1stride, void* d_flags, void* d_accepted, const KeyCtx<Paired<BN254>>& key);
extern template void launch_verify_id_paired_g1<BLS12_381>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, const u32* g1ws, size_t g1stride, void* d_flags, void* d_accepted, const KeyCtx<Paired<BLS12_381>>& key);
extern template void launch_agg_final_paired<BN254>(elp_ctx* c, hipStream_t stream, const void* F, const void* s2_std);
extern template void launch_agg_final_paired<BLS12_381>(elp_ctx* c, hipStream_t stream, const void* F, const void* s2_std);
extern template void launch_verify_id_agg_paired<BLS12_381>(elp_ctx* c, hipStream_t stream, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad,
                                                            const void* d_ad_off, size_t ad_len, const AggSeed& seed, uint8_t* nizk_flags, void* deltas, void* sig2s,
                                                            void* wave_prod);
#endif

// the pairing check on FOUR lanes per item (elp/pair4.h, elpasso_pair4.h; translation units elpasso_<curve>_pair4.hip): reads K and `todo` like k_pair_rest
template <class B>
void launch_pair4(hipStream_t stream, const void* gg_lines, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags,
                  void* d_accepted);
template <class B>
void launch_vid_mid(hipStream_t stream, const KeyCtx<B>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off,
                    size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, const void* pre);
#ifndef ELP_PAIR4_TU
extern template void launch_vid_mid<BN254>(hipStream_t stream, const KeyCtx<BN254>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, const void* pre);
extern template void launch_vid_mid<BLS12_381>(hipStream_t stream, const KeyCtx<BLS12_381>& key, size_t n, const void* d_records, int words, uint64_t mask, int retr, const void* d_ad, const void* d_ad_off, size_t ad_len, uint8_t* nizk_ok, const uint8_t* kvalid, const u32* kws, size_t kstride, uint8_t* pair_ok, const void* pre);
extern template void launch_pair4<BN254>(hipStream_t stream, const void* gg_lines, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags, void* d_accepted);
extern template void launch_pair4<BLS12_381>(hipStream_t stream, const void* gg_lines, size_t n, const void* d_records, int words, const uint8_t* todo, const u32* kws, size_t kstride, uint8_t* d_flags, void* d_accepted);
#endif
// batch sizes served by the four-lane pairing check by default (ELP_OPT_PAIR4 = 1), from the A/B sweeps of profiles/r05_four_lane.md: up to 16 384 items the quads
// of a batch run ONE wave per SIMD (256 workgroups of 64 items); el_passo_verify_id from 3 073 items (below, the interpreter's one launch is as fast or faster:
// 2 048 items 3.64 vs 3.75 ms, 4 096 items 3.87 vs 4.65), PS verification above the interpreter's 4 096 (4 096 items: 2.90 vs 3.47 ms; 5 120: 4.73 vs 3.52)
#ifndef ELP_PAIR4_MAX
#define ELP_PAIR4_MAX 16384
#endif
#ifndef ELP_PAIR4_VID_FROM
#define ELP_PAIR4_VID_FROM 3072
#endif

template <class C>
int elp_verify_id_batch_dev_t(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad,
                            const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted) {
  int rc = check_fused(c, mask, need_rp(retr));
  if (rc) return rc;
  if (n == 0) return ELP_OK;
  const int H = popcount_mask(mask, c->A);
  if (H < (retr ? 2 : 1)) return ELP_ERR_ARG;  // rs[0] (and rs[1]) are the responses of attributes 0 (and 1), src/ps-verifier.cc:95,107
  const int words = verify_id_record_words<C>(c->A, H, retr != 0);
  if constexpr (CoopBuild<C>::value && SmallBuild<C>::value) {
    // mid-size batches (round 5): the same job kernels for the NIZK half, the pairing check on FOUR lanes per item (k_vid_mid / k_pair4) -- between the batches the
    // interpreter serves best and the batches that fill the chip at one or two lanes per item
    const bool mid = c->pair4 == 2 ? n <= (size_t)(8 * ELP_PAIR4_MAX)
                                   : (c->pair4 == 1 && n <= (size_t)ELP_PAIR4_MAX && (n > (size_t)ELP_PAIR4_VID_FROM || !c->coop));      // interpreter off: four lanes beat two at every size of the range
    const bool small = !mid && c->coop && n <= (c->vid_coop_max ? c->vid_coop_max : (size_t)(C::IS_BN ? 9216 : 8192));
    if (small || mid) {
      // small batch: NIZK half with four job lanes per item (k_vid_nizk4), pairing check on 32 / 64 lanes per item (k_pair_coop)
      const void* consts = mid ? (const void*)c : coop_consts_for<C>(c, (hipStream_t)stream);      // the interpreter's constants: not needed by the four-lane check
      const size_t lanes = (size_t)grid_for(n) * ELP_BLOCK;
      const size_t k_bytes = (lanes * (size_t)vid_k_words<C>() * 4 + 255) & ~(size_t)255;
      void* extra = nullptr;
      const size_t pre_bytes = (lanes * 2 * sizeof(Jac<F2<C>>) + 255) & ~(size_t)255;      // fixed-base parts of V_k and K per item (k_vid_fixed_coop)
      const size_t psi_bytes = lanes * (size_t)(24 * vtab_entry_words<F2<C>>()) * 4;        // psi^j images of the multiples of k (k_vid_ktab -> the G2 job): a multiple of 16 per lane
      // four lanes per G2 job (round 5): batches of the one-launch path up to 512 items cut their NIZK half into workgroups of 16 items; the G1 job lanes of those
      // workgroups -- 64 per 16 items -- keep a table slice each, so the table workspace is provided for 4 x the lanes
      const bool quadx = elp_quad_g2_on && !mid && !(c->overlap) && c->use_vtab && n <= c->small_dense_from && n <= c->small_one_max && n <= (size_t)elp_quad_g2_max();
      const size_t lanes_tab = quadx && ((n + 15) / 16) * 64 > lanes ? ((n + 15) / 16) * 64 : lanes;
      KeyCtx<C> key = make_key_ws<C>(c, (hipStream_t)stream, lanes_tab, k_bytes + pre_bytes + ((4 * lanes + 255) & ~(size_t)255) + psi_bytes, &extra);
      if (quadx && key.vtab) key.flags |= KEY_QUAD_G2;
      if (consts && extra) {
        // ... where the NIZK workgroup is the critical path of the call (one round of pairing workgroups: a lone call 2.36 -> 2.23 ms, 64 items 2.48 -> 2.31, 1 024 items
        // 2.69 -> 2.42); above, the 24 extra entries per item only cost k_vid_prep time (4 096 items: 4.96 against 4.62 ms)
        if (n <= c->small_dense_from && !mid) key.vpsi = (u32*)((uint8_t*)extra + k_bytes + pre_bytes + ((4 * lanes + 255) & ~(size_t)255));
        // k_vid_fixed_coop (fixed-base sums, K) and k_vid_ktab (multiples of k) -> k_vid_nizk4 and k_pair_coop -> k_vid_combine.  The pairing check only
        // needs K, the NIZK half only the sums and the table: with ELP_OPT_STREAM_OVERLAP the two pairs of kernels run side by side on two streams
        //   caller's stream:  memset, k_vid_fixed_coop ------------------(e0)  wait(e3) k_vid_nizk4 ............ wait(e1) k_vid_combine
        //   second stream  :  wait(e2) k_vid_ktab (e3)        wait(e0) k_pair_coop, k_pair_rest (e1)
        // (measured: 64 items 2.6 instead of 4.6 ms); by default everything is queued in that order on the caller's stream.
        hipStream_t st = (hipStream_t)stream;
        u32* kws = (u32*)extra;
        void* pre = (uint8_t*)extra + k_bytes;
        uint8_t* nizk_ok = (uint8_t*)extra + k_bytes + pre_bytes;
        uint8_t* done = nizk_ok + lanes;
        uint8_t* kvalid = done + lanes;
        uint8_t* pair_ok = kvalid + lanes;
        hipStream_t js = st;
        const bool overlap = c->overlap && !mid;
        if (overlap) {
          if (!c->jstream) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->jstream, hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->jev[0], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->jev[1], hipEventDisableTiming));
          }
          if (!c->jev[2]) {
            HIPCHK(c, hipEventCreateWithFlags(&c->jev[2], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->jev[3], hipEventDisableTiming));
          }
          js = c->jstream;
        }
        HIPCHK(c, hipMemsetAsync(done, 0, lanes, st));
        if (overlap) {
          HIPCHK(c, hipEventRecord(c->jev[2], st));                        // what the caller queued before this call (the records) precedes the second stream's work
          HIPCHK(c, hipStreamWaitEvent(js, c->jev[2], 0));
        }
        if (!overlap) {
          launch_vid_prep<C>(st, key, n, d_records, words, mask, retr, pre, kws, lanes, kvalid);      // both as workgroup ranges of one launch
        } else {
          launch_vid_ktab<C>(js, key, n, d_records, words, retr);
          HIPCHK(c, hipEventRecord(c->jev[3], js));
          launch_vid_fixed_coop<C>(st, key, n, d_records, words, mask, retr, pre, kws, lanes, kvalid);
        }
        if (overlap) {
          HIPCHK(c, hipEventRecord(c->jev[0], st));
          HIPCHK(c, hipStreamWaitEvent(js, c->jev[0], 0));
          HIPCHK(c, hipStreamWaitEvent(st, c->jev[3], 0));
        }
        if (mid && c->mid_two_launches) {
          launch_vid_nizk4<C>(st, n, d_records, words, mask, retr, d_ad, d_ad_off, ad_len, nizk_ok, kws, lanes, key, pre, 1);
          launch_pair4<C>(st, key.gg_lines, n, d_records, words, kvalid, kws, lanes, pair_ok, nullptr);
        } else if (mid) {
          launch_vid_mid<C>(st, key, n, d_records, words, mask, retr, d_ad, d_ad_off, ad_len, nizk_ok, kvalid, kws, lanes, pair_ok, pre);      // both halves as workgroup ranges of one launch
        } else if (overlap) {
          launch_vid_nizk4<C>(st, n, d_records, words, mask, retr, d_ad, d_ad_off, ad_len, nizk_ok, kws, lanes, key, pre, 1);
          key.vtab = nullptr;
          launch_pair_coop<C>(js, key, consts, n, d_records, words, kvalid, kws, lanes, pair_ok, done, nullptr, ELP_REST_BY_CALLER);
          HIPCHK(c, hipEventRecord(c->jev[1], js));
          HIPCHK(c, hipStreamWaitEvent(st, c->jev[1], 0));
          launch_pair_rest<C>(st, key, n, d_records, words, kvalid, kws, lanes, pair_ok, done, nullptr);      // large private frame: on the caller's queue (see launch_pair_coop)
        } else if (n <= c->small_one_max) {
          // default: NIZK half and pairing check as workgroup ranges of ONE launch (k_vid_small) -- side by side without a second stream.  Its pairing
          // workgroups carry the NIZK half's register budget (one wave per SIMD); with all four of their waves interpreting they process items at the rate of
          // k_pair_coop's two waves per SIMD, so the one launch serves every batch of the cooperative path (ELP_SMALL_ONE_MAX: A/B against two launches).
          bool dense = false;
          if constexpr (Small2Build<C>::value) dense = n > c->small_dense_from;
          if (dense) {
            if constexpr (Small2Build<C>::value) {
              launch_vid_small2<C>(st, key, consts, n, d_records, words, mask, retr, d_ad, d_ad_off, ad_len, nizk_ok, kvalid, kws, lanes, pair_ok, done, pre);
              KeyCtx<C> k2 = key;
              k2.vtab = nullptr;
              launch_pair_rest<C>(st, k2, n, d_records, words, kvalid, kws, lanes, pair_ok, done, nullptr);
            }
          } else {
            launch_vid_small<C>(st, key, consts, n, d_records, words, mask, retr, d_ad, d_ad_off, ad_len, nizk_ok, kvalid, kws, lanes, pair_ok, done, pre);
          }
        } else {
          launch_vid_nizk4<C>(st, n, d_records, words, mask, retr, d_ad, d_ad_off, ad_len, nizk_ok, kws, lanes, key, pre, 1);
          key.vtab = nullptr;
          launch_pair_coop<C>(st, key, consts, n, d_records, words, kvalid, kws, lanes, pair_ok, done, nullptr, nullptr);
        }
        launch_vid_combine<C>(st, n, nizk_ok, pair_ok, d_flags, d_accepted);
        HIPCHK(c, hipGetLastError());
        return ELP_OK;
      }
    }
  }
  if constexpr (PairedBuild<C>::value) {
    const size_t np = layout_split(c, n);       // items [0, np): plain kernel; [np, n): paired kernel
    if (np < n) {
      const uint8_t* recs2 = (const uint8_t*)d_records + np * (size_t)words * 4;
      const u32* off2 = d_ad_off ? (const u32*)d_ad_off + np : nullptr;
      bool launched = false;
      if (c->split == 3) {
        // G1 jobs (three commitments + the subgroup tests, one lane per job, plain layout) as a kernel of their own, then the paired kernel over Fp2
        const size_t m = n - np;
        const size_t stride = (size_t)grid_for(m) * ELP_BLOCK;
        const size_t g1_bytes = (stride * (size_t)g1jobs_ws_words<C>() * 4 + 255) & ~(size_t)255;
        void* extra = nullptr;
        // one workspace serves both launches (they run one after the other on this stream): the paired kernel's lanes x its slice is the larger need
        KeyCtx<Paired<C>> pkey = make_key_ws<Paired<C>>(c, (hipStream_t)stream, (size_t)grid_for_paired(m) * ELP_BLOCK, g1_bytes, &extra);
        if (extra) {
          KeyCtx<C> jkey = make_key<C>(c);
          jkey.vtab = (u32*)pkey.vtab;       // 3 m slices of 8 G1 entries: smaller than the paired kernel's 2 m slices (pipeline.h vtab_words)
          static_assert(3 * 8 * vtab_entry_words<F1<C>>() <= 2 * vtab_words<Paired<C>>(), "the job kernel's tables fit the paired kernel's workspace");
          launch_vid_g1jobs<C>((hipStream_t)stream, m, recs2, words, mask, retr, (u32*)extra, stride, jkey);
          launch_verify_id_paired_g1<C>(c, (hipStream_t)stream, m, recs2, words, mask, retr, d_ad, off2, ad_len, (const u32*)extra, stride, (uint8_t*)d_flags + np,
                                        d_accepted, pkey);
          launched = true;
        }
      }
      if (!launched)
      launch_verify_id_paired<C>(c, (hipStream_t)stream, n - np, recs2, words, mask, retr, d_ad, off2, ad_len, (uint8_t*)d_flags + np, d_accepted);
      HIPCHK(c, hipGetLastError());
      if (np == 0) return ELP_OK;
      n = np;
    }
  }
  if constexpr (SplitBuild<C>::value) {
    if (c->split == 2) {
      // k_vid_g2 (this stream) || k_vid_g1 (second stream), then k_vid_pair2
      hipStream_t st = (hipStream_t)stream;
      if (!c->jstream) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->jstream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->jev[0], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->jev[1], hipEventDisableTiming));
      }
      const size_t lanes = (size_t)grid_for(n) * ELP_BLOCK;
      const size_t ws_bytes = (lanes * (size_t)vid_ws_words<C>() * 4 + 255) & ~(size_t)255;
      void* extra = nullptr;
      KeyCtx<C> key = make_key_ws<C>(c, st, lanes, ws_bytes + lanes, &extra);
      if (!extra) {
        c->err = "no device memory for the verification workspace";
        return ELP_ERR_HIP;
      }
      u32* ws = (u32*)extra;
      uint8_t* ok_g2 = (uint8_t*)extra + ws_bytes;
      HIPCHK(c, hipEventRecord(c->jev[0], st));                 // everything queued before this call (the records) precedes the G1 job too
      HIPCHK(c, hipStreamWaitEvent(c->jstream, c->jev[0], 0));
      launch_vid_g2<C>(st, n, d_records, words, mask, retr, ok_g2, ws, lanes, key);
      launch_vid_g1<C>(c->jstream, n, d_records, words, mask, retr, ws, lanes, key);
      HIPCHK(c, hipEventRecord(c->jev[1], c->jstream));
      HIPCHK(c, hipStreamWaitEvent(st, c->jev[1], 0));
      key.vtab = nullptr;
      hipLaunchKernelGGL((k_vid_pair2<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, st, key, (const u32*)d_records, words, (u64)mask, retr,
                         (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, (const uint8_t*)ok_g2, (const u32*)ws, lanes, (uint8_t*)d_flags,
                         (unsigned long long*)d_accepted, n);
      HIPCHK(c, hipGetLastError());
      return ELP_OK;
    }
    if (c->split == 1) {
      // two phases: k_vid_nizk (two job waves per 64 items, two waves per SIMD) -> per-item verdict + K in the workspace -> k_vid_pair
      const size_t lanes = (size_t)grid_for(n) * ELP_BLOCK;
      const size_t k_bytes = (lanes * (size_t)vid_k_words<C>() * 4 + 255) & ~(size_t)255;
      void* extra = nullptr;
      KeyCtx<C> key = make_key_ws<C>(c, (hipStream_t)stream, lanes, k_bytes + lanes, &extra);
      if (!extra) {
        c->err = "no device memory for the verification workspace";
        return ELP_ERR_HIP;
      }
      u32* kws = (u32*)extra;
      uint8_t* nizk_ok = (uint8_t*)extra + k_bytes;
      launch_vid_nizk<C>(c, (hipStream_t)stream, n, d_records, words, mask, retr, d_ad, d_ad_off, ad_len, nizk_ok, kws, lanes, key, nullptr);
      key.vtab = nullptr;
      hipLaunchKernelGGL((k_vid_pair<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, (hipStream_t)stream, key, (const u32*)d_records, words,
                         (const uint8_t*)nizk_ok, (const u32*)kws, lanes, (uint8_t*)d_flags, (unsigned long long*)d_accepted, n);
      HIPCHK(c, hipGetLastError());
      return ELP_OK;
    }
  }
  if constexpr (SplitBuild<C>::value) {
    if (c->stage_records && words <= ELP_STAGE_CAP && (words & 3) == 0 && ((uintptr_t)d_records & 15) == 0) {
      launch_verify_id_staged<C>((hipStream_t)stream, n, d_records, words, mask, retr, d_ad, d_ad_off, ad_len, d_flags, d_accepted,
                                 make_key_ws<C>(c, (hipStream_t)stream, (size_t)grid_for(n) * ELP_BLOCK));
      HIPCHK(c, hipGetLastError());
      return ELP_OK;
    }
  }
  hipLaunchKernelGGL((k_verify_id<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, (hipStream_t)stream, make_key_ws<C>(c, (hipStream_t)stream, (size_t)grid_for(n) * ELP_BLOCK),
                     (const u32*)d_records, words, (u64)mask, retr, (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len,
                     (uint8_t*)d_flags, (unsigned long long*)d_accepted, n);
  HIPCHK(c, hipGetLastError());
  return ELP_OK;
}
// Wire messages on the small / mid-size paths (round 5): decode into records with job-uniform waves (pipeline.h wire_decode_job: workgroup b runs job b % 7 of the
// items [64 (b / 7), 64 (b / 7) + 64)), then the record path, then the AND with the decoder's verdicts.
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_wire_decode(int A, const uint8_t* msgs, const u32* msg_off, int retr, u32* recs, int rec_words, uint8_t* okj, unsigned long long* mask_agg,
                                                size_t n) {
  const int job = (int)(blockIdx.x % WIRE_DECODE_JOBS);
  const size_t i = (size_t)(blockIdx.x / WIRE_DECODE_JOBS) * ELP_BLOCK + threadIdx.x;
  if (i >= n) return;
  u64 mask = 0;
  const bool ok = wire_decode_job<C>(job, A, msgs + msg_off[i], (size_t)(msg_off[i + 1] - msg_off[i]), retr != 0, recs + i * (size_t)rec_words, &mask);
  okj[(size_t)job * n + i] = ok ? 1 : 0;
  if (job == WIRE_DECODE_JOBS - 1 && ok) {
    atomicAnd(&mask_agg[0], (unsigned long long)mask);
    atomicOr(&mask_agg[1], (unsigned long long)mask);
  }
}
template <class C>
__global__ void ELP_LAUNCH_BOUNDS k_wire_combine(const uint8_t* verdict, const uint8_t* okj, uint8_t* flags, unsigned long long* accepted, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n) {
    ok = verdict[i] != 0;
    for (int j = 0; j < WIRE_DECODE_JOBS; j++) ok = ok && okj[(size_t)j * n + i] != 0;
    flags[i] = ok ? 1 : 0;
  }
  count_accept(ok, accepted);
}
template <class C>
int elp_verify_id_wire_batch_dev_t(elp_ctx* c, void* stream, size_t n, const void* d_msgs, const void* d_msg_off, int retr,
                                          const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags, void* d_accepted) {
  int rc = check_fused(c, 0, need_rp(retr));
  if (rc) return rc;
  if (n == 0) return ELP_OK;
  if constexpr (CoopBuild<C>::value && SmallBuild<C>::value) {
    // Batches of the sizes the record paths serve better than a round of the fused wire kernels (a lone message: ~3 ms instead of ~9): decode, check that all
    // messages hide the same attributes (the record kernels take one mask per launch; one 16-byte read-back), verify as records, AND with the decoder's verdicts.
    // Mixed patterns, and batches in which no message parses, take the fused kernels below.
    if (c->wire_decode && (c->coop || c->pair4) && n <= (size_t)ELP_PAIR4_MAX) {
      const int words = verify_id_record_words<C>(c->A, retr ? 2 : 1, retr != 0);      // the record's size does not depend on the number of hidden attributes
      const size_t rec_bytes = (n * (size_t)words * 4 + 255) & ~(size_t)255, ok_bytes = (((size_t)WIRE_DECODE_JOBS + 1) * n + 255) & ~(size_t)255;
      const size_t need = rec_bytes + ok_bytes + 256;
      elp_ctx::VtabWs* ww = nullptr;
      for (auto& e : c->wire_ws)
        if (e.stream == (hipStream_t)stream) ww = &e;
      if (!ww) {
        c->wire_ws.push_back({(hipStream_t)stream, nullptr, 0});
        ww = &c->wire_ws.back();
      }
      if (ww->bytes < need) {
        if (ww->p) HIPCHK(c, hipFree(ww->p));        // hipFree waits for the work that may still read it
        ww->p = nullptr;
        ww->bytes = 0;
        HIPCHK(c, hipMalloc(&ww->p, need + need / 4));
        ww->bytes = need + need / 4;
      }
      uint8_t* const wsb = (uint8_t*)ww->p;
      hipStream_t st = (hipStream_t)stream;
      u32* recs = (u32*)wsb;
      uint8_t* okj = wsb + rec_bytes;
      uint8_t* verdict = okj + (size_t)WIRE_DECODE_JOBS * n;
      unsigned long long* agg = (unsigned long long*)(wsb + rec_bytes + ok_bytes);
      HIPCHK(c, hipMemsetAsync(recs, 0, rec_bytes, st));       // a message that does not decode leaves an all-zero (rejected) record, not stale memory
      HIPCHK(c, hipMemsetAsync(agg, 0xff, 8, st));
      HIPCHK(c, hipMemsetAsync(agg + 1, 0, 8, st));
      hipLaunchKernelGGL((k_wire_decode<C>), dim3(grid_for(n) * WIRE_DECODE_JOBS), dim3(ELP_BLOCK), 0, st, c->A, (const uint8_t*)d_msgs, (const u32*)d_msg_off, retr, recs, words,
                         okj, agg, n);
      if (!c->wire_mask_host) HIPCHK(c, hipHostMalloc((void**)&c->wire_mask_host, 16, hipHostMallocDefault));      // page-locked: the read-back is one DMA, not a staged copy
      unsigned long long* const h = c->wire_mask_host;
      h[0] = 0;
      h[1] = 1;
      HIPCHK(c, hipMemcpyAsync(h, agg, 16, hipMemcpyDeviceToHost, st));
      HIPCHK(c, hipStreamSynchronize(st));      // the ONE synchronisation of this entry point (include/elpasso.h ELP_OPT_WIRE_DECODE)
      if (h[0] == h[1]) {          // one hidden pattern (a batch without a single well-formed message leaves ~0 != 0)
        rc = elp_verify_id_batch_dev_t<C>(c, stream, n, recs, (uint64_t)h[0], retr, d_ad, d_ad_off, ad_len, verdict, nullptr);
        if (rc) return rc;
        hipLaunchKernelGGL((k_wire_combine<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, st, (const uint8_t*)verdict, (const uint8_t*)okj, (uint8_t*)d_flags,
                           (unsigned long long*)d_accepted, n);
        HIPCHK(c, hipGetLastError());
        return ELP_OK;
      }
    }
  }
  if constexpr (PairedBuild<C>::value) {
    const size_t np = layout_split(c, n);       // messages [0, np): plain kernel; [np, n): paired kernel
    if (np < n) {
      launch_verify_id_wire_paired<C>(c, (hipStream_t)stream, n - np, d_msgs, (const u32*)d_msg_off + np, retr, d_ad,
                                      d_ad_off ? (const u32*)d_ad_off + np : nullptr, ad_len, (uint8_t*)d_flags + np, d_accepted);
      HIPCHK(c, hipGetLastError());
      if (np == 0) return ELP_OK;
      n = np;
    }
  }
  hipLaunchKernelGGL((k_verify_id_wire<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, (hipStream_t)stream, make_key_ws<C>(c, (hipStream_t)stream, (size_t)grid_for(n) * ELP_BLOCK),
                     (const uint8_t*)d_msgs, (const u32*)d_msg_off, retr, (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len,
                     (uint8_t*)d_flags, (unsigned long long*)d_accepted, n);
  HIPCHK(c, hipGetLastError());
  return ELP_OK;
}

template <class C>
int elp_ps_verify_batch_dev_t(elp_ctx* c, void* stream, size_t n, const void* d_records, int nattr, void* d_flags, void* d_accepted) {
  int rc = check_fused(c, 0);
  if (rc) return rc;
  if (nattr < 0 || nattr > c->A) return ELP_ERR_ARG;
  if (n == 0) return ELP_OK;
  if constexpr (CoopBuild<C>::value) {
    const bool small = c->coop && n <= c->coop_max && c->pair4 != 2;
    const bool mid = c->pair4 == 2 ? n <= (size_t)(8 * ELP_PAIR4_MAX) : (c->pair4 == 1 && !small && n <= (size_t)ELP_PAIR4_MAX);      // four lanes per item above the interpreter's range
    if (small || mid) {
      // small batch: K per item on eight lanes (three table sums), then the pairing check on 32 lanes per item (interpreter) or on four (k_pair4)
      const void* consts = mid ? (const void*)c : coop_consts_for<C>(c, (hipStream_t)stream);
      const size_t lanes = (size_t)grid_for(n) * ELP_BLOCK;
      const size_t k_bytes = (lanes * (size_t)vid_k_words<C>() * 4 + 255) & ~(size_t)255;
      void* extra = nullptr;
      const int use_vtab = c->use_vtab;
      c->use_vtab = 0;                                   // no tables of multiples on this path: only the state region of the workspace
      KeyCtx<C> key = make_key_ws<C>(c, (hipStream_t)stream, lanes, k_bytes + 2 * lanes, &extra);
      c->use_vtab = use_vtab;
      if (consts && extra) {
        u32* kws = (u32*)extra;
        uint8_t* todo = (uint8_t*)extra + k_bytes;
        uint8_t* done = todo + lanes;
        HIPCHK(c, hipMemsetAsync(done, 0, lanes, (hipStream_t)stream));
        const int words = 4 * C::N + 8 * nattr;
        launch_ps_k<C>((hipStream_t)stream, key, n, d_records, words, nattr, todo, kws, lanes);
        bool row16 = false;
        if constexpr (Pair16Build<C>::value) row16 = c->pair16 != 0 && !mid && n <= c->pair16_max && n >= c->pair16_min;
        if (row16) {
          if constexpr (Pair16Build<C>::value)
            launch_pair16<C>((hipStream_t)stream, key.gg_lines, n, d_records, words, todo, kws, lanes, (uint8_t*)d_flags, d_accepted);
        } else if (mid)
          launch_pair4<C>((hipStream_t)stream, key.gg_lines, n, d_records, words, todo, kws, lanes, (uint8_t*)d_flags, d_accepted);
        else
          launch_pair_coop<C>((hipStream_t)stream, key, consts, n, d_records, words, todo, kws, lanes, (uint8_t*)d_flags, done, d_accepted, nullptr);
        HIPCHK(c, hipGetLastError());
        return ELP_OK;
      }
    }
  }
  if constexpr (PairedBuild<C>::value) {
    const size_t np = layout_split(c, n);
    if (np < n) {
      const uint8_t* recs2 = (const uint8_t*)d_records + np * (size_t)(4 * C::N + 8 * nattr) * 4;
      launch_ps_verify_paired<C>(c, (hipStream_t)stream, n - np, recs2, nattr, (uint8_t*)d_flags + np, d_accepted);
      HIPCHK(c, hipGetLastError());
      if (np == 0) return ELP_OK;
      n = np;
    }
  }
  hipLaunchKernelGGL((k_ps_verify<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, (hipStream_t)stream, make_key<C>(c),
                     (const u32*)d_records, 4 * C::N + 8 * nattr, nattr, (uint8_t*)d_flags, (unsigned long long*)d_accepted, n);
  HIPCHK(c, hipGetLastError());
  return ELP_OK;
}
template <class C>
int elp_provide_id_batch_dev_t(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, const void* d_ad,
                             const void* d_ad_off, size_t ad_len, void* d_sigs, void* d_flags, void* d_accepted) {
  int rc = check_fused(c, mask, NEED_SK);
  if (rc) return rc;
  if (n == 0) return ELP_OK;
  const int H = popcount_mask(mask, c->A);
  const int words = provide_id_record_words<C>(c->A, H);
  hipLaunchKernelGGL((k_provide_id<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, (hipStream_t)stream, make_key<C>(c),
                     (const u32*)d_records, words, (u64)mask, (const uint8_t*)d_ad, (const u32*)d_ad_off, (u32)ad_len, (u32*)d_sigs,
                     (uint8_t*)d_flags, (unsigned long long*)d_accepted, n);
  HIPCHK(c, hipGetLastError());
  return ELP_OK;
}

template <class C>
int elp_prove_id_batch_dev_t(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, int retr, const void* d_ad,
                             const void* d_ad_off, size_t ad_len, void* d_proofs, void* d_flags, void* d_accepted) {
  int rc = check_fused(c, mask, need_rp(retr));
  if (rc) return rc;
  if (n == 0) return ELP_OK;
  const int H = popcount_mask(mask, c->A);
  if (H < (retr ? 2 : 1)) return ELP_ERR_ARG;
  hipLaunchKernelGGL((k_prove_id<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, (hipStream_t)stream, make_key<C>(c),
                     (const u32*)d_records, prove_id_record_words<C>(c->A, H, retr != 0), (u64)mask, retr, (const uint8_t*)d_ad,
                     (const u32*)d_ad_off, (u32)ad_len, (u32*)d_proofs, verify_id_record_words<C>(c->A, H, retr != 0),
                     (uint8_t*)d_flags, (unsigned long long*)d_accepted, n);
  HIPCHK(c, hipGetLastError());
  return ELP_OK;
}
template <class C>
int elp_request_id_batch_dev_t(elp_ctx* c, void* stream, size_t n, const void* d_records, uint64_t mask, const void* d_ad,
                               const void* d_ad_off, size_t ad_len, void* d_requests) {
  int rc = check_fused(c, mask);
  if (rc) return rc;
  if (n == 0) return ELP_OK;
  const int H = popcount_mask(mask, c->A);
  hipLaunchKernelGGL((k_request_id<C>), dim3(grid_for(n)), dim3(ELP_BLOCK), 0, (hipStream_t)stream, make_key<C>(c),
                     (const u32*)d_records, request_id_record_words<C>(c->A, H), (u64)mask, (const uint8_t*)d_ad,
                     (const u32*)d_ad_off, (u32)ad_len, (u32*)d_requests, request_id_out_words<C>(H), n);
  HIPCHK(c, hipGetLastError());
  return ELP_OK;
}

// host-buffer wrappers: stage inputs, call the _dev entry point on the context stream, copy results back
static inline int stage_ad(elp_ctx* c, size_t n, const uint8_t* ad, const uint32_t* ad_off, size_t ad_len, DevBuf& dad, DevBuf& doff,
                    const void** p_ad, const void** p_off) {
  size_t total = ad_off ? ad_off[n] : ad_len;
  HIPCHK(c, dad.alloc(total));
  if (total) HIPCHK(c, hipMemcpyAsync(dad.p, ad, total, hipMemcpyHostToDevice, c->stream));
  *p_ad = dad.p;
  *p_off = nullptr;
  if (ad_off) {
    HIPCHK(c, doff.alloc((n + 1) * 4));
    HIPCHK(c, hipMemcpyAsync(doff.p, ad_off, (n + 1) * 4, hipMemcpyHostToDevice, c->stream));
    *p_off = doff.p;
  }
  return ELP_OK;
}

// error exit of a host-buffer entry point: copies and kernels may already be queued on the context's stream(s) and still use the DevBufs that the exit returns to
// the process-wide block cache (another context could be handed the block at once), so drain the device first
static inline int sync_fail(int rc) {
  (void)hipDeviceSynchronize();
  return rc;
}
template <class C>
int elp_provide_id_batch_t(elp_ctx* c, size_t n, const uint8_t* records, uint64_t mask, const uint8_t* ad, const uint32_t* ad_off,
                         size_t ad_len, uint8_t* sigs, uint8_t* flags, uint64_t* accepted) {
  int rc = check_fused(c, mask, NEED_SK);
  if (rc) return sync_fail(rc);
  if (accepted) *accepted = 0;
  if (n == 0) return ELP_OK;
  if (!records || !flags || !sigs || (!ad && (ad_off ? ad_off[n] : ad_len))) return ELP_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t rsz = elp_provide_id_record_size(c->curve, c->A, popcount_mask(mask, c->A));
  DevBuf drec, dad, doff, dfl, dcnt, dsig;
  const void *pad, *poff;
  HIPCHK(c, drec.alloc(n * rsz));
  HIPCHK(c, dfl.alloc(n));
  HIPCHK(c, dcnt.alloc(8));
  HIPCHK(c, dsig.alloc(n * 2 * Sizes<C>::G1));
  HIPCHK(c, hipMemcpyAsync(drec.p, records, n * rsz, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(dcnt.p, 0, 8, c->stream));
  rc = stage_ad(c, n, ad, ad_off, ad_len, dad, doff, &pad, &poff);
  if (rc) return sync_fail(rc);
  rc = elp_provide_id_batch_dev(c, c->stream, n, drec.p, mask, pad, poff, ad_len, dsig.p, dfl.p, dcnt.p);
  if (rc) return sync_fail(rc);
  uint64_t cnt = 0;
  HIPCHK(c, hipMemcpyAsync(flags, dfl.p, n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(sigs, dsig.p, n * 2 * Sizes<C>::G1, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&cnt, dcnt.p, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (accepted) *accepted = cnt;
  return ELP_OK;
}

template <class C>
int elp_bench_op_t(elp_ctx* c, int op, size_t lanes, int iters, float* ms) {
  if (!c || !ms || lanes == 0 || op < 0 || op > 23) return ELP_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf out;
  HIPCHK(c, out.alloc(lanes * 4));
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  hipLaunchKernelGGL((k_bench_op<C>), dim3(grid_for(lanes)), dim3(ELP_BLOCK), 0, c->stream, op, (u32*)out.p, 1, lanes);  // warm-up
  HIPCHK(c, hipEventRecord(e0, c->stream));
  hipLaunchKernelGGL((k_bench_op<C>), dim3(grid_for(lanes)), dim3(ELP_BLOCK), 0, c->stream, op, (u32*)out.p, iters, lanes);
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipEventSynchronize(e1));
  HIPCHK(c, hipEventElapsedTime(ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ELP_OK;
}

template <class C>
int elp_bench_fp_mul_t(elp_ctx* c, size_t lanes, int iters, float* ms) {
  if (!c || !ms || lanes == 0) return ELP_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf out;
  HIPCHK(c, out.alloc(lanes * 4));
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  hipLaunchKernelGGL((k_bench_fp_mul<C>), dim3(grid_for(lanes)), dim3(ELP_BLOCK), 0, c->stream, (u32*)out.p, 1, lanes);  // warm-up
  HIPCHK(c, hipEventRecord(e0, c->stream));
  hipLaunchKernelGGL((k_bench_fp_mul<C>), dim3(grid_for(lanes)), dim3(ELP_BLOCK), 0, c->stream, (u32*)out.p, iters, lanes);
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipEventSynchronize(e1));
  HIPCHK(c, hipEventElapsedTime(ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ELP_OK;
}



