// Measurement tool (round 6): where the time of k_msm_buckets goes.  The library's kernel beside three cut-down copies of it on the same inputs:
//   mode 0  the kernel as built into the library (histogram, scan, counting sort, size-ranked bucket sums since the last build of round 6)
//   mode 1  histogram + scan + counting sort only (every lane stores an empty sum)
//   mode 2  bucket sums only: lane b adds M / 256 points at fixed positions (no histogram, no sort, uniform trip count)
//   mode 7  the library's kernel on the layout of split scalars (window-major digits, sign bytes, negated points)
//   mode 6  buckets dealt to the lanes by size (see k_ranked)
//   mode 3  as mode 0 with the two global passes over the scalars reading a byte array laid out per window (one byte per point)
// Inputs are random field elements (not curve points: the formulas do not care), scalars random bytes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DELP_FP6_INLINE=1 -I ps-signature-and-el-passo_amd/csrc -I include tools/ubench_msm.hip -o build/ubench_msm
#include "elpasso_impl.h"
#include <stdio.h>
#include <vector>
using namespace elp;
typedef F1<BN254> F;

template <int MODE>
__global__ void ELP_MSM_LAUNCH_BOUNDS k_variant(const Aff<F>* pts, const uint8_t* scalars, const uint8_t* digits, size_t n, int S, Jac<F>* partial) {
  __shared__ unsigned cnt[256];
  __shared__ unsigned start[256];
  __shared__ unsigned wave_tot[4];
  __shared__ unsigned short idx[ELP_MSM_SLICE];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int w = blockIdx.x / S, s = blockIdx.x % S;
  const size_t lo = n * (size_t)s / S, hi = n * (size_t)(s + 1) / S;
  const int M = (int)(hi - lo);
  Jac<F> acc;
  jac_set_inf(acc);
  if (MODE == 2) {
    const int per = M / 256;
    for (int t = 0; t < per; t++) jac_madd<F>(acc, acc, pts[lo + (size_t)tid * per + t]);
    partial[(size_t)blockIdx.x * 256 + tid] = acc;
    return;
  }
  cnt[tid] = 0;
  __syncthreads();
  for (int j = tid; j < M; j += ELP_MSM_TPB) {
    unsigned d = MODE == 3 ? digits[(size_t)w * n + lo + j] : scalars[(lo + j) * 32 + w];
    if (d != 0 && (MODE == 3 || !aff_is_inf(pts[lo + j]))) atomicAdd(&cnt[d], 1u);
  }
  __syncthreads();
  unsigned c0 = cnt[tid], x = c0;
  for (int d = 1; d < 64; d <<= 1) {
    unsigned y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  if (lane == 63) wave_tot[wv] = x;
  __syncthreads();
  unsigned base = 0;
  for (int k = 0; k < wv; k++) base += wave_tot[k];
  const unsigned my_start = base + x - c0;
  start[tid] = my_start;
  __syncthreads();
  cnt[tid] = my_start;
  __syncthreads();
  for (int j = tid; j < M; j += ELP_MSM_TPB) {
    unsigned d = MODE == 3 ? digits[(size_t)w * n + lo + j] : scalars[(lo + j) * 32 + w];
    if (d != 0 && (MODE == 3 || !aff_is_inf(pts[lo + j]))) idx[atomicAdd(&cnt[d], 1u)] = (unsigned short)j;
  }
  __syncthreads();
  if (MODE != 1) {
    const unsigned t1 = start[tid] + c0;
    for (unsigned t = start[tid]; t < t1; t++) jac_madd<F>(acc, acc, pts[lo + idx[t]]);
  } else if (c0 == 77777) {
    acc.X.v[0] = (i32)idx[start[tid]];
  }
  partial[(size_t)blockIdx.x * 256 + tid] = acc;
}


// mode 4 / 5: LPB = 2 / 4 lanes per bucket (512 / 1 024 threads per workgroup): a bucket's run of the sorted list is split evenly over its lanes, their sums meet through shuffles
template <int LPB>
__global__ void __launch_bounds__(256 * LPB) k_split(const Aff<F>* pts, const uint8_t* scalars, size_t n, int S, Jac<F>* partial) {
  __shared__ unsigned cnt[256];
  __shared__ unsigned start[256];
  __shared__ unsigned cnt0[256];
  __shared__ unsigned wave_tot[4];
  __shared__ unsigned short idx[ELP_MSM_SLICE];
  constexpr int TPB = 256 * LPB;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int w = blockIdx.x / S, s = blockIdx.x % S;
  const size_t lo = n * (size_t)s / S, hi = n * (size_t)(s + 1) / S;
  const int M = (int)(hi - lo);
  if (tid < 256) cnt[tid] = 0;
  __syncthreads();
  for (int j = tid; j < M; j += TPB) {
    unsigned d = scalars[(lo + j) * 32 + w];
    if (d != 0 && !aff_is_inf(pts[lo + j])) atomicAdd(&cnt[d], 1u);
  }
  __syncthreads();
  unsigned c0 = tid < 256 ? cnt[tid] : 0, x = c0;
  for (int d = 1; d < 64; d <<= 1) {
    unsigned y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  if (tid < 256 && lane == 63) wave_tot[wv] = x;
  __syncthreads();
  if (tid < 256) {
    unsigned base = 0;
    for (int k = 0; k < wv; k++) base += wave_tot[k];
    const unsigned my_start = base + x - c0;
    start[tid] = my_start;
    cnt0[tid] = c0;
    cnt[tid] = my_start;
  }
  __syncthreads();
  for (int j = tid; j < M; j += TPB) {
    unsigned d = scalars[(lo + j) * 32 + w];
    if (d != 0 && !aff_is_inf(pts[lo + j])) idx[atomicAdd(&cnt[d], 1u)] = (unsigned short)j;
  }
  __syncthreads();
  const int b = tid / LPB, part = tid % LPB;
  const unsigned cb = cnt0[b], sb = start[b];
  const unsigned t0 = sb + (cb * (unsigned)part) / LPB, t1 = sb + (cb * (unsigned)(part + 1)) / LPB;
  Jac<F> acc;
  jac_set_inf(acc);
  for (unsigned t = t0; t < t1; t++) jac_madd<F>(acc, acc, pts[lo + idx[t]]);
  for (int d = 1; d < LPB; d <<= 1) {
    Jac<F> o;
    for (int i = 0; i < BN254::NL; i++) {
      o.X.v[i] = __shfl_xor(acc.X.v[i], d);
      o.Y.v[i] = __shfl_xor(acc.Y.v[i], d);
      o.Z.v[i] = __shfl_xor(acc.Z.v[i], d);
    }
    jac_add<F>(acc, acc, o);
  }
  if (part == 0) partial[(size_t)blockIdx.x * 256 + b] = acc;
}

// mode 6: as the library's kernel, with the buckets dealt to the lanes in the order of their sizes (lane t takes the bucket of rank t): a wave's 64 buckets are then
// of similar length, so the waves of a workgroup finish at different times and release their SIMD slots -- for batches that are work, not latency
__global__ void ELP_MSM_LAUNCH_BOUNDS k_ranked(const Aff<F>* pts, const uint8_t* scalars, size_t n, int S, Jac<F>* partial) {
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
  cnt[tid] = 0;
  __syncthreads();
  for (int j = tid; j < M; j += ELP_MSM_TPB) {
    unsigned d = scalars[(lo + j) * 32 + w];
    if (d != 0 && !aff_is_inf(pts[lo + j])) atomicAdd(&cnt[d], 1u);
  }
  __syncthreads();
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
  {                                                          // rank of this bucket among the 256 by size, largest first (ties by index)
    unsigned rank = 0;
    for (int k = 0; k < 256; k++) {
      const unsigned ck = cnt0[k];
      rank += (ck > c0 || (ck == c0 && k < tid)) ? 1u : 0u;
    }
    perm[rank] = (unsigned char)tid;
  }
  __syncthreads();
  cnt[tid] = my_start;
  __syncthreads();
  for (int j = tid; j < M; j += ELP_MSM_TPB) {
    unsigned d = scalars[(lo + j) * 32 + w];
    if (d != 0 && !aff_is_inf(pts[lo + j])) idx[atomicAdd(&cnt[d], 1u)] = (unsigned short)j;
  }
  __syncthreads();
  const int b = perm[tid];
  Jac<F> acc;
  jac_set_inf(acc);
  const unsigned t0 = start[b], t1 = t0 + cnt0[b];
  for (unsigned t = t0; t < t1; t++) jac_madd<F>(acc, acc, pts[lo + idx[t]]);
  partial[(size_t)blockIdx.x * 256 + b] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 65536;
  const int NW = argc > 2 ? atoi(argv[2]) : 16;
  std::vector<uint8_t> ks(n * 32), dg((size_t)32 * n);
  std::vector<Aff<F>> pts(n);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  for (auto& b : ks) b = (uint8_t)rnd();
  for (size_t i = 0; i < n; i++)
    for (int w = 0; w < 32; w++) dg[(size_t)w * n + i] = ks[i * 32 + w];
  for (auto& p : pts)
    for (int i = 0; i < BN254::NL; i++) {
      p.x.v[i] = (i32)(rnd() & 0x7FFFFFF);
      p.y.v[i] = (i32)(rnd() & 0x7FFFFFF);
    }
  Aff<F>* dp; uint8_t *dk, *dd; Jac<F>* part;
  CK(hipMalloc(&dp, n * sizeof(Aff<F>)));
  CK(hipMalloc(&dk, n * 32));
  CK(hipMalloc(&dd, n * 32));
  uint8_t* dk36;
  CK(hipMalloc(&dk36, n * 36));
  {
    std::vector<uint8_t> k36(n * 36);
    for (auto& b : k36) b = (uint8_t)rnd();
    CK(hipMemcpy(dk36, k36.data(), n * 36, hipMemcpyHostToDevice));
  }
  if (NW > 32) { printf("at most 32 windows (the plain scalars have 32 bytes)\n"); return 1; }
  CK(hipMalloc(&part, (size_t)32 * 256 * 256 * sizeof(Jac<F>)));
  CK(hipMemcpy(dp, pts.data(), n * sizeof(Aff<F>), hipMemcpyHostToDevice));
  u32* draw_pts = nullptr;
  int* dbad_pts = nullptr;
  if (argc > 3) {      // real curve points (std affine, 64 bytes each, e.g. written by tools/probes/msm_points.py) instead of random field elements
    FILE* fh = fopen(argv[3], "rb");
    std::vector<uint8_t> raw(n * 64);
    if (!fh || fread(raw.data(), 64, n, fh) != n) { printf("cannot read %zu points from %s\n", n, argv[3]); return 1; }
    fclose(fh);
    u32* draw; int* dbad;
    CK(hipMalloc(&draw, n * 64));
    CK(hipMalloc(&dbad, 4));
    CK(hipMemset(dbad, 0, 4));
    CK(hipMemcpy(draw, raw.data(), n * 64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_msm_prepare<BN254, 1>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, draw, (void*)dp, dbad, n);
    int hb = 0;
    CK(hipMemcpy(&hb, dbad, 4, hipMemcpyDeviceToHost));
    printf("real points: %zu loaded, %d invalid\n", n, hb);
    draw_pts = draw;
    dbad_pts = dbad;
  }
  CK(hipMemcpy(dk, ks.data(), n * 32, hipMemcpyHostToDevice));
  CK(hipMemcpy(dd, dg.data(), n * 32, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int S : {8, 16, 32, 64, 128}) {
    if ((size_t)S * 8192 < n || n / S < 256) continue;
    for (int mode = 0; mode < 8; mode++) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; rep++) {
        if (draw_pts && getenv("UB_REPREPARE")) hipLaunchKernelGGL((k_msm_prepare<BN254, 1>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, draw_pts, (void*)dp, dbad_pts, n);      // as in the library: the points are rewritten before every sum
        CK(hipEventRecord(e0, 0));
        if (mode == 0) hipLaunchKernelGGL((k_msm_buckets<F>), dim3(NW * S), dim3(ELP_MSM_TPB), 0, 0, dp, dk, n, S, part, 32, -1, 0);
        if (mode == 7) hipLaunchKernelGGL((k_msm_buckets<F>), dim3(NW * S), dim3(ELP_MSM_TPB), 0, 0, dp, dk36, n, S, part, 0, 2 * ELP_MSM_GLV_HW, ELP_MSM_GLV_HW);
        if (mode == 1) hipLaunchKernelGGL((k_variant<1>), dim3(NW * S), dim3(ELP_MSM_TPB), 0, 0, dp, dk, dd, n, S, part);
        if (mode == 2) hipLaunchKernelGGL((k_variant<2>), dim3(NW * S), dim3(ELP_MSM_TPB), 0, 0, dp, dk, dd, n, S, part);
        if (mode == 3) hipLaunchKernelGGL((k_variant<3>), dim3(NW * S), dim3(ELP_MSM_TPB), 0, 0, dp, dk, dd, n, S, part);
        if (mode == 4) hipLaunchKernelGGL((k_split<2>), dim3(NW * S), dim3(512), 0, 0, dp, dk, n, S, part);
        if (mode == 5) hipLaunchKernelGGL((k_split<4>), dim3(NW * S), dim3(1024), 0, 0, dp, dk, n, S, part);
        if (mode == 6) hipLaunchKernelGGL(k_ranked, dim3(NW * S), dim3(ELP_MSM_TPB), 0, 0, dp, dk, n, S, part);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
      }
      printf("n=%zu windows=%d slices=%3d (%5zu points per workgroup, %5.1f per bucket)  mode %d: %8.1f us\n", n, NW, S, n / S, (double)n / S / 255.0, mode, best * 1e3);
    }
  }
  return 0;
}
