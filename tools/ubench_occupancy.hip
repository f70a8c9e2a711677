// Measurement tool: how many waves of a 64-lane, ~240-VGPR kernel with S bytes of private (scratch) memory per lane does the
// runtime keep resident per SIMD?  Each wave records HW_ID / XCC_ID and its start / end time; the host computes the maximum overlap
// per SIMD.  hipcc --offload-arch=gfx950 -O2 tools/ubench_occupancy.hip -o /tmp/ubench_occ && /tmp/ubench_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>

struct Rec {
  unsigned long long t0, t1;
  unsigned hw, xcc;
};

template <int SCRATCH_WORDS>
__global__ void __launch_bounds__(64) k_occ(Rec* recs, unsigned* sink, unsigned long long spin_ticks, int idx) {
  unsigned priv[SCRATCH_WORDS > 0 ? SCRATCH_WORDS : 1];
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long r0 = wall_clock64();
  if (SCRATCH_WORDS > 0) {
    for (int i = 0; i < SCRATCH_WORDS; i += 61) priv[(i + idx) % SCRATCH_WORDS] = i * threadIdx.x;
  }
  asm volatile("v_mov_b32 v230, 0" ::: "v230");   // force a ~236-VGPR allocation like k_verify_id
  unsigned acc = 0;
  while (wall_clock64() - r0 < spin_ticks) {
    if (SCRATCH_WORDS > 0) acc += priv[(acc + idx) % SCRATCH_WORDS];
    acc = acc * 1664525u + 1013904223u;
  }
  unsigned long long r1 = wall_clock64();
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    Rec r;
    r.t0 = r0;
    r.t1 = r1;
    r.hw = hw;
    r.xcc = xcc;
    recs[blockIdx.x] = r;
  }
  if (acc == 12345 && t0 == 1) sink[0] = acc;
}

template <int W>
static void run(int waves) {
  Rec* d;
  unsigned* sink;
  hipMalloc(&d, sizeof(Rec) * waves);
  hipMalloc(&sink, 4);
  hipLaunchKernelGGL((k_occ<W>), dim3(waves), dim3(64), 0, 0, d, sink, 100000ull /* 1 ms at 100 MHz */, 3);
  hipError_t e = hipDeviceSynchronize();
  std::vector<Rec> h(waves);
  hipMemcpy(h.data(), d, sizeof(Rec) * waves, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (auto& r : h) {
    unsigned simd = (r.hw >> 4) & 3, cu = (r.hw >> 8) & 15, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7, xcc = r.xcc & 15;
    unsigned key = (xcc << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd;
    ev[key].push_back({r.t0, +1});
    ev[key].push_back({r.t1, -1});
    tmin = std::min(tmin, r.t0);
    tmax = std::max(tmax, r.t1);
  }
  int best = 0;
  std::map<int, int> hist;
  for (auto& kv : ev) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end());
    int cur = 0, mx = 0;
    for (auto& p : v) {
      cur += p.second;
      mx = std::max(mx, cur);
    }
    hist[mx]++;
    best = std::max(best, mx);
  }
  printf("scratch %6d B/lane  waves %5d  SIMDs used %4zu  max resident waves/SIMD %d  elapsed %.2f ms (%s)  histogram:", W * 4, waves,
         ev.size(), best, (tmax - tmin) / 100000.0, hipGetErrorString(e));
  for (auto& kv : hist) printf(" %dw:%d", kv.first, kv.second);
  printf("\n");
  hipFree(d);
  hipFree(sink);
}

int main() {
  for (int waves : {1024, 4096}) {
    run<0>(waves);
    run<512>(waves);
    run<1400>(waves);
    run<2800>(waves);
    run<5600>(waves);
  }
  return 0;
}
