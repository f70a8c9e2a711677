// Measurement tool: where do the two 256-thread workgroups of a lone k_vid_small launch land, and what does placement cost?  A kernel of two workgroups -- A: four
// waves running a fixed dependent chain of 64-bit multiply-adds (the NIZK job waves), B: one such wave + three idle ones (the pairing interpreter) -- is launched
// 16 times, with four single-workgroup kernels in between as in the library's call sequence.  Every wave records XCC / SE / CU / SIMD and its cycle and wall-clock
// times.  hipcc --offload-arch=gfx950 -O2 tools/ubench_lone.hip -o /tmp/ubench_lone && /tmp/ubench_lone
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

struct Rec {
  unsigned long long w0, w1, c0, c1;
  unsigned hw, xcc;
};
__device__ __forceinline__ unsigned long long chain(unsigned long long acc, int iters) {
  unsigned a = (unsigned)acc | 1u, b = 0x9e3779b9u;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) acc = (unsigned long long)a * b + acc, a += (unsigned)(acc >> 33), b ^= (unsigned)acc;
  }
  return acc;
}
__global__ void __launch_bounds__(256) k_two(Rec* recs, unsigned long long* sink, int iters, int launch) {
  const int wave = threadIdx.x >> 6;
  const bool works = blockIdx.x == 0 || wave == 0;
  unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  unsigned long long acc = threadIdx.x + 1;
  if (works) acc = chain(acc, blockIdx.x == 0 && wave == 0 ? iters : (iters * 2) / 3);
  unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {
    Rec r;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(r.hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(r.xcc));
    r.w0 = w0; r.w1 = w1; r.c0 = c0; r.c1 = c1;
    recs[(launch * 2 + blockIdx.x) * 4 + wave] = r;
  }
  if (acc == 42) sink[0] = acc;
}
__global__ void __launch_bounds__(64) k_one(unsigned long long* sink, int iters) {
  unsigned long long acc = chain(threadIdx.x + 7, iters);
  if (acc == 42) sink[0] = acc;
}
int main(int argc, char** argv) {
  const int L = 16;
  const int idle_iters = argc > 1 ? atoi(argv[1]) : 2000;
  Rec* d;
  unsigned long long* sink;
  hipMalloc(&d, sizeof(Rec) * L * 8);
  hipMalloc(&sink, 8);
  hipEvent_t e0[L], e1[L];
  for (int i = 0; i < L; i++) hipEventCreate(&e0[i]), hipEventCreate(&e1[i]);
  for (int rep = 0; rep < 2; rep++)
    for (int i = 0; i < L; i++) {
      // two single-workgroup kernels of ~0.3 ms and ~2 ms in front (k_vid_fixed_coop, k_vid_ktab in the library): every XCD but the first idles meanwhile
      for (int q = 0; q < 2; q++) hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, sink, idle_iters);
      hipEventRecord(e0[i], 0);
      hipLaunchKernelGGL(k_two, dim3(2), dim3(256), 0, 0, d, sink, 16000, i);
      hipEventRecord(e1[i], 0);
      for (int q = 0; q < 2; q++) hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, sink, 100);
    }
  hipDeviceSynchronize();
  std::vector<Rec> h(L * 8);
  hipMemcpy(h.data(), d, sizeof(Rec) * L * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < L; i++) {
    float ms;
    hipEventElapsedTime(&ms, e0[i], e1[i]);
    printf("launch %2d  %.3f ms |", i, ms);
    for (int b = 0; b < 2; b++)
      for (int w = 0; w < 4; w++) {
        Rec& r = h[(i * 2 + b) * 4 + w];
        unsigned simd = (r.hw >> 4) & 3, cu = (r.hw >> 8) & 15, se = (r.hw >> 13) & 7;
        if (b == 0 || w == 0)
          printf(" %c%d x%u.s%u.c%02u.m%u %5.2fms %4.2fGHz |", b ? 'B' : 'A', w, r.xcc & 15, se, cu, simd, (r.w1 - r.w0) / 100000.0, (r.c1 - r.c0) / ((r.w1 - r.w0) * 10.0) / 1000.0);
      }
    printf("\n");
  }
  return 0;
}
