// Measurement tool: does straight-line code larger than the instruction cache slow the Montgomery multiplier down?
// K chained fp_mul are fully unrolled into one loop body (about 2.3 KB of code each); total work is constant.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ps-signature-and-el-passo_amd/csrc tools/ubench_icache.hip -o build/ubench_icache
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "elp/fp.h"
#include "elp/params_bn254.h"
using namespace elp;

template <int K>
struct Chain {
  static __device__ __forceinline__ void run(Fp<BN254>& a, Fp<BN254>& b) {
    a = fp_mul<BN254>(a, b);
    b = fp_mul<BN254>(b, a);
    Chain<K - 2>::run(a, b);
  }
};
template <>
struct Chain<0> {
  static __device__ __forceinline__ void run(Fp<BN254>&, Fp<BN254>&) {}
};

template <int K>
__global__ void __launch_bounds__(64) k_chain(u32* out, int iters, u32 seed) {
  Fp<BN254> a, b;
  for (int i = 0; i < BN254::NL; i++) {
    a.v[i] = (i32)((seed + threadIdx.x * 7 + i * 13) & 0xfffffff);
    b.v[i] = (i32)((seed * 3 + blockIdx.x + i * 5) & 0xfffffff);
  }
  for (int it = 0; it < iters; it++) {
    Chain<K>::run(a, b);
  }
  u32 x = 0;
  for (int i = 0; i < BN254::NL; i++) x ^= (u32)a.v[i] ^ (u32)b.v[i];
  out[blockIdx.x * 64 + threadIdx.x] = x;
}

template <int K>
static void run(int waves_per_simd) {
  const int total = 1 << 15;   // fp_mul per lane
  int waves = 1024 * waves_per_simd;
  u32* d;
  hipMalloc(&d, waves * 64 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_chain<K>), dim3(waves), dim3(64), 0, 0, d, total / K, 12345u);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_chain<K>), dim3(waves), dim3(64), 0, 0, d, total / K, 12345u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("K=%4d (%7.1f KB body)  waves/SIMD=%d  %.3f ms  %.1f ns per fp_mul per wave-slot\n", K, K * 2.3, waves_per_simd, ms,
         ms * 1e6 / total / waves_per_simd);
  hipFree(d);
}

int main() {
  for (int w : {1, 2}) {
    run<2>(w);
    run<16>(w);
    run<64>(w);
    run<128>(w);
  }
  return 0;
}
