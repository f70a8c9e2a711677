// Measurement tool: cost of the tower building blocks in registers (no memory traffic), at 1 and 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ps-signature-and-el-passo_amd/csrc tools/ubench_tower.hip -o build/ubench_tower
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ELP_FP6_INLINE 1
#include "elp/tower.h"
#include "elp/params_bn254.h"
using namespace elp;
typedef BN254 C;

template <int OP>
__global__ void __launch_bounds__(64) k_op(u32* out, int iters, u32 seed) {
  Fp2<C> x, y, z, w;
  for (int i = 0; i < C::NL; i++) {
    x.c0.v[i] = (i32)((seed + threadIdx.x * 7 + i * 13) & 0xfffffff);
    x.c1.v[i] = (i32)((seed * 5 + threadIdx.x * 3 + i * 11) & 0xfffffff);
    y.c0.v[i] = (i32)((seed * 3 + blockIdx.x + threadIdx.x * 11 + i * 5) & 0xfffffff);
    y.c1.v[i] = (i32)((seed * 7 + blockIdx.x + threadIdx.x * 13 + i * 17) & 0xfffffff);
  }
  z = y;
  w = x;
  for (int it = 0; it < iters; it++) {
    if (OP == 0) {  // 2 fp_mul
      x.c0 = fp_mul<C>(x.c0, y.c0);
      y.c0 = fp_mul<C>(y.c0, x.c0);
    } else if (OP == 1) {  // 2 fp2_mul (3 products + 5 add/sub each)
      fp2_mul<C>(x, x, y);
      fp2_mul<C>(y, y, x);
    } else if (OP == 2) {  // 2 fp2_sqr
      fp2_sqr<C>(x, x);
      fp2_sqr<C>(y, y);
    } else if (OP == 3) {  // 8 carried additions (4 independent chains)
      x.c0 = fp_add(x.c0, y.c0); x.c1 = fp_sub(x.c1, y.c1); y.c0 = fp_add(y.c0, z.c0); y.c1 = fp_sub(y.c1, z.c1);
      z.c0 = fp_add(z.c0, w.c0); z.c1 = fp_sub(z.c1, w.c1); w.c0 = fp_add(w.c0, x.c0); w.c1 = fp_sub(w.c1, x.c1);
    } else if (OP == 4) {  // 8 lazy additions + 1 carry each 4
      x.c0 = fp_add_lazy(x.c0, y.c0); x.c1 = fp_sub_lazy(x.c1, y.c1); y.c0 = fp_add_lazy(y.c0, z.c0); y.c1 = fp_sub_lazy(y.c1, z.c1);
      z.c0 = fp_add_lazy(z.c0, w.c0); z.c1 = fp_sub_lazy(z.c1, w.c1); w.c0 = fp_add_lazy(w.c0, x.c0); w.c1 = fp_sub_lazy(w.c1, x.c1);
      if ((it & 3) == 3) { fp_carry(x.c0); fp_carry(x.c1); fp_carry(y.c0); fp_carry(y.c1); fp_carry(z.c0); fp_carry(z.c1); fp_carry(w.c0); fp_carry(w.c1); }
    } else if (OP == 5) {  // fp6_mul on (x,y,z)
      Fp6<C> a, b;
      a.c0 = x; a.c1 = y; a.c2 = z; b.c0 = w; b.c1 = x; b.c2 = y;
      fp6_mul<C>(a, a, b);
      x = a.c0; y = a.c1; z = a.c2;
    } else if (OP == 6) {  // 2 fp_sqr
      x.c0 = fp_sqr<C>(x.c0);
      y.c0 = fp_sqr<C>(y.c0);
    }
  }
  u32 acc = 0;
  for (int i = 0; i < C::NL; i++) acc ^= (u32)x.c0.v[i] ^ (u32)x.c1.v[i] ^ (u32)y.c0.v[i] ^ (u32)y.c1.v[i] ^ (u32)z.c0.v[i] ^ (u32)z.c1.v[i] ^ (u32)w.c0.v[i] ^ (u32)w.c1.v[i];
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int OP>
static void run(const char* name, int waves_per_simd, int iters) {
  int waves = 1024 * waves_per_simd;
  u32* d;
  hipMalloc(&d, waves * 64 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_op<OP>), dim3(waves), dim3(64), 0, 0, d, iters, 12345u);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_op<OP>), dim3(waves), dim3(64), 0, 0, d, iters, 12345u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s waves/SIMD=%d  %8.3f ms  %9.1f ns per iteration per wave-slot\n", name, waves_per_simd, ms, ms * 1e6 / iters / waves_per_simd);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("2 fp_mul", w, 4000);
    run<6>("2 fp_sqr", w, 4000);
    run<1>("2 fp2_mul", w, 2000);
    run<2>("2 fp2_sqr", w, 2000);
    run<3>("8 fp_add/sub carried", w, 4000);
    run<4>("8 lazy add/sub + 2 carries", w, 4000);
    run<5>("fp6_mul", w, 500);
  }
  return 0;
}
