// VALU instruction-cost micro-benchmark for gfx950 (used to design fp_mul): measures cycles per wave-instruction for the
// integer ops a big-number multiplier is made of, at 1, 2, 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 64
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ void k(uint32_t* out, int iters) {
  uint32_t a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 + 99;
  uint64_t d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  uint32_t x = out[0], y = out[1] | 1;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REP; r++) {
      if (OP == 0) {  // 4 independent v_mad_u64_u32 chains
        asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x), "v"(y) : "vcc");
      } else if (OP == 1) {  // v_mul_lo_u32
        asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));
      } else if (OP == 2) {  // v_mul_hi_u32
        asm volatile("v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));
      } else if (OP == 3) {  // v_add_co_u32 + v_addc_co_u32 pairs
        asm volatile("v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_add_co_u32 %2, vcc, %2, %4\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y) : "vcc");
      } else if (OP == 4) {  // v_lshl_add_u64
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n\tv_lshl_add_u64 %1, %1, 0, %2\n\tv_lshl_add_u64 %2, %2, 0, %3\n\tv_lshl_add_u64 %3, %3, 0, %0"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
      } else if (OP == 5) {  // v_mov_b32
        asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      } else if (OP == 6) {  // v_add_u32
        asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));
      } else if (OP == 7) {  // v_mad_u32_u24
        asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n\tv_mad_u32_u24 %1, %1, %4, %2\n\tv_mad_u32_u24 %2, %2, %4, %3\n\tv_mad_u32_u24 %3, %3, %4, %0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));
      } else if (OP == 8) {  // v_mul_hi_u32_u24
        asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n\tv_mul_hi_u32_u24 %1, %1, %4\n\tv_mul_hi_u32_u24 %2, %2, %4\n\tv_mul_hi_u32_u24 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));
      } else if (OP == 9) {  // v_fma_f64
        asm volatile("v_fma_f64 %0, %0, %0, %1\n\tv_fma_f64 %1, %1, %1, %2\n\tv_fma_f64 %2, %2, %2, %3\n\tv_fma_f64 %3, %3, %3, %0"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
      } else if (OP == 10) {  // dependent chain of v_mad_u64_u32 (latency)
        asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x), "v"(y) : "vcc");
      } else if (OP == 11) {  // v_add3_u32
        asm volatile("v_add3_u32 %0, %0, %4, %1\n\tv_add3_u32 %1, %1, %4, %2\n\tv_add3_u32 %2, %2, %4, %3\n\tv_add3_u32 %3, %3, %4, %0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y));
      } else if (OP == 12) {  // v_mad_i32_i24 with sgpr-free: v_mad_u64_u32 with carry use: mad + addc alternating
        asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x), "v"(y), "v"(a0), "v"(a1) : "vcc");
      }
    }
  }
  long long t1 = clock64();
  uint32_t acc = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)d0 ^ (uint32_t)d1 ^ (uint32_t)d2 ^ (uint32_t)d3 ^ (uint32_t)(d0 >> 32);
  out[2 + blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}

template <int OP>
int run(const char* name, uint32_t* d, int waves_per_simd) {
  const int iters = 2000;
  int blocks = 256 * 4 * waves_per_simd;  // 64-thread blocks: one wave each
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 10);
  CHK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters);
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  long long cyc; CHK(hipMemcpy(&cyc, ((long long*)d) + (1 << 20), 8, hipMemcpyDeviceToHost));
  double ninstr = (double)iters * REP * 4;   // wave-instructions per wave
  // per-SIMD issue cost: waves_per_simd waves share one SIMD
  printf("%-28s waves/SIMD=%d  %.3f ms  wave-cycles/instr=%.2f  SIMD-cycles/instr=%.2f (clock64 ticks: %lld)\n", name, waves_per_simd, ms,
         (double)cyc / ninstr, (double)cyc / ninstr / waves_per_simd, cyc);
  return 0;
}

int main() {
  uint32_t* d;
  CHK(hipMalloc(&d, (size_t)(1 << 20) * 8 + 64));
  CHK(hipMemset(d, 1, (size_t)(1 << 20) * 8 + 64));
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_mad_u64_u32 (4 chains)", d, w);
    run<10>("v_mad_u64_u32 (dependent)", d, w);
    run<12>("v_mad_u64_u32 + v_addc", d, w);
    run<1>("v_mul_lo_u32", d, w);
    run<2>("v_mul_hi_u32", d, w);
    run<3>("v_add_co/v_addc_co", d, w);
    run<4>("v_lshl_add_u64", d, w);
    run<5>("v_mov_b32", d, w);
    run<6>("v_add_u32", d, w);
    run<11>("v_add3_u32", d, w);
    run<7>("v_mad_u32_u24", d, w);
    run<8>("v_mul_hi_u32_u24", d, w);
    run<9>("v_fma_f64", d, w);
  }
  return 0;
}
