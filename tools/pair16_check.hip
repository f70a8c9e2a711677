// Debugging aid of the row-of-16 pairing check (round 6): runs k_pair16 on the vectors of tools/pair16_vectors.h (tools/gen_row16.py --vectors), dumps the slots of
// workgroup 0 before chosen program counters in canonical form, prints the verdicts.  tools/gen_row16.py --compare <output> compares the dumps with its simulator.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ps-signature-and-el-passo_amd/csrc tools/pair16_check.hip -o tools/pair16_check.bin
#include <stdio.h>
#include <stdlib.h>
#include "elpasso_pair16.h"
#include "../../tools/pair16_vectors.h"
using namespace elp;
typedef BN254 B;

__global__ void k_prep(LineCoef<B>* lines, u32* kws, size_t stride, u32* recs, uint8_t* todo, const unsigned* gg, const unsigned* kk, const unsigned* s1, const unsigned* s2,
                       const unsigned* bad2, int* ok) {
  Aff<F2<B>> q, aK;
  int good = g2_load<B>(q, gg) ? 1 : 0;
  good &= g2_load<B>(aK, kk) ? 1 : 0;
  ml_precompute<B>(lines, q);
  for (int i = 0; i < 4; i++) {
    vid_store_k<B>(kws, stride, (size_t)i, aK);
    for (int w = 0; w < 16; w++) {
      recs[i * 32 + w] = s1[w];
      recs[i * 32 + 16 + w] = (i == 1) ? bad2[w] : (i == 2 ? 0u : s2[w]);      // item 1: wrong sig2; item 2: sig2 = infinity (e(sig1, K) != 1); item 3 = item 0
    }
    todo[i] = 1;
  }
  *ok = good;
}
__global__ void k_canon(u32* out, const i32* dump, int nslots) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= nslots) return;
  Fp<B> v;
  for (int i = 0; i < B::NL; i++) v.v[i] = dump[s * R16_NLP + i];
  StdFp<B> t = fp_to_std<B>(v);
  for (int i = 0; i < B::N; i++) out[s * B::N + i] = t.w[i];
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

int main(int argc, char** argv) {
  LineCoef<B>* lines;
  u32 *kws, *recs, *canon;
  uint8_t *todo, *flags;
  unsigned *dgg, *dk, *ds1, *ds2, *dbad;
  int* ok;
  i32* dump;
  const size_t stride = 64;
  CK(hipMalloc(&lines, 80 * sizeof(LineCoef<B>)));
  CK(hipMalloc(&kws, 36 * stride * 4));
  CK(hipMalloc(&recs, 4 * 32 * 4));
  CK(hipMalloc(&todo, 64));
  CK(hipMalloc(&flags, 64));
  CK(hipMalloc(&ok, 4));
  const int nsl = R16_ROWS * R16_ROW_SLOTS;
  CK(hipMalloc(&dump, nsl * R16_NLP * 4));
  CK(hipMalloc(&canon, nsl * B::N * 4));
  CK(hipMalloc(&dgg, 128)); CK(hipMalloc(&dk, 128)); CK(hipMalloc(&ds1, 64)); CK(hipMalloc(&ds2, 64)); CK(hipMalloc(&dbad, 64));
  CK(hipMemcpy(dgg, V_GG, 128, hipMemcpyHostToDevice)); CK(hipMemcpy(dk, V_K, 128, hipMemcpyHostToDevice));
  CK(hipMemcpy(ds1, V_SIG1, 64, hipMemcpyHostToDevice)); CK(hipMemcpy(ds2, V_SIG2, 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dbad, V_BAD2, 64, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_prep, dim3(1), dim3(1), 0, 0, lines, kws, stride, recs, todo, dgg, dk, ds1, ds2, dbad, ok);
  CK(hipDeviceSynchronize());
  int hok = 0;
  CK(hipMemcpy(&hok, ok, 4, hipMemcpyDeviceToHost));
  printf("# inputs decode: %d\n", hok);
  CK(hipMemset(flags, 9, 64));
  hipLaunchKernelGGL(k_pair16, dim3(1), dim3(64), 0, 0, (const LineMem<B>*)lines, (const u32*)recs, 32, (const uint8_t*)todo, (const u32*)kws, stride, flags, (unsigned long long*)nullptr,
                     (size_t)4, -1, (i32*)nullptr);
  CK(hipDeviceSynchronize());
  uint8_t hf[4];
  CK(hipMemcpy(hf, flags, 4, hipMemcpyDeviceToHost));
  printf("# verdicts (valid, wrong sig2, sig2 = infinity, valid): %d %d %d %d   expected 1 0 0 1\n", hf[0], hf[1], hf[2], hf[3]);
  for (int a = 1; a < argc; a++) {
    const int pc = atoi(argv[a]);
    CK(hipMemset(dump, 0, nsl * R16_NLP * 4));
    hipLaunchKernelGGL(k_pair16, dim3(1), dim3(64), 0, 0, (const LineMem<B>*)lines, (const u32*)recs, 32, (const uint8_t*)todo, (const u32*)kws, stride, flags,
                       (unsigned long long*)nullptr, (size_t)4, pc, dump);
    hipLaunchKernelGGL(k_canon, dim3((nsl + 63) / 64), dim3(64), 0, 0, canon, (const i32*)dump, nsl);
    CK(hipDeviceSynchronize());
    static u32 h[R16_ROWS * R16_ROW_SLOTS * 8];
    CK(hipMemcpy(h, canon, nsl * B::N * 4, hipMemcpyDeviceToHost));
    for (int s = 0; s < R16_ROW_SLOTS; s++) {        // row 0 = the valid item
      printf("pc %d slot %d ", pc, s);
      for (int w = B::N - 1; w >= 0; w--) printf("%08x", h[s * B::N + w]);
      printf("\n");
    }
  }
  return 0;
}
