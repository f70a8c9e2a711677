// Measurement tool (round 5; VERDICT r4 #1a): the Fp12-level building blocks of the pairing on ONE, TWO and FOUR lanes per item -- plain layout
// (elp/tower.h), paired layout (Paired<>, elp/common.h) and the quad layout (elp/quad.h) -- register-resident dependent chains, no memory traffic.
// For every operation and layout it prints the latency per operation of a lone wave per SIMD (what a batch that does not fill the chip pays) and the
// throughput in items' operations per second with 1, 2 and 4 waves per SIMD.  Every quad / paired routine is first CHECKED against the plain-layout
// routine on the same inputs (bit-exact after canonicalisation).  The gate of VERDICT r4 #1: four lanes >= 1.7 x lower latency per item than two lanes
// AND >= 0.75 of the two-lane throughput.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ps-signature-and-el-passo_amd/csrc [-DUB_BLS=1] tools/ubench_tower.hip -o build/ubench_tower
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifndef ELP_FP6_INLINE
#define ELP_FP6_INLINE 1
#endif
#include "elp/pairing.h"
#include "elp/quad.h"
#include "elp/params_bn254.h"
#include "elp/params_bls12_381.h"
using namespace elp;
#if UB_BLS
typedef BLS12_381 B;
#define CURVE_NAME "BLS12-381"
#else
typedef BN254 B;
#define CURVE_NAME "BN254"
#endif
typedef Paired<B> P;

#ifndef UB_WAVES
#define UB_WAVES 1
#endif

enum { OP_MUL = 0, OP_CYC = 1, OP_COMP = 2, OP_LINE = 3, OP_EXP = 4, OP_SQR = 5, NOPS = 6 };
static const char* OP_NAME[NOPS] = {"fp12 product", "cyclotomic squaring (GS)", "compressed squaring (Karabina)", "sparse line product", "f^|z| (GS chain)", "fp12 squaring (Miller)"};

// deterministic pseudo-random field elements from a seed (any carried limbs are a valid Montgomery-form value)
__device__ __forceinline__ Fp<B> rnd_fp(u32& s) {
  Fp<B> r;
  for (int i = 0; i < B::NL; i++) {
    s = s * 1664525u + 1013904223u;
    r.v[i] = (i32)(s >> (33 - B::LB)) - (1 << (B::LB - 2));
  }
  r.v[B::NL - 1] >>= 8;       // keep the value below a few p
  return r;
}
__device__ __forceinline__ Fp2<B> rnd_fp2(u32& s) {
  Fp2<B> r;
  r.c0 = rnd_fp(s);
  r.c1 = rnd_fp(s);
  return r;
}
__device__ __forceinline__ void rnd_fp12(Fp12<B>& f, u32& s) {
  f.c0.c0 = rnd_fp2(s); f.c0.c1 = rnd_fp2(s); f.c0.c2 = rnd_fp2(s);
  f.c1.c0 = rnd_fp2(s); f.c1.c1 = rnd_fp2(s); f.c1.c2 = rnd_fp2(s);
}
// an element of the cyclotomic subgroup: f^((p^6 - 1)(p^2 + 1))
__device__ void to_cyclotomic(Fp12<B>& f) {
  Fp12<B> t0, t1, g;
  fp12_inv<B>(t0, f);
  fp12_conj(t1, f);
  fp12_mul<B>(g, t1, t0);
  fp12_frob<B>(t0, g, 2);
  fp12_mul<B>(f, t0, g);
}
__device__ void canon12(u32* out, const Fp12<B>& f) {
  const Fp2<B>* c[6] = {&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2};
  for (int k = 0; k < 6; k++) {
    StdFp<B> a = fp_to_std<B>(c[k]->c0), b = fp_to_std<B>(c[k]->c1);
    for (int i = 0; i < B::N; i++) {
      out[(2 * k) * B::N + i] = a.w[i];
      out[(2 * k + 1) * B::N + i] = b.w[i];
    }
  }
}
// ---- the operations, one body per layout (LANES = 1, 2, 4).  `f` is the running value, `g` a second operand, (la, lb, lc) a line.
template <int LANES>
struct Ops;
template <>
struct Ops<1> {
  typedef B C;
  typedef Fp12<B> V;
  typedef CycComp<B> K;
  static ELP_INL void load(V& r, const Fp12<B>& m) { r = m; }
  static ELP_INL void store(Fp12<B>& m, const V& a) { m = a; }
  static ELP_INL Fp2<C> load2(const Fp2<B>& m) { return m; }
  static ELP_INL void mul(V& r, const V& a, const V& b) { fp12_mul<C>(r, a, b); }
  static ELP_INL void sqr(V& r, const V& a) { fp12_sqr<C>(r, a); }
  static ELP_INL void cyc(V& r, const V& a) { fp12_cyc_sqr<C>(r, a); }
  static ELP_INL void line(V& f, const Fp2<C>& a, const Fp2<C>& b, const Fp2<C>& c) { fp12_mul_by_line<C>(f, a, b, c); }
  static ELP_INL void expz(V& r, const V& a) { fp12_exp_u64_gs<C>(r, a, C::ZABS, nullptr); }
  static ELP_INL void to_comp(K& k, const V& a) { fp12_to_comp<C>(k, a); }
  static ELP_INL void comp_sqr(K& k) { cyc_comp_sqr_inl<C>(k, k); }
  static ELP_INL void comp_into(V& a, const K& k) { a.c1.c0 = k.z2; a.c0.c2 = k.z3; a.c0.c1 = k.z4; a.c1.c2 = k.z5; }
};
template <>
struct Ops<2> {
  typedef P C;
  typedef Fp12<P> V;
  typedef CycComp<P> K;
  static ELP_INL void load(V& r, const Fp12<B>& m) { fp12_from_mem<C>(r, m); }
  static ELP_INL void store(Fp12<B>& m, const V& a) {   // each lane writes its own components of the plain-layout value in memory
    const Fp2<C>* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
    Fp2<B>* d[6] = {&m.c0.c0, &m.c0.c1, &m.c0.c2, &m.c1.c0, &m.c1.c1, &m.c1.c2};
    for (int k = 0; k < 6; k++) (pair_odd() ? d[k]->c1 : d[k]->c0) = fp_cast<B>(c[k]->c);
  }
  static ELP_INL Fp2<C> load2(const Fp2<B>& m) { return fp2_from_mem<C>(m); }
  static ELP_INL void mul(V& r, const V& a, const V& b) { fp12_mul<C>(r, a, b); }
  static ELP_INL void sqr(V& r, const V& a) { fp12_sqr<C>(r, a); }
  static ELP_INL void cyc(V& r, const V& a) { fp12_cyc_sqr<C>(r, a); }
  static ELP_INL void line(V& f, const Fp2<C>& a, const Fp2<C>& b, const Fp2<C>& c) { fp12_mul_by_line<C>(f, a, b, c); }
  static ELP_INL void expz(V& r, const V& a) { fp12_exp_u64_gs<C>(r, a, C::ZABS, nullptr); }
  static ELP_INL void to_comp(K& k, const V& a) { fp12_to_comp<C>(k, a); }
  static ELP_INL void comp_sqr(K& k) { cyc_comp_sqr_inl<C>(k, k); }
  static ELP_INL void comp_into(V& a, const K& k) { a.c1.c0 = k.z2; a.c0.c2 = k.z3; a.c0.c1 = k.z4; a.c1.c2 = k.z5; }
};
template <>
struct Ops<4> {
  typedef P C;
  typedef Fp12Q<P> V;
  typedef CycCompQ<P> K;
  static ELP_INL void load(V& r, const Fp12<B>& m) { fp12q_from_plain<C>(r, m); }
  static ELP_INL void store(Fp12<B>& m, const V& a) {
    Fp6<B>& half = quad_hi() ? m.c1 : m.c0;
    (pair_odd() ? half.c0.c1 : half.c0.c0) = fp_cast<B>(a.h.c0.c);
    (pair_odd() ? half.c1.c1 : half.c1.c0) = fp_cast<B>(a.h.c1.c);
    (pair_odd() ? half.c2.c1 : half.c2.c0) = fp_cast<B>(a.h.c2.c);
  }
  static ELP_INL Fp2<C> load2(const Fp2<B>& m) { return fp2_from_mem<C>(m); }
  static ELP_INL void mul(V& r, const V& a, const V& b) { fp12q_mul<C>(r, a, b); }
  static ELP_INL void sqr(V& r, const V& a) { fp12q_sqr<C>(r, a); }
  static ELP_INL void cyc(V& r, const V& a) { fp12q_cyc_sqr<C>(r, a); }
  static ELP_INL void line(V& f, const Fp2<C>& a, const Fp2<C>& b, const Fp2<C>& c) { fp12q_mul_by_line<C>(f, a, b, c); }
  static ELP_INL void expz(V& r, const V& a) { fp12q_exp_u64_gs<C>(r, a, C::ZABS); }
  static ELP_INL void to_comp(K& k, const V& a) { fp12q_to_comp<C>(k, a); }
  static ELP_INL void comp_sqr(K& k) { cyc_compq_sqr<C>(k, k); }
  static ELP_INL void comp_into(V& a, const K& k) {      // back into the Fp12Q slots: low (c1, c2) = (z4, z3), high (c0, c2) = (z2, z5)
    const bool hi = quad_hi();
    const Fp2<C> other = fp2_quad_swap(k.x0);             // low receives z4, high receives z2
    a.h.c0 = fp2_select(hi, other, a.h.c0);
    a.h.c1 = fp2_select(hi, a.h.c1, other);
    a.h.c2 = k.x1;
  }
};

// operands of an item, prepared by a kernel of their own in the plain layout (so that the timed kernels contain the measured routine and nothing else)
struct ItemIn {
  Fp12<B> f, g;
  Fp2<B> la, lb, lc;
};
__global__ void __launch_bounds__(64) k_setup(ItemIn* in, unsigned items, u32 seed, int cyclotomic) {
  const unsigned item = blockIdx.x * 64 + threadIdx.x;
  if (item >= items) return;
  u32 s = seed ^ (item * 2654435761u);
  ItemIn x;
  rnd_fp12(x.f, s);
  rnd_fp12(x.g, s);
  if (cyclotomic) to_cyclotomic(x.f);
  x.la = rnd_fp2(s);
  x.lb = rnd_fp2(s);
  x.lc = rnd_fp2(s);
  in[item] = x;
}
__global__ void __launch_bounds__(64) k_canon(u32* out, const Fp12<B>* res, unsigned items) {
  const unsigned item = blockIdx.x * 64 + threadIdx.x;
  if (item < items) canon12(out + (size_t)item * 12 * B::N, res[item]);
}
// one kernel per (operation, layout): `iters` dependent repetitions on the item's value
template <int OP, int LANES>
__global__ void __launch_bounds__(64, UB_WAVES) k_op(Fp12<B>* out, const ItemIn* in, unsigned items, int iters, int write) {
  typedef Ops<LANES> O;
  const unsigned lane = blockIdx.x * 64 + threadIdx.x;
  const unsigned item = (lane / LANES) % items;
  typename O::V f, g;
  O::load(f, in[item].f);
  O::load(g, in[item].g);
  const auto la = O::load2(in[item].la), lb = O::load2(in[item].lb), lc = O::load2(in[item].lc);
  if (OP == OP_COMP) {
    typename O::K k;
    O::to_comp(k, f);
    ELP_NOUNROLL
    for (int it = 0; it < iters; it++) O::comp_sqr(k);
    O::comp_into(f, k);
  } else {
    ELP_NOUNROLL
    for (int it = 0; it < iters; it++) {
      if (OP == OP_MUL) O::mul(f, f, g);
      else if (OP == OP_SQR) O::sqr(f, f);
      else if (OP == OP_CYC) O::cyc(f, f);
      else if (OP == OP_LINE) O::line(f, la, lb, lc);
      else if (OP == OP_EXP) O::expz(f, f);
    }
  }
  if (write && lane / LANES < items) O::store(out[item], f);
}

#define HIPCHK(x)                                                                       \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)

template <int OP, int LANES>
static void launch(Fp12<B>* out, const ItemIn* in, unsigned items, int blocks, int iters, int write) {
  hipLaunchKernelGGL((k_op<OP, LANES>), dim3(blocks), dim3(64), 0, 0, out, in, items, iters, write);
}
typedef void (*launch_fn)(Fp12<B>*, const ItemIn*, unsigned, int, int, int);
static launch_fn LAUNCH[NOPS][3] = {
    {launch<OP_MUL, 1>, launch<OP_MUL, 2>, launch<OP_MUL, 4>},    {launch<OP_CYC, 1>, launch<OP_CYC, 2>, launch<OP_CYC, 4>},
    {launch<OP_COMP, 1>, launch<OP_COMP, 2>, launch<OP_COMP, 4>}, {launch<OP_LINE, 1>, launch<OP_LINE, 2>, launch<OP_LINE, 4>},
    {launch<OP_EXP, 1>, launch<OP_EXP, 2>, launch<OP_EXP, 4>},    {launch<OP_SQR, 1>, launch<OP_SQR, 2>, launch<OP_SQR, 4>}};
static const int LANES_OF[3] = {1, 2, 4};

int main(int argc, char** argv) {
  const int check_only = argc > 1 && !strcmp(argv[1], "check");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, 0));
  const int simds = prop.multiProcessorCount * 4;
  printf("# ubench_tower: %s, %s, %d CUs (%d SIMDs), build for %d wave(s) per SIMD\n", CURVE_NAME, prop.name, prop.multiProcessorCount, simds, UB_WAVES);
  const size_t words = 12 * B::N;
  // ---- parity of the layouts: 256 items, a few iterations, canonical words compared with the one-lane routine
  const unsigned items = 256;
  ItemIn* d_in[2];        // [0]: random operands, [1]: f in the cyclotomic subgroup
  for (int cyc = 0; cyc < 2; cyc++) {
    HIPCHK(hipMalloc(&d_in[cyc], items * sizeof(ItemIn)));
    hipLaunchKernelGGL(k_setup, dim3(items / 64), dim3(64), 0, 0, d_in[cyc], items, 20211u, cyc);
  }
  HIPCHK(hipDeviceSynchronize());
  {
    Fp12<B>* d_res;
    u32* d_can;
    HIPCHK(hipMalloc(&d_res, items * sizeof(Fp12<B>)));
    HIPCHK(hipMalloc(&d_can, items * words * 4));
    u32* h[3];
    int bad = 0;
    for (int op = 0; op < NOPS; op++) {
      const int iters = op == OP_EXP ? 1 : 3;
      const int cyc = op == OP_CYC || op == OP_COMP || op == OP_EXP;
      for (int l = 0; l < 3; l++) {
        HIPCHK(hipMemset(d_res, 0, items * sizeof(Fp12<B>)));
        LAUNCH[op][l](d_res, d_in[cyc], items, items * LANES_OF[l] / 64, iters, 1);
        hipLaunchKernelGGL(k_canon, dim3(items / 64), dim3(64), 0, 0, d_can, d_res, items);
        HIPCHK(hipDeviceSynchronize());
        h[l] = (u32*)malloc(items * words * 4);
        HIPCHK(hipMemcpy(h[l], d_can, items * words * 4, hipMemcpyDeviceToHost));
      }
      int nz = 0;
      for (size_t i = 0; i < items * words; i++) nz |= h[0][i] != 0;
      const int e2 = memcmp(h[0], h[1], items * words * 4) != 0, e4 = memcmp(h[0], h[2], items * words * 4) != 0;
      printf("check %-32s two lanes %s, four lanes %s%s\n", OP_NAME[op], e2 ? "DIFFER" : "equal", e4 ? "DIFFER" : "equal", nz ? "" : "  (all-zero output?)");
      bad |= e2 | e4 | !nz;
      for (int l = 0; l < 3; l++) free(h[l]);
    }
    HIPCHK(hipFree(d_res));
    HIPCHK(hipFree(d_can));
    if (bad) {
      printf("PARITY FAILED\n");
      return 1;
    }
    if (check_only) return 0;
  }
  // ---- timing
  Fp12<B>* d = nullptr;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  printf("# op | lanes per item | waves per SIMD | ms | ns per op per item (latency of the chain) | M item-ops per second (whole chip)\n");
  for (int op = 0; op < NOPS; op++) {
    const int iters = op == OP_EXP ? 8 : op == OP_MUL || op == OP_SQR ? 400 : 600;
    const int cyc = op == OP_CYC || op == OP_COMP || op == OP_EXP;
    for (int l = 0; l < 3; l++) {
      for (int w = 1; w <= UB_WAVES; w *= 2) {
        const int blocks = simds * w;
        // slope between two chain lengths: the set-up of the operands (random values, the easy part that puts them into the cyclotomic subgroup) drops out
        float t[2];
        for (int half = 0; half < 2; half++) {
          const int n = half ? iters : iters / 2;
          LAUNCH[op][l](d, d_in[cyc], items, blocks, n, 0);
          HIPCHK(hipDeviceSynchronize());
          float b = 1e30f;
          for (int rep = 0; rep < 3; rep++) {
            HIPCHK(hipEventRecord(e0));
            LAUNCH[op][l](d, d_in[cyc], items, blocks, n, 0);
            HIPCHK(hipEventRecord(e1));
            HIPCHK(hipEventSynchronize(e1));
            float ms;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < b) b = ms;
          }
          t[half] = b;
        }
        const float best = (t[1] - t[0]) * 2;        // time of `iters` operations
        const double items = (double)blocks * 64 / LANES_OF[l];
        printf("%-32s lanes=%d waves=%d  %8.3f ms  %10.1f ns/op  %9.2f Mop/s\n", OP_NAME[op], LANES_OF[l], w, best, best * 1e6 / iters, items * iters / best / 1e3);
        fflush(stdout);
      }
    }
  }
  return 0;
}
