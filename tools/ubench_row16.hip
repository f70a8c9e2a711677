// Measurement tool (round 6; VERDICT r5 #3, the gate in front of a compiled 8- / 16-lane pairing check for BASELINE config 2): the Fp12-level building
// blocks of the pairing with ONE ITEM PER DPP ROW -- 12 of the 16 lanes of a row hold one base-field coefficient each of f = sum_k f_k w^k, f_k in Fp2
// (lane q = 2k + c: c = 0 real, c = 1 imaginary part), Fp12 = Fp2[w]/(w^6 - xi) -- against the FOUR-lanes-per-item layout of elp/quad.h.
//
// Scheme ("every lane computes one output coefficient as ONE inner product"): an operation first publishes its operands in LDS (the value, xi times the
// value, the second operand / the line), then every lane fetches the 2 x NT base-field operands its coefficient needs -- lane-dependent ADDRESSES and
// signs from a table in registers, the same instruction stream on every lane -- and evaluates one NT-term inner product with a single Montgomery reduction
// (quad.h fp_dot).  No interpretation, no divergence; all lanes of an item sit in one wave, so the two barriers per operation are wave-local.
//   fp12 product              12 terms per lane      h_k = sum_{i+j=k} f_i g_j + xi sum_{i+j=k+6} f_i g_j
//   fp12 squaring (Miller)     8 terms per lane      the same with the symmetric pairs merged (one operand doubled)
//   cyclotomic squaring (GS)   4 terms per lane      Granger-Scott over Fp4 = Fp2[s]/(s^2 - xi), s = w^3: f = A + B w + C w^2,
//                                                    f^2 = (3A^2 - 2conj(A)) + (3 s C^2 + 2 conj(B)) w + (3 B^2 - 2 conj(C)) w^2
//   sparse line product        6 terms per lane      f (a + b w + c w^3)
// Every routine is first CHECKED against the one-lane routines of elp/tower.h on the same inputs (bit-exact after canonicalisation); then the latency per
// operation of a lone wave per SIMD is printed next to the quad layout's, and the ratio the gate asks for (>= 1.6 x lower latency than four lanes).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ps-signature-and-el-passo_amd/csrc [-DUB_BLS=1] tools/ubench_row16.hip -o build/ubench_row16
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
static_assert(B::TWIST_D || true, "");

constexpr int NL = B::NL;
constexpr int NLP = (NL + 3) & ~3;               // limbs of a slot padded to whole 16-byte words
enum { V_F = 0, V_XF = 1, V_G = 2, NVAL = 3 };   // LDS values of a row: f, xi f, the second operand (or the line: coefficients 0..5 = a.re a.im b.re b.im c.re c.im)
constexpr int SLOT_ZERO = NVAL * 12;             // a slot of zeros (padding terms)
constexpr int NSLOT = NVAL * 12 + 1;
enum { OP_MUL = 0, OP_SQR = 1, OP_CYC = 2, OP_LINE = 3, NOPS = 4 };
static const char* OP_NAME[NOPS] = {"fp12 product", "fp12 squaring (Miller)", "cyclotomic squaring (GS)", "sparse line product"};
constexpr int NT_OF[NOPS] = {12, 8, 4, 6};
// term descriptor: a-slot | b-slot << 8 | negate << 16 | double << 17
struct Tables {
  u32 t[NOPS][16][12];
};
__constant__ Tables d_tbl;

// ---- host: the term tables from the algebra
static int slot(int val, int k, int c) { return val * 12 + 2 * k + c; }
static void add_fp2_product(u32* row, int& n, int c, int va, int ka, int vb, int kb, bool dbl, int budget, int& weight) {
  // coefficient c of (A * B), A = value va coefficient ka, B = value vb coefficient kb:  re = A.re B.re - A.im B.im,  im = A.re B.im + A.im B.re
  const u32 d = dbl ? (1u << 17) : 0;
  if (c == 0) {
    row[n++] = (u32)slot(va, ka, 0) | ((u32)slot(vb, kb, 0) << 8) | d;
    row[n++] = (u32)slot(va, ka, 1) | ((u32)slot(vb, kb, 1) << 8) | (1u << 16) | d;
  } else {
    row[n++] = (u32)slot(va, ka, 0) | ((u32)slot(vb, kb, 1) << 8) | d;
    row[n++] = (u32)slot(va, ka, 1) | ((u32)slot(vb, kb, 0) << 8) | d;
  }
  weight += dbl ? 4 : 2;
  if (weight > budget) {
    fprintf(stderr, "term table exceeds the accumulator headroom\n");
    exit(3);
  }
}
static void build_tables(Tables& T) {
  const int budget = B::HEADROOM - 1;
  for (int op = 0; op < NOPS; op++)
    for (int q = 0; q < 16; q++) {
      u32* row = T.t[op][q];
      int n = 0, weight = 0;
      const int k = (q < 12 ? q : 11) >> 1, c = (q < 12 ? q : 11) & 1;
      if (op == OP_MUL) {
        for (int i = 0; i < 6; i++) add_fp2_product(row, n, c, i <= k ? V_F : V_XF, i, V_G, (k - i + 6) % 6, false, budget, weight);
      } else if (op == OP_SQR) {
        for (int i = 0; i < 6; i++)
          for (int j = i; j < 6; j++)
            if ((i + j) % 6 == k) add_fp2_product(row, n, c, i + j >= 6 ? V_XF : V_F, i, V_F, j, i != j, budget, weight);
      } else if (op == OP_CYC) {
        // h_0 = 3 (f0^2 + xi f3^2) - 2 f0   h_3 = 3 (2 f0 f3) + 2 f3   h_1 = 3 xi (2 f2 f5) + 2 f1   h_4 = 3 (f2^2 + xi f5^2) - 2 f4
        // h_2 = 3 (f1^2 + xi f4^2) - 2 f2   h_5 = 3 (2 f1 f4) + 2 f5        (the factor 3 and the linear term are applied to the inner product)
        static const int X[6] = {0, 2, 1, 0, 2, 1}, Y[6] = {3, 5, 4, 3, 5, 4};
        if (k == 0 || k == 4 || k == 2) {
          add_fp2_product(row, n, c, V_F, X[k], V_F, X[k], false, budget, weight);
          add_fp2_product(row, n, c, V_XF, Y[k], V_F, Y[k], false, budget, weight);
        } else {
          add_fp2_product(row, n, c, k == 1 ? V_XF : V_F, X[k], V_F, Y[k], true, budget, weight);
        }
      } else {
        // h_k = a f_k + b F_{k-1} + c F_{k-3}, F_j = f_j for j >= 0 and xi f_{j+6} for j < 0; the line sits in value V_G: "coefficients" 0, 1, 2 = a, b, c
        add_fp2_product(row, n, c, V_F, k, V_G, 0, false, budget, weight);
        add_fp2_product(row, n, c, k - 1 >= 0 ? V_F : V_XF, (k - 1 + 6) % 6, V_G, 1, false, budget, weight);
        add_fp2_product(row, n, c, k - 3 >= 0 ? V_F : V_XF, (k - 3 + 6) % 6, V_G, 2, false, budget, weight);
      }
      if (n > NT_OF[op]) {
        fprintf(stderr, "op %d lane %d: %d terms\n", op, q, n);
        exit(3);
      }
      while (n < 12) row[n++] = (u32)SLOT_ZERO | ((u32)SLOT_ZERO << 8);
    }
}

// ---- device
__device__ __forceinline__ Fp<B> rnd_fp(u32& s) {
  Fp<B> r;
  for (int i = 0; i < NL; i++) {
    s = s * 1664525u + 1013904223u;
    r.v[i] = (i32)(s >> (33 - B::LB)) - (1 << (B::LB - 2));
  }
  r.v[NL - 1] >>= 8;
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
// reference chains on one lane per item (elp/tower.h)
template <int OP>
__global__ void __launch_bounds__(64) k_ref(Fp12<B>* out, const ItemIn* in, unsigned items, int iters) {
  const unsigned item = blockIdx.x * 64 + threadIdx.x;
  if (item >= items) return;
  Fp12<B> f = in[item].f, g = in[item].g;
  for (int it = 0; it < iters; it++) {
    if (OP == OP_MUL) fp12_mul<B>(f, f, g);
    else if (OP == OP_SQR) fp12_sqr<B>(f, f);
    else if (OP == OP_CYC) fp12_cyc_sqr<B>(f, f);
    else fp12_mul_by_line<B>(f, in[item].la, in[item].lb, in[item].lc);
  }
  out[item] = f;
}
// the quad layout (elp/quad.h) for the latency comparison of the SAME lease
template <int OP>
__global__ void __launch_bounds__(64, 1) k_quad(Fp12<B>* out, const ItemIn* in, unsigned items, int iters, int write) {
  const unsigned lane = blockIdx.x * 64 + threadIdx.x;
  const unsigned item = (lane / 4) % items;
  Fp12Q<P> f, g;
  fp12q_from_plain<P>(f, in[item].f);
  fp12q_from_plain<P>(g, in[item].g);
  const Fp2<P> la = fp2_from_mem<P>(in[item].la), lb = fp2_from_mem<P>(in[item].lb), lc = fp2_from_mem<P>(in[item].lc);
  ELP_NOUNROLL
  for (int it = 0; it < iters; it++) {
    if (OP == OP_MUL) fp12q_mul<P>(f, f, g);
    else if (OP == OP_SQR) fp12q_sqr<P>(f, f);
    else if (OP == OP_CYC) fp12q_cyc_sqr<P>(f, f);
    else fp12q_mul_by_line<P>(f, la, lb, lc);
  }
  if (write && lane / 4 < items) out[item].c0.c0.c0 = fp_cast<B>(f.h.c0.c);      // (a sink: the quad layout's results are checked by tools/ubench_tower.hip)
}

// Fp index of w-basis coefficient (k, c) inside the plain-layout Fp12 (c0 = (c0.c0, c0.c1, c0.c2) <-> w^0, w^2, w^4; c1 <-> w^1, w^3, w^5)
__device__ __forceinline__ int plain_index(int k, int c) { return (k & 1) * 6 + (k >> 1) * 2 + c; }
__device__ __forceinline__ Fp<B> row_pair_swap(const Fp<B>& a) {      // the other component of the lane's Fp2 coefficient: DPP quad_perm [1,0,3,2]
  Fp<B> r;
  ELP_UNROLL
  for (int i = 0; i < NL; i++) r.v[i] = __builtin_amdgcn_update_dpp(0, a.v[i], 0xB1, 0xF, 0xF, true);
  return r;
}
__device__ __forceinline__ void lds_put(i32* L, int s, const Fp<B>& a) {
  ELP_UNROLL
  for (int i = 0; i < NL; i++) L[s * NLP + i] = a.v[i];
}
__device__ __forceinline__ Fp<B> lds_get(const i32* L, int s) {
  Fp<B> r;
  ELP_UNROLL
  for (int i = 0; i < NL; i++) r.v[i] = L[s * NLP + i];
  return r;
}
// one inner product of NT terms for this lane's coefficient from the row's LDS values
template <int NT>
__device__ __forceinline__ Fp<B> row_dot(const i32* L, const u32 (&tb)[12]) {
  Fp<B> a[NT], b[NT];
  ELP_UNROLL
  for (int t = 0; t < NT; t++) {
    a[t] = lds_get(L, (int)(tb[t] & 0xFF));
    b[t] = lds_get(L, (int)((tb[t] >> 8) & 0xFF));
    const i32 m = -(i32)((tb[t] >> 16) & 1u);            // 0 / -1: conditional negation as (x ^ m) - m, doubling as a shift
    const u32 sh = (tb[t] >> 17) & 1u;
    ELP_UNROLL
    for (int i = 0; i < NL; i++) b[t].v[i] = ((b[t].v[i] << sh) ^ m) - m;
  }
  return fp_dot<B, NT>(a, b);
}
// publish the lane's coefficient of f and of xi f:  xi (x + y i) = (x - y) + (x + y) i
__device__ __forceinline__ void row_publish(i32* L, int q, int c, bool active, const Fp<B>& fo) {
  const Fp<B> p = row_pair_swap(fo);
  const Fp<B> xf = c == 0 ? fp_sub<B>(fo, p) : fp_add<B>(p, fo);
  if (active) {
    lds_put(L, V_F * 12 + q, fo);
    lds_put(L, V_XF * 12 + q, xf);
  }
}

template <int OP>
__global__ void __launch_bounds__(64, 1) k_row(Fp12<B>* out, const ItemIn* in, unsigned items, int iters, int write) {
  constexpr int NT = NT_OF[OP];
  __shared__ __attribute__((aligned(16))) i32 lds[4][NSLOT * NLP];
  const int row = (int)(threadIdx.x >> 4), q0 = (int)(threadIdx.x & 15);
  const bool active = q0 < 12;
  const int q = active ? q0 : 11, k = q >> 1, c = q & 1;
  const unsigned item = ((blockIdx.x * 64 + threadIdx.x) / 16) % items;
  i32* L = lds[row];
  u32 tb[12];
  ELP_UNROLL
  for (int t = 0; t < 12; t++) tb[t] = d_tbl.t[OP][q0][t];
  const Fp<B>* fplain = reinterpret_cast<const Fp<B>*>(&in[item].f);
  const Fp<B>* gplain = reinterpret_cast<const Fp<B>*>(&in[item].g);
  Fp<B> fo = fplain[plain_index(k, c)];
  if (q0 == 0) lds_put(L, SLOT_ZERO, fp_zero<B>());
  if (OP == OP_MUL) {
    if (active) lds_put(L, V_G * 12 + q, gplain[plain_index(k, c)]);
  } else if (OP == OP_LINE) {
    const Fp<B>* lp = reinterpret_cast<const Fp<B>*>(&in[item].la);      // la, lb, lc are contiguous: 6 base-field values
    if (q0 < 6) lds_put(L, V_G * 12 + q0, lp[q0]);
  }
  ELP_NOUNROLL
  for (int it = 0; it < iters; it++) {
    row_publish(L, q, c, active, fo);
    __syncthreads();
    Fp<B> r = row_dot<NT>(L, tb);
    if (OP == OP_CYC) {
      // h = 3 * (inner product) -/+ 2 f_k: minus for k = 0, 2, 4 (the real parts of A, B, C in Fp4), plus for k = 3, 1, 5
      const i32 s2 = (k & 1) ? 2 : -2;
      ELP_UNROLL
      for (int i = 0; i < NL; i++) r.v[i] = 3 * r.v[i] + s2 * fo.v[i];
      fp_carry<B>(r);
      fp_reduce_weak<B>(r);
    }
    __syncthreads();
    fo = r;
  }
  if (write && active) reinterpret_cast<Fp<B>*>(&out[item])[plain_index(k, c)] = fo;
}

#define HIPCHK(x)                                                                       \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)

typedef void (*kfn)(Fp12<B>*, const ItemIn*, unsigned, int, int);
typedef void (*rfn)(Fp12<B>*, const ItemIn*, unsigned, int);
static kfn ROW[NOPS] = {k_row<OP_MUL>, k_row<OP_SQR>, k_row<OP_CYC>, k_row<OP_LINE>};
static kfn QUAD[NOPS] = {k_quad<OP_MUL>, k_quad<OP_SQR>, k_quad<OP_CYC>, k_quad<OP_LINE>};
static rfn REF[NOPS] = {k_ref<OP_MUL>, k_ref<OP_SQR>, k_ref<OP_CYC>, k_ref<OP_LINE>};

int main(int argc, char** argv) {
  const int check_only = argc > 1 && !strcmp(argv[1], "check");
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, 0));
  const int simds = prop.multiProcessorCount * 4;
  printf("# ubench_row16: %s, %s, %d CUs (%d SIMDs); one item per 16-lane row (12 lanes active) against four lanes per item; lone wave per SIMD\n", CURVE_NAME, prop.name,
         prop.multiProcessorCount, simds);
  static Tables T;
  build_tables(T);
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_tbl), &T, sizeof T));
  const size_t words = 12 * B::N;
  const unsigned items = 256;
  ItemIn* d_in[2];
  for (int cyc = 0; cyc < 2; cyc++) {
    HIPCHK(hipMalloc(&d_in[cyc], items * sizeof(ItemIn)));
    hipLaunchKernelGGL(k_setup, dim3(items / 64), dim3(64), 0, 0, d_in[cyc], items, 20216u, cyc);
  }
  HIPCHK(hipDeviceSynchronize());
  Fp12<B>* d_res;
  u32* d_can;
  HIPCHK(hipMalloc(&d_res, items * sizeof(Fp12<B>)));
  HIPCHK(hipMalloc(&d_can, items * words * 4));
  int bad = 0;
  for (int op = 0; op < NOPS; op++) {
    const int cyc = op == OP_CYC;
    u32* h[2];
    for (int l = 0; l < 2; l++) {
      HIPCHK(hipMemset(d_res, 0, items * sizeof(Fp12<B>)));
      if (l == 0)
        hipLaunchKernelGGL(REF[op], dim3(items / 64), dim3(64), 0, 0, d_res, d_in[cyc], items, 3);
      else
        hipLaunchKernelGGL(ROW[op], dim3(items * 16 / 64), dim3(64), 0, 0, d_res, d_in[cyc], items, 3, 1);
      hipLaunchKernelGGL(k_canon, dim3(items / 64), dim3(64), 0, 0, d_can, d_res, items);
      HIPCHK(hipDeviceSynchronize());
      h[l] = (u32*)malloc(items * words * 4);
      HIPCHK(hipMemcpy(h[l], d_can, items * words * 4, hipMemcpyDeviceToHost));
    }
    int nz = 0;
    for (size_t i = 0; i < items * words; i++) nz |= h[0][i] != 0;
    const int e = memcmp(h[0], h[1], items * words * 4) != 0;
    printf("check %-28s one item per row %s the one-lane routine%s\n", OP_NAME[op], e ? "DIFFERS from" : "equals", nz ? "" : "  (all-zero output?)");
    bad |= e | !nz;
    free(h[0]);
    free(h[1]);
  }
  if (bad) {
    printf("PARITY FAILED\n");
    return 1;
  }
  if (check_only) return 0;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  printf("# op | layout | ms | ns per op per item (latency of the chain, lone wave per SIMD) | M item-ops per second (whole chip, one wave per SIMD)\n");
  for (int op = 0; op < NOPS; op++) {
    const int iters = 400, cyc = op == OP_CYC;
    double lat[2] = {0, 0};
    for (int l = 0; l < 2; l++) {
      kfn fn = l ? ROW[op] : QUAD[op];
      const int lanes = l ? 16 : 4;
      float t[2];
      for (int half = 0; half < 2; half++) {
        const int n = half ? iters : iters / 2;
        hipLaunchKernelGGL(fn, dim3(simds), dim3(64), 0, 0, d_res, d_in[cyc], items, n, 0);
        HIPCHK(hipDeviceSynchronize());
        float b = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
          HIPCHK(hipEventRecord(e0));
          hipLaunchKernelGGL(fn, dim3(simds), dim3(64), 0, 0, d_res, d_in[cyc], items, n, 0);
          HIPCHK(hipEventRecord(e1));
          HIPCHK(hipEventSynchronize(e1));
          float ms;
          HIPCHK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < b) b = ms;
        }
        t[half] = b;
      }
      const float best = (t[1] - t[0]) * 2;
      const double nitems = (double)simds * 64 / lanes;
      lat[l] = best * 1e6 / iters;
      printf("%-28s %-22s %8.3f ms  %10.1f ns/op  %9.2f Mop/s\n", OP_NAME[op], l ? "row of 16 (12 active)" : "four lanes (quad.h)", best, lat[l], nitems * iters / best / 1e3);
      fflush(stdout);
    }
    printf("%-28s latency ratio four lanes / row: %.2f  (gate: >= 1.6)\n", OP_NAME[op], lat[0] / lat[1]);
  }
  return 0;
}
