/*
 * elp_oracle.c -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; it is never on the
 * product path (the product is the HIP library behind include/elpasso.h and has no CPU fallback).
 *
 * What is restated, in the reference's own structure (one independent scalar multiplication per term, two full
 * pairings compared in GT, no fixed-base tables, no batching):
 *   el_passo_verify_id                       /root/reference/src/ps-verifier.cc:37-138
 *   el_passo_verify_id_without_id_retrieval  /root/reference/src/ps-verifier.cc:140-212
 *   prepare_hybrid_verification              /root/reference/src/ps-verifier.cc:214-229
 *   PSVerifier::verify                       /root/reference/src/ps-verifier.cc:13-35
 *   el_passo_provide_id / nizk_verify / sign_hybrid / sign_commitment   /root/reference/src/ps-signer.cc:63-146
 * The arithmetic underneath lives in herumi/mcl (third-parties/mcl, an un-vendored submodule whose sources and pinned
 * commit are absent from /root/reference; approx. v1.2x).  Its published algorithms are restated here for the curve
 * the reference actually runs on (mcl's default BN254: p = 36z^4+36z^3+24z^2+6z+1, z = -(2^62+2^55+1), y^2 = x^3+2,
 * Fp2 = Fp[i]/(i^2+1), xi = 1+i, D-type twist): Montgomery Fp, Fp2/Fp6/Fp12 tower, Jacobian G1/G2 with width-5 NAF
 * scalar multiplication, optimal ate Miller loop, final exponentiation, little-endian serialisation with y-parity
 * flag, Fr::setHashOf masking, Shallue-van de Woestijne hashAndMapToG1, SHA-256.
 *
 * Pinning: tests/test_oracle_golden.py checks this library against every golden vector captured from the reference's
 * own prebuilt wasm (tests/golden/bn254_*.json) and against the independent big-int model oracle/pymodel.py.
 *
 * Representation: 4 x 64-bit limbs, unsigned __int128 products (deliberately different from the HIP path's 9 x 29-bit signed
 * limbs; the Miller loop uses Jacobian line formulas, the HIP path homogeneous ones).
 *
 * Second build, -DELPO_BLS12_381 (libelp_oracle_bls.so): the same restatement over BLS12-381 (6 x 64-bit limbs, y^2 = x^3 + 4, M-type twist
 * y^2 = x^3 + 4 xi, Miller loop over |z| = 0xd201000000010000, final exponentiation by the Hayashida-Hayasaka-Teruya chain AND, as a
 * self-check, by plain square-and-multiply with the integer (p^4-p^2+1)/r).  Pinning: the reference's prebuilt wasm is mcl built with
 * MCL_MAX_BIT_SIZE=384 (/root/reference/Makefile:65); with mcl's default CurveParam overwritten in its linear memory before initPairing()
 * (oracle/wasm_curve.js) it runs the whole protocol on BLS12-381, and tests/test_oracle_bls_golden.py checks this build against every vector
 * so produced (tests/golden/bls12_381_*.json): verdicts of the RP and IdP modules, hashAndMapToG1 as mcl defines it for this curve (SHA-512
 * Fp::setHashOf, Shallue-van de Woestijne map with b = 4, cofactor (z-1)^2/3) and mcl's GLV G1::mul on points outside G1.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------------ Fp */
#ifdef ELPO_BLS12_381
#define NL 6            /* 64-bit limbs of a field element */
#define FB 48           /* bytes of a field element on the wire / in records */
#define PBITS 381
#define RBITS 255
#define CURVE_B_INT 4
typedef struct { u64 v[NL]; } fp;
static const u64 P[NL] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const u64 RORD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const u64 ZABS = 0xd201000000010000ull; /* |z|, z < 0 */
static const u64 ZM1D3_ABS = 0x460055555555aaabull;      /* |z - 1| / 3 */
static const u64 G1_COFACTOR[4] = {0x8c00aaab0000aaabull, 0x396c8c005555e156ull, 0, 0};   /* (z - 1)^2 / 3 */
/* (p^4 - p^2 + 1) / r, 1268 bits: the hard part of the final exponentiation as one integer (self-check of the chain) */
static const u64 HARD_EXP[20] = {0xe516c3f438e3ba79ull, 0xfa9912aae208ccf1ull, 0x905ce937335d5b68ull, 0xc71a2629b0dea236ull, 0x83774940996754c8ull,
                                 0x21d160aeb6a1e799ull, 0x2ed0b283ed237db4ull, 0x915c97f36c6f1821ull, 0x67f17fcbde783765ull, 0x2378b9039096d1b7ull,
                                 0x7988f8761bdc51dcull, 0x2076995003fc77a1ull, 0x827eca0ba621315bull, 0xe5a72bce8d63cb9full, 0xf68f7764c28b6f8aull,
                                 0x2f230063cf081517ull, 0x94506632528d6a9aull, 0xd3cde88eeb996ca3ull, 0xc0bd38c3195c899eull, 0x000f686b3d807d01ull};
#else
#define NL 4
#define FB 32
#define PBITS 254
#define RBITS 254
#define CURVE_B_INT 2
typedef struct { u64 v[NL]; } fp;
static const u64 P[NL] = {0xa700000000000013ull, 0x6121000000000013ull, 0xba344d8000000008ull, 0x2523648240000001ull};
static const u64 RORD[4] = {0xa10000000000000dull, 0xff9f800000000010ull, 0xba344d8000000007ull, 0x2523648240000001ull};
static const u64 ZABS = 0x4080000000000001ull; /* |z|, z < 0 */
#endif
static u64 PINV;                                /* -p^-1 mod 2^64 */
static fp FP_ONE, FP_R2, FP_ZERO;
static int g_init = 0;

static int geq_p(const u64* a) {
  for (int i = NL - 1; i >= 0; i--) {
    if (a[i] > P[i]) return 1;
    if (a[i] < P[i]) return 0;
  }
  return 1;
}
static void sub_p(u64* a) {
  u128 br = 0;
  for (int i = 0; i < NL; i++) {
    u128 t = (u128)a[i] - P[i] - br;
    a[i] = (u64)t;
    br = (t >> 64) & 1;
  }
}
static void fp_add(fp* r, const fp* a, const fp* b) {
  u128 c = 0;
  for (int i = 0; i < NL; i++) {
    c += (u128)a->v[i] + b->v[i];
    r->v[i] = (u64)c;
    c >>= 64;
  }
  if (geq_p(r->v)) sub_p(r->v);      /* p < 2^(64 NL - 1) on both curves: the sum of two reduced values has no carry out */
}
static void fp_sub(fp* r, const fp* a, const fp* b) {
  u128 br = 0;
  u64 t[NL];
  for (int i = 0; i < NL; i++) {
    u128 d = (u128)a->v[i] - b->v[i] - br;
    t[i] = (u64)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < NL; i++) {
      c += (u128)t[i] + P[i];
      t[i] = (u64)c;
      c >>= 64;
    }
  }
  memcpy(r->v, t, 8 * NL);
}
static int fp_is_zero(const fp* a) {
  u64 t = 0;
  for (int i = 0; i < NL; i++) t |= a->v[i];
  return t == 0;
}
static int fp_eq(const fp* a, const fp* b) { return memcmp(a->v, b->v, 8 * NL) == 0; }
static void fp_neg(fp* r, const fp* a) {
  if (fp_is_zero(a)) { *r = *a; return; }
  fp_sub(r, &FP_ZERO, a);
}
static void fp_dbl(fp* r, const fp* a) { fp_add(r, a, a); }
/* Montgomery multiplication, operand scanning (separate product and reduction passes) */
static void fp_mul(fp* r, const fp* a, const fp* b) {
  u64 t[2 * NL + 1] = {0};
  for (int i = 0; i < NL; i++) {
    u128 c = 0;
    for (int j = 0; j < NL; j++) {
      c += (u128)a->v[j] * b->v[i] + t[i + j];
      t[i + j] = (u64)c;
      c >>= 64;
    }
    t[i + NL] = (u64)c;
  }
  for (int i = 0; i < NL; i++) {
    u64 m = t[i] * PINV;
    u128 c = 0;
    for (int j = 0; j < NL; j++) {
      c += (u128)m * P[j] + t[i + j];
      t[i + j] = (u64)c;
      c >>= 64;
    }
    for (int k = i + NL; c && k < 2 * NL + 1; k++) {
      c += t[k];
      t[k] = (u64)c;
      c >>= 64;
    }
  }
  u64 o[NL];
  for (int i = 0; i < NL; i++) o[i] = t[NL + i];
  if (t[2 * NL] || geq_p(o)) sub_p(o);
  memcpy(r->v, o, 8 * NL);
}
static void fp_sqr(fp* r, const fp* a) { fp_mul(r, a, a); }
static void fp_pow(fp* r, const fp* a, const u64* e, int nl) {
  fp acc = FP_ONE, base = *a;
  for (int i = nl * 64 - 1; i >= 0; i--) {
    fp_sqr(&acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(&acc, &acc, &base);
  }
  *r = acc;
}
static void sub_small(u64* o, const u64* a, u64 k) { /* o = a - k */
  u128 br = k;
  for (int i = 0; i < NL; i++) {
    u128 t = (u128)a[i] - br;
    o[i] = (u64)t;
    br = (t >> 64) & 1;
  }
}
static void shr(u64* a, int s) {
  for (int i = 0; i < NL; i++) a[i] = (a[i] >> s) | (i < NL - 1 ? a[i + 1] << (64 - s) : 0);
}
static void fp_inv(fp* r, const fp* a) { /* a^(p-2) */
  u64 e[NL];
  sub_small(e, P, 2);
  fp_pow(r, a, e, NL);
}
static int fp_sqrt(fp* r, const fp* a) { /* p = 3 mod 4: a^((p+1)/4) */
  u64 e[NL];
  u128 c = 1;
  for (int i = 0; i < NL; i++) {
    c += P[i];
    e[i] = (u64)c;
    c >>= 64;
  }
  shr(e, 2);
  fp t, s;
  fp_pow(&t, a, e, NL);
  fp_sqr(&s, &t);
  int ok = fp_eq(&s, a);   /* r may alias a */
  *r = t;
  return ok;
}
static int fp_legendre(const fp* a) {
  if (fp_is_zero(a)) return 0;
  u64 e[NL];
  sub_small(e, P, 1);
  shr(e, 1);
  fp t;
  fp_pow(&t, a, e, NL);
  return fp_eq(&t, &FP_ONE) ? 1 : -1;
}
static void fp_from_u64(fp* r, u64 x) {
  fp t;
  memset(&t, 0, sizeof t);
  t.v[0] = x;
  fp_mul(r, &t, &FP_R2);
}
static void fp_from_le(fp* r, const uint8_t* b) { /* canonical LE bytes -> Montgomery (no range check) */
  fp t;
  memcpy(t.v, b, FB);
  fp_mul(r, &t, &FP_R2);
}
static void fp_to_le(uint8_t* b, const fp* a) {
  fp one, t;
  memset(&one, 0, sizeof one);
  one.v[0] = 1;
  fp_mul(&t, a, &one);
  memcpy(b, t.v, FB);
}
static int le_lt_p(const uint8_t* b) {
  u64 t[NL];
  memcpy(t, b, FB);
  return !geq_p(t);
}

/* ------------------------------------------------------------------------------------------------ Fp2 */
typedef struct { fp a, b; } fp2;
static fp2 F2_ZERO, F2_ONE;
static void fp2_add(fp2* r, const fp2* x, const fp2* y) { fp_add(&r->a, &x->a, &y->a); fp_add(&r->b, &x->b, &y->b); }
static void fp2_sub(fp2* r, const fp2* x, const fp2* y) { fp_sub(&r->a, &x->a, &y->a); fp_sub(&r->b, &x->b, &y->b); }
static void fp2_neg(fp2* r, const fp2* x) { fp_neg(&r->a, &x->a); fp_neg(&r->b, &x->b); }
static void fp2_dbl(fp2* r, const fp2* x) { fp2_add(r, x, x); }
static void fp2_conj(fp2* r, const fp2* x) { r->a = x->a; fp_neg(&r->b, &x->b); }
static int fp2_is_zero(const fp2* x) { return fp_is_zero(&x->a) && fp_is_zero(&x->b); }
static int fp2_eq(const fp2* x, const fp2* y) { return fp_eq(&x->a, &y->a) && fp_eq(&x->b, &y->b); }
static void fp2_mul(fp2* r, const fp2* x, const fp2* y) {
  fp t0, t1, s0, s1, u;
  fp_mul(&t0, &x->a, &y->a);
  fp_mul(&t1, &x->b, &y->b);
  fp_add(&s0, &x->a, &x->b);
  fp_add(&s1, &y->a, &y->b);
  fp_mul(&u, &s0, &s1);
  fp_sub(&r->a, &t0, &t1);
  fp_sub(&u, &u, &t0);
  fp_sub(&r->b, &u, &t1);
}
static void fp2_sqr(fp2* r, const fp2* x) {
  fp s, d, m;
  fp_add(&s, &x->a, &x->b);
  fp_sub(&d, &x->a, &x->b);
  fp_mul(&m, &x->a, &x->b);
  fp_mul(&r->a, &s, &d);
  fp_dbl(&r->b, &m);
}
static void fp2_mul_fp(fp2* r, const fp2* x, const fp* s) { fp_mul(&r->a, &x->a, s); fp_mul(&r->b, &x->b, s); }
static void fp2_mul_xi(fp2* r, const fp2* x) { /* (a+bi)(1+i) */
  fp t;
  fp_sub(&t, &x->a, &x->b);
  fp_add(&r->b, &x->a, &x->b);
  r->a = t;
}
static void fp2_inv(fp2* r, const fp2* x) {
  fp n, t;
  fp_sqr(&n, &x->a);
  fp_sqr(&t, &x->b);
  fp_add(&n, &n, &t);
  fp_inv(&n, &n);
  fp_mul(&r->a, &x->a, &n);
  fp_mul(&t, &x->b, &n);
  fp_neg(&r->b, &t);
}
static void fp2_pow(fp2* r, const fp2* x, const u64* e, int nl) {
  fp2 acc = F2_ONE, base = *x;
  for (int i = nl * 64 - 1; i >= 0; i--) {
    fp2_sqr(&acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) fp2_mul(&acc, &acc, &base);
  }
  *r = acc;
}
static int fp2_sqrt(fp2* r, const fp2* x) {
  if (fp_is_zero(&x->b)) {
    fp s;
    if (fp_sqrt(&s, &x->a)) { r->a = s; r->b = FP_ZERO; return 1; }
    fp na;
    fp_neg(&na, &x->a);
    if (!fp_sqrt(&s, &na)) return 0;
    r->a = FP_ZERO;
    r->b = s;
    return 1;
  }
  fp n, t, u, half, two;
  fp_sqr(&n, &x->a);
  fp_sqr(&t, &x->b);
  fp_add(&n, &n, &t);
  if (!fp_sqrt(&n, &n)) return 0;
  fp_from_u64(&two, 2);
  fp_inv(&half, &two);
  fp_add(&t, &x->a, &n);
  fp_mul(&t, &t, &half);
  if (!fp_sqrt(&u, &t)) {
    fp_sub(&t, &x->a, &n);
    fp_mul(&t, &t, &half);
    if (!fp_sqrt(&u, &t)) return 0;
  }
  fp d;
  fp_dbl(&d, &u);
  fp_inv(&d, &d);
  r->a = u;
  fp_mul(&r->b, &x->b, &d);
  return 1;
}

/* ------------------------------------------------------------------------------------------------ Fp6, Fp12 */
typedef struct { fp2 c0, c1, c2; } fp6;      /* Fp2[v]/(v^3 - xi) */
typedef struct { fp6 c0, c1; } fp12;         /* Fp6[w]/(w^2 - v)  */
static fp2 FROB[4][6];                        /* FROB[n][k] = xi^(k (p^n - 1)/6), n = 1..3 */
static fp2 TWIST_B;                           /* b/xi (BN254, D-type twist) or b xi (BLS12-381, M-type twist) */

static void fp6_add(fp6* r, const fp6* x, const fp6* y) { fp2_add(&r->c0, &x->c0, &y->c0); fp2_add(&r->c1, &x->c1, &y->c1); fp2_add(&r->c2, &x->c2, &y->c2); }
static void fp6_sub(fp6* r, const fp6* x, const fp6* y) { fp2_sub(&r->c0, &x->c0, &y->c0); fp2_sub(&r->c1, &x->c1, &y->c1); fp2_sub(&r->c2, &x->c2, &y->c2); }
static void fp6_neg(fp6* r, const fp6* x) { fp2_neg(&r->c0, &x->c0); fp2_neg(&r->c1, &x->c1); fp2_neg(&r->c2, &x->c2); }
static void fp6_mul_v(fp6* r, const fp6* x) {
  fp2 t;
  fp2_mul_xi(&t, &x->c2);
  r->c2 = x->c1;
  r->c1 = x->c0;
  r->c0 = t;
}
static void fp6_mul(fp6* r, const fp6* x, const fp6* y) { /* schoolbook with Karatsuba cross terms */
  fp2 t0, t1, t2, a, b, s, r0, r1, r2;
  fp2_mul(&t0, &x->c0, &y->c0);
  fp2_mul(&t1, &x->c1, &y->c1);
  fp2_mul(&t2, &x->c2, &y->c2);
  fp2_add(&a, &x->c1, &x->c2); fp2_add(&b, &y->c1, &y->c2); fp2_mul(&s, &a, &b);
  fp2_sub(&s, &s, &t1); fp2_sub(&s, &s, &t2); fp2_mul_xi(&s, &s); fp2_add(&r0, &s, &t0);
  fp2_add(&a, &x->c0, &x->c1); fp2_add(&b, &y->c0, &y->c1); fp2_mul(&s, &a, &b);
  fp2_sub(&s, &s, &t0); fp2_sub(&s, &s, &t1); fp2_mul_xi(&a, &t2); fp2_add(&r1, &s, &a);
  fp2_add(&a, &x->c0, &x->c2); fp2_add(&b, &y->c0, &y->c2); fp2_mul(&s, &a, &b);
  fp2_sub(&s, &s, &t0); fp2_sub(&s, &s, &t2); fp2_add(&r2, &s, &t1);
  r->c0 = r0; r->c1 = r1; r->c2 = r2;
}
static void fp6_sqr(fp6* r, const fp6* x) { fp6_mul(r, x, x); }
static void fp6_mul_fp2(fp6* r, const fp6* x, const fp2* s) { fp2_mul(&r->c0, &x->c0, s); fp2_mul(&r->c1, &x->c1, s); fp2_mul(&r->c2, &x->c2, s); }
static void fp6_inv(fp6* r, const fp6* x) {
  fp2 A, B, C, t, F;
  fp2_sqr(&A, &x->c0); fp2_mul(&t, &x->c1, &x->c2); fp2_mul_xi(&t, &t); fp2_sub(&A, &A, &t);
  fp2_sqr(&B, &x->c2); fp2_mul_xi(&B, &B); fp2_mul(&t, &x->c0, &x->c1); fp2_sub(&B, &B, &t);
  fp2_sqr(&C, &x->c1); fp2_mul(&t, &x->c0, &x->c2); fp2_sub(&C, &C, &t);
  fp2 u, v;
  fp2_mul(&u, &x->c2, &B); fp2_mul(&v, &x->c1, &C); fp2_add(&u, &u, &v); fp2_mul_xi(&u, &u);
  fp2_mul(&F, &x->c0, &A); fp2_add(&F, &F, &u);
  fp2_inv(&F, &F);
  fp2_mul(&r->c0, &A, &F); fp2_mul(&r->c1, &B, &F); fp2_mul(&r->c2, &C, &F);
}
static void fp12_one(fp12* r) { memset(r, 0, sizeof *r); r->c0.c0 = F2_ONE; }
static int fp12_eq(const fp12* x, const fp12* y) {
  return fp2_eq(&x->c0.c0, &y->c0.c0) && fp2_eq(&x->c0.c1, &y->c0.c1) && fp2_eq(&x->c0.c2, &y->c0.c2) &&
         fp2_eq(&x->c1.c0, &y->c1.c0) && fp2_eq(&x->c1.c1, &y->c1.c1) && fp2_eq(&x->c1.c2, &y->c1.c2);
}
static void fp12_mul(fp12* r, const fp12* x, const fp12* y) {
  fp6 t0, t1, a, b, s;
  fp6_mul(&t0, &x->c0, &y->c0);
  fp6_mul(&t1, &x->c1, &y->c1);
  fp6_add(&a, &x->c0, &x->c1);
  fp6_add(&b, &y->c0, &y->c1);
  fp6_mul(&s, &a, &b);
  fp6_sub(&s, &s, &t0);
  fp6_sub(&r->c1, &s, &t1);
  fp6_mul_v(&t1, &t1);
  fp6_add(&r->c0, &t0, &t1);
}
static void fp12_sqr(fp12* r, const fp12* x) {
  fp6 t, a, b, s;
  fp6_mul(&t, &x->c0, &x->c1);
  fp6_add(&a, &x->c0, &x->c1);
  fp6_mul_v(&b, &x->c1);
  fp6_add(&b, &b, &x->c0);
  fp6_mul(&s, &a, &b);
  fp6_sub(&s, &s, &t);
  fp6_mul_v(&a, &t);
  fp6_sub(&r->c0, &s, &a);
  fp6_add(&r->c1, &t, &t);
}
static void fp12_conj(fp12* r, const fp12* x) { r->c0 = x->c0; fp6_neg(&r->c1, &x->c1); }
static void fp12_inv(fp12* r, const fp12* x) {
  fp6 t0, t1;
  fp6_sqr(&t0, &x->c0);
  fp6_sqr(&t1, &x->c1);
  fp6_mul_v(&t1, &t1);
  fp6_sub(&t0, &t0, &t1);
  fp6_inv(&t1, &t0);
  fp6_mul(&r->c0, &x->c0, &t1);
  fp6_mul(&t0, &x->c1, &t1);
  fp6_neg(&r->c1, &t0);
}
static void fp12_frob(fp12* r, const fp12* x, int n) { /* coefficient of w^k, k = 2i + j for v^i w^j */
  fp2* o[6] = {&r->c0.c0, &r->c1.c0, &r->c0.c1, &r->c1.c1, &r->c0.c2, &r->c1.c2};
  const fp2* in[6] = {&x->c0.c0, &x->c1.c0, &x->c0.c1, &x->c1.c1, &x->c0.c2, &x->c1.c2};
  for (int k = 0; k < 6; k++) {
    fp2 t = *in[k];
    if (n & 1) fp2_conj(&t, &t);
    if (k) fp2_mul(&t, &t, &FROB[n][k]);
    *o[k] = t;
  }
}
/* f * line.  A line through points of the twist, evaluated at P = (x_P, y_P), is  a y_P  -  (...) x_P  +  c  with a, b, c in Fp2 (ml_dbl / ml_add pass
   a y_P, b x_P and c).  D-type twist (BN254; untwist (x, y) -> (x w^2, y w^3)):  a + b w + c w^3.  M-type twist (BLS12-381; untwist (x, y) ->
   (x / w^2, y / w^3)): the same line times w^3 -- an element of Fp4 = Fp2(w^3), which the final exponentiation maps to 1 --  c + b w^2 + a w^3. */
static void fp12_mul_line(fp12* f, const fp2* a, const fp2* b, const fp2* c) {
  fp12 l;
  memset(&l, 0, sizeof l);
#ifdef ELPO_BLS12_381
  l.c0.c0 = *c;
  l.c0.c1 = *b;   /* w^2 = v */
  l.c1.c1 = *a;   /* w^3 = v w */
#else
  l.c0.c0 = *a;
  l.c1.c0 = *b;
  l.c1.c1 = *c;
#endif
  fp12_mul(f, f, &l);
}
#ifdef ELPO_BLS12_381
static void fp12_pow_limbs(fp12* r, const fp12* x, const u64* e, int nl) {   /* plain square-and-multiply, e != 0 */
  fp12 acc;
  int started = 0;
  for (int i = nl * 64 - 1; i >= 0; i--) {
    if (started) fp12_sqr(&acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) {
      if (started) fp12_mul(&acc, &acc, x); else { acc = *x; started = 1; }
    }
  }
  *r = acc;
}
#endif
static void fp12_pow_u64(fp12* r, const fp12* x, u64 e) {
  fp12 acc = *x;
  int top = 63;
  while (!((e >> top) & 1)) top--;
  for (int i = top - 1; i >= 0; i--) {
    fp12_sqr(&acc, &acc);
    if ((e >> i) & 1) fp12_mul(&acc, &acc, x);
  }
  *r = acc;
}

/* ------------------------------------------------------------------------------------------------ groups */
typedef struct { fp x, y; int inf; } g1a;
typedef struct { fp X, Y, Z; } g1j;        /* Z = 0: infinity */
typedef struct { fp2 x, y; int inf; } g2a;
typedef struct { fp2 X, Y, Z; } g2j;
static fp CURVE_B;

/* The group law is written once with macros over the field type */
#define DEFINE_GROUP(G, F, FZERO, FONE)                                                                         \
  static int G##_is_inf(const G##j* p) { return F##_is_zero(&p->Z); }                                           \
  static void G##_set_inf(G##j* p) { p->X = FONE; p->Y = FONE; p->Z = FZERO; }                                  \
  static void G##_from_aff(G##j* r, const G##a* p) {                                                            \
    if (p->inf) { G##_set_inf(r); return; }                                                                     \
    r->X = p->x; r->Y = p->y; r->Z = FONE;                                                                      \
  }                                                                                                             \
  static void G##_dbl(G##j* r, const G##j* p) { /* dbl-2009-l */                                                \
    if (G##_is_inf(p)) { *r = *p; return; }                                                                     \
    F A, B, C, D, E, Fq, t, X3, Y3, Z3;                                                                         \
    F##_sqr(&A, &p->X); F##_sqr(&B, &p->Y); F##_sqr(&C, &B);                                                    \
    F##_add(&t, &p->X, &B); F##_sqr(&t, &t); F##_sub(&t, &t, &A); F##_sub(&t, &t, &C); F##_dbl(&D, &t);         \
    F##_dbl(&E, &A); F##_add(&E, &E, &A); F##_sqr(&Fq, &E);                                                     \
    F##_dbl(&t, &D); F##_sub(&X3, &Fq, &t);                                                                     \
    F##_sub(&t, &D, &X3); F##_mul(&Y3, &E, &t); F##_dbl(&C, &C); F##_dbl(&C, &C); F##_dbl(&C, &C);              \
    F##_sub(&Y3, &Y3, &C);                                                                                      \
    F##_mul(&Z3, &p->Y, &p->Z); F##_dbl(&Z3, &Z3);                                                              \
    r->X = X3; r->Y = Y3; r->Z = Z3;                                                                            \
  }                                                                                                             \
  static void G##_add(G##j* r, const G##j* p, const G##j* q) { /* add-2007-bl */                                \
    if (G##_is_inf(p)) { *r = *q; return; }                                                                     \
    if (G##_is_inf(q)) { *r = *p; return; }                                                                     \
    F Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, X3, Y3, Z3;                                                \
    F##_sqr(&Z1Z1, &p->Z); F##_sqr(&Z2Z2, &q->Z);                                                               \
    F##_mul(&U1, &p->X, &Z2Z2); F##_mul(&U2, &q->X, &Z1Z1);                                                     \
    F##_mul(&S1, &p->Y, &q->Z); F##_mul(&S1, &S1, &Z2Z2);                                                       \
    F##_mul(&S2, &q->Y, &p->Z); F##_mul(&S2, &S2, &Z1Z1);                                                       \
    F##_sub(&H, &U2, &U1); F##_sub(&rr, &S2, &S1);                                                              \
    if (F##_is_zero(&H)) {                                                                                      \
      if (F##_is_zero(&rr)) { G##_dbl(r, p); } else { G##_set_inf(r); }                                         \
      return;                                                                                                   \
    }                                                                                                           \
    F##_dbl(&rr, &rr);                                                                                          \
    F##_dbl(&I, &H); F##_sqr(&I, &I); F##_mul(&J, &H, &I); F##_mul(&V, &U1, &I);                                \
    F##_sqr(&X3, &rr); F##_sub(&X3, &X3, &J); F##_dbl(&t, &V); F##_sub(&X3, &X3, &t);                           \
    F##_sub(&t, &V, &X3); F##_mul(&Y3, &rr, &t); F##_mul(&t, &S1, &J); F##_dbl(&t, &t); F##_sub(&Y3, &Y3, &t);  \
    F##_add(&Z3, &p->Z, &q->Z); F##_sqr(&Z3, &Z3); F##_sub(&Z3, &Z3, &Z1Z1); F##_sub(&Z3, &Z3, &Z2Z2);          \
    F##_mul(&Z3, &Z3, &H);                                                                                      \
    r->X = X3; r->Y = Y3; r->Z = Z3;                                                                            \
  }                                                                                                             \
  static void G##_neg(G##j* r, const G##j* p) { r->X = p->X; r->Z = p->Z; F##_neg(&r->Y, &p->Y); }              \
  static void G##_to_aff(G##a* r, const G##j* p) {                                                              \
    if (G##_is_inf(p)) { memset(r, 0, sizeof *r); r->inf = 1; return; }                                         \
    F zi, zi2;                                                                                                  \
    F##_inv(&zi, &p->Z); F##_sqr(&zi2, &zi);                                                                    \
    F##_mul(&r->x, &p->X, &zi2); F##_mul(&zi2, &zi2, &zi); F##_mul(&r->y, &p->Y, &zi2);                         \
    r->inf = 0;                                                                                                 \
  }                                                                                                             \
  /* G::mul: width-5 NAF, the shape of mcl's generic window method (no GLV/GLS) */                              \
  static void G##_mul(G##j* r, const G##a* p, const u64* k) {                                                   \
    signed char naf[260];                                                                                       \
    u64 e[5] = {k[0], k[1], k[2], k[3], 0};                                                                     \
    int n = 0;                                                                                                  \
    while (e[0] | e[1] | e[2] | e[3] | e[4]) {                                                                  \
      int d = 0;                                                                                                \
      if (e[0] & 1) {                                                                                           \
        d = (int)(e[0] & 31);                                                                                   \
        if (d >= 16) d -= 32;                                                                                   \
        if (d > 0) { u128 br = (u64)d; for (int i = 0; i < 5; i++) { u128 t = (u128)e[i] - br; e[i] = (u64)t; br = (t >> 64) & 1; } } \
        else { u128 c = (u64)(-d); for (int i = 0; i < 5; i++) { c += e[i]; e[i] = (u64)c; c >>= 64; } }        \
      }                                                                                                         \
      naf[n++] = (signed char)d;                                                                                \
      for (int i = 0; i < 5; i++) e[i] = (e[i] >> 1) | (i < 4 ? e[i + 1] << 63 : 0);                            \
    }                                                                                                           \
    G##j tbl[8], p2, acc, t;                                                                                    \
    G##_from_aff(&tbl[0], p);                                                                                   \
    G##_dbl(&p2, &tbl[0]);                                                                                      \
    for (int i = 1; i < 8; i++) G##_add(&tbl[i], &tbl[i - 1], &p2);                                             \
    G##_set_inf(&acc);                                                                                          \
    for (int i = n - 1; i >= 0; i--) {                                                                          \
      G##_dbl(&acc, &acc);                                                                                      \
      if (naf[i] > 0) G##_add(&acc, &acc, &tbl[naf[i] >> 1]);                                                   \
      else if (naf[i] < 0) { G##_neg(&t, &tbl[(-naf[i]) >> 1]); G##_add(&acc, &acc, &t); }                      \
    }                                                                                                           \
    *r = acc;                                                                                                   \
  }

DEFINE_GROUP(g1, fp, FP_ZERO, FP_ONE)
DEFINE_GROUP(g2, fp2, F2_ZERO, F2_ONE)

/* G1::mul as mcl evaluates it (every G1::mul call site of the reference goes through this one).
   BN254: E(Fp) = G1, any correct method gives [k]P: the window method above.
   BLS12-381: mcl's G1::mul is the GLV method with psi(x, y) = (beta x, y) = [L] on G1, L = z^2 - 1: k mod r is split by plain division,
   k = a + b L (0 <= a < L), and the result is [a]P + [b]psi(P).  Inside G1 that is [k]P.  For a point of E(Fp) outside G1 -- mcl's default
   accepts one on deserialisation -- it is not (psi fixes the order-3 points (0, +-2): that component is multiplied by a + b), and the
   reference's verdict on such inputs follows this split: pinned by 20 crafted proofs / requests in tests/golden/bls12_381_oracle_{edge,requests}.json. */
#ifdef ELPO_BLS12_381
static const u64 GLV_L[2] = {0x00000000ffffffffull, 0xac45a4010001a402ull};                 /* z^2 - 1 */
static const u64 GLV_BETA[NL] = {0x8bfd00000000aaacull, 0x409427eb4f49fffdull, 0x897d29650fb85f9bull, 0xaa0d857d89759ad4ull, 0xec02408663d4de85ull,
                                 0x1a0111ea397fe699ull};
static void g1_mul_ref(g1j* r, const g1a* p, const u64* k) {
  u64 e[4] = {k[0], k[1], k[2], k[3]};
  for (int rep = 0; rep < 3; rep++) {                 /* k < 2^256 < 3 r */
    int ge = 1;
    for (int i = 3; i >= 0; i--) { if (e[i] > RORD[i]) break; if (e[i] < RORD[i]) { ge = 0; break; } }
    if (!ge) break;
    u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 t = (u128)e[i] - RORD[i] - br; e[i] = (u64)t; br = (t >> 64) & 1; }
  }
  /* b, a = divmod(e, L): schoolbook binary long division (a < L < 2^128 throughout) */
  u64 q[4] = {0, 0, 0, 0};
  u128 rem = 0;
  const u128 Lv = ((u128)GLV_L[1] << 64) | GLV_L[0];
  for (int i = 255; i >= 0; i--) {
    int top = (int)(rem >> 127);
    rem = (rem << 1) | ((e[i >> 6] >> (i & 63)) & 1);
    if (top || rem >= Lv) { rem -= Lv; q[i >> 6] |= 1ull << (i & 63); }
  }
  u64 a[4] = {(u64)rem, (u64)(rem >> 64), 0, 0};
  g1a psi = *p;
  fp beta;
  fp_from_le(&beta, (const uint8_t*)GLV_BETA);
  fp_mul(&psi.x, &psi.x, &beta);
  g1j ra, rb;
  g1_mul(&ra, p, a);
  g1_mul(&rb, &psi, q);
  if (p->inf) { g1_set_inf(r); return; }
  g1_add(r, &ra, &rb);
}
#else
static void g1_mul_ref(g1j* r, const g1a* p, const u64* k) { g1_mul(r, p, k); }
#endif

static int g1_on_curve(const g1a* p) {
  if (p->inf) return 1;
  fp l, r;
  fp_sqr(&l, &p->y);
  fp_sqr(&r, &p->x); fp_mul(&r, &r, &p->x); fp_add(&r, &r, &CURVE_B);
  return fp_eq(&l, &r);
}
static int g2_on_curve(const g2a* p) {
  if (p->inf) return 1;
  fp2 l, r;
  fp2_sqr(&l, &p->y);
  fp2_sqr(&r, &p->x); fp2_mul(&r, &r, &p->x); fp2_add(&r, &r, &TWIST_B);
  return fp2_eq(&l, &r);
}

/* ------------------------------------------------------------------------------------------------ pairing */
/* pi_p on the twist: (conj(x) gamma_{1,2}, conj(y) gamma_{1,3}); pi_p^2: (x gamma_{2,2}, y gamma_{2,3}) */
static void g2_frob(g2a* r, const g2a* q, int n) {
  fp2 x = q->x, y = q->y;
  if (n & 1) { fp2_conj(&x, &x); fp2_conj(&y, &y); }
  fp2_mul(&r->x, &x, &FROB[n][2]);
  fp2_mul(&r->y, &y, &FROB[n][3]);
  r->inf = q->inf;
}
/* Jacobian tangent:  l = Z3 Z^2 y_P - 3 X^2 Z^2 x_P w + (3 X^3 - 2 Y^2) w^3 ;  T <- 2T */
static void ml_dbl(fp12* f, g2j* T, const g1a* p) {
  fp2 X2, Y2, Z2, t, a, b, c, X3c;
  fp2_sqr(&X2, &T->X); fp2_sqr(&Y2, &T->Y); fp2_sqr(&Z2, &T->Z);
  fp2_mul(&X3c, &X2, &T->X);
  fp2_dbl(&c, &X3c); fp2_add(&c, &c, &X3c);          /* 3 X^3 */
  fp2_dbl(&t, &Y2); fp2_sub(&c, &c, &t);             /* - 2 Y^2 */
  fp2_dbl(&b, &X2); fp2_add(&b, &b, &X2); fp2_mul(&b, &b, &Z2); fp2_neg(&b, &b);   /* -3 X^2 Z^2 */
  g2j T2;
  g2_dbl(&T2, T);
  fp2_mul(&a, &T2.Z, &Z2);                            /* Z3 Z^2 */
  fp2_mul_fp(&a, &a, &p->y);
  fp2_mul_fp(&b, &b, &p->x);
  fp12_mul_line(f, &a, &b, &c);
  *T = T2;
}
/* chord through T and affine Q:  l = Z3 y_P - r x_P w + (r x_Q - Z3 y_Q) w^3,  r = y_Q Z^3 - Y, Z3 = Z (x_Q Z^2 - X) */
static void ml_add(fp12* f, g2j* T, const g2a* q, const g1a* p) {
  fp2 Z2, Z3, H, rr, Z3n, a, b, c, t;
  fp2_sqr(&Z2, &T->Z);
  fp2_mul(&Z3, &Z2, &T->Z);
  fp2_mul(&H, &q->x, &Z2); fp2_sub(&H, &H, &T->X);
  fp2_mul(&rr, &q->y, &Z3); fp2_sub(&rr, &rr, &T->Y);
  fp2_mul(&Z3n, &T->Z, &H);
  fp2_mul(&c, &rr, &q->x); fp2_mul(&t, &Z3n, &q->y); fp2_sub(&c, &c, &t);
  fp2_mul_fp(&a, &Z3n, &p->y);
  fp2_neg(&b, &rr); fp2_mul_fp(&b, &b, &p->x);
  fp12_mul_line(f, &a, &b, &c);
  g2j Q;
  g2_from_aff(&Q, q);
  g2_add(T, T, &Q);
}
#ifdef ELPO_BLS12_381
static void miller_loop(fp12* f, const g1a* p, const g2a* q) {
  fp12_one(f);
  if (p->inf || q->inf) return;
  /* optimal ate on BLS12: f_{|z|,Q}(P), conjugated because z < 0; no Frobenius steps */
  int top = 63;
  while (!((ZABS >> top) & 1)) top--;
  g2j T;
  g2_from_aff(&T, q);
  for (int i = top - 1; i >= 0; i--) {
    fp12_sqr(f, f);
    ml_dbl(f, &T, p);
    if ((ZABS >> i) & 1) ml_add(f, &T, q, p);
  }
  fp12_conj(f, f);
}
static void fp12_pow_z(fp12* r, const fp12* x) { /* x^z, z = -|z|, x unitary after the easy part */
  fp12_pow_u64(r, x, ZABS);
  fp12_conj(r, r);
}
/* hard part by the chain of Hayashida, Hayasaka and Teruya: (p^4-p^2+1)/r = ((z-1)^2/3) (z+p) (z^2+p^2-1) + 1 */
static void final_exp_hard_chain(fp12* r, const fp12* f) {
  fp12 a, b, c, t;
  fp12_pow_u64(&a, f, ZM1D3_ABS); fp12_conj(&a, &a);                      /* f^((z-1)/3), (z-1)/3 < 0 */
  fp12_pow_z(&t, &a); fp12_conj(&b, &a); fp12_mul(&a, &t, &b);            /* ^(z-1) */
  fp12_pow_z(&t, &a); fp12_frob(&b, &a, 1); fp12_mul(&b, &b, &t);         /* ^(z+p) */
  fp12_pow_z(&t, &b); fp12_pow_z(&t, &t); fp12_frob(&c, &b, 2); fp12_mul(&c, &c, &t);
  fp12_conj(&t, &b); fp12_mul(&c, &c, &t);                                /* ^(z^2+p^2-1) */
  fp12_mul(r, &c, f);
}
static void final_exp_easy(fp12* f, const fp12* fin) {
  fp12 t0, t1;
  fp12_inv(&t0, fin);
  fp12_conj(&t1, fin);
  fp12_mul(f, &t1, &t0);
  fp12_frob(&t0, f, 2);
  fp12_mul(f, &t0, f);
}
static void final_exp(fp12* r, const fp12* fin) {
  fp12 f;
  final_exp_easy(&f, fin);
  final_exp_hard_chain(r, &f);
}
/* self-check: the chain equals the plain power by the integer (p^4-p^2+1)/r */
int elpo_selftest_final_exp(const uint8_t* P_, const uint8_t* Q_);
#else
static void miller_loop(fp12* f, const g1a* p, const g2a* q) {
  fp12_one(f);
  if (p->inf || q->inf) return;
  /* s = |6z + 2| = 6|z| - 2, plain binary expansion */
  u128 s = (u128)6 * ZABS - 2;
  int top = 127;
  while (!((s >> top) & 1)) top--;
  g2j T;
  g2_from_aff(&T, q);
  for (int i = top - 1; i >= 0; i--) {
    fp12_sqr(f, f);
    ml_dbl(f, &T, p);
    if ((s >> i) & 1) ml_add(f, &T, q, p);
  }
  /* z < 0: f <- conj(f), T <- -T ; then the two Frobenius line steps of the BN optimal ate pairing */
  fp12_conj(f, f);
  g2_neg(&T, &T);
  g2a q1, q2;
  g2_frob(&q1, q, 1);
  g2_frob(&q2, q, 2);
  fp2_neg(&q2.y, &q2.y);
  ml_add(f, &T, &q1, p);
  ml_add(f, &T, &q2, p);
}
static void fp12_pow_z(fp12* r, const fp12* x) { /* x^z, z = -|z|, x unitary after the easy part */
  fp12_pow_u64(r, x, ZABS);
  fp12_conj(r, r);
}
static void final_exp(fp12* r, const fp12* fin) {
  fp12 f, t0, t1;
  fp12_inv(&t0, fin);
  fp12_conj(&t1, fin);
  fp12_mul(&f, &t1, &t0);
  fp12_frob(&t0, &f, 2);
  fp12_mul(&f, &t0, &f);
  /* hard part (p^4-p^2+1)/r = p^3 + (6z^2+1) p^2 + (-36z^3-18z^2-12z+1) p + (-36z^3-30z^2-18z-2)  [Devegili et al.] */
  fp12 fz, fz2, fz3, y0, y1, y2, y3, y4, y5, y6, T0, T1, t;
  fp12_pow_z(&fz, &f); fp12_pow_z(&fz2, &fz); fp12_pow_z(&fz3, &fz2);
  fp12_frob(&y0, &f, 1); fp12_frob(&t, &f, 2); fp12_mul(&y0, &y0, &t); fp12_frob(&t, &f, 3); fp12_mul(&y0, &y0, &t);
  fp12_conj(&y1, &f);
  fp12_frob(&y2, &fz2, 2);
  fp12_frob(&y3, &fz, 1); fp12_conj(&y3, &y3);
  fp12_frob(&t, &fz2, 1); fp12_mul(&y4, &fz, &t); fp12_conj(&y4, &y4);
  fp12_conj(&y5, &fz2);
  fp12_frob(&t, &fz3, 1); fp12_mul(&y6, &fz3, &t); fp12_conj(&y6, &y6);
  fp12_sqr(&T0, &y6); fp12_mul(&T0, &T0, &y4); fp12_mul(&T0, &T0, &y5);
  fp12_mul(&T1, &y3, &y5); fp12_mul(&T1, &T1, &T0);
  fp12_mul(&T0, &T0, &y2);
  fp12_sqr(&T1, &T1); fp12_mul(&T1, &T1, &T0); fp12_sqr(&T1, &T1);
  fp12_mul(&T0, &T1, &y1); fp12_mul(&T1, &T1, &y0);
  fp12_sqr(&T0, &T0); fp12_mul(r, &T0, &T1);
}
#endif
static void pairing(fp12* r, const g1a* p, const g2a* q) {
  fp12 f;
  miller_loop(&f, p, q);
  final_exp(r, &f);
}

/* ------------------------------------------------------------------------------------------------ SHA-256 */
typedef struct { uint32_t h[8]; uint8_t buf[64]; u64 len; } sha256_t;
static const uint32_t SK[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha_block(sha256_t* s, const uint8_t* b) {
  uint32_t w[64], a, bb, c, d, e, f, g, h;
  for (int i = 0; i < 16; i++) w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  a = s->h[0]; bb = s->h[1]; c = s->h[2]; d = s->h[3]; e = s->h[4]; f = s->h[5]; g = s->h[6]; h = s->h[7];
  for (int i = 0; i < 64; i++) {
    uint32_t t1 = h + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + SK[i] + w[i];
    uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
  }
  s->h[0] += a; s->h[1] += bb; s->h[2] += c; s->h[3] += d; s->h[4] += e; s->h[5] += f; s->h[6] += g; s->h[7] += h;
}
static void sha_init(sha256_t* s) {
  static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(s->h, iv, 32);
  s->len = 0;
}
static void sha_update(sha256_t* s, const uint8_t* p, size_t n) {
  while (n--) {
    s->buf[s->len++ & 63] = *p++;
    if ((s->len & 63) == 0) sha_block(s, s->buf);
  }
}
static void sha_final(sha256_t* s, uint8_t out[32]) {
  u64 bits = s->len * 8;
  uint8_t pad = 0x80, z = 0, lb[8];
  sha_update(s, &pad, 1);
  while ((s->len & 63) != 56) sha_update(s, &z, 1);
  for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
  sha_update(s, lb, 8);
  for (int i = 0; i < 8; i++) { out[4 * i] = s->h[i] >> 24; out[4 * i + 1] = s->h[i] >> 16; out[4 * i + 2] = s->h[i] >> 8; out[4 * i + 3] = s->h[i]; }
}
static void sha_update_hex(sha256_t* s, const uint8_t* p, size_t n) { /* serializeToHexStr: lowercase hex */
  static const char hx[] = "0123456789abcdef";
  for (size_t i = 0; i < n; i++) {
    uint8_t c[2] = {(uint8_t)hx[p[i] >> 4], (uint8_t)hx[p[i] & 15]};
    sha_update(s, c, 2);
  }
}

/* ------------------------------------------------------------------------------------------------ mcl conventions */
static int u256_geq(const u64* a, const u64* m) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > m[i]) return 1;
    if (a[i] < m[i]) return 0;
  }
  return 1;
}
/* Fr::setHashOf (and, on BN254, Fp::setHashOf): LE(SHA-256(m)) masked to the bit length of the modulus; if still >= modulus, to one bit less */
static void set_hash_of(u64 out[4], const uint8_t* msg, size_t len, const u64* mod) {
  sha256_t s;
  uint8_t d[32];
  sha_init(&s); sha_update(&s, msg, len); sha_final(&s, d);
  memcpy(out, d, 32);
  const int bits = RBITS;               /* on BN254 p and r have the same bit length */
  out[3] &= (1ull << (bits - 192)) - 1;
  if (u256_geq(out, mod)) out[3] &= (1ull << (bits - 193)) - 1;
}
static void g1_ser(uint8_t out[FB], const g1a* p) {
  if (p->inf) { memset(out, 0, FB); return; }
  uint8_t y[FB];
  fp_to_le(out, &p->x); fp_to_le(y, &p->y);
  if (y[0] & 1) out[FB - 1] |= 0x80;
}
static void g2_ser(uint8_t out[2 * FB], const g2a* p) {
  if (p->inf) { memset(out, 0, 2 * FB); return; }
  uint8_t y[FB];
  fp_to_le(out, &p->x.a); fp_to_le(out + FB, &p->x.b); fp_to_le(y, &p->y.a);
  if (y[0] & 1) out[2 * FB - 1] |= 0x80;
}
static int g1_de(g1a* p, const uint8_t in[FB]) {
  uint8_t t[FB];
  memcpy(t, in, FB);
  int any = 0;
  for (int i = 0; i < FB; i++) any |= t[i];
  memset(p, 0, sizeof *p);
  if (!any) { p->inf = 1; return 1; }
  int odd = t[FB - 1] >> 7;
  t[FB - 1] &= 0x7f;
  if (!le_lt_p(t)) return 0;
  fp_from_le(&p->x, t);
  fp rhs;
  fp_sqr(&rhs, &p->x); fp_mul(&rhs, &rhs, &p->x); fp_add(&rhs, &rhs, &CURVE_B);
  if (!fp_sqrt(&p->y, &rhs)) return 0;
  uint8_t y[FB];
  fp_to_le(y, &p->y);
  if ((y[0] & 1) != odd) fp_neg(&p->y, &p->y);
  return 1;
}
static int g2_de(g2a* p, const uint8_t in[2 * FB]) {
  uint8_t t[2 * FB];
  memcpy(t, in, 2 * FB);
  int any = 0;
  for (int i = 0; i < 2 * FB; i++) any |= t[i];
  memset(p, 0, sizeof *p);
  if (!any) { p->inf = 1; return 1; }
  int odd = t[2 * FB - 1] >> 7;
  t[2 * FB - 1] &= 0x7f;
  if (!le_lt_p(t) || !le_lt_p(t + FB)) return 0;
  fp_from_le(&p->x.a, t); fp_from_le(&p->x.b, t + FB);
  fp2 rhs;
  fp2_sqr(&rhs, &p->x); fp2_mul(&rhs, &rhs, &p->x); fp2_add(&rhs, &rhs, &TWIST_B);
  if (!fp2_sqrt(&p->y, &rhs)) return 0;
  uint8_t y[FB];
  fp_to_le(y, &p->y.a);
  if ((y[0] & 1) != odd) fp2_neg(&p->y, &p->y);
  return 1;
}
/* ------------------------------------------------------------------------------------------------ SHA-512 (Fp::setHashOf of a field wider than 256 bits) */
#ifdef ELPO_BLS12_381
static const u64 SK512[80] = {
    0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull, 0x59f111f1b605d019ull,
    0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull, 0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull,
    0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull, 0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull,
    0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull, 0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull,
    0x983e5152ee66dfabull, 0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
    0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull, 0x53380d139d95b3dfull,
    0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull, 0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull,
    0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull, 0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull,
    0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull, 0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull,
    0x5b9cca4f7763e373ull, 0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
    0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull, 0xd186b8c721c0c207ull,
    0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull, 0x113f9804bef90daeull, 0x1b710b35131c471bull,
    0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull, 0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull,
    0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};
#define ROR64(x, n) (((x) >> (n)) | ((x) << (64 - (n))))
static void sha512_block(u64 h[8], const uint8_t* b) {
  u64 w[80], v[8];
  for (int i = 0; i < 16; i++) {
    w[i] = 0;
    for (int j = 0; j < 8; j++) w[i] = (w[i] << 8) | b[8 * i + j];
  }
  for (int i = 16; i < 80; i++) {
    u64 s0 = ROR64(w[i - 15], 1) ^ ROR64(w[i - 15], 8) ^ (w[i - 15] >> 7), s1 = ROR64(w[i - 2], 19) ^ ROR64(w[i - 2], 61) ^ (w[i - 2] >> 6);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  memcpy(v, h, 64);
  for (int i = 0; i < 80; i++) {
    u64 t1 = v[7] + (ROR64(v[4], 14) ^ ROR64(v[4], 18) ^ ROR64(v[4], 41)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + SK512[i] + w[i];
    u64 t2 = (ROR64(v[0], 28) ^ ROR64(v[0], 34) ^ ROR64(v[0], 39)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
    v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
  }
  for (int i = 0; i < 8; i++) h[i] += v[i];
}
static void sha512(uint8_t out[64], const uint8_t* msg, size_t len) {
  u64 h[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
              0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
  size_t off = 0;
  for (; off + 128 <= len; off += 128) sha512_block(h, msg + off);
  uint8_t tail[256];
  size_t rem = len - off, tl = rem + 17 <= 128 ? 128 : 256;
  memset(tail, 0, sizeof tail);
  memcpy(tail, msg + off, rem);
  tail[rem] = 0x80;
  u64 bits = (u64)len * 8;                       /* the upper 64 bits of the 128-bit length field stay 0 */
  for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
  for (size_t o = 0; o < tl; o += 128) sha512_block(h, tail + o);
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(h[i] >> (56 - 8 * j));
}
#endif

/* Fp::setHashOf: the digest (SHA-256 for a modulus of at most 256 bits, SHA-512 for the 381-bit field of BLS12-381 -- mcl picks the hash by the
   bit size of the modulus), its first FB bytes as an LE integer, masked to PBITS bits and to one bit less if still >= p. */
static void fp_set_hash_of(fp* r, const uint8_t* msg, size_t len) {
  u64 t[NL];
#ifdef ELPO_BLS12_381
  uint8_t d[64];
  sha512(d, msg, len);
#else
  uint8_t d[32];
  sha256_t s;
  sha_init(&s); sha_update(&s, msg, len); sha_final(&s, d);
#endif
  memcpy(t, d, FB);
  t[NL - 1] &= (1ull << (PBITS - 64 * (NL - 1))) - 1;
  if (geq_p(t)) t[NL - 1] &= (1ull << (PBITS - 1 - 64 * (NL - 1))) - 1;
  fp_from_le(r, (const uint8_t*)t);
}
/* hashAndMapToG1 (call sites src/ps-verifier.cc:94,186, src/ps-requester.cc:185,336): t = Fp::setHashOf(msg), the Shallue-van de Woestijne map
   (mcl MapTo::calcBN; mcl's default, non-ETH mode uses it for BLS12-381 too), then -- BLS12-381 only -- multiplication by the G1 cofactor
   (z-1)^2/3.  Pinned on both curves by the reference wasm's proofs (tests/golden/{bn254,bls12_381}_oracle_flows.json, section hash_to_g1:
   32 service names, all six branch / sign cases). */
static fp SVDW_C1, SVDW_C2;
#ifdef ELPO_BLS12_381
static void g1_mul(g1j* r, const g1a* p, const u64* k);
#endif
static void hash_and_map_g1(g1a* out, const uint8_t* msg, size_t len) {
  fp t, w, x, y, one = FP_ONE, tmp;
  fp_set_hash_of(&t, msg, len);
  int neg = fp_legendre(&t) < 0;
  fp_sqr(&w, &t); fp_add(&w, &w, &CURVE_B); fp_add(&w, &w, &one);
  fp_inv(&w, &w); fp_mul(&w, &w, &t); fp_mul(&w, &w, &SVDW_C1);
  for (int i = 0; i < 3; i++) {
    if (i == 0) { fp_mul(&tmp, &t, &w); fp_sub(&x, &SVDW_C2, &tmp); }
    else if (i == 1) { fp_neg(&x, &x); fp_sub(&x, &x, &one); }
    else { fp_sqr(&tmp, &w); fp_inv(&tmp, &tmp); fp_add(&x, &tmp, &one); }
    fp_sqr(&tmp, &x); fp_mul(&tmp, &tmp, &x); fp_add(&tmp, &tmp, &CURVE_B);
    if (fp_sqrt(&y, &tmp)) break;
  }
  if (neg) fp_neg(&y, &y);
#ifdef ELPO_BLS12_381
  g1a pt;
  memset(&pt, 0, sizeof pt);
  pt.x = x; pt.y = y;
  g1j j;
  g1_mul(&j, &pt, G1_COFACTOR);
  g1_to_aff(out, &j);
#else
  out->x = x; out->y = y; out->inf = 0;
#endif
}

/* ------------------------------------------------------------------------------------------------ init */
void elpo_init(void) {
  if (g_init) return;
  /* -p^-1 mod 2^64 by Newton iteration */
  u64 inv = 1;
  for (int i = 0; i < 6; i++) inv *= 2 - P[0] * inv;
  PINV = (u64)0 - inv;
  memset(&FP_ZERO, 0, sizeof FP_ZERO);
  /* R mod p and R^2 mod p by repeated doubling of 1 (R = 2^(64 NL)) */
  fp t;
  memset(&t, 0, sizeof t);
  t.v[0] = 1;
  for (int i = 0; i < 64 * NL; i++) fp_add(&t, &t, &t);
  FP_ONE = t;
  for (int i = 0; i < 64 * NL; i++) fp_add(&t, &t, &t);
  FP_R2 = t;
  F2_ZERO.a = FP_ZERO; F2_ZERO.b = FP_ZERO;
  F2_ONE.a = FP_ONE; F2_ONE.b = FP_ZERO;
  fp_from_u64(&CURVE_B, CURVE_B_INT);
  fp2 xi = {FP_ONE, FP_ONE}, xinv, b2 = {CURVE_B, FP_ZERO};
#ifdef ELPO_BLS12_381
  (void)xinv;
  fp2_mul(&TWIST_B, &b2, &xi);          /* M-type twist: b' = b xi */
#else
  fp2_inv(&xinv, &xi);
  fp2_mul(&TWIST_B, &b2, &xinv);        /* D-type twist: b' = b / xi */
#endif
  /* gamma_{1,k} = xi^(k (p-1)/6); gamma_{2,k} = gamma_{1,k} conj(gamma_{1,k}); gamma_{3,k} = gamma_{1,k} * gamma_{2,k}^p...
     computed directly: gamma_{n,k} = gamma_{1,k}^(1 + p + ... + p^(n-1)), with x^p = conj(x) in Fp2 */
  u64 e[NL];
  sub_small(e, P, 1);
  { /* e = (p-1)/6 */
    u128 rem = 0;
    for (int i = NL - 1; i >= 0; i--) {
      u128 cur = (rem << 64) | e[i];
      e[i] = (u64)(cur / 6);
      rem = cur % 6;
    }
  }
  fp2 g1;
  fp2_pow(&g1, &xi, e, NL);
  for (int k = 0; k < 6; k++) {
    fp2 a = F2_ONE;
    for (int j = 0; j < k; j++) fp2_mul(&a, &a, &g1);
    fp2 ac, a2, a3;
    fp2_conj(&ac, &a);
    fp2_mul(&a2, &a, &ac);        /* a^(1+p) */
    fp2_conj(&a3, &a2);           /* a^(p+p^2) */
    fp2_mul(&a3, &a3, &a);        /* a^(1+p+p^2) */
    FROB[1][k] = a; FROB[2][k] = a2; FROB[3][k] = a3;
  }
  /* SvdW constants: c1 = sqrt(-3) = (-3)^((p+1)/4), the root fp_sqrt returns (BN254: ...0004, BLS12-381: ...fffdfffd; both pinned by the
     golden vectors), c2 = (c1 - 1)/2 */
  fp m3, two, half;
  fp_from_u64(&m3, 3); fp_neg(&m3, &m3);
  fp_sqrt(&SVDW_C1, &m3);
  fp_from_u64(&two, 2); fp_inv(&half, &two);
  fp_sub(&SVDW_C2, &SVDW_C1, &FP_ONE); fp_mul(&SVDW_C2, &SVDW_C2, &half);
  g_init = 1;
}

/* ------------------------------------------------------------------------------------------------ byte-level API */
#define G1B (2 * FB)
#define G2B (4 * FB)
/* Formats are those of include/elpasso.h: G1 = x|y (FB + FB bytes LE, zeros = infinity; FB = 32 / 48), G2 = x.a|x.b|y.a|y.b, Fr = 32 LE. */
static int g1_load(g1a* p, const uint8_t* b) {
  int any = 0;
  for (int i = 0; i < 2 * FB; i++) any |= b[i];
  memset(p, 0, sizeof *p);
  if (!any) { p->inf = 1; return 1; }
  if (!le_lt_p(b) || !le_lt_p(b + FB)) return 0;
  fp_from_le(&p->x, b); fp_from_le(&p->y, b + FB);
  return g1_on_curve(p);
}
static int g2_load(g2a* p, const uint8_t* b) {
  int any = 0;
  for (int i = 0; i < 4 * FB; i++) any |= b[i];
  memset(p, 0, sizeof *p);
  if (!any) { p->inf = 1; return 1; }
  for (int i = 0; i < 4; i++) if (!le_lt_p(b + FB * i)) return 0;
  fp_from_le(&p->x.a, b); fp_from_le(&p->x.b, b + FB); fp_from_le(&p->y.a, b + 2 * FB); fp_from_le(&p->y.b, b + 3 * FB);
  return g2_on_curve(p);
}
static void g1_store(uint8_t* b, const g1a* p) {
  if (p->inf) { memset(b, 0, 2 * FB); return; }
  fp_to_le(b, &p->x); fp_to_le(b + FB, &p->y);
}
static void g2_store(uint8_t* b, const g2a* p) {
  if (p->inf) { memset(b, 0, 4 * FB); return; }
  fp_to_le(b, &p->x.a); fp_to_le(b + FB, &p->x.b); fp_to_le(b + 2 * FB, &p->y.a); fp_to_le(b + 3 * FB, &p->y.b);
}
static void k_load(u64 k[4], const uint8_t* b) { memcpy(k, b, 32); }

int elpo_g1_mul(const uint8_t* P_, const uint8_t* k_, uint8_t* out) {
  g1a p, r; g1j j; u64 k[4];
  if (!g1_load(&p, P_)) return 0;
  k_load(k, k_); g1_mul_ref(&j, &p, k); g1_to_aff(&r, &j); g1_store(out, &r);
  return 1;
}
int elpo_g2_mul(const uint8_t* P_, const uint8_t* k_, uint8_t* out) {
  g2a p, r; g2j j; u64 k[4];
  if (!g2_load(&p, P_)) return 0;
  k_load(k, k_); g2_mul(&j, &p, k); g2_to_aff(&r, &j); g2_store(out, &r);
  return 1;
}
int elpo_g1_add(const uint8_t* a_, const uint8_t* b_, uint8_t* out) {
  g1a a, b, r; g1j ja, jb;
  if (!g1_load(&a, a_) || !g1_load(&b, b_)) return 0;
  g1_from_aff(&ja, &a); g1_from_aff(&jb, &b); g1_add(&ja, &ja, &jb); g1_to_aff(&r, &ja); g1_store(out, &r);
  return 1;
}
int elpo_g2_add(const uint8_t* a_, const uint8_t* b_, uint8_t* out) {
  g2a a, b, r; g2j ja, jb;
  if (!g2_load(&a, a_) || !g2_load(&b, b_)) return 0;
  g2_from_aff(&ja, &a); g2_from_aff(&jb, &b); g2_add(&ja, &ja, &jb); g2_to_aff(&r, &ja); g2_store(out, &r);
  return 1;
}
int elpo_g1_decompress(const uint8_t* w, uint8_t* out) { g1a p; if (!g1_de(&p, w)) return 0; g1_store(out, &p); return 1; }
int elpo_g2_decompress(const uint8_t* w, uint8_t* out) { g2a p; if (!g2_de(&p, w)) return 0; g2_store(out, &p); return 1; }
int elpo_g1_compress(const uint8_t* a_, uint8_t* w) { g1a p; if (!g1_load(&p, a_)) return 0; g1_ser(w, &p); return 1; }
int elpo_g2_compress(const uint8_t* a_, uint8_t* w) { g2a p; if (!g2_load(&p, a_)) return 0; g2_ser(w, &p); return 1; }
void elpo_hash_to_g1(const uint8_t* msg, size_t len, uint8_t* out) { g1a p; hash_and_map_g1(&p, msg, len); g1_store(out, &p); }
void elpo_fr_set_hash_of(const uint8_t* msg, size_t len, uint8_t* out) { u64 k[4]; set_hash_of(k, msg, len, RORD); memcpy(out, k, 32); }
static void gt_store(uint8_t* out, const fp12* f) {
  const fp2* e[6] = {&f->c0.c0, &f->c0.c1, &f->c0.c2, &f->c1.c0, &f->c1.c1, &f->c1.c2};
  for (int i = 0; i < 6; i++) { fp_to_le(out + 2 * FB * i, &e[i]->a); fp_to_le(out + 2 * FB * i + FB, &e[i]->b); }
}
int elpo_pairing(const uint8_t* P_, const uint8_t* Q_, uint8_t* gt) {
  g1a p; g2a q; fp12 e;
  if (!g1_load(&p, P_) || !g2_load(&q, Q_)) return 0;
  pairing(&e, &p, &q); gt_store(gt, &e);
  return 1;
}

int elpo_curve(void) {
#ifdef ELPO_BLS12_381
  return 1;
#else
  return 0;
#endif
}
#ifdef ELPO_BLS12_381
/* 1 iff, for e = f_{z,Q}(P), the hard part computed by the chain equals e_easy^((p^4-p^2+1)/r) computed by plain square-and-multiply */
int elpo_selftest_final_exp(const uint8_t* P_, const uint8_t* Q_) {
  elpo_init();
  g1a p; g2a q; fp12 f, e, a, b;
  if (!g1_load(&p, P_) || !g2_load(&q, Q_)) return -1;
  miller_loop(&f, &p, &q);
  final_exp_easy(&e, &f);
  final_exp_hard_chain(&a, &e);
  fp12_pow_limbs(&b, &e, HARD_EXP, 20);
  return fp12_eq(&a, &b);
}
#endif

/* ---- key container: plain copies of the points, no precomputation (reference structure) */
typedef struct {
  int A;
  g1a g, *Yi, hs, g_eg, apk, h, skX;
  g2a gg, XX, *YYi;
} elpo_key;

/* g1_bases: (A+6) points in the order of include/elpasso.h (g, Y_i, H1(svc), g_eg, authority_pk, h, X);
   g2_bases: (A+2) points (gg, XX, YY_i). */
elpo_key* elpo_key_new(int A, const uint8_t* g1_bases, const uint8_t* g2_bases) {
  elpo_init();
  elpo_key* k = (elpo_key*)calloc(1, sizeof *k);
  k->A = A;
  k->Yi = (g1a*)calloc(A, sizeof(g1a));
  k->YYi = (g2a*)calloc(A, sizeof(g2a));
  int ok = g1_load(&k->g, g1_bases);
  for (int i = 0; i < A; i++) ok &= g1_load(&k->Yi[i], g1_bases + G1B * (1 + i));
  ok &= g1_load(&k->hs, g1_bases + G1B * (A + 1));
  ok &= g1_load(&k->g_eg, g1_bases + G1B * (A + 2));
  ok &= g1_load(&k->apk, g1_bases + G1B * (A + 3));
  ok &= g1_load(&k->h, g1_bases + G1B * (A + 4));
  ok &= g1_load(&k->skX, g1_bases + G1B * (A + 5));
  ok &= g2_load(&k->gg, g2_bases);
  ok &= g2_load(&k->XX, g2_bases + G2B);
  for (int i = 0; i < A; i++) ok &= g2_load(&k->YYi[i], g2_bases + G2B * (2 + i));
  if (!ok) { free(k->Yi); free(k->YYi); free(k); return 0; }
  return k;
}
void elpo_key_free(elpo_key* k) { if (k) { free(k->Yi); free(k->YYi); free(k); } }

static void g1_mul_add(g1j* acc, const g1a* base, const u64* k) { g1j t; g1_mul_ref(&t, base, k); g1_add(acc, acc, &t); }
static void g2_mul_add(g2j* acc, const g2a* base, const u64* k) { g2j t; g2_mul(&t, base, k); g2_add(acc, acc, &t); }
static void fr_one_minus(u64 out[4], const u64 c[4]) { /* (1 - c) mod r, c < r */
  u64 one[4] = {1, 0, 0, 0};
  u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)one[i] - c[i] - br; out[i] = (u64)t; br = (t >> 64) & 1; }
  if (br) { u128 cy = 0; for (int i = 0; i < 4; i++) { cy += (u128)out[i] + RORD[i]; out[i] = (u64)cy; cy >>= 64; } }
}
static void challenge(u64 out[4], sha256_t* s, const uint8_t* ad, size_t adl) {
  uint8_t d[32];
  sha_update(s, ad, adl);
  sha_final(s, d);                         /* digest_engine.digest(associated_data) */
  set_hash_of(out, d, 32, RORD);           /* _local_c.setHashOf(_c_str) */
}

#ifdef ELPO_BLS12_381
/* Project policy on this curve (G1 cofactor (z-1)^2/3 != 1; include/elpasso.h ELP_OPT_SUBGROUP_CHECK): prover-supplied G1 points must lie in the
   order-r subgroup.  Checked here by the definition, [r]P == O, with plain double-and-add (the HIP path uses the endomorphism test). */
static int elpo_subgroup_check;
static int g1_in_subgroup(const g1a* p) {
  if (p->inf || !elpo_subgroup_check) return 1;
  g1j acc, base;
  g1_set_inf(&acc);
  g1_from_aff(&base, p);
  for (int i = 255; i >= 0; i--) {
    g1_dbl(&acc, &acc);
    if ((RORD[i >> 6] >> (i & 63)) & 1) g1_add(&acc, &acc, &base);
  }
  return g1_is_inf(&acc);
}
#else
static int g1_in_subgroup(const g1a* p) { (void)p; return 1; }     /* BN254: E(Fp) has prime order r */
#endif

/* sigma_1 must not be the identity of G1 (ps-verifier.cc:16-18).  With a G1 cofactor that is asked of the order-r component: a point whose order divides the
   cofactor pairs to 1 with everything, so sig1 = T (order 3), sig2 = O would satisfy the pairing equation for any K.  Policy: sig1 != O and sig1 in G1. */
static int sig1_admissible(const g1a* s) { return !s->inf && g1_in_subgroup(s); }
/* the library's ELP_OPT_STRICT_SIGNATURE for el_passo_verify_id: 0 (default here) = the reference's behaviour, 1 = sig1 must be admissible as in PSVerifier::verify */
static int elpo_strict = 0;
void elpo_set_strict(int on) { elpo_strict = on; }
/* the library's ELP_OPT_SUBGROUP_CHECK (BLS12-381): 1 (default here, as in the library) = the project's policy above, a DELIBERATE divergence from the
   reference; 0 = the reference's behaviour -- mcl's default does not test the order of a deserialised G1 point; together with g1_mul_ref this
   build then reproduces every verdict of the reference's wasm on points outside G1 (tests/test_oracle_bls.py) */
static int elpo_subgroup_check = 1;
void elpo_set_subgroup_check(int on) { elpo_subgroup_check = on; }

/* record: sig1 | sig2 | phi | [E1 | E2] | k | c | rs[..] | m[..]   (same as elp_verify_id_batch) */
int elpo_verify_id(const elpo_key* key, const uint8_t* rec, uint64_t hidden_mask, int retr, const uint8_t* ad, size_t adl) {
  const int A = key->A;
  int H = 0;
  for (int i = 0; i < A; i++) H += (hidden_mask >> i) & 1;
  const int nrs = H + (retr ? 2 : 1);
  g1a sig1, sig2, phi, E1, E2;
  g2a kk;
  const uint8_t* p = rec;
  int ok = g1_load(&sig1, p); p += G1B;
  ok &= g1_load(&sig2, p); p += G1B;
  ok &= g1_load(&phi, p); p += G1B;
  if (retr) { ok &= g1_load(&E1, p); p += G1B; ok &= g1_load(&E2, p); p += G1B; }
  ok &= g2_load(&kk, p); p += G2B;
  if (!ok) return 0;
  if (elpo_strict && !sig1_admissible(&sig1)) return 0;
  if (!g1_in_subgroup(&phi) || (retr && (!g1_in_subgroup(&E1) || !g1_in_subgroup(&E2)))) return 0;
  u64 c[4], s[4];
  k_load(c, p); p += 32;
  const uint8_t* rs = p; p += 32 * nrs;
  const uint8_t* ms = p;
  if (u256_geq(c, RORD)) return 0;                             /* a challenge >= r can never equal the recomputed one */
  /* V_k = k^c * prod YY_j^{r_j} * gg^{r_t} * XX^{1-c}         ps-verifier.cc:72-88 */
  g2j Vk;
  g2_mul(&Vk, &kk, c);
  int j = 0;
  for (int i = 0; i < A; i++)
    if ((hidden_mask >> i) & 1) { k_load(s, rs + 32 * j); j++; g2_mul_add(&Vk, &key->YYi[i], s); }
  k_load(s, rs + 32 * (retr ? nrs - 2 : nrs - 1));
  g2_mul_add(&Vk, &key->gg, s);
  fr_one_minus(s, c);
  g2_mul_add(&Vk, &key->XX, s);
  /* V_phi = phi^c * H1(svc)^{r_0}                              ps-verifier.cc:91-96 */
  g1j Vphi, VE1, VE2;
  g1_mul_ref(&Vphi, &phi, c);
  k_load(s, rs);
  g1_mul_add(&Vphi, &key->hs, s);
  if (retr) {
    u64 re[4];
    k_load(re, rs + 32 * (nrs - 1));
    g1_mul_ref(&VE1, &E1, c); g1_mul_add(&VE1, &key->g_eg, re);                 /* :99-101 */
    g1_mul_ref(&VE2, &E2, c); g1_mul_add(&VE2, &key->apk, re);                  /* :104-108 */
    k_load(s, rs + 32); g1_mul_add(&VE2, &key->h, s);
  }
  /* c' = Hr(SHA256(hex(k) hex(phi) [hex(E1) hex(E2)] hex(V_k) hex(V_phi) [hex(V_E1) hex(V_E2)] ad))   :111-122 */
  sha256_t sh;
  uint8_t b1[FB], b2[2 * FB];
  g2a aVk; g1a a1;
  sha_init(&sh);
  g2_ser(b2, &kk); sha_update_hex(&sh, b2, 2 * FB);
  g1_ser(b1, &phi); sha_update_hex(&sh, b1, FB);
  if (retr) { g1_ser(b1, &E1); sha_update_hex(&sh, b1, FB); g1_ser(b1, &E2); sha_update_hex(&sh, b1, FB); }
  g2_to_aff(&aVk, &Vk); g2_ser(b2, &aVk); sha_update_hex(&sh, b2, 2 * FB);
  g1_to_aff(&a1, &Vphi); g1_ser(b1, &a1); sha_update_hex(&sh, b1, FB);
  if (retr) {
    g1_to_aff(&a1, &VE1); g1_ser(b1, &a1); sha_update_hex(&sh, b1, FB);
    g1_to_aff(&a1, &VE2); g1_ser(b1, &a1); sha_update_hex(&sh, b1, FB);
  }
  u64 c2[4];
  challenge(c2, &sh, ad, adl);
  if (memcmp(c2, c, 32) != 0) return 0;                                    /* :128-130 */
  /* K = k * prod_{revealed} YY_i^{m_i} ; e(sig1, K) == e(sig2, gg)         :133-137, :214-229 */
  g2j K;
  g2_from_aff(&K, &kk);
  j = 0;
  for (int i = 0; i < A; i++)
    if (!((hidden_mask >> i) & 1)) { k_load(s, ms + 32 * j); j++; g2_mul_add(&K, &key->YYi[i], s); }
  g2a aK;
  g2_to_aff(&aK, &K);
  fp12 lhs, rhs;
  pairing(&lhs, &sig1, &aK);
  pairing(&rhs, &sig2, &key->gg);
  return fp12_eq(&lhs, &rhs);
}

/* record: sig1 | sig2 | m[nattr]        ps-verifier.cc:13-35 */
int elpo_ps_verify(const elpo_key* key, const uint8_t* rec, int nattr) {
  g1a sig1, sig2;
  if (!g1_load(&sig1, rec) || !g1_load(&sig2, rec + G1B)) return 0;
  if (!sig1_admissible(&sig1)) return 0;                                  /* ps-verifier.cc:16-18, in the order-r component */
  g2j K;
  g2_from_aff(&K, &key->XX);
  u64 s[4];
  for (int i = 0; i < nattr; i++) { k_load(s, rec + 2 * G1B + 32 * i); g2_mul_add(&K, &key->YYi[i], s); }
  g2a aK;
  g2_to_aff(&aK, &K);
  fp12 lhs, rhs;
  pairing(&lhs, &sig1, &aK);
  pairing(&rhs, &sig2, &key->gg);
  return fp12_eq(&lhs, &rhs);
}

/* record: A | c | rs[H+1] | m[A-H] | u ; out: sig1 | sig2        ps-signer.cc:63-146 */
int elpo_provide_id(const elpo_key* key, const uint8_t* rec, uint64_t hidden_mask, const uint8_t* ad, size_t adl, uint8_t* out) {
  const int A = key->A;
  int H = 0;
  for (int i = 0; i < A; i++) H += (hidden_mask >> i) & 1;
  memset(out, 0, 2 * G1B);
  g1a Ac;
  if (!g1_load(&Ac, rec)) return 0;
  if (!g1_in_subgroup(&Ac)) return 0;
  const uint8_t* p = rec + G1B;
  u64 c[4], s[4], u[4];
  k_load(c, p); p += 32;
  const uint8_t* rs = p; p += 32 * (H + 1);
  const uint8_t* ms = p; p += 32 * (A - H);
  k_load(u, p);
  g1j V;
  g1_mul_ref(&V, &Ac, c);                                                     /* :83 */
  k_load(s, rs); g1_mul_add(&V, &key->g, s);                              /* :85-86 */
  int j = 1;
  for (int i = 0; i < A; i++)
    if ((hidden_mask >> i) & 1) { k_load(s, rs + 32 * j); j++; g1_mul_add(&V, &key->Yi[i], s); }   /* :88-94 */
  sha256_t sh;
  uint8_t b1[FB];
  g1a aV;
  sha_init(&sh);
  g1_ser(b1, &Ac); sha_update_hex(&sh, b1, FB);
  g1_to_aff(&aV, &V); g1_ser(b1, &aV); sha_update_hex(&sh, b1, FB);
  u64 c2[4];
  challenge(c2, &sh, ad, adl);
  if (memcmp(c2, c, 32) != 0) return 0;                                   /* :106-108 */
  g1j Ap;
  g1_from_aff(&Ap, &Ac);
  if (A != 1) {                                                           /* quirk :115-117 */
    j = 0;
    for (int i = 0; i < A; i++)
      if (!((hidden_mask >> i) & 1)) { k_load(s, ms + 32 * j); j++; g1_mul_add(&Ap, &key->Yi[i], s); }
  }
  g1j s1, s2, X;
  g1_mul_ref(&s1, &key->g, u);                                                /* :138 */
  g1_from_aff(&X, &key->skX);
  g1_add(&Ap, &Ap, &X);                                                   /* :140 */
  g1a aAp, r1, r2;
  g1_to_aff(&aAp, &Ap);
  g1_mul_ref(&s2, &aAp, u);                                                   /* :141 */
  g1_to_aff(&r1, &s1); g1_to_aff(&r2, &s2);
  g1_store(out, &r1); g1_store(out + G1B, &r2);
  return 1;
}

/* batch drivers for the CPU baseline (OpenMP over items when available) */
long elpo_verify_id_batch(const elpo_key* key, long n, const uint8_t* recs, size_t stride, uint64_t mask, int retr,
                          const uint8_t* ad, size_t adl, uint8_t* flags, int nthreads) {
  long acc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : acc)
#endif
  for (long i = 0; i < n; i++) {
    int ok = elpo_verify_id(key, recs + (size_t)i * stride, mask, retr, ad, adl);
    if (flags) flags[i] = (uint8_t)ok;
    acc += ok;
  }
  (void)nthreads;
  return acc;
}
