/* TEST INFRASTRUCTURE ONLY (never shipped, never loaded by the product): the handful of C-ABI entry points that
 * csrc/host/elp_mcl_compat.cc calls, answered by the CPU oracle (oracle/elp_oracle.c) instead of the GPU.  It lets the reference's
 * own test programs (compiled unchanged against elp_mcl_compat.h, oracle/Makefile `dropin-cpu`) run in the GPU-less build container,
 * which checks the stand-in layer's SEMANTICS (aliasing, encodings, Fr arithmetic) independently of the HIP kernels. */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/elpasso.h"

void elpo_init(void);
int elpo_g1_mul(const uint8_t*, const uint8_t*, uint8_t*);
int elpo_g2_mul(const uint8_t*, const uint8_t*, uint8_t*);
int elpo_g1_add(const uint8_t*, const uint8_t*, uint8_t*);
int elpo_g2_add(const uint8_t*, const uint8_t*, uint8_t*);
int elpo_g1_decompress(const uint8_t*, uint8_t*);
int elpo_g2_decompress(const uint8_t*, uint8_t*);
void elpo_hash_to_g1(const uint8_t*, size_t, uint8_t*);
int elpo_pairing(const uint8_t*, const uint8_t*, uint8_t*);

struct elp_ctx {
  int dummy;
};
static struct elp_ctx g_ctx;

int elp_init(int curve, int device, elp_ctx** out) {
  (void)device;
  if (curve != ELP_CURVE_BN254) return ELP_ERR_ARG;
  elpo_init();
  *out = &g_ctx;
  return ELP_OK;
}
void elp_destroy(elp_ctx* c) { (void)c; }
const char* elp_last_error(const elp_ctx* c) { (void)c; return "oracle shim"; }
int elp_set_option(elp_ctx* c, int o, int v) { (void)c; (void)o; (void)v; return ELP_OK; }
#define LOOP(n, body) for (size_t i = 0; i < (n); i++) { body; }
int elp_g1_mul(elp_ctx* c, size_t n, const uint8_t* p, const uint8_t* k, uint8_t* o) { (void)c; LOOP(n, if (!elpo_g1_mul(p + 64 * i, k + 32 * i, o + 64 * i)) memset(o + 64 * i, 0, 64)) return ELP_OK; }
int elp_g2_mul(elp_ctx* c, size_t n, const uint8_t* p, const uint8_t* k, uint8_t* o) { (void)c; LOOP(n, if (!elpo_g2_mul(p + 128 * i, k + 32 * i, o + 128 * i)) memset(o + 128 * i, 0, 128)) return ELP_OK; }
int elp_g1_add(elp_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* o) { (void)c; LOOP(n, if (!elpo_g1_add(a + 64 * i, b + 64 * i, o + 64 * i)) memset(o + 64 * i, 0, 64)) return ELP_OK; }
int elp_g2_add(elp_ctx* c, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* o) { (void)c; LOOP(n, if (!elpo_g2_add(a + 128 * i, b + 128 * i, o + 128 * i)) memset(o + 128 * i, 0, 128)) return ELP_OK; }
int elp_g1_decompress(elp_ctx* c, size_t n, const uint8_t* w, uint8_t* o, uint8_t* ok) { (void)c; LOOP(n, ok[i] = (uint8_t)elpo_g1_decompress(w + 32 * i, o + 64 * i)) return ELP_OK; }
int elp_g2_decompress(elp_ctx* c, size_t n, const uint8_t* w, uint8_t* o, uint8_t* ok) { (void)c; LOOP(n, ok[i] = (uint8_t)elpo_g2_decompress(w + 64 * i, o + 128 * i)) return ELP_OK; }
int elp_hash_to_g1(elp_ctx* c, size_t n, const uint8_t* msgs, const uint32_t* off, uint8_t* o) { (void)c; LOOP(n, elpo_hash_to_g1(msgs + off[i], off[i + 1] - off[i], o + 64 * i)) return ELP_OK; }
int elp_pairing(elp_ctx* c, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt) { (void)c; LOOP(n, if (!elpo_pairing(g1 + 64 * i, g2 + 128 * i, gt + 384 * i)) return ELP_ERR_POINT) return ELP_OK; }
