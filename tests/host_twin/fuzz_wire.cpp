// TEST INFRASTRUCTURE (never shipped): sanitizer harness for the code that parses UNTRUSTED bytes.
//
// The device-side wire parser WireSrc<C> (csrc/elp/pipeline.h: T-L-V walk of IdProof::toBufferString() messages, src/ps-encoding.cc:451-489 in the
// reference, whose own parser has undefined behaviour on malformed input, :377) is ELP_HD code; an out-of-bounds read on the GPU is silent, so the same
// header is compiled here for the host with -fsanitize=address,undefined (tests/test_fuzz_wire.py builds and runs this file) and fed mutated messages.
// Every message is copied into a heap block of EXACTLY its length, so a read one byte past the end is an AddressSanitizer report.
//
// Input  (argv[1]):  u32 A | u32 ncases | ncases x { u32 len | u8 retr | bytes[len] }
// Output (argv[2]):  ncases x { u8 accepted | u8 digest[32] }   accepted = WireSrc::open (structure, lengths, scalar ranges, point decoding);
//                    digest = XOR of every response rs(j) and of Fr::setHashOf of every revealed attribute, read back through the accessors the
//                    verification uses (they walk the message again), XOR the canonical bytes of k / phi / E1 / E2 (ser_k, ser_g1) folded to 32 bytes.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "elp/pipeline.h"
#include "elp/params_bn254.h"

using namespace elp;
typedef BN254 C;

static uint32_t rd32(FILE* f) {
  uint32_t v = 0;
  if (fread(&v, 4, 1, f) != 1) {
    fprintf(stderr, "fuzz_wire: truncated corpus\n");
    exit(2);
  }
  return v;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* in = fopen(argv[1], "rb");
  FILE* out = fopen(argv[2], "wb");
  if (!in || !out) return 2;
  const int A = (int)rd32(in);
  const uint32_t n = rd32(in);
  size_t accepted = 0;
  for (uint32_t t = 0; t < n; t++) {
    const uint32_t len = rd32(in);
    const int retr = fgetc(in);
    uint8_t* msg = (uint8_t*)malloc(len ? len : 1);          // exactly `len` usable bytes (len == 0: one byte the parser must not touch either way)
    if (len && fread(msg, 1, len, in) != len) return 2;
    uint8_t rec[33];
    memset(rec, 0, sizeof rec);
    WireSrc<C> src;
    Aff<F1<C>> sig1, sig2, phi, E1, E2;
    Aff<F2<C>> kk;
    Scalar c;
    const bool ok = src.open(msg, len, A, retr != 0, sig1, sig2, phi, E1, E2, kk, c);
    if (ok) {
      accepted++;
      rec[0] = 1;
      uint8_t* d = rec + 1;
      for (int j = 0; j < src.nrs(); j++) {
        const Scalar r = src.rs(j);
        for (int q = 0; q < 8; q++)
          for (int b = 0; b < 4; b++) d[4 * q + b] ^= (uint8_t)(r.v[q] >> (8 * b));
      }
      for (int i = 0; i < A; i++)
        if (!src.hidden(i)) {
          const Scalar m = src.next_revealed_hash(i);
          for (int q = 0; q < 8; q++)
            for (int b = 0; b < 4; b++) d[4 * q + b] ^= (uint8_t)(m.v[q] >> (8 * b));
        }
      uint8_t buf[2 * C::FBYTES];
      src.ser_k(buf);
      for (int i = 0; i < 2 * C::FBYTES; i++) d[i & 31] ^= buf[i];
      for (int w = 0; w < (retr ? 3 : 1); w++) {
        src.ser_g1(w, buf);
        for (int i = 0; i < C::FBYTES; i++) d[i & 31] ^= buf[i];
      }
      for (int q = 0; q < 8; q++)
        for (int b = 0; b < 4; b++) d[4 * q + b] ^= (uint8_t)(c.v[q] >> (8 * b));
    }
    fwrite(rec, 1, sizeof rec, out);
    free(msg);
  }
  fclose(out);
  fclose(in);
  fprintf(stderr, "fuzz_wire: %u cases, %zu accepted\n", n, accepted);
  return 0;
}
