// TEST INFRASTRUCTURE (never shipped): sanitizer harness for the host layer's message parsers, PSBuffer::parse* and the four fromBufferString()
// (csrc/host/ps-encoding.cc; reference: src/ps-encoding.cc:136-162 parseVar, :377-489 the message parsers, which read out of bounds on truncated input).
// Built by tests/test_fuzz_wire.py with -fsanitize=address,undefined from the product's own sources (ps-encoding.cc, elp_mcl_compat.cc); the C-ABI calls
// behind G1 / G2 ::deserialize are answered by the CPU oracle through tests/cpu_shim (no GPU in the build container).  Each message is parsed as every
// message type; a parser may throw std::exception (the reference's PSBuffer::at does) or return an object -- it may not touch memory it does not own.
//
// Input (argv[1]): the corpus format of fuzz_wire.cpp.   Output (argv[2]): ncases x u8 bit mask, bit t set = parser t returned an object
// (0 IdProof, 1 PSCredRequest, 2 PSCredential, 3 PSPubKey).
#include <stdio.h>
#include <stdlib.h>

#include <exception>

#include "ps-encoding.h"

static uint32_t rd32(FILE* f) {
  uint32_t v = 0;
  if (fread(&v, 4, 1, f) != 1) exit(2);
  return v;
}
template <class Fn>
static int tryParse(Fn fn) {
  try {
    fn();
    return 1;
  } catch (const std::exception&) {
    return 0;
  }
}
int main(int argc, char** argv) {
  if (argc < 3) return 2;
  initPairing();
  FILE* in = fopen(argv[1], "rb");
  FILE* out = fopen(argv[2], "wb");
  if (!in || !out) return 2;
  (void)rd32(in);   // A: the host parsers do not know the key
  const uint32_t n = rd32(in);
  size_t objs = 0;
  for (uint32_t t = 0; t < n; t++) {
    const uint32_t len = rd32(in);
    (void)fgetc(in);
    PSBuffer b;
    b.resize(len);
    if (len && fread(b.data(), 1, len, in) != len) return 2;
    b.shrink_to_fit();                                    // capacity == size: the vector's block ends where the message ends
    int m = 0;
    m |= tryParse([&] { IdProof p = IdProof::fromBufferString(b); (void)p; }) << 0;
    m |= tryParse([&] { PSCredRequest r = PSCredRequest::fromBufferString(b); (void)r; }) << 1;
    m |= tryParse([&] { PSCredential c = PSCredential::fromBufferString(b); (void)c; }) << 2;
    m |= tryParse([&] { PSPubKey k = PSPubKey::fromBufferString(b); (void)k; }) << 3;
    // base64 both ways on arbitrary bytes
    PSBuffer rt = PSBuffer::fromBase64(b.toBase64());
    if (rt != b) {
      fprintf(stderr, "fuzz_psbuffer: base64 round trip changed case %u\n", t);
      return 3;
    }
    objs += m & 1;
    fputc(m, out);
  }
  fclose(out);
  fclose(in);
  fprintf(stderr, "fuzz_psbuffer: %u cases, %zu parsed as IdProof\n", n, objs);
  return 0;
}
