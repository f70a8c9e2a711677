// T-L-V + base64 codec (wire format decoded in SURVEY.md Appendix C; reference: src/ps-encoding.cc:98-489).
// Written from the format description: one generic element writer/reader over a serialiser functor.
#include "ps-encoding.h"

#include <string.h>

namespace {

const char kB64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";

int b64Value(unsigned char c) {
  if (c >= 'A' && c <= 'Z') return c - 'A';
  if (c >= 'a' && c <= 'z') return c - 'a' + 26;
  if (c >= '0' && c <= '9') return c - '0' + 52;
  if (c == '+') return 62;
  if (c == '/') return 63;
  return -1;
}

template <class T>
size_t wireOf(const T& v, uint8_t* tmp, size_t cap) {
  return v.serialize(tmp, cap);
}

// element body (without the type byte): var-length | bytes
template <class T>
void putBody(PSBuffer& b, const T& v) {
  uint8_t tmp[128];
  size_t n = wireOf(v, tmp, sizeof tmp);
  b.appendVar(n);
  b.insert(b.end(), tmp, tmp + n);
}
template <class T>
size_t getBody(const PSBuffer& b, size_t off, T& v) {
  size_t n = 0;
  size_t used = b.parseVar(off, n);
  if (off + used + n > b.size()) throw std::out_of_range("PSBuffer: truncated element");
  v.deserialize(b.data() + off + used, n);   // like the reference, a failed decode leaves v unchanged
  return used + n;
}
template <class T>
void putElement(PSBuffer& b, PSEncodingType t, const T& v, bool withType) {
  if (withType) b.appendType(t);
  putBody(b, v);
}
template <class T>
size_t getElement(const PSBuffer& b, size_t off, PSEncodingType want, T& v, bool withType) {
  size_t used = 0;
  if (withType) {
    PSEncodingType t;
    used += b.parseType(off, t);
    if (t != want) return 0;
  }
  return used + getBody(b, off + used, v);
}
template <class T>
void putList(PSBuffer& b, PSEncodingType t, const std::vector<T>& vs) {
  b.appendType(t);
  b.appendVar(vs.size());
  for (const T& v : vs) putBody(b, v);
}
template <class T>
size_t getList(const PSBuffer& b, size_t off, PSEncodingType want, std::vector<T>& vs) {
  PSEncodingType t;
  size_t used = b.parseType(off, t);
  if (t != want) return 0;
  size_t count = 0;
  used += b.parseVar(off + used, count);
  for (size_t i = 0; i < count; i++) {
    T v;
    used += getBody(b, off + used, v);
    vs.push_back(v);
  }
  return used;
}

}  // namespace

PSBuffer PSBuffer::fromBase64(const std::string& s) {
  PSBuffer out;
  uint32_t acc = 0;
  int bits = 0;
  for (unsigned char c : s) {
    int v = b64Value(c);
    if (v < 0) break;  // '=' padding or a foreign character ends the payload
    acc = (acc << 6) | (uint32_t)v;
    bits += 6;
    if (bits >= 8) {
      bits -= 8;
      out.push_back((uint8_t)(acc >> bits));
    }
  }
  return out;
}

std::string PSBuffer::toBase64() {
  std::string out;
  size_t n = size();
  out.reserve((n + 2) / 3 * 4);
  for (size_t i = 0; i < n; i += 3) {
    uint32_t w = (uint32_t)(*this)[i] << 16;
    if (i + 1 < n) w |= (uint32_t)(*this)[i + 1] << 8;
    if (i + 2 < n) w |= (*this)[i + 2];
    out += kB64[(w >> 18) & 63];
    out += kB64[(w >> 12) & 63];
    out += i + 1 < n ? kB64[(w >> 6) & 63] : '=';
    out += i + 2 < n ? kB64[w & 63] : '=';
  }
  return out;
}

void PSBuffer::appendType(PSEncodingType type) { push_back(static_cast<uint8_t>(type)); }
size_t PSBuffer::parseType(size_t offset, PSEncodingType& type) const {
  type = static_cast<PSEncodingType>(at(offset));
  return 1;
}
void PSBuffer::appendVar(size_t var) {
  if (var < 253) {
    push_back((uint8_t)var);
  } else if (var <= 0xFFFF) {
    push_back(253);
    push_back((uint8_t)(var >> 8));
    push_back((uint8_t)var);
  }  // larger values are not representable in this format (the reference drops them too)
}
size_t PSBuffer::parseVar(size_t offset, size_t& var) const {
  uint8_t first = at(offset);
  if (first < 253) {
    var = first;
    return 1;
  }
  if (first == 253) {
    var = ((size_t)at(offset + 1) << 8) | at(offset + 2);
    return 3;
  }
  return 0;
}

void PSBuffer::appendG1Element(const G1& g, bool withType) { putElement(*this, PSEncodingType::G1, g, withType); }
size_t PSBuffer::parseG1Element(size_t offset, G1& g, bool withType) const { return getElement(*this, offset, PSEncodingType::G1, g, withType); }
void PSBuffer::appendG2Element(const G2& g, bool withType) { putElement(*this, PSEncodingType::G2, g, withType); }
size_t PSBuffer::parseG2Element(size_t offset, G2& g, bool withType) const { return getElement(*this, offset, PSEncodingType::G2, g, withType); }
void PSBuffer::appendFrElement(const Fr& f, bool withType) { putElement(*this, PSEncodingType::Fr, f, withType); }
size_t PSBuffer::parseFrElement(size_t offset, Fr& f, bool withType) const { return getElement(*this, offset, PSEncodingType::Fr, f, withType); }
void PSBuffer::appendG1List(const std::vector<G1>& gs) { putList(*this, PSEncodingType::G1List, gs); }
size_t PSBuffer::parseG1List(size_t offset, std::vector<G1>& gs) const { return getList(*this, offset, PSEncodingType::G1List, gs); }
void PSBuffer::appendG2List(const std::vector<G2>& gs) { putList(*this, PSEncodingType::G2List, gs); }
size_t PSBuffer::parseG2List(size_t offset, std::vector<G2>& gs) const { return getList(*this, offset, PSEncodingType::G2List, gs); }
void PSBuffer::appendFrList(const std::vector<Fr>& fs) { putList(*this, PSEncodingType::FrList, fs); }
size_t PSBuffer::parseFrList(size_t offset, std::vector<Fr>& fs) const { return getList(*this, offset, PSEncodingType::FrList, fs); }

void PSBuffer::appendStrList(const std::vector<std::string>& strs) {
  appendType(PSEncodingType::StrList);
  appendVar(strs.size());
  for (const std::string& s : strs) {
    appendVar(s.size());
    insert(end(), s.begin(), s.end());
  }
}
size_t PSBuffer::parseStrList(size_t offset, std::vector<std::string>& strs) const {
  PSEncodingType t;
  size_t used = parseType(offset, t);
  if (t != PSEncodingType::StrList) return 0;
  size_t count = 0;
  used += parseVar(offset + used, count);
  for (size_t i = 0; i < count; i++) {
    size_t len = 0;
    used += parseVar(offset + used, len);
    if (offset + used + len > size()) throw std::out_of_range("PSBuffer: truncated string list");
    strs.emplace_back(reinterpret_cast<const char*>(data() + offset + used), len);
    used += len;
  }
  return used;
}

// ---- messages: PSCredential = G1 G1 ; PSPubKey = G1 G2 G2 G1List G2List ; PSCredRequest = G1 Fr FrList StrList ;
//      IdProof = G1 G1 G2 G1 Fr FrList StrList [G1 G1]
PSBuffer PSCredential::toBufferString() {
  PSBuffer b;
  b.appendG1Element(sig1);
  b.appendG1Element(sig2);
  return b;
}
PSCredential PSCredential::fromBufferString(const PSBuffer& b) {
  PSCredential c;
  size_t off = b.parseG1Element(0, c.sig1);
  b.parseG1Element(off, c.sig2);
  return c;
}
PSBuffer PSPubKey::toBufferString() {
  PSBuffer b;
  b.appendG1Element(g);
  b.appendG2Element(gg);
  b.appendG2Element(XX);
  b.appendG1List(Yi);
  b.appendG2List(YYi);
  return b;
}
PSPubKey PSPubKey::fromBufferString(const PSBuffer& b) {
  PSPubKey k;
  size_t off = b.parseG1Element(0, k.g);
  off += b.parseG2Element(off, k.gg);
  off += b.parseG2Element(off, k.XX);
  off += b.parseG1List(off, k.Yi);
  b.parseG2List(off, k.YYi);
  return k;
}
PSBuffer PSCredRequest::toBufferString() {
  PSBuffer b;
  b.appendG1Element(A);
  b.appendFrElement(c);
  b.appendFrList(rs);
  b.appendStrList(attributes);
  return b;
}
PSCredRequest PSCredRequest::fromBufferString(const PSBuffer& b) {
  PSCredRequest r;
  size_t off = b.parseG1Element(0, r.A);
  off += b.parseFrElement(off, r.c);
  off += b.parseFrList(off, r.rs);
  b.parseStrList(off, r.attributes);
  return r;
}
PSBuffer IdProof::toBufferString() {
  PSBuffer b;
  b.appendG1Element(sig1);
  b.appendG1Element(sig2);
  b.appendG2Element(k);
  b.appendG1Element(phi);
  b.appendFrElement(c);
  b.appendFrList(rs);
  b.appendStrList(attributes);
  if (E1 && E2) {
    b.appendG1Element(*E1);
    b.appendG1Element(*E2);
  }
  return b;
}
IdProof IdProof::fromBufferString(const PSBuffer& b) {
  IdProof p;
  size_t off = b.parseG1Element(0, p.sig1);
  off += b.parseG1Element(off, p.sig2);
  off += b.parseG2Element(off, p.k);
  off += b.parseG1Element(off, p.phi);
  off += b.parseFrElement(off, p.c);
  off += b.parseFrList(off, p.rs);
  off += b.parseStrList(off, p.attributes);
  if (off < b.size()) {
    G1 e1, e2;
    off += b.parseG1Element(off, e1);
    b.parseG1Element(off, e2);
    p.E1 = e1;
    p.E2 = e2;
  }
  return p;
}
