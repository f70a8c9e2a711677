"""Sanitizer + fuzz pass over everything that parses untrusted bytes (sanitizers run on the CPU build only).

* tests/host_twin/fuzz_wire.cpp -- the DEVICE wire parser `WireSrc` (csrc/elp/pipeline.h: T-L-V walk, point decompression, attribute hashing; the code behind
  elp_verify_id_wire_batch) compiled for the host with -fsanitize=address,undefined and fed mutated golden `IdProof` messages, each in a heap block of exactly
  its own length.  Its verdict and the values it reads back through the accessors must equal a strict restatement of the documented format written here on top
  of the model's codec (oracle/pymodel.py), with zero sanitizer reports.  The reference's parser has undefined behaviour on such input
  (src/ps-encoding.cc:136-162, :377); an out-of-bounds read on the GPU would be silent.
* tests/host_twin/fuzz_psbuffer.cpp -- the host layer's `PSBuffer::parse*` / `fromBufferString` (csrc/host/ps-encoding.cc) on the same corpus: may throw, may
  not touch foreign memory.

Mutations: bit flips, byte overwrites, truncation at every offset, FD-length (3-byte) forms on every length byte, oversize counts and lengths, splices of
two messages, random garbage.  >= 10 000 cases per run; `hypothesis` drives extra mutation programs on top of the seeded bulk."""
import base64
import os
import random
import struct
import subprocess

import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from elp_testlib import BN254, ROOT, Mcl, load_golden

M = Mcl(BN254)
HT = os.path.join(ROOT, "tests", "host_twin")
CSRC = os.path.join(ROOT, "ps-signature-and-el-passo_amd", "csrc")
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
L = 32


def _build(out, cmd):
    srcs = [a for a in cmd if a.endswith((".cpp", ".cc", ".c"))]
    newest = max(os.path.getmtime(s) for s in srcs + [os.path.join(CSRC, "elp", "pipeline.h"), os.path.join(CSRC, "elp", "encode.h")])
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.check_call(cmd + ["-o", out])
    return out


@pytest.fixture(scope="module")
def fuzz_wire_bin(tmp_path_factory):
    out = os.path.join(HT, "fuzz_wire.san")
    return _build(out, ["g++"] + SAN + ["-I", CSRC, os.path.join(HT, "fuzz_wire.cpp")])


@pytest.fixture(scope="module")
def fuzz_psbuffer_bin():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libelp_oracle.so"])
    host = os.path.join(CSRC, "host")
    out = os.path.join(HT, "fuzz_psbuffer.san")
    return _build(out, ["g++"] + SAN + ["-I", host, os.path.join(HT, "fuzz_psbuffer.cpp"), os.path.join(host, "ps-encoding.cc"),
                                        os.path.join(host, "elp_mcl_compat.cc"), "-x", "c", os.path.join(ROOT, "tests", "cpu_shim", "elp_oracle_shim.c"),
                                        "-x", "none", "-L", os.path.join(ROOT, "oracle"), "-lelp_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])


# ---- strict restatement of what WireSrc::open accepts (pipeline.h "wire ingest" comment; encode.h g1_deserialize / g2_deserialize)
def _var(b, off):
    if off >= len(b):
        return None
    f = b[off]
    if f < 253:
        return f, off + 1
    if f == 253 and off + 2 < len(b):
        return (b[off + 1] << 8) | b[off + 2], off + 3
    return None


def _elem(b, off, typ, want):
    if off >= len(b) or b[off] != typ:
        return None
    v = _var(b, off + 1)
    if v is None or v[0] != want or v[1] + want > len(b):
        return None
    return bytes(b[v[1]:v[1] + want]), v[1] + want


def _g1_ok(bs):
    if bs == bytes(L):
        return True
    x = int.from_bytes(bs, "little") & ((1 << 255) - 1)
    return x < M.p and M.F.sqrt((x * x * x + M.cv.b) % M.p) is not None


def _g2_decode(bs):
    """None = rejected, else the canonical flag of the decoded point (False for infinity)"""
    if bs == bytes(2 * L):
        return False
    odd = bool(bs[2 * L - 1] & 0x80)
    xa = int.from_bytes(bs[:L], "little")
    xb = int.from_bytes(bs[L:], "little") & ((1 << 255) - 1)
    if xa >= M.p or xb >= M.p:
        return None
    F = M.F
    x = (xa, xb)
    y = F.f2_sqrt(F.f2_add(F.f2_mul(F.f2_sqr(x), x), F.b2))
    if y is None:
        return None
    if bool(y[0] & 1) != odd:
        y = F.f2_neg(y)
    return odd and y[0] != 0


def wire_parse(msg, A, retr):
    """None if the documented format rejects the message, else its fields (sig1, sig2, k, phi, c, rs, attributes, E1, E2, canonical flag of k) as raw bytes / integers"""
    b = bytes(msg)
    off = 0
    parts = []
    for typ, want in ((1, L), (1, L), (2, 2 * L), (1, L), (3, 32)):
        e = _elem(b, off, typ, want)
        if e is None:
            return None
        parts.append(e[0])
        off = e[1]
    s1, s2, kk, phi, cc = parts
    if off >= len(b) or b[off] != 6:
        return None
    v = _var(b, off + 1)
    if v is None or v[0] > 64:
        return None
    nrs, off = v
    rs = []
    for _ in range(nrs):
        v = _var(b, off)
        if v is None or v[0] != 32 or v[1] + 32 > len(b):
            return None
        r = int.from_bytes(b[v[1]:v[1] + 32], "little")
        if r >= M.r:
            return None
        rs.append(r)
        off = v[1] + 32
    if off >= len(b) or b[off] != 7:
        return None
    v = _var(b, off + 1)
    if v is None or v[0] != A:
        return None
    off = v[1]
    attrs = []
    for _ in range(A):
        v = _var(b, off)
        if v is None or v[1] + v[0] > len(b):
            return None
        attrs.append(b[v[1]:v[1] + v[0]])
        off = v[1] + v[0]
    H = sum(1 for a in attrs if len(a) == 0)
    if nrs != H + (2 if retr else 1) or H < (2 if retr else 1):
        return None
    e1 = e2 = None
    if retr:
        e = _elem(b, off, 1, L)
        if e is None:
            return None
        e1, off = e
        e = _elem(b, off, 1, L)
        if e is None:
            return None
        e2, off = e
    c = int.from_bytes(cc, "little")
    if c >= M.r:
        return None
    if retr and not (_g1_ok(e1) and _g1_ok(e2)):
        return None
    if not (_g1_ok(s1) and _g1_ok(s2) and _g1_ok(phi)):
        return None
    kflag = _g2_decode(kk)
    if kflag is None:
        return None
    return s1, s2, kk, phi, c, rs, attrs, e1, e2, kflag


def wire_open_model(msg, A, retr):
    """(accepted, digest) exactly as tests/host_twin/fuzz_wire.cpp reports them"""
    f = wire_parse(msg, A, retr)
    if f is None:
        return False, None
    s1, s2, kk, phi, c, rs, attrs, e1, e2, kflag = f
    d = bytearray(32)

    def fold(bs):
        for i, x in enumerate(bs):
            d[i & 31] ^= x
    for r in rs:
        fold(r.to_bytes(32, "little"))
    for a in attrs:
        if len(a):
            fold(M.fr_hash(bytes(a)).to_bytes(32, "little"))
    kc = bytearray(kk)
    kc[2 * L - 1] = (kc[2 * L - 1] & 0x7F) | (0x80 if kflag else 0)
    fold(kc)
    fold(phi)
    if retr:
        fold(e1)
        fold(e2)
    fold(c.to_bytes(32, "little"))
    return True, bytes(d)


# ---- corpus: the reference's own wire messages (tests/golden/*, produced by the reference's wasm)
def golden_messages():
    out = []   # (A, retr, bytes)
    d = load_golden("bn254_oracle_flows.json")
    for s in d["scenarios"]:
        for p in s["proofs"]:
            for c in p["cases"][:2]:
                out.append((s["A"], 0, base64.b64decode(c["proof"])))
    w = load_golden("bn254_oracle_with_retrieval.json")
    for r in w["runs"]:
        out.append((len(r["attr_values"]), 1, base64.b64decode(r["proof"])))
    return out


def length_bytes(msg, A, retr):
    """offsets of every length / count byte of a well-formed message (for the FD-form and oversize mutations)"""
    offs, off = [], 0
    for want in (L, L, 2 * L, L, 32):
        offs.append(off + 1)
        off += 2 + want
    offs.append(off + 1)          # FrList count
    n = msg[off + 1]
    off += 2
    for _ in range(n):
        offs.append(off)
        off += 33
    offs.append(off + 1)          # StrList count
    off += 2
    for _ in range(A):
        offs.append(off)
        off += 1 + msg[off]
    if retr:
        offs += [off + 1, off + 2 + L + 1]
    return offs


def mutate(rnd, base, other):
    A, retr, msg = base
    m = bytearray(msg)
    op = rnd.randrange(12)
    if op == 0 and len(m):                                        # bit flip(s)
        for _ in range(rnd.choice((1, 1, 1, 2, 5))):
            i = rnd.randrange(len(m))
            m[i] ^= 1 << rnd.randrange(8)
    elif op == 1:                                                 # truncate anywhere
        m = m[:rnd.randrange(len(m) + 1)]
    elif op == 2 and len(m):                                      # overwrite a byte with an interesting value
        m[rnd.randrange(len(m))] = rnd.choice((0, 1, 2, 3, 6, 7, 0x20, 0x40, 0x7F, 0x80, 252, 253, 254, 255))
    elif op == 3:                                                 # FD form of a length byte: FD 00 xx (accepted where the value still matches), FD hi lo
        o = rnd.choice(length_bytes(msg, A, retr))
        v = m[o]
        hi = rnd.choice((0, 0, 0, 1, 0xFF))
        m[o:o + 1] = bytes([253, hi, v])
    elif op == 4:                                                 # oversize / undersize count or length
        o = rnd.choice(length_bytes(msg, A, retr))
        m[o] = rnd.choice((0, 1, m[o] + 1 & 0xFF, max(0, m[o] - 1), 64, 65, 200, 252, 254, 255))
    elif op == 5:                                                 # splice: head of one message, tail of another
        o2 = other[2]
        m = m[:rnd.randrange(len(m) + 1)] + bytearray(o2[rnd.randrange(len(o2) + 1):])
    elif op == 6:                                                 # garbage of plausible length
        m = bytearray(rnd.getrandbits(8) for _ in range(rnd.choice((0, 1, 2, 3, 33, 34, 100, len(m)))))
    elif op == 7:                                                 # append bytes
        m += bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 40)))
    elif op == 8 and len(m) > 4:                                  # delete a slice
        i = rnd.randrange(len(m) - 1)
        del m[i:i + rnd.randrange(1, min(40, len(m) - i))]
    elif op == 9:                                                 # a field at its range edge: x = p, p - 1, r, r - 1, all ones
        vals = (M.p, M.p - 1, M.r, M.r - 1, (1 << 256) - 1, (1 << 255) - 1, 0, 1 << 255)
        offs = [2, 2 + L + 2, 2 * (L + 2) + 2, 2 * (L + 2) + 2 + L, 2 * (L + 2) + 2 * L + 2 + 2, 3 * (L + 2) + 2 * L + 2 + 2]
        o = rnd.choice(offs)
        if o + 32 <= len(m):
            m[o:o + 32] = rnd.choice(vals).to_bytes(32, "little")
    elif op == 10:                                                # unchanged (must accept) / wrong retrieval flag
        retr = rnd.choice((retr, retr, 1 - retr))
    else:                                                         # wrong attribute count for the key
        A = rnd.choice((A, A + 1, max(1, A - 1)))
    return A, retr, bytes(m)


def run_cases(binary, tmp, cases_by_A, tag):
    res = []
    for A, cases in cases_by_A.items():
        blob = struct.pack("<II", A, len(cases)) + b"".join(struct.pack("<IB", len(m), retr) + m for retr, m in cases)
        fin, fout = os.path.join(tmp, "%s_%d.in" % (tag, A)), os.path.join(tmp, "%s_%d.out" % (tag, A))
        open(fin, "wb").write(blob)
        r = subprocess.run([binary, fin, fout], capture_output=True, text=True, env=ENV, timeout=900)
        assert r.returncode == 0, "sanitizer report or crash (exit %d):\n%s" % (r.returncode, r.stderr[-4000:])
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
        res.append((A, cases, open(fout, "rb").read()))
    return res


def check_wire(binary, tmp, triples, tag):
    by_A = {}
    for A, retr, m in triples:
        by_A.setdefault(A, []).append((retr, m))
    acc = 0
    for A, cases, raw in run_cases(binary, tmp, by_A, tag):
        assert len(raw) == 33 * len(cases)
        for j, (retr, m) in enumerate(cases):
            ok, dig = wire_open_model(m, A, retr)
            got = raw[33 * j]
            assert got == int(ok), "parser verdict %d != model %d for A=%d retr=%d msg=%s" % (got, ok, A, retr, m.hex())
            if ok:
                acc += 1
                assert raw[33 * j + 1:33 * j + 33] == dig, "accessor read-back differs for A=%d retr=%d msg=%s" % (A, retr, m.hex())
    return acc


def test_wire_parser_fuzz_10k_cases_under_sanitizers(fuzz_wire_bin, tmp_path):
    rnd = random.Random(20241004)
    gold = golden_messages()
    assert len(gold) >= 20
    triples = [(A, retr, m) for A, retr, m in gold]                       # the untouched messages: all accepted
    for A, retr, m in gold[:6]:                                            # truncation at EVERY offset of a few messages
        triples += [(A, retr, m[:k]) for k in range(len(m))]
    for A, retr, m in gold[:8]:                                            # the 3-byte length form on EVERY length byte (value-preserving: FD 00 xx)
        for o in length_bytes(m, A, retr):
            triples.append((A, retr, m[:o] + bytes([253, 0, m[o]]) + m[o + 1:]))
    while len(triples) < 12000:
        triples.append(mutate(rnd, rnd.choice(gold), rnd.choice(gold)))
    acc = check_wire(fuzz_wire_bin, str(tmp_path), triples, "bulk")
    assert acc >= len(gold) + 50                                           # unchanged messages, FD 00 xx forms and harmless mutations are accepted
    assert len(triples) >= 10000


@settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
@given(st.lists(st.tuples(st.integers(0, 10**9), st.integers(0, 63)), min_size=40, max_size=40))
def test_wire_parser_fuzz_hypothesis(fuzz_wire_bin, tmp_path_factory, prog):
    """hypothesis chooses (seed, base message) pairs; every pair becomes a chain of 1-3 mutations"""
    gold = golden_messages()
    triples = []
    for seed, bi in prog:
        rnd = random.Random(seed)
        cur = gold[bi % len(gold)]
        for _ in range(1 + seed % 3):
            cur = mutate(rnd, cur, rnd.choice(gold)) if len(cur[2]) > 120 and _wellformed(cur) else mutate(rnd, gold[bi % len(gold)], rnd.choice(gold))
        triples.append(cur)
    check_wire(fuzz_wire_bin, str(tmp_path_factory.mktemp("hyp")), triples, "hyp")


def _wellformed(t):
    try:
        length_bytes(t[2], t[0], t[1])
        return wire_open_model(t[2], t[0], t[1])[0]
    except IndexError:
        return False


def test_host_psbuffer_parsers_under_sanitizers(fuzz_psbuffer_bin, tmp_path):
    rnd = random.Random(77)
    gold = golden_messages()
    d = load_golden("bn254_oracle_flows.json")
    # other message types the host layer parses: public keys and credential requests
    extra = [(s["A"], 0, base64.b64decode(s["pk"])) for s in d["scenarios"]]
    extra += [(s["A"], 0, base64.b64decode(rq["request"])) for s in d["scenarios"] for rq in s["requests"] if "request" in rq]
    pool = gold + extra
    triples = list(pool)
    for A, retr, m in pool[:4] + extra[:2]:
        triples += [(A, retr, m[:k]) for k in range(len(m))]
    while len(triples) < 10000:
        base = rnd.choice(pool)
        t = mutate(rnd, base, rnd.choice(pool)) if base in gold else (base[0], 0, _generic_mutation(rnd, base[2], rnd.choice(pool)[2]))
        triples.append(t)
    by_A = {0: [(retr, m) for _, retr, m in triples]}
    (_, cases, raw), = run_cases(fuzz_psbuffer_bin, str(tmp_path), by_A, "psb")
    assert len(raw) == len(cases)
    # the untouched golden proofs parse as IdProof (bit 0)
    for j in range(len(gold)):
        assert raw[j] & 1


def _generic_mutation(rnd, m, other):
    m = bytearray(m)
    op = rnd.randrange(6)
    if op == 0 and len(m):
        m[rnd.randrange(len(m))] ^= 1 << rnd.randrange(8)
    elif op == 1:
        m = m[:rnd.randrange(len(m) + 1)]
    elif op == 2 and len(m):
        m[rnd.randrange(len(m))] = rnd.choice((0, 1, 4, 5, 6, 7, 0x20, 0x40, 252, 253, 254, 255))
    elif op == 3:
        m = m[:rnd.randrange(len(m) + 1)] + bytearray(other[rnd.randrange(len(other) + 1):])
    elif op == 4 and len(m) > 4:
        i = rnd.randrange(len(m) - 1)
        del m[i:i + rnd.randrange(1, min(40, len(m) - i))]
    else:
        m += bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 40)))
    return bytes(m)
