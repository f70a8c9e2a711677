"""Shared test helpers: byte packing of records for the C-ABI / host twin from oracle-model objects."""
import base64
import ctypes
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.pymodel import BLS12_381, BN254, Codec, Mcl, Protocol, scalar_stream  # noqa: E402

# standard BLS12-381 generators (zkcrypto / IETF pairing-friendly-curves draft)
BLS_G1 = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
          0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)
BLS_G2 = ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
           0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
          (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
           0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def fb(x, n=32):
    return int(x).to_bytes(n, "little")


def ib(b):
    return int.from_bytes(b, "little")


def g1b(P, n=32):
    return bytes(2 * n) if P is None else fb(P[0], n) + fb(P[1], n)


def g2b(P, n=32):
    return bytes(4 * n) if P is None else fb(P[0][0], n) + fb(P[0][1], n) + fb(P[1][0], n) + fb(P[1][1], n)


def g1u(b, n=32):
    x, y = ib(b[:n]), ib(b[n:2 * n])
    return None if x == 0 and y == 0 else (x, y)


def g2u(b, n=32):
    v = [ib(b[n * i:n * i + n]) for i in range(4)]
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))


def hidden_mask(attributes):
    """bit i set <=> attributes[i] is hidden (empty placeholder)."""
    m = 0
    for i, a in enumerate(attributes):
        if len(a) == 0:
            m |= 1 << i
    return m


def pack_verify_id(m, pr):
    """sig1 | sig2 | phi | [E1 | E2] | k | c | rs | m_revealed   (csrc/elp/pipeline.h verify_id_item)."""
    n = m.fb
    out = g1b(pr.sig1, n) + g1b(pr.sig2, n) + g1b(pr.phi, n)
    if pr.has_E:
        out += g1b(pr.E1, n) + g1b(pr.E2, n)
    out += g2b(pr.k, n) + fb(pr.c)
    for r in pr.rs:
        out += fb(r)
    for a in pr.attributes:
        if len(a):
            out += fb(m.fr_hash(bytes(a)))
    return out


def pack_ps_verify(m, cred, all_attributes):
    out = g1b(cred.sig1, m.fb) + g1b(cred.sig2, m.fb)
    for a in all_attributes:
        out += fb(m.fr_hash(a.encode() if isinstance(a, str) else bytes(a)))
    return out


def pack_provide_id(m, rq, u):
    out = g1b(rq.A, m.fb) + fb(rq.c)
    for r in rq.rs:
        out += fb(r)
    for a in rq.attributes:
        if len(a):
            out += fb(m.fr_hash(bytes(a)))
    return out + fb(u)


def pack_request_id(m, attrs, rnd):
    """m[A] | t | rho_0 | rho[H]   (csrc/elp/pipeline.h request_id_item); rnd in the order Protocol.request_id draws it."""
    return b"".join(fb(m.fr_hash(bytes(a))) for a, _ in attrs) + b"".join(fb(x) for x in rnd)


def pack_prove_id(m, cred, attrs, rnd):
    """sig1 | sig2 | m[A] | t | r | [eps] | rho[H] | rho_t | [rho_eps]   (csrc/elp/pipeline.h prove_id_item)."""
    return g1b(cred.sig1, m.fb) + g1b(cred.sig2, m.fb) + b"".join(fb(m.fr_hash(bytes(a))) for a, _ in attrs) + b"".join(fb(x) for x in rnd)


def g1_bases(m, pk, svc=None, g_eg=None, apk=None, h=None, skX=None):
    """0 = g, 1+i = Y_i, A+1 = H1(service), A+2 = g_eg, A+3 = authority_pk, A+4 = h, A+5 = X."""
    hs = m.hash_to_g1(svc) if svc is not None else None
    pts = [pk.g] + list(pk.Yi) + [hs, g_eg, apk, h, skX]
    return b"".join(g1b(P, m.fb) for P in pts)


def g2_bases(m, pk):
    return b"".join(g2b(P, m.fb) for P in [pk.gg, pk.XX] + list(pk.YYi))


_twin = None


def build_twin(so, flags):
    """g++ build of tests/host_twin/twin.cpp: one object per curve, compiled in parallel, linked into `so`."""
    import subprocess
    src = os.path.join(ROOT, "tests", "host_twin", "twin.cpp")
    inc = os.path.join(ROOT, "ps-signature-and-el-passo_amd", "csrc")
    objs, procs = [], []
    for part in (1, 2):
        obj = "%s.part%d.o" % (so, part)
        objs.append(obj)
        procs.append(subprocess.Popen(["g++", "-std=c++17", "-fPIC", "-DTWIN_PART=%d" % part, "-I", inc] + list(flags) + ["-c", "-o", obj, src]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("host twin build failed")
    subprocess.check_call(["g++", "-shared", "-o", so] + objs + ["-lpthread"])
    for o in objs:
        os.remove(o)


def twin():
    """Host (g++) build of the device headers — test-only library, built on demand."""
    global _twin
    if _twin is None and os.environ.get("ELP_TWIN_LIB"):       # a pre-built variant of the twin (tests/test_sanitized_arithmetic.py: the ASan / UBSan build)
        _twin = ctypes.CDLL(os.environ["ELP_TWIN_LIB"])
        for n in ("twin_bn254_ctx_new", "twin_bls_ctx_new"):
            getattr(_twin, n).restype = ctypes.c_void_p
    if _twin is None:
        so = os.path.join(ROOT, "tests", "host_twin", "libtwin.so")
        src = os.path.join(ROOT, "tests", "host_twin", "twin.cpp")
        inc = os.path.join(ROOT, "ps-signature-and-el-passo_amd", "csrc")
        newest = max(os.path.getmtime(os.path.join(dp, f)) for dp, _, fs in os.walk(inc) for f in fs if f.endswith(".h"))
        newest = max(newest, os.path.getmtime(src))
        if not os.path.exists(so) or os.path.getmtime(so) < newest:
            build_twin(so, ["-O2"])
        _twin = ctypes.CDLL(so)
        for n in ("twin_bn254_ctx_new", "twin_bls_ctx_new"):
            getattr(_twin, n).restype = ctypes.c_void_p
    return _twin


_oracle = None


def oracle():
    """The C oracle (oracle/elp_oracle.c), built on demand with the recipe in oracle/Makefile."""
    global _oracle
    if _oracle is None:
        import subprocess
        od = os.path.join(ROOT, "oracle")
        so = os.path.join(od, "libelp_oracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(od, "elp_oracle.c")):
            subprocess.check_call(["make", "-C", od, "-s"])
        L = ctypes.CDLL(so)
        L.elpo_key_new.restype = ctypes.c_void_p
        L.elpo_key_new.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
        L.elpo_key_free.argtypes = [ctypes.c_void_p]
        L.elpo_verify_id.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        L.elpo_ps_verify.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.elpo_provide_id.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.elpo_verify_id_batch.restype = ctypes.c_long
        L.elpo_verify_id_batch.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64,
                                           ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
        L.elpo_hash_to_g1.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.elpo_fr_set_hash_of.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.elpo_init()
        _oracle = L
    return _oracle


_oracle_bls = None


def oracle_bls():
    """The C oracle built for BLS12-381 (oracle/elp_oracle.c -DELPO_BLS12_381 -> libelp_oracle_bls.so): same entry points, 48-byte coordinates.
    Pinned to the reference's own wasm run on this curve by tests/test_oracle_bls_golden.py (tests/golden/bls12_381_*.json)."""
    global _oracle_bls
    if _oracle_bls is None:
        import subprocess
        od = os.path.join(ROOT, "oracle")
        so = os.path.join(od, "libelp_oracle_bls.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(od, "elp_oracle.c")):
            subprocess.check_call(["make", "-C", od, "-s"])
        L = ctypes.CDLL(so)
        L.elpo_key_new.restype = ctypes.c_void_p
        L.elpo_key_new.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
        L.elpo_key_free.argtypes = [ctypes.c_void_p]
        L.elpo_verify_id.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        L.elpo_ps_verify.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.elpo_provide_id.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.elpo_verify_id_batch.restype = ctypes.c_long
        L.elpo_verify_id_batch.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64,
                                           ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
        L.elpo_hash_to_g1.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.elpo_fr_set_hash_of.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.elpo_init()
        assert L.elpo_curve() == 1
        _oracle_bls = L
    return _oracle_bls


def oracle_key(m, pk, **kw):
    L = oracle()
    h = L.elpo_key_new(len(pk.Yi), g1_bases(m, pk, **kw), g2_bases(m, pk))
    assert h, "oracle rejected the key"
    return ctypes.c_void_p(h)


class OracleBackedCtx:
    """Stand-in for elpasso.Context in CPU tests of host-side logic (synthetic generator, host layer): the same method
    surface, every group operation answered by the C oracle.  Lives in tests/ only."""

    F, G1, G2, GT = 32, 64, 128, 384
    curve = 0

    def __init__(self):
        self.L = oracle()
        self.b1 = {}
        self.b2 = {}
        self.A = 0

    def hash_to_g1(self, msgs):
        out = b""
        o = ctypes.create_string_buffer(64)
        for m_ in msgs:
            self.L.elpo_hash_to_g1(bytes(m_), len(m_), o)
            out += o.raw
        return out

    def _mul(self, fn, sz, pts, ks):
        out = b""
        o = ctypes.create_string_buffer(sz)
        for i in range(len(pts) // sz):
            assert fn(pts[i * sz:(i + 1) * sz], ks[32 * i:32 * i + 32], o)
            out += o.raw
        return out

    def g1_mul(self, pts, ks):
        return self._mul(self.L.elpo_g1_mul, 64, pts, ks)

    def g2_mul(self, pts, ks):
        return self._mul(self.L.elpo_g2_mul, 128, pts, ks)

    def set_pubkey(self, g, gg, XX, Yi, YYi, window_bits=0):
        A = len(Yi) // 64
        self.A = A
        self.b1 = {0: g}
        for i in range(A):
            self.b1[1 + i] = Yi[64 * i:64 * i + 64]
        self.b2 = {0: gg, 1: XX}
        for i in range(A):
            self.b2[2 + i] = YYi[128 * i:128 * i + 128]

    def set_rp(self, service, apk=None, g=None, h=None):
        A = self.A
        self.b1[A + 1] = self.hash_to_g1([service])
        self.b1[A + 2], self.b1[A + 3], self.b1[A + 4] = g or bytes(64), apk or bytes(64), h or bytes(64)

    def set_signer_secret(self, X):
        self.b1[self.A + 5] = X

    def _msm(self, bases, mulfn, addfn, sz, ids, scalars):
        nt = len(ids)
        n = len(scalars) // (32 * nt)
        out = b""
        o = ctypes.create_string_buffer(sz)
        for i in range(n):
            acc = bytes(sz)
            for t, b in enumerate(ids):
                k = scalars[32 * (i * nt + t):32 * (i * nt + t) + 32]
                assert mulfn(bases[b], k, o)
                term = o.raw
                assert addfn(acc, term, o)
                acc = o.raw
            out += acc
        return out

    def g1_msm_fixed(self, ids, scalars):
        return self._msm(self.b1, self.L.elpo_g1_mul, self.L.elpo_g1_add, 64, ids, scalars)

    def g2_msm_fixed(self, ids, scalars):
        return self._msm(self.b2, self.L.elpo_g2_mul, self.L.elpo_g2_add, 128, ids, scalars)

    def key_handle(self):
        A = self.A
        g1 = b"".join(self.b1.get(i, bytes(64)) for i in range(A + 6))
        g2 = b"".join(self.b2[i] for i in range(A + 2))
        h = self.L.elpo_key_new(A, g1, g2)
        assert h
        return ctypes.c_void_p(h)
