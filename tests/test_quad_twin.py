"""The FOUR-LANES-PER-ITEM layer (csrc/elp/quad.h, pair4.h; round 5) on the CPU: the host twin runs the four lanes of a quad as four threads with rendezvous
exchanges and compares every Fp12-level routine with the one-lane routine of tower.h / pairing.h on the same inputs (canonical words, bit for bit), and the
four-lane pairing check e(sig1, K) e(-sig2, gg) == 1 (src/ps-verifier.cc:31-34, 132-137) with the one-lane check and with the big-int model's verdict --
accepting and rejecting signatures, points at infinity.  BN254 (the parity curve) and BLS12-381 (against the model, itself pinned by tests/test_oracle_bls_golden.py)."""
import base64
import ctypes
import os
import random
import subprocess

import pytest

from elp_testlib import BLS12_381, BLS_G1, BLS_G2, BN254, Codec, Mcl, ROOT, fb, g1b, g2b, load_golden

_lib = None


def quad_twin(flags=("-O2",), name="libtwin_quad.so"):
    global _lib
    if os.environ.get("ELP_TWINQ_LIB"):          # a pre-built variant (tests/test_sanitized_arithmetic.py: the ASan / UBSan build)
        return ctypes.CDLL(os.environ["ELP_TWINQ_LIB"])
    so = os.path.join(ROOT, "tests", "host_twin", name)
    src = os.path.join(ROOT, "tests", "host_twin", "twin_quad.cpp")
    inc = os.path.join(ROOT, "ps-signature-and-el-passo_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(dp, f)) for dp, _, fs in os.walk(inc) for f in fs if f.endswith(".h"))
    newest = max(newest, os.path.getmtime(src))
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        objs, procs = [], []
        for curve in (0, 1):                 # the two curves compile in parallel
            obj = "%s.c%d.o" % (so, curve)
            objs.append(obj)
            procs.append(subprocess.Popen(["g++", "-std=c++17", "-fPIC", "-DTWINQ_CURVE=%d" % curve, "-I", inc] + list(flags) + ["-c", "-o", obj, src]))
        for p in procs:
            if p.wait() != 0:
                raise RuntimeError("quad twin build failed")
        subprocess.check_call(["g++", "-shared", "-o", so] + objs + ["-lpthread"])
        for o in objs:
            os.remove(o)
    return ctypes.CDLL(so)


@pytest.fixture(scope="module")
def L():
    return quad_twin()


CURVES = [("bn254", BN254, 32), ("bls", BLS12_381, 48)]
OPS = [(0, 0, "product"), (1, 0, "squaring"), (2, 0, "Granger-Scott squaring"), (3, 1, "one compressed squaring"), (3, 7, "seven compressed squarings"), (4, 1, "Frobenius"),
       (4, 2, "Frobenius^2"), (4, 3, "Frobenius^3"), (5, 0, "inverse"), (6, 0, "f^z, compressed chain"), (7, 0, "sparse line product"), (8, 0, "conjugate"), (9, 0, "f^|z|, GS chain")]


@pytest.mark.parametrize("name,curve,N", CURVES)
def test_fp12_routines_on_four_lanes_equal_one_lane(L, name, curve, N):
    m = Mcl(curve)
    rnd = random.Random(5)
    fn = getattr(L, "twinq_%s_op" % name)
    for op, n, what in OPS:
        for rep in range(2 if op in (6, 9) else 3):
            f = b"".join(fb(rnd.randrange(m.p), N) for _ in range(12))
            g = b"".join(fb(rnd.randrange(m.p), N) for _ in range(12))
            oq, op_ = ctypes.create_string_buffer(12 * N), ctypes.create_string_buffer(12 * N)
            assert fn(op, n, f, g, oq, op_) == 1, what
            assert oq.raw == op_.raw, (name, what, rep)
            assert any(oq.raw)


@pytest.mark.parametrize("name,curve,N", CURVES)
def test_pairing_check_on_four_lanes(L, name, curve, N):
    """PS-signature shaped inputs: sig1 = [u] g, sig2 = [u k] g with K = [k] gg verifies; tampering with either side, or a wrong K, does not; points at infinity
    contribute 1 (sig1 = sig2 = O accepts at this level -- the admissibility rule of sig1 sits above the pairing, pipeline.h sig1_admissible)."""
    m = Mcl(curve)
    G = m.G
    rnd = random.Random(11)
    if name == "bls":
        g1, gg = BLS_G1, BLS_G2
    else:                                     # the generators of a golden public key (the reference's own key material)
        pk = Codec(m).pk_decode(base64.b64decode(load_golden("bn254_oracle_flows.json")["scenarios"][0]["pk"]))
        g1, gg = pk.g, pk.gg
    fn = getattr(L, "twinq_%s_pair_check" % name)
    cases = []
    for _ in range(2):
        u, k = rnd.randrange(1, m.r), rnd.randrange(1, m.r)
        s1, K = G.g1_mul(g1, u), G.g2_mul(gg, k)
        s2 = G.g1_mul(s1, k)
        cases += [(s1, s2, K, 1), (s1, G.g1_mul(s2, 2), K, 0), (G.g1_mul(s1, 3), s2, K, 0), (s1, s2, G.g2_mul(K, 5), 0)]
    u = rnd.randrange(1, m.r)
    s1 = G.g1_mul(g1, u)
    cases += [(None, None, G.g2_mul(gg, 9), 1), (s1, None, G.g2_mul(gg, 9), 0), (None, s1, G.g2_mul(gg, 9), 0), (s1, None, None, 1), (s1, s1, gg, 1)]
    for s1, s2, K, want in cases:
        rc = fn(g1b(s1, N), g1b(s2, N), g2b(K, N), g2b(gg, N))
        assert rc == (want | (want << 1)), (name, rc, want)
