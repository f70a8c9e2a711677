"""GPU tests of the host C++ protocol layer (csrc/host: PSSigner / PSRequester / PSVerifier with the reference's API over
the C-ABI).  Run on the MI355X box."""
import base64
import ctypes
import importlib
import os
import subprocess

import pytest

from elp_testlib import BN254, Codec, Mcl, Protocol, fb, load_golden, scalar_stream

pytestmark = pytest.mark.gpu
M = Mcl(BN254)
CD = Codec(M)
PR = Protocol(M)


@pytest.fixture(scope="module")
def host():
    b = importlib.import_module("ps-signature-and-el-passo_amd.build")
    L = ctypes.CDLL(b.build_host())
    L.elph_last_error.restype = ctypes.c_char_p
    c, cp, sz = ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t
    L.elph_verify_id_b64.argtypes = [cp, cp, cp, cp, c, cp, cp, cp]
    L.elph_user_name_b64.argtypes = [cp, cp, sz]
    L.elph_ps_verify_b64.argtypes = [cp, cp, cp]
    L.elph_prove_id_b64.argtypes = [cp, cp, cp, cp, cp, c, cp, cp, cp, cp, sz, cp, sz]
    L.elph_request_id_b64.argtypes = [cp, cp, cp, cp, sz, cp, sz]
    assert L.elph_init(0) == 0, L.elph_last_error()
    L.elph_set_strict_signature(0)      # bit-for-bit reference verdicts (golden case sig_both_zero); default-strict: test_default_strict_signature_through_reference_api below
    return L


def test_cpp_flow_tests():
    b = importlib.import_module("ps-signature-and-el-passo_amd.build")
    exe = b.build_cpp_tests()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL OK" in r.stdout


def test_cpp_flow_tests_on_bls12_381():
    """The same PSSigner / PSRequester / PSVerifier flows through the reference API with initPairing(BLS12_381): request -> issue ->
    unblind -> verify -> randomise -> prove (with and without id-retrieval) -> verify, wire round trips, negative cases."""
    b = importlib.import_module("ps-signature-and-el-passo_amd.build")
    exe = b.build_cpp_tests()
    r = subprocess.run([exe, "bls12_381"], capture_output=True, text=True, timeout=900)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL OK" in r.stdout


def test_golden_verdicts_through_reference_api(host):
    d = load_golden("bn254_oracle_flows.json")
    n = 0
    for s in d["scenarios"][:2] + d["scenarios"][4:]:
        for p in s["proofs"][:2]:
            for c in p["cases"]:
                got = host.elph_verify_id_b64(s["pk"].encode(), c["proof"].encode(), c["ad"].encode(), c["svc"].encode(), 0, b"", b"", b"")
                assert got in (0, 1), host.elph_last_error()
                assert bool(got) == c["expect"], (s["name"], c["label"])
                n += 1
            out = ctypes.create_string_buffer(512)
            assert host.elph_user_name_b64(p["cases"][0]["proof"].encode(), out, 512) > 0
            assert out.value.decode() == p["username"]
    assert n > 60
    w = load_golden("bn254_oracle_with_retrieval.json")
    for r in w["runs"]:
        assert host.elph_verify_id_b64(r["pk"].encode(), r["proof"].encode(), b"hello", b"service", 1, b"ghi", b"abc", b"jkl") == 1
        assert host.elph_verify_id_b64(r["pk"].encode(), r["proof"].encode(), b"hellO", b"service", 1, b"ghi", b"abc", b"jkl") == 0
        assert host.elph_verify_id_b64(r["pk"].encode(), r["proof"].encode(), b"hello", b"service", 1, b"ghi", b"abc", b"abc") == 0
    # reference-issued credentials verify as plain PS signatures
    s = d["scenarios"][0]
    spec = " ".join("%s N" % v for v in s["attr_values"]).encode()
    assert host.elph_ps_verify_b64(s["pk"].encode(), s["requests"][0]["unblinded"].encode(), spec) == 1
    assert host.elph_ps_verify_b64(s["pk"].encode(), s["requests"][0]["credential"].encode(), spec) == 0


@pytest.mark.parametrize("retr", [0, 1])
def test_host_prover_bit_exact_vs_oracle(host, retr):
    """el_passo_request_id / el_passo_prove_id with an injected random source must emit byte-identical wire messages to the
    oracle model, and the oracle must accept them."""
    seed, A, H = 777, 4, 2
    gg = CD.pk_decode(base64.b64decode(load_golden("bn254_oracle_flows.json")["scenarios"][0]["pk"])).gg
    g = M.hash_to_g1("abc")
    x = scalar_stream(seed, 0, M.r)
    ys = [scalar_stream(seed, 1 + i, M.r) for i in range(A)]
    pk, skX = PR.key_gen(g, gg, x, ys)
    pk_b64 = base64.b64encode(CD.pk_encode(pk))
    attrs = [(b"s-value", True), (b"gamma-value", True), (b"tp", False), (b"other", False)]
    spec = " ".join("%s %s" % (a.decode(), "Y" if h else "N") for a, h in attrs).encode()
    # request_id
    rnd = [scalar_stream(seed, 50 + j, M.r) for j in range(2 + H)]
    rq, t1 = PR.request_id(pk, attrs, b"ad1", rnd)
    out = ctypes.create_string_buffer(4096)
    n = host.elph_request_id_b64(pk_b64, spec, b"ad1", b"".join(fb(v) for v in rnd), len(rnd), out, 4096)
    assert n > 0, host.elph_last_error()
    assert base64.b64decode(out.value) == CD.req_encode(rq)
    # credential (oracle-issued), then prove
    cred = PR.unblind(PR.provide_id(pk, skX, rq, b"ad1", scalar_stream(seed, 99, M.r)), t1)
    cred_b64 = base64.b64encode(CD.cred_encode(cred))
    rnd = [scalar_stream(seed, 200 + j, M.r) for j in range(3 + H + 2)]
    apk, h = M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    if retr:
        want = PR.prove_id(pk, cred, attrs, b"sess", b"service", apk, g, h, rnd, with_retrieval=True)
        use = rnd
    else:
        use = rnd[:2] + rnd[3:3 + H + 1]
        want = PR.prove_id(pk, cred, attrs, b"sess", b"service", None, None, None, use, with_retrieval=False)
    n = host.elph_prove_id_b64(pk_b64, cred_b64, spec, b"sess", b"service", retr, b"ghi", b"abc", b"jkl", b"".join(fb(v) for v in use),
                               len(use), out, 4096)
    assert n > 0, host.elph_last_error()
    assert base64.b64decode(out.value) == CD.proof_encode(want)
    if retr:
        assert PR.verify_id(pk, want, b"sess", b"service", apk, g, h)
    else:
        assert PR.verify_id_noretr(pk, want, b"sess", b"service")
    assert host.elph_verify_id_b64(pk_b64, out.value, b"sess", b"service", retr, b"ghi", b"abc", b"jkl") == 1


def test_key_gen_from_equals_oracle(host):
    """a8: PSSigner::key_gen (src/ps-signer.cc:29-55) with injected secrets -> wire bytes equal the model's key."""
    d = load_golden("bn254_oracle_flows.json")
    for si, A in ((0, 3), (1, 8)):
        tpl_b64 = d["scenarios"][si]["pk"]
        tpl = CD.pk_decode(base64.b64decode(tpl_b64))
        x = scalar_stream(4242, 0, M.r)
        ys = [scalar_stream(4242, 1 + i, M.r) for i in range(A)]
        want, _ = PR.key_gen(tpl.g, tpl.gg, x, ys)
        out = ctypes.create_string_buffer(8192)
        host.elph_key_gen_b64.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        n = host.elph_key_gen_b64(tpl_b64.encode(), fb(x), b"".join(fb(y) for y in ys), A, out, 8192)
        assert n > 0, host.elph_last_error()
        assert base64.b64decode(out.value) == CD.pk_encode(want)


def test_unblind_and_randomize_equal_oracle(host):
    """a11: PSRequester::unblind_credential / randomize_credential (src/ps-requester.cc:101-113,139-148) on a reference-issued
    (blinded) credential with the RNG seam: byte-identical to the model; the unblinded one equals the reference's own unblinding
    only when t1 is the reference's, so here t1 is ours and the model is the comparison."""
    seed, A, H = 991, 4, 2
    gg = CD.pk_decode(base64.b64decode(load_golden("bn254_oracle_flows.json")["scenarios"][0]["pk"])).gg
    g = M.hash_to_g1("abc")
    x = scalar_stream(seed, 0, M.r)
    ys = [scalar_stream(seed, 1 + i, M.r) for i in range(A)]
    pk, skX = PR.key_gen(g, gg, x, ys)
    pk_b64 = base64.b64encode(CD.pk_encode(pk))
    attrs = [(b"s-value", True), (b"gamma-value", True), (b"tp", False), (b"other", False)]
    spec = " ".join("%s %s" % (a.decode(), "Y" if h else "N") for a, h in attrs).encode()
    rnd = [scalar_stream(seed, 50 + j, M.r) for j in range(2 + H + 1)]          # t1, rho_0, rho_j, then t of randomize
    rq, t1 = PR.request_id(pk, attrs, b"ad1", rnd[:2 + H])
    blinded = PR.provide_id(pk, skX, rq, b"ad1", scalar_stream(seed, 99, M.r))
    want_ub = PR.unblind(blinded, t1)
    want_rz = PR.randomize(want_ub, rnd[2 + H])
    assert PR.ps_verify(pk, want_ub, [a for a, _ in attrs]) and PR.ps_verify(pk, want_rz, [a for a, _ in attrs])
    o1, o2 = ctypes.create_string_buffer(512), ctypes.create_string_buffer(512)
    cp, sz = ctypes.c_char_p, ctypes.c_size_t
    host.elph_unblind_randomize_b64.argtypes = [cp, cp, cp, cp, sz, cp, cp, sz, cp, sz]
    rc = host.elph_unblind_randomize_b64(pk_b64, spec, b"ad1", b"".join(fb(v) for v in rnd), len(rnd),
                                         base64.b64encode(CD.cred_encode(blinded)), o1, 512, o2, 512)
    assert rc == 0, host.elph_last_error()
    assert base64.b64decode(o1.value) == CD.cred_encode(want_ub)
    assert base64.b64decode(o2.value) == CD.cred_encode(want_rz)


def test_default_strict_signature_through_reference_api(host):
    """Library default: sig1 == infinity is rejected by el_passo_verify_id (the reference accepts sig1 = sig2 = infinity)."""
    d = load_golden("bn254_oracle_flows.json")
    s = d["scenarios"][0]
    c = next(c for c in s["proofs"][0]["cases"] if c["label"] == "sig_both_zero")
    o = next(c for c in s["proofs"][0]["cases"] if c["label"] == "original")
    assert c["expect"] is True
    args = (s["pk"].encode(), c["proof"].encode(), c["ad"].encode(), c["svc"].encode(), 0, b"", b"", b"")
    assert host.elph_verify_id_b64(*args) == 1                      # module fixture: reference-compatible mode
    host.elph_set_strict_signature(1)
    try:
        assert host.elph_verify_id_b64(*args) == 0
        assert host.elph_verify_id_b64(s["pk"].encode(), o["proof"].encode(), o["ad"].encode(), o["svc"].encode(), 0, b"", b"", b"") == 1
    finally:
        host.elph_set_strict_signature(0)
