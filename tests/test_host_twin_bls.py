"""BLS12-381 instantiation of the device headers, checked on the CPU (host twin) against the big-int model.
There is no reference oracle for this curve (the reference runs on BN254): parity here means
"equal to the independent big-int model" plus algebraic known-answer properties (bilinearity, group order)."""
import ctypes
import random

import pytest

from elp_testlib import (BLS12_381, BLS_G1, BLS_G2, Codec, Mcl, Protocol, fb, g1_bases, g1b, g1u, g2_bases, g2b, g2u, hidden_mask, ib,
                         pack_provide_id, pack_ps_verify, pack_verify_id, scalar_stream, twin)

M = Mcl(BLS12_381)
PR = Protocol(M)
G = M.G
N = 48
W_TEST = 4


@pytest.fixture(scope="module")
def L():
    return twin()


def test_fp_and_hash(L):
    rnd = random.Random(1)
    p = M.p
    o = ctypes.create_string_buffer(N)
    for a, b in [(rnd.randrange(p), rnd.randrange(p)) for _ in range(200)] + [(p - 1, p - 1), (0, 5), (1, 1), (p - 1, 1)]:
        L.twin_bls_fp_mul(fb(a, N), fb(b, N), o)
        assert ib(o.raw) == a * b % p
    for _ in range(20):
        a = rnd.randrange(1, p)
        L.twin_bls_fp_inv(fb(a, N), o)
        assert ib(o.raw) == pow(a, -1, p)
        assert L.twin_bls_fp_sqrt(fb(a * a % p, N), o) and ib(o.raw) in (a, p - a)
    o2 = ctypes.create_string_buffer(2 * N)
    for s in ("abc", "ghi", "jkl", "service", ""):
        L.twin_bls_hash_to_g1(s.encode(), len(s), o2)
        P = M.hash_to_g1(s)
        assert g1u(o2.raw, N) == P and G.g1_on_curve(P) and G.g1_mul(P, M.r - 1) == G.g1_neg(P)


def test_group_ops(L):
    rnd = random.Random(2)
    o, o2 = ctypes.create_string_buffer(2 * N), ctypes.create_string_buffer(4 * N)
    for k in [0, 1, 2, 15, 16, M.r - 1, M.r, rnd.randrange(M.r), rnd.randrange(2**256)]:
        assert L.twin_bls_g1_mul(g1b(BLS_G1, N), fb(k), o) and g1u(o.raw, N) == G.g1_mul(BLS_G1, k)
    for k in [0, 1, 5, M.r - 1, rnd.randrange(M.r)]:
        assert L.twin_bls_g2_mul(g2b(BLS_G2, N), fb(k), o2) and g2u(o2.raw, N) == G.g2_mul(BLS_G2, k)
        assert L.twin_bls_g2_mul_gls_psi(g2b(BLS_G2, N), fb(k), o2) == 1 and g2u(o2.raw, N) == G.g2_mul(BLS_G2, k)      # GLS with the psi-images read from a table (WsTabPsi)
    P = G.g1_mul(BLS_G1, 5)
    for a, b in [(P, BLS_G1), (P, P), (P, G.g1_neg(P)), (P, None), (None, None)]:
        assert L.twin_bls_g1_add(g1b(a, N), g1b(b, N), o) and g1u(o.raw, N) == G.g1_add(a, b)
    Q = G.g2_mul(BLS_G2, 7)
    for a, b in [(Q, BLS_G2), (Q, Q), (Q, G.g2_neg(Q))]:
        assert L.twin_bls_g2_add(g2b(a, N), g2b(b, N), o2) and g2u(o2.raw, N) == G.g2_add(a, b)
    assert L.twin_bls_g1_decompress(M.g1_ser(G.g1_neg(P)), o) and g1u(o.raw, N) == G.g1_neg(P)
    assert L.twin_bls_g2_decompress(M.g2_ser(G.g2_neg(Q)), o2) and g2u(o2.raw, N) == G.g2_neg(Q)


def test_pairing_equals_model_and_is_bilinear(L):
    P, Q = G.g1_mul(BLS_G1, 12345), G.g2_mul(BLS_G2, 6789)
    og = ctypes.create_string_buffer(12 * N)
    e = G.pairing(P, Q)
    want = b"".join(fb(e[k][0], N) + fb(e[k][1], N) for k in [0, 2, 4, 1, 3, 5])
    assert L.twin_bls_pairing(g1b(P, N), g2b(Q, N), og, 0) == 1 and og.raw == want
    assert L.twin_bls_pairing(g1b(P, N), g2b(Q, N), og, 2) == 1
    assert L.twin_bls_pairing_fixedq(g1b(P, N), g2b(Q, N), og) == 1 and og.raw == want
    # e(aP, Q) == e(P, aQ)
    a = 998877665544332211
    og2 = ctypes.create_string_buffer(12 * N)
    L.twin_bls_pairing(g1b(G.g1_mul(P, a), N), g2b(Q, N), og, 0)
    L.twin_bls_pairing(g1b(P, N), g2b(G.g2_mul(Q, a), N), og2, 0)
    assert og.raw == og2.raw


def test_protocol_flows(L):
    seed, A, H = 4242, 4, 2
    g, gg = M.hash_to_g1("abc"), BLS_G2
    pk, skX = PR.key_gen(g, gg, scalar_stream(seed, 0, M.r), [scalar_stream(seed, 1 + i, M.r) for i in range(A)])
    apk, h = M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    ctx = ctypes.c_void_p(L.twin_bls_ctx_new(A, W_TEST, g1_bases(M, pk, svc="service", g_eg=g, apk=apk, h=h, skX=skX), g2_bases(M, pk)))
    assert ctx.value
    attrs = [(b"s-value", True), (b"gamma-value", True), (b"tp", False), (b"other", False)]
    rq, t1 = PR.request_id(pk, attrs, b"ad", [scalar_stream(seed, 50 + j, M.r) for j in range(2 + H)])
    u = scalar_stream(seed, 99, M.r)
    want = PR.provide_id(pk, skX, rq, b"ad", u)
    out = ctypes.create_string_buffer(4 * N)
    assert L.twin_bls_provide_id(ctx, pack_provide_id(M, rq, u), ctypes.c_uint64(3), b"ad", 2, out) == 1
    assert out.raw == g1b(want.sig1, N) + g1b(want.sig2, N)
    assert L.twin_bls_provide_id(ctx, pack_provide_id(M, rq, u), ctypes.c_uint64(3), b"ae", 2, out) == 0
    cred = PR.unblind(want, t1)
    vals = [a for a, _ in attrs]
    assert PR.ps_verify(pk, cred, vals)
    assert L.twin_bls_ps_verify(ctx, pack_ps_verify(M, cred, vals), A) == 1
    assert L.twin_bls_ps_verify(ctx, pack_ps_verify(M, want, vals), A) == 0
    rnd = [scalar_stream(seed, 200 + j, M.r) for j in range(3 + H + 2)]
    pr = PR.prove_id(pk, cred, attrs, b"sess", b"service", apk, g, h, rnd)
    assert PR.verify_id(pk, pr, b"sess", b"service", apk, g, h)
    mask = ctypes.c_uint64(hidden_mask(pr.attributes))
    assert L.twin_bls_verify_id(ctx, pack_verify_id(M, pr), mask, 1, b"sess", 4) == 1
    assert L.twin_bls_verify_id(ctx, pack_verify_id(M, pr), mask, 1, b"sesS", 4) == 0
    import copy
    bad = copy.copy(pr)
    bad.sig2 = G.g1_add(pr.sig2, g)
    assert not PR.verify_id(pk, bad, b"sess", b"service", apk, g, h)
    assert L.twin_bls_verify_id(ctx, pack_verify_id(M, bad), mask, 1, b"sess", 4) == 0
    pr2 = PR.prove_id(pk, cred, attrs, b"sess", b"service", None, None, None, rnd[:2] + rnd[3:3 + H + 1], with_retrieval=False)
    assert PR.verify_id_noretr(pk, pr2, b"sess", b"service")
    assert L.twin_bls_verify_id(ctx, pack_verify_id(M, pr2), mask, 0, b"sess", 4) == 1
    # the same verdicts from the two-lanes-per-item layout (two threads here, every exchange a rendezvous)
    for fn in (L.twin_blsp_verify_id, L.twin_blsp_verify_id_g1split, L.twin_bls_verify_id_jobs4):      # g1split: ELP_OPT_SPLIT_PHASES = 3; jobs4: the small-batch form (k_vid_nizk4)
        assert fn(ctx, pack_verify_id(M, pr), mask, 1, b"sess", 4) == 1
        assert fn(ctx, pack_verify_id(M, pr), mask, 1, b"sesS", 4) == 0
        assert fn(ctx, pack_verify_id(M, bad), mask, 1, b"sess", 4) == 0
        assert fn(ctx, pack_verify_id(M, pr2), mask, 0, b"sess", 4) == 1
    assert L.twin_blsp_ps_verify(ctx, pack_ps_verify(M, cred, vals), A) == 1
    assert L.twin_blsp_ps_verify(ctx, pack_ps_verify(M, want, vals), A) == 0
    P, Q = G.g1_mul(BLS_G1, 321), G.g2_mul(BLS_G2, 654)
    og, og2 = ctypes.create_string_buffer(12 * N), ctypes.create_string_buffer(12 * N)
    assert L.twin_blsp_pairing(g1b(P, N), g2b(Q, N), og) == 1 and L.twin_bls_pairing(g1b(P, N), g2b(Q, N), og2, 0) == 1 and og.raw == og2.raw
