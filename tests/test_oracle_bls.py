"""The C oracle's BLS12-381 build (oracle/elp_oracle.c -DELPO_BLS12_381) against the independent big-int model (oracle/pymodel.py).
These tests establish that TWO independently written implementations (C: 6 x 64-bit Montgomery limbs, Jacobian lines, HHT chain; Python:
big integers, affine formulas, plain exponentiation) agree bit for bit on primitives, GT values, issued signatures and verdicts --
the pair then serves as the checker of the HIP path's BLS12-381 instantiation (tests/test_gpu_bls.py) and as its CPU baseline.  Both are
pinned to the reference's own wasm run on this curve in tests/test_oracle_bls_golden.py; the subgroup / strict-signature tests below state
the library's POLICY on points outside G1, which differs from the reference's behaviour on purpose (same file)."""
import copy
import ctypes
import random

from elp_testlib import (BLS12_381, BLS_G1, BLS_G2, Mcl, Protocol, fb, g1_bases, g1b, g1u, g2_bases, g2b, g2u, hidden_mask, oracle_bls,
                         pack_provide_id, pack_ps_verify, pack_verify_id, scalar_stream)

M = Mcl(BLS12_381)
PR = Protocol(M)
G = M.G
N = 48


def test_primitives_equal_the_model():
    L = oracle_bls()
    rnd = random.Random(5)
    o, o2 = ctypes.create_string_buffer(2 * N), ctypes.create_string_buffer(4 * N)
    for k in [0, 1, 2, 31, 32, M.r - 1, M.r, rnd.randrange(M.r), rnd.randrange(2**256)]:
        assert L.elpo_g1_mul(g1b(BLS_G1, N), fb(k), o) and g1u(o.raw, N) == G.g1_mul(BLS_G1, k)
    for k in [0, 1, 7, M.r - 1, rnd.randrange(M.r)]:
        assert L.elpo_g2_mul(g2b(BLS_G2, N), fb(k), o2) and g2u(o2.raw, N) == G.g2_mul(BLS_G2, k)
    P, Q = G.g1_mul(BLS_G1, 5), G.g2_mul(BLS_G2, 7)
    for a, b in [(P, BLS_G1), (P, P), (P, G.g1_neg(P)), (P, None), (None, None)]:
        assert L.elpo_g1_add(g1b(a, N), g1b(b, N), o) and g1u(o.raw, N) == G.g1_add(a, b)
    for a, b in [(Q, BLS_G2), (Q, Q), (Q, G.g2_neg(Q))]:
        assert L.elpo_g2_add(g2b(a, N), g2b(b, N), o2) and g2u(o2.raw, N) == G.g2_add(a, b)
    # wire encodings both ways
    w1, w2 = ctypes.create_string_buffer(N), ctypes.create_string_buffer(2 * N)
    for X in (P, G.g1_neg(P)):
        assert L.elpo_g1_compress(g1b(X, N), w1) and w1.raw == M.g1_ser(X)
        assert L.elpo_g1_decompress(M.g1_ser(X), o) and g1u(o.raw, N) == X
    for X in (Q, G.g2_neg(Q)):
        assert L.elpo_g2_compress(g2b(X, N), w2) and w2.raw == M.g2_ser(X)
        assert L.elpo_g2_decompress(M.g2_ser(X), o2) and g2u(o2.raw, N) == X
    assert not L.elpo_g1_mul(g1b((1, 1), N), fb(3), o)               # not on the curve
    # hashes: Fr::setHashOf masking to 255 / 254 bits, and mcl's hashAndMapToG1 for the curve (SHA-512 setHashOf, SvdW, cofactor)
    h = ctypes.create_string_buffer(32)
    for s in (b"", b"abc", b"attribute-7", b"x" * 100):
        L.elpo_fr_set_hash_of(s, len(s), h)
        assert int.from_bytes(h.raw, "little") == M.fr_hash(s)
    for s in ("abc", "ghi", "jkl", "service", ""):
        L.elpo_hash_to_g1(s.encode(), len(s), o)
        assert g1u(o.raw, N) == M.hash_to_g1(s)


def test_pairing_value_bilinearity_and_final_exponentiation_self_check():
    L = oracle_bls()
    P, Q = G.g1_mul(BLS_G1, 12345), G.g2_mul(BLS_G2, 6789)
    og, og2 = ctypes.create_string_buffer(12 * N), ctypes.create_string_buffer(12 * N)
    e = G.pairing(P, Q)
    want = b"".join(fb(e[k][0], N) + fb(e[k][1], N) for k in [0, 2, 4, 1, 3, 5])
    assert L.elpo_pairing(g1b(P, N), g2b(Q, N), og) == 1 and og.raw == want       # GT bytes equal the model's (model: plain exponentiation)
    a = 998877665544332211
    L.elpo_pairing(g1b(G.g1_mul(P, a), N), g2b(Q, N), og)
    L.elpo_pairing(g1b(P, N), g2b(G.g2_mul(Q, a), N), og2)
    assert og.raw == og2.raw
    # the HHT chain of the hard part == plain square-and-multiply by the 1268-bit integer (p^4 - p^2 + 1) / r
    assert L.elpo_selftest_final_exp(g1b(P, N), g2b(Q, N)) == 1
    # e(P, O) = e(O, Q) = 1
    one = fb(1, N) + bytes(11 * N)
    assert L.elpo_pairing(g1b(None, N), g2b(Q, N), og) == 1 and og.raw == one


def test_protocol_flows_equal_the_model():
    L = oracle_bls()
    seed, A, H = 777, 4, 2
    g, gg = M.hash_to_g1("abc"), BLS_G2
    pk, skX = PR.key_gen(g, gg, scalar_stream(seed, 0, M.r), [scalar_stream(seed, 1 + i, M.r) for i in range(A)])
    apk, h = M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    key = ctypes.c_void_p(L.elpo_key_new(A, g1_bases(M, pk, svc="service", g_eg=g, apk=apk, h=h, skX=skX), g2_bases(M, pk)))
    assert key.value
    attrs = [(b"s-value", True), (b"gamma-value", True), (b"tp", False), (b"other", False)]
    rq, t1 = PR.request_id(pk, attrs, b"ad", [scalar_stream(seed, 50 + j, M.r) for j in range(2 + H)])
    u = scalar_stream(seed, 99, M.r)
    want = PR.provide_id(pk, skX, rq, b"ad", u)
    out = ctypes.create_string_buffer(4 * N)
    assert L.elpo_provide_id(key, pack_provide_id(M, rq, u), 3, b"ad", 2, out) == 1
    assert out.raw == g1b(want.sig1, N) + g1b(want.sig2, N)                       # issued signature, bit for bit
    assert L.elpo_provide_id(key, pack_provide_id(M, rq, u), 3, b"ae", 2, out) == 0
    cred = PR.unblind(want, t1)
    vals = [a for a, _ in attrs]
    assert L.elpo_ps_verify(key, pack_ps_verify(M, cred, vals), A) == 1
    assert L.elpo_ps_verify(key, pack_ps_verify(M, want, vals), A) == 0
    rnd = [scalar_stream(seed, 200 + j, M.r) for j in range(3 + H + 2)]
    pr = PR.prove_id(pk, cred, attrs, b"sess", b"service", apk, g, h, rnd)
    mask = hidden_mask(pr.attributes)
    assert PR.verify_id(pk, pr, b"sess", b"service", apk, g, h)
    assert L.elpo_verify_id(key, pack_verify_id(M, pr), mask, 1, b"sess", 4) == 1
    assert L.elpo_verify_id(key, pack_verify_id(M, pr), mask, 1, b"sesS", 4) == 0
    for tamper in ("sig2", "phi", "c", "rs"):
        bad = copy.copy(pr)
        if tamper == "sig2":
            bad.sig2 = G.g1_add(pr.sig2, g)
        elif tamper == "phi":
            bad.phi = G.g1_add(pr.phi, g)
        elif tamper == "c":
            bad.c = pr.c ^ 1
        else:
            bad.rs = [pr.rs[0] ^ 1] + list(pr.rs[1:])
        assert L.elpo_verify_id(key, pack_verify_id(M, bad), mask, 1, b"sess", 4) == int(PR.verify_id(pk, bad, b"sess", b"service", apk, g, h)) == 0
    pr2 = PR.prove_id(pk, cred, attrs, b"sess", b"service", None, None, None, rnd[:2] + rnd[3:3 + H + 1], with_retrieval=False)
    assert L.elpo_verify_id(key, pack_verify_id(M, pr2), mask, 0, b"sess", 4) == 1
    L.elpo_key_free(key)


def _small_order_point(order, seed=1):
    """A point of E(Fp) of the given prime order dividing the G1 cofactor (z-1)^2/3 = 3 * 11^2 * 10177^2 * 859267^2 * 52437899^2."""
    # E(Fp) = Z/((z-1)/3) x Z/((z-1) r): the exponent of the group is |z-1| r, so [|z-1| r / order] R has order `order` or is O
    h1 = (BLS12_381.z - 1) ** 2 // 3
    assert h1 % order == 0
    n = abs(BLS12_381.z - 1) * M.r
    x = seed
    while True:
        rhs = (x * x * x + 4) % M.p
        y = pow(rhs, (M.p + 1) // 4, M.p)
        if y * y % M.p == rhs:
            T = G.g1_mul_plain((x, y), n // order)
            if T is not None:
                assert G.g1_mul_plain(T, order) is None and G.g1_on_curve(T)
                return T
        x += 1


def test_off_subgroup_g1_inputs_are_rejected():
    """Project policy on BLS12-381 (G1 cofactor != 1): phi, E1, E2 (and the commitment of a request) outside the order-r subgroup reject the item -- in the
    model ([r]P == O), in the C oracle ([r]P by double-and-add) and in the HIP path's endomorphism test (host twin), on crafted inputs: a valid proof whose
    phi / E1 / E2 got a component of order 3 or 11 added."""
    from elp_testlib import twin
    L, T = oracle_bls(), twin()
    seed, A, H = 991, 4, 2
    g, gg = M.hash_to_g1("abc"), BLS_G2
    pk, skX = PR.key_gen(g, gg, scalar_stream(seed, 0, M.r), [scalar_stream(seed, 1 + i, M.r) for i in range(A)])
    apk, h = M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    bases = (g1_bases(M, pk, svc="service", g_eg=g, apk=apk, h=h, skX=skX), g2_bases(M, pk))
    key = ctypes.c_void_p(L.elpo_key_new(A, *bases))
    tctx = ctypes.c_void_p(T.twin_bls_ctx_new(A, 4, *bases))
    assert key.value and tctx.value
    t3, t11 = _small_order_point(3), _small_order_point(11, seed=7)
    # the membership test itself: G1 points pass, cofactor-order points and mixtures fail
    for P, want in [(BLS_G1, 1), (G.g1_mul(BLS_G1, 123456789), 1), (g, 1), (None, 1), (t3, 0), (t11, 0), (G.g1_add(BLS_G1, t3), 0), (G.g1_add(g, t11), 0)]:
        assert T.twin_bls_g1_in_subgroup(g1b(P, N)) == want
        assert PR.in_g1(P) == bool(want)
    attrs = [(b"s-value", True), (b"gamma-value", True), (b"tp", False), (b"other", False)]
    rq, t1 = PR.request_id(pk, attrs, b"ad", [scalar_stream(seed, 50 + j, M.r) for j in range(2 + H)])
    u = scalar_stream(seed, 99, M.r)
    cred = PR.unblind(PR.provide_id(pk, skX, rq, b"ad", u), t1)
    rnd = [scalar_stream(seed, 200 + j, M.r) for j in range(3 + H + 2)]
    pr = PR.prove_id(pk, cred, attrs, b"sess", b"service", apk, g, h, rnd)
    mask = hidden_mask(pr.attributes)
    rec = pack_verify_id(M, pr)
    assert L.elpo_verify_id(key, rec, mask, 1, b"sess", 4) == 1 and T.twin_bls_verify_id(tctx, rec, ctypes.c_uint64(mask), 1, b"sess", 4) == 1
    for field, t in (("phi", t3), ("phi", t11), ("E1", t3), ("E2", t11)):
        bad = copy.copy(pr)
        setattr(bad, field, G.g1_add(getattr(pr, field), t))
        assert G.g1_on_curve(getattr(bad, field))
        rb = pack_verify_id(M, bad)
        assert not PR.verify_id(pk, bad, b"sess", b"service", apk, g, h)
        assert L.elpo_verify_id(key, rb, mask, 1, b"sess", 4) == 0
        assert T.twin_bls_verify_id(tctx, rb, ctypes.c_uint64(mask), 1, b"sess", 4) == 0
        assert T.twin_blsp_verify_id(tctx, rb, ctypes.c_uint64(mask), 1, b"sess", 4) == 0          # the two-lanes-per-item layout
        assert T.twin_blsp_verify_id_g1split(tctx, rb, ctypes.c_uint64(mask), 1, b"sess", 4) == 0  # G1 jobs kernel + paired body (ELP_OPT_SPLIT_PHASES = 3)
        assert T.twin_bls_verify_id_jobs4(tctx, rb, ctypes.c_uint64(mask), 1, b"sess", 4) == 0     # small batches: four job lanes, one subgroup test each
    # a request whose commitment left the subgroup is not signed
    badrq = copy.copy(rq)
    badrq.A = G.g1_add(rq.A, t3)
    out = ctypes.create_string_buffer(4 * N)
    assert PR.provide_id(pk, skX, badrq, b"ad", u) is None
    assert L.elpo_provide_id(key, pack_provide_id(M, badrq, u), 3, b"ad", 2, out) == 0
    assert T.twin_bls_provide_id(tctx, pack_provide_id(M, badrq, u), ctypes.c_uint64(3), b"ad", 2, out) == 0
    L.elpo_key_free(key)


def test_cofactor_sig1_forgery_is_rejected():
    """Round-3 advisor finding: on BLS12-381 a point T of E(Fp) whose order divides the G1 cofactor pairs to 1 with everything, so (sig1, sig2) = (T, O) satisfies
    e(sig1, K) = e(sig2, gg) for ANY K although sig1 is not the point at infinity -- with an honest NIZK over a k built from the public key alone that is a universal
    forgery.  Rule (csrc/elp/pipeline.h sig1_admissible): sigma_1 must be != O AND in G1 -- always in PSVerifier::verify, under ELP_OPT_STRICT_SIGNATURE (the
    library's default) in el_passo_verify_id.  Model, C oracle and host twin (both layouts) agree on the crafted inputs; with the option off the reference's
    lenient behaviour (which BN254's golden vectors pin) is what all of them show."""
    from elp_testlib import twin
    L, T = oracle_bls(), twin()
    seed, A, H = 4242, 4, 2
    g, gg = M.hash_to_g1("abc"), BLS_G2
    pk, skX = PR.key_gen(g, gg, scalar_stream(seed, 0, M.r), [scalar_stream(seed, 1 + i, M.r) for i in range(A)])
    apk, h = M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    bases = (g1_bases(M, pk, svc="service", g_eg=g, apk=apk, h=h, skX=skX), g2_bases(M, pk))
    key = ctypes.c_void_p(L.elpo_key_new(A, *bases))
    tctx = ctypes.c_void_p(T.twin_bls_ctx_new(A, 4, *bases))
    T.twin_bls_ctx_set_flags.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.elpo_set_strict.argtypes = [ctypes.c_int]
    t3 = _small_order_point(3)
    attrs = [(b"s-value", True), (b"gamma-value", True), (b"tp", False), (b"other", False)]
    rq, t1 = PR.request_id(pk, attrs, b"ad", [scalar_stream(seed, 50 + j, M.r) for j in range(2 + H)])
    cred = PR.unblind(PR.provide_id(pk, skX, rq, b"ad", scalar_stream(seed, 99, M.r)), t1)
    pr = PR.prove_id(pk, cred, attrs, b"sess", b"service", apk, g, h, [scalar_stream(seed, 200 + j, M.r) for j in range(3 + H + 2)])
    mask = hidden_mask(pr.attributes)
    forged = copy.copy(pr)
    forged.sig1, forged.sig2 = t3, None                       # needs no credential at all: the NIZK half does not involve the signature
    mixed = copy.copy(pr)
    mixed.sig1 = G.g1_add(pr.sig1, t3)                        # honest signature with a cofactor component: pairs like the honest one
    cases = [(pr, 1, 1), (forged, 1, 0), (mixed, 1, 0)]       # (proof, verdict with the option off, verdict with it on)
    try:
        for strict in (0, 1):
            PR.strict = bool(strict)
            L.elpo_set_strict(strict)
            T.twin_bls_ctx_set_flags(tctx, strict)            # KEY_STRICT_SIG = 1
            for q, lenient, hard in cases:
                want = hard if strict else lenient
                rec = pack_verify_id(M, q)
                assert PR.verify_id(pk, q, b"sess", b"service", apk, g, h) == bool(want)
                assert L.elpo_verify_id(key, rec, mask, 1, b"sess", 4) == want
                assert T.twin_bls_verify_id(tctx, rec, ctypes.c_uint64(mask), 1, b"sess", 4) == want
                assert T.twin_blsp_verify_id(tctx, rec, ctypes.c_uint64(mask), 1, b"sess", 4) == want
                assert T.twin_blsp_verify_id_g1split(tctx, rec, ctypes.c_uint64(mask), 1, b"sess", 4) == want
                assert T.twin_bls_verify_id_jobs4(tctx, rec, ctypes.c_uint64(mask), 1, b"sess", 4) == want      # small-batch form: the sig1 test is role 0's
        # strict, but the caller vouches for subgroup membership (ELP_OPT_SUBGROUP_CHECK = 0): only the infinity test is left, as documented
        T.twin_bls_ctx_set_flags(tctx, 1 | 2)
        assert T.twin_blsp_verify_id(tctx, pack_verify_id(M, forged), ctypes.c_uint64(mask), 1, b"sess", 4) == 1
    finally:
        PR.strict = False
        L.elpo_set_strict(0)
        T.twin_bls_ctx_set_flags(tctx, 0)
    # PSVerifier::verify: sigma_1 = T with sigma_2 = O verifies against every attribute set unless the order-r component is asked
    allattrs = [a for a, _ in attrs]
    from oracle.pymodel import Credential
    good, bad = pack_ps_verify(M, cred, allattrs), pack_ps_verify(M, Credential(t3, None), allattrs)
    assert PR.ps_verify(pk, cred, allattrs) and not PR.ps_verify(pk, Credential(t3, None), allattrs)
    for fn, ctx in ((L.elpo_ps_verify, key), (T.twin_bls_ps_verify, tctx), (T.twin_blsp_ps_verify, tctx)):
        assert fn(ctx, good, A) == 1 and fn(ctx, bad, A) == 0
    L.elpo_key_free(key)
    T.twin_bls_ctx_free(tctx)
