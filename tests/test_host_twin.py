"""CPU-side checks of the device formulas: the arithmetic headers under csrc/elp/ are compiled for the host (g++) into a
TEST-ONLY library and compared with the oracle.  This is how the HIP code is debugged in a container with no GPU; the
product never loads this library."""
import base64
import ctypes
import random

import pytest

from elp_testlib import (BN254, Codec, Mcl, Protocol, fb, g1_bases, g1b, g1u, g2_bases, g2b, g2u, hidden_mask, ib, load_golden,
                         pack_provide_id, pack_ps_verify, pack_verify_id, scalar_stream, twin)

M = Mcl(BN254)
CD = Codec(M)
PR = Protocol(M)
G = M.G
W_TEST = 4   # small windows keep the host-side table build fast


@pytest.fixture(scope="module")
def L():
    return twin()


def _ctx(L, pk, W=W_TEST, **kw):
    h = L.twin_bn254_ctx_new(len(pk.Yi), W, g1_bases(M, pk, **kw), g2_bases(M, pk))
    assert h
    return ctypes.c_void_p(h)


def test_fp_arith(L):
    rnd = random.Random(1)
    p = M.p
    o = ctypes.create_string_buffer(32)
    cases = [(rnd.randrange(p), rnd.randrange(p)) for _ in range(300)] + [(p - 1, p - 1), (0, 5), (1, 1), (p - 1, 1), (0, 0)]
    for a, b in cases:
        L.twin_bn254_fp_mul(fb(a), fb(b), o)
        assert ib(o.raw) == a * b % p
    for a, _ in cases[:50]:
        L.twin_bn254_fp_inv(fb(a), o)
        assert ib(o.raw) == pow(a, -1, p)
        ok = L.twin_bn254_fp_sqrt(fb(a * a % p), o)
        assert ok and ib(o.raw) in (a, p - a)
    o2 = ctypes.create_string_buffer(64)
    F = M.F
    for _ in range(20):
        x = (rnd.randrange(p), rnd.randrange(p))
        sq = F.f2_sqr(x)
        assert L.twin_bn254_fp2_sqrt(fb(sq[0]) + fb(sq[1]), o2)
        r = (ib(o2.raw[:32]), ib(o2.raw[32:]))
        assert F.f2_sqr(r) == sq
    # elements of Fp (a square of Fp: a real root; a non-square: a purely imaginary one), zero, and non-squares of Fp2 (rejected)
    for a0 in [0, 1, 4, p - 1, p - 4, 3, rnd.randrange(p), rnd.randrange(p)]:
        assert L.twin_bn254_fp2_sqrt(fb(a0) + fb(0), o2)
        r = (ib(o2.raw[:32]), ib(o2.raw[32:]))
        assert F.f2_sqr(r) == (a0, 0) and (r[0] == 0 or r[1] == 0)
    rejected = 0
    for _ in range(40):
        x = (rnd.randrange(p), rnd.randrange(1, p))
        is_sq = pow((x[0] * x[0] + x[1] * x[1]) % p, (p - 1) // 2, p) == 1        # a square of Fp2 iff its norm is one of Fp
        ok = L.twin_bn254_fp2_sqrt(fb(x[0]) + fb(x[1]), o2)
        assert bool(ok) == is_sq
        if ok:
            assert F.f2_sqr((ib(o2.raw[:32]), ib(o2.raw[32:]))) == x
        rejected += not ok
    assert 5 < rejected < 35


def test_hash_to_g1_all_branches(L):
    o = ctypes.create_string_buffer(64)
    seen = set()
    for s in ["abc", "ghi", "jkl", "service", "svc"] + [chr(97 + i) for i in range(26)]:
        L.twin_bn254_hash_to_g1(s.encode(), len(s), o)
        assert g1u(o.raw) == M.hash_to_g1(s), s
        seen.add(M._last_branch)
    assert len(seen) == 6


def test_group_ops(L):
    rnd = random.Random(2)
    d = load_golden("bn254_oracle_flows.json")
    pk = CD.pk_decode(base64.b64decode(d["scenarios"][0]["pk"]))
    g, gg = pk.g, pk.gg
    o, o2 = ctypes.create_string_buffer(64), ctypes.create_string_buffer(128)
    for k in [0, 1, 2, 3, 15, 16, 17, M.r - 1, M.r, rnd.randrange(M.r), rnd.randrange(2**256)]:
        assert L.twin_bn254_g1_mul(g1b(g), fb(k), o)
        assert g1u(o.raw) == G.g1_mul(g, k)
    for k in [0, 1, 2, 5, M.r - 1, rnd.randrange(M.r)]:
        assert L.twin_bn254_g2_mul(g2b(gg), fb(k), o2)
        assert g2u(o2.raw) == G.g2_mul(gg, k)
    P = G.g1_mul(g, 5)
    for a, b in [(P, g), (P, P), (P, G.g1_neg(P)), (P, None), (None, g), (None, None)]:
        assert L.twin_bn254_g1_add(g1b(a), g1b(b), o)
        assert g1u(o.raw) == G.g1_add(a, b)
    assert not L.twin_bn254_g1_mul(fb(1) + fb(1), fb(1), o)      # (1,1) is not on the curve
    assert L.twin_bn254_g1_decompress(M.g1_ser(G.g1_neg(g)), o) and g1u(o.raw) == G.g1_neg(g)
    for Q in (pk.XX, G.g2_neg(pk.XX), pk.YYi[0]):
        assert L.twin_bn254_g2_decompress(M.g2_ser(Q), o2) and g2u(o2.raw) == Q


def test_pairing_value_and_cyclotomic(L):
    d = load_golden("bn254_oracle_flows.json")
    pk = CD.pk_decode(base64.b64decode(d["scenarios"][0]["pk"]))
    P, Q = G.g1_mul(pk.g, 12345), G.g2_mul(pk.gg, 6789)
    og = ctypes.create_string_buffer(384)
    e = G.pairing(P, Q)
    want = b"".join(fb(e[k][0]) + fb(e[k][1]) for k in [0, 2, 4, 1, 3, 5])
    assert L.twin_bn254_pairing(g1b(P), g2b(Q), og, 0) == 1 and og.raw == want
    assert L.twin_bn254_pairing(g1b(P), g2b(Q), og, 2) == 1           # Granger-Scott squaring == generic squaring
    assert L.twin_bn254_pairing_fixedq(g1b(P), g2b(Q), og) == 1 and og.raw == want


def test_two_miller_loops_on_one_accumulator(L):
    """pairing.h miller_loop_two (aggregated verification with two items per lane, round 5): the Miller value of two variable pairs walked together -- one squaring of the
    accumulator per step, the two lines multiplied pairwise -- equals the product of the two single loops bit for bit; a pair that is not live contributes 1 (it walks the
    loop on a copy of the other pair and feeds the neutral line); after the final exponentiation the value is the model's e(P0, Q0) e(P1, Q1)."""
    d = load_golden("bn254_oracle_flows.json")
    pk = CD.pk_decode(base64.b64decode(d["scenarios"][0]["pk"]))
    o2, op, og = ctypes.create_string_buffer(384), ctypes.create_string_buffer(384), ctypes.create_string_buffer(384)
    zero = bytes(384)
    rnd = random.Random(7)
    for _ in range(3):
        P0, Q0 = G.g1_mul(pk.g, rnd.randrange(1, M.r)), G.g2_mul(pk.gg, rnd.randrange(1, M.r))
        P1, Q1 = G.g1_mul(pk.g, rnd.randrange(1, M.r)), G.g2_mul(pk.gg, rnd.randrange(1, M.r))
        for mask in (3, 1, 2, 0):
            assert L.twin_bn254_miller_two(g1b(P0), g2b(Q0), g1b(P1), g2b(Q1), mask, o2, op) == 1
            assert o2.raw == op.raw and o2.raw != zero, mask
        L.twin_bn254_miller_two(g1b(P0), g2b(Q0), g1b(P1), g2b(Q1), 3, o2, op)
        L.twin_bn254_fp12_op(6, 0, o2.raw, o2.raw, og)            # final exponentiation of the joint Miller value
        e = G.F.f12_mul(G.pairing(P0, Q0), G.pairing(P1, Q1))
        assert og.raw == b"".join(fb(e[k][0]) + fb(e[k][1]) for k in [0, 2, 4, 1, 3, 5])
    # a point at infinity in a live pair: contributes 1 as in the single loop
    inf1, inf2 = bytes(64), bytes(128)
    assert L.twin_bn254_miller_two(inf1, g2b(Q0), g1b(P1), g2b(Q1), 3, o2, op) == 1 and o2.raw == op.raw
    assert L.twin_bn254_miller_two(g1b(P0), g2b(Q0), g1b(P1), inf2, 3, o2, op) == 1 and o2.raw == op.raw
    assert L.twin_bn254_miller_two(inf1, inf2, inf1, inf2, 3, o2, op) == 1 and o2.raw == op.raw


def test_verify_id_golden(L):
    d = load_golden("bn254_oracle_flows.json")
    n = 0
    for s in d["scenarios"]:
        if s["A"] > 8:
            continue
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        ctxs = {}
        for p in s["proofs"][:2]:
            for c in p["cases"]:
                if c["svc"] not in ctxs:
                    ctxs[c["svc"]] = _ctx(L, pk, svc=c["svc"])
                P = CD.proof_decode(base64.b64decode(c["proof"]))
                ad = c["ad"].encode()
                got = L.twin_bn254_verify_id(ctxs[c["svc"]], pack_verify_id(M, P), ctypes.c_uint64(hidden_mask(P.attributes)), 0, ad, len(ad))
                assert bool(got) == c["expect"], (s["name"], c["label"])
                # the two-phase form (two job roles + pairing phase) gives the same verdict
                got = L.twin_bn254_verify_id_split(ctxs[c["svc"]], pack_verify_id(M, P), ctypes.c_uint64(hidden_mask(P.attributes)), 0, ad, len(ad))
                assert bool(got) == c["expect"], ("split", s["name"], c["label"])
                # the small-batch form (fixed-base sums first, four job roles, pairing phase) as well
                got = L.twin_bn254_verify_id_jobs4(ctxs[c["svc"]], pack_verify_id(M, P), ctypes.c_uint64(hidden_mask(P.attributes)), 0, ad, len(ad))
                assert got == int(c["expect"]), ("jobs4 / jobs5", got, s["name"], c["label"])
                n += 1
    assert n > 80


def test_verify_id_with_retrieval_golden(L):
    w = load_golden("bn254_oracle_with_retrieval.json")
    for r in w["runs"]:
        pk = CD.pk_decode(base64.b64decode(r["pk"]))
        g, apk, h = M.hash_to_g1(r["g_seed"]), M.hash_to_g1(r["authority_pk_seed"]), M.hash_to_g1(r["h_seed"])
        P = CD.proof_decode(base64.b64decode(r["proof"]))
        ctx = _ctx(L, pk, svc=r["svc"], g_eg=g, apk=apk, h=h)
        rec = pack_verify_id(M, P)
        for ad, exp in ((b"hello", 1), (b"hellO", 0)):
            assert L.twin_bn254_verify_id(ctx, rec, ctypes.c_uint64(hidden_mask(P.attributes)), 1, ad, len(ad)) == exp
            assert L.twin_bn254_verify_id_split(ctx, rec, ctypes.c_uint64(hidden_mask(P.attributes)), 1, ad, len(ad)) == exp
            assert L.twin_bn254_verify_id_jobs4(ctx, rec, ctypes.c_uint64(hidden_mask(P.attributes)), 1, ad, len(ad)) == exp


def test_ps_verify_and_provide_id(L):
    seed = 20211
    A, H = 4, 2
    d = load_golden("bn254_oracle_flows.json")
    gg = CD.pk_decode(base64.b64decode(d["scenarios"][0]["pk"])).gg
    g = M.hash_to_g1("abc")
    x = scalar_stream(seed, 0, M.r)
    ys = [scalar_stream(seed, 1 + i, M.r) for i in range(A)]
    pk, skX = PR.key_gen(g, gg, x, ys)
    ctx = _ctx(L, pk, skX=skX)
    out = ctypes.create_string_buffer(128)
    for n in range(3):
        attrs = [(("a%d-%d" % (i, n)).encode(), i < H) for i in range(A)]
        rnd = [scalar_stream(seed, 100 + 10 * n + j, M.r) for j in range(2 + H)]
        ad = b"ad%d" % n
        rq, t1 = PR.request_id(pk, attrs, ad, rnd)
        if n == 2:
            rq.c ^= 1
        u = scalar_stream(seed, 1000 + n, M.r)
        want = PR.provide_id(pk, skX, rq, ad, u)
        got = L.twin_bn254_provide_id(ctx, pack_provide_id(M, rq, u), ctypes.c_uint64((1 << H) - 1), ad, len(ad), out)
        assert bool(got) == (want is not None)
        if want is not None:
            assert out.raw == g1b(want.sig1) + g1b(want.sig2)
            ub = PR.unblind(want, t1)
            vals = [a for a, _ in attrs]
            assert L.twin_bn254_ps_verify(ctx, pack_ps_verify(M, ub, vals), A) == 1
            assert L.twin_bn254_ps_verify(ctx, pack_ps_verify(M, want, vals), A) == 0     # still blinded
        else:
            assert out.raw == bytes(128)


def test_exceptional_group_law_cases_take_the_exact_path(L):
    """A valid proof built so that the K accumulation meets P + P: the optimistic (untested) formulas produce a degenerate
    accumulator, the lane must redo the section with the exact formulas and ACCEPT, like the oracle."""
    import importlib
    from elp_testlib import OracleBackedCtx, oracle
    synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
    O = oracle()
    octx = OracleBackedCtx()
    A, H = 4, 2
    wl = synth.Workload(octx, A)
    n = 4
    recs, mask, expect = wl.verify_id_batch(n, H, first_item=0, corrupt_every=0, degenerate_items=(1, 2), window_bits=W_TEST)
    assert list(expect) == [1] * n
    g1 = b"".join(octx.b1.get(i, bytes(64)) for i in range(A + 6))
    g2 = b"".join(octx.b2[i] for i in range(A + 2))
    h = L.twin_bn254_ctx_new(A, W_TEST, g1, g2)
    assert h
    okey = octx.key_handle()
    rsz = len(recs) // n
    for i in range(n):
        rec = recs[i * rsz:(i + 1) * rsz]
        assert O.elpo_verify_id(okey, rec, mask, 1, b"hello", 5) == 1
        assert L.twin_bn254_verify_id(ctypes.c_void_p(h), rec, ctypes.c_uint64(mask), 1, b"hello", 5) == 1, i


def test_verify_id_from_wire_messages_golden(L):
    """Device-side wire ingest (T-L-V parse + decompression + attribute hashing): the reference's own messages, byte for byte."""
    d = load_golden("bn254_oracle_flows.json")
    n = 0
    for s in d["scenarios"][:2] + d["scenarios"][4:]:
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        ctxs = {}
        for p in s["proofs"][:2]:
            for c in p["cases"]:
                if c["svc"] not in ctxs:
                    ctxs[c["svc"]] = _ctx(L, pk, svc=c["svc"])
                raw = base64.b64decode(c["proof"])
                ad = c["ad"].encode()
                assert bool(L.twin_bn254_verify_id_wire(ctxs[c["svc"]], raw, len(raw), 0, ad, len(ad))) == c["expect"], (s["name"], c["label"])
                n += 1
        # truncations and garbage never crash and never verify
        raw = base64.b64decode(s["proofs"][0]["cases"][0]["proof"])
        key = ctxs[s["proofs"][0]["cases"][0]["svc"]]
        ad = s["proofs"][0]["cases"][0]["ad"].encode()
        for cut in (0, 1, 2, 33, 70, len(raw) // 2, len(raw) - 1):
            assert L.twin_bn254_verify_id_wire(key, raw[:cut], cut, 0, ad, len(ad)) == 0
        assert L.twin_bn254_verify_id_wire(key, raw + b"\x01", len(raw) + 1, 0, ad, len(ad)) in (0, 1)
        assert L.twin_bn254_verify_id_wire(key, raw, len(raw), 1, ad, len(ad)) == 0       # no E1/E2 in the message
    assert n > 60
    w = load_golden("bn254_oracle_with_retrieval.json")
    g, apk, h = M.hash_to_g1("abc"), M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    for r in w["runs"]:
        pk = CD.pk_decode(base64.b64decode(r["pk"]))
        ctx = _ctx(L, pk, svc=r["svc"], g_eg=g, apk=apk, h=h)
        raw = base64.b64decode(r["proof"])
        assert L.twin_bn254_verify_id_wire(ctx, raw, len(raw), 1, b"hello", 5) == 1
        assert L.twin_bn254_verify_id_wire(ctx, raw, len(raw), 1, b"hellO", 5) == 0


def test_glv_gls_scalar_multiplication(L):
    """Endomorphism-accelerated variable-base multiplication ([k]P through phi / psi) equals plain scalar multiplication."""
    rnd = random.Random(3)
    d = load_golden("bn254_oracle_flows.json")
    pk = CD.pk_decode(base64.b64decode(d["scenarios"][0]["pk"]))
    P, Q = G.g1_mul(pk.g, 777), G.g2_mul(pk.gg, 999)
    o, o2 = ctypes.create_string_buffer(64), ctypes.create_string_buffer(128)
    ks = [0, 1, 2, 15, 16, M.r - 1, M.r, M.r + 1, 2**256 - 1, (M.r - 1) // 2] + [rnd.randrange(M.r) for _ in range(10)]
    for k in ks:
        assert L.twin_bn254_g1_mul_glv(g1b(P), fb(k), o) and g1u(o.raw) == G.g1_mul(P, k)
        assert L.twin_bn254_g2_mul_gls(g2b(Q), fb(k), o2) and g2u(o2.raw) == G.g2_mul(Q, k)
        assert L.twin_bn254_g2_mul_gls_psi(g2b(Q), fb(k), o2) == 1 and g2u(o2.raw) == G.g2_mul(Q, k)      # psi-images read from the table (the G2 job of small batches); == 1: the four one-dimension multiplications of the quad form add up to the same point
    assert L.twin_bn254_g1_mul_glv(bytes(64), fb(5), o) and g1u(o.raw) is None


def test_batch_prover_items_bit_exact_vs_model(L):
    """request_id_item / prove_id_item (user side on the device) against the oracle model with the same injected randomness:
    byte-identical outputs, and the prover's output record is accepted by the verifier item."""
    seed, A, H = 31337, 4, 2
    d = load_golden("bn254_oracle_flows.json")
    gg = CD.pk_decode(base64.b64decode(d["scenarios"][0]["pk"])).gg
    g = M.hash_to_g1("abc")
    pk, skX = PR.key_gen(g, gg, scalar_stream(seed, 0, M.r), [scalar_stream(seed, 1 + i, M.r) for i in range(A)])
    apk, h = M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    ctx = _ctx(L, pk, svc="service", g_eg=g, apk=apk, h=h, skX=skX)
    attrs = [(b"s-value", True), (b"gamma-value", True), (b"tp", False), (b"other", False)]
    ms = [M.fr_hash(a) for a, _ in attrs]
    mask = ctypes.c_uint64(0b0011)
    rnd = [scalar_stream(seed, 50 + j, M.r) for j in range(2 + H)]
    rq, t1 = PR.request_id(pk, attrs, b"ad1", rnd)
    out = ctypes.create_string_buffer(64 + 32 * (2 + H))
    L.twin_bn254_request_id(ctx, b"".join(fb(x) for x in ms + rnd), mask, b"ad1", 3, out)
    assert out.raw == g1b(rq.A) + fb(rq.c) + b"".join(fb(x) for x in rq.rs)
    cred = PR.unblind(PR.provide_id(pk, skX, rq, b"ad1", scalar_stream(seed, 99, M.r)), t1)
    for retr in (1, 0):
        rnd = [scalar_stream(seed, 200 + j, M.r) for j in range(3 + H + 2)]
        use = rnd if retr else rnd[:2] + rnd[3:3 + H + 1]
        want = PR.prove_id(pk, cred, attrs, b"sess", b"service", apk if retr else None, g, h, use, with_retrieval=bool(retr))
        rec = g1b(cred.sig1) + g1b(cred.sig2) + b"".join(fb(x) for x in ms + use)
        o = ctypes.create_string_buffer(len(pack_verify_id(M, want)))
        assert L.twin_bn254_prove_id(ctx, rec, mask, retr, b"sess", 4, o) == 1
        assert o.raw == pack_verify_id(M, want)
        assert L.twin_bn254_verify_id(ctx, o.raw, mask, retr, b"sess", 4) == 1
    # attribute 0 must be hidden
    assert L.twin_bn254_prove_id(ctx, rec, ctypes.c_uint64(0b0110), 0, b"sess", 4, o) == 0


def test_divstep_inversion_against_model_both_curves(L):
    """fp_inv is the Bernstein-Yang divstep inversion (csrc/elp/fp.h): every result must equal pow(a, -1, p), inv(0) = 0, on both
    fields, for edge values (0, 1, 2, p-1, p-2, (p+-1)/2, powers of two, values with long runs of ones) and 400 random ones."""
    from elp_testlib import BLS12_381
    rnd = random.Random(2024)
    for pfx, cv, nb in (("twin_bn254", BN254, 32), ("twin_bls", BLS12_381, 48)):
        p = cv.p
        inv = getattr(L, pfx + "_fp_inv")
        o = ctypes.create_string_buffer(nb)
        vals = [0, 1, 2, 3, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, (1 << 29) - 1, 1 << 29, 1 << 30, (1 << 200) % p, (1 << (p.bit_length() - 1)) - 1,
                (1 << (p.bit_length() - 1)), p - (1 << 100)] + [rnd.randrange(p) for _ in range(400)] + [rnd.randrange(1 << 40) for _ in range(20)]
        for a in vals:
            inv(fb(a, nb), o)
            got = ib(o.raw)
            assert got == (pow(a, -1, p) if a else 0), (pfx, hex(a))


# ---- paired layout (two lanes per item; the lanes are two threads here and every DPP exchange is a rendezvous)
def test_paired_layout_primitives(L):
    pk = CD.pk_decode(base64.b64decode(load_golden("bn254_oracle_flows.json")["scenarios"][0]["pk"]))
    P, Q = G.g1_mul(pk.g, 12345), G.g2_mul(pk.gg, 6789)
    og, og2 = ctypes.create_string_buffer(384), ctypes.create_string_buffer(384)
    assert L.twin_bn254p_pairing(g1b(P), g2b(Q), og) == 1 and L.twin_bn254_pairing(g1b(P), g2b(Q), og2, 0) == 1
    assert og.raw == og2.raw                      # GT bytes of the pair == GT bytes of the single lane (== the model's, tested above)
    o2 = ctypes.create_string_buffer(128)
    rnd = random.Random(5)
    for k in [1, 2, M.r - 1, rnd.randrange(M.r), rnd.randrange(M.r)]:
        assert L.twin_bn254p_g2_mul_gls(g2b(Q), fb(k), o2) and g2u(o2.raw) == G.g2_mul(Q, k)
    for R in (Q, G.g2_neg(Q), pk.XX, None):
        assert L.twin_bn254p_g2_decompress(M.g2_ser(R), o2) and g2u(o2.raw) == R


def test_paired_layout_verify_id_golden(L):
    """verify_id_item_paired (the body of k_verify_id_paired) on the golden verdicts, with and without id-retrieval, and PS verification."""
    d = load_golden("bn254_oracle_flows.json")
    n = 0
    for s in d["scenarios"][:2]:
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        ctxs = {}
        for p in s["proofs"][:2]:
            for c in p["cases"]:
                if c["svc"] not in ctxs:
                    ctxs[c["svc"]] = _ctx(L, pk, svc=c["svc"].encode())
                P = CD.proof_decode(base64.b64decode(c["proof"]))
                ad = c["ad"].encode()
                got = L.twin_bn254p_verify_id(ctxs[c["svc"]], pack_verify_id(M, P), ctypes.c_uint64(hidden_mask(P.attributes)), 0, ad, len(ad))
                assert bool(got) == c["expect"], (s["name"], c["label"])
                if n % 3 == 0:      # KEY_PHASE_MIX: the pairing check before the NIZK half (the order half of a launch's workgroups take under ELP_PHASE_MIX)
                    L.twin_bn254_ctx_set_flags(ctxs[c["svc"]], 4)
                    got = L.twin_bn254p_verify_id(ctxs[c["svc"]], pack_verify_id(M, P), ctypes.c_uint64(hidden_mask(P.attributes)), 0, ad, len(ad))
                    L.twin_bn254_ctx_set_flags(ctxs[c["svc"]], 0)
                    assert bool(got) == c["expect"], (s["name"], c["label"], "pairing first")
                # ELP_OPT_SPLIT_PHASES = 3: the G1 jobs as a kernel of their own (vid_g1_job), then the paired body over their output
                got = L.twin_bn254p_verify_id_g1split(ctxs[c["svc"]], pack_verify_id(M, P), ctypes.c_uint64(hidden_mask(P.attributes)), 0, ad, len(ad))
                assert bool(got) == c["expect"], (s["name"], c["label"], "g1split")
                n += 1
    assert n >= 40
    import copy
    for r in load_golden("bn254_oracle_with_retrieval.json")["runs"][:2]:
        pk = CD.pk_decode(base64.b64decode(r["pk"]))
        g, apk, h = M.hash_to_g1(r["g_seed"]), M.hash_to_g1(r["authority_pk_seed"]), M.hash_to_g1(r["h_seed"])
        P = CD.proof_decode(base64.b64decode(r["proof"]))
        ctx = _ctx(L, pk, svc=r["svc"].encode(), g_eg=g, apk=apk, h=h)
        mask = ctypes.c_uint64(hidden_mask(P.attributes))
        for fn in (L.twin_bn254p_verify_id, L.twin_bn254p_verify_id_g1split, "pairing first"):
            if fn == "pairing first":
                L.twin_bn254_ctx_set_flags(ctx, 4)
                fn = L.twin_bn254p_verify_id
            assert fn(ctx, pack_verify_id(M, P), mask, 1, b"hello", 5) == 1
            assert fn(ctx, pack_verify_id(M, P), mask, 1, b"hellO", 5) == 0
            for fld in ("E1", "E2", "phi", "sig1", "sig2"):
                Q = copy.copy(P)
                setattr(Q, fld, G.g1_add(getattr(P, fld), pk.g))
                assert fn(ctx, pack_verify_id(M, Q), mask, 1, b"hello", 5) == 0, fld
            Q = copy.copy(P)
            Q.rs = list(P.rs)
            Q.rs[1] = (Q.rs[1] + 1) % M.r           # the response only the odd lane's job (V_E2) reads
            assert fn(ctx, pack_verify_id(M, Q), mask, 1, b"hello", 5) == 0
    s = d["scenarios"][0]
    pk = CD.pk_decode(base64.b64decode(s["pk"]))
    ctx = _ctx(L, pk)
    ub = CD.cred_decode(base64.b64decode(s["requests"][0]["unblinded"]))
    bl = CD.cred_decode(base64.b64decode(s["requests"][0]["credential"]))
    assert L.twin_bn254p_ps_verify(ctx, pack_ps_verify(M, ub, s["attr_values"]), 3) == 1
    assert L.twin_bn254p_ps_verify(ctx, pack_ps_verify(M, bl, s["attr_values"]), 3) == 0


def test_paired_layout_wire_ingest(L):
    """verify_id_wire_item_paired: T-L-V parse on both lanes, the G1 decompressions split between the lanes, k decompressed by the pair."""
    d = load_golden("bn254_oracle_flows.json")
    for s in d["scenarios"][:2]:
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        ctxs = {}
        for p in s["proofs"][:1]:
            for c in p["cases"]:
                if c["svc"] not in ctxs:
                    ctxs[c["svc"]] = _ctx(L, pk, svc=c["svc"].encode())
                raw, ad = base64.b64decode(c["proof"]), c["ad"].encode()
                assert bool(L.twin_bn254p_verify_id_wire(ctxs[c["svc"]], raw, len(raw), 0, ad, len(ad))) == c["expect"], (s["name"], c["label"])
    r = load_golden("bn254_oracle_with_retrieval.json")["runs"][0]
    pk = CD.pk_decode(base64.b64decode(r["pk"]))
    g, apk, h = M.hash_to_g1(r["g_seed"]), M.hash_to_g1(r["authority_pk_seed"]), M.hash_to_g1(r["h_seed"])
    ctx = _ctx(L, pk, svc=r["svc"].encode(), g_eg=g, apk=apk, h=h)
    raw = base64.b64decode(r["proof"])
    assert L.twin_bn254p_verify_id_wire(ctx, raw, len(raw), 1, b"hello", 5) == 1
    assert L.twin_bn254p_verify_id_wire(ctx, raw, len(raw), 1, b"hellO", 5) == 0
    for cut in (1, 2, 35, 100, len(raw) - 1):
        assert L.twin_bn254p_verify_id_wire(ctx, raw[:cut], cut, 1, b"hello", 5) == 0


def test_aggregation_multiplier_in_glv_form(L):
    """g1_mul_pair64_with (the multiplier a + b lam of aggregated verification, 64 shared doublings): the result equals [a]P + [b]phi(P) computed by the model,
    phi(x, y) = (beta x, y) for ONE of the two primitive cube roots of unity (the same one for every input), on both curves; edge multipliers included."""
    from elp_testlib import BLS12_381
    for pfx, curve, N in (("twin_bn254", BN254, 32), ("twin_bls", BLS12_381, 48)):
        Mc = Mcl(curve)
        G = Mc.G
        p = curve.p
        betas = [b for b in (pow(g, (p - 1) // 3, p) for g in range(2, 12)) if b != 1][:1]
        betas.append(betas[0] * betas[0] % p)
        fn = getattr(L, pfx + "_g1_mul_pair64")
        which = None
        P0 = Mc.hash_to_g1("pair64")
        cases = [(1, 0), (0, 1), (2**64 - 1, 2**64 - 1), (0x8888888888888888, 0x7777777777777777), (0, 0)]
        cases += [(scalar_stream(5, 2 * i, 2**64), scalar_stream(5, 2 * i + 1, 2**64)) for i in range(6)]
        for j, (a, b) in enumerate(cases):
            P = G.g1_mul(P0, j + 1)
            k = (ctypes.c_uint32 * 4)(a & 0xFFFFFFFF, a >> 32, b & 0xFFFFFFFF, b >> 32)
            out = ctypes.create_string_buffer(2 * N)
            assert fn(g1b(P, N), k, out) == 1
            got = out.raw
            want = []
            for beta in betas:
                Q = (P[0] * beta % p, P[1])
                want.append(g1b(G.g1_add(G.g1_mul(P, a), G.g1_mul(Q, b)), N))
            if a == 0 and b == 0:
                assert got == bytes(2 * N)
                continue
            if which is None and want[0] != want[1]:
                which = 0 if got == want[0] else 1
            if which is not None:
                assert got == want[which], (pfx, j)
            else:
                assert got in want
        assert which is not None


def test_wire_messages_decode_into_the_records_of_the_record_path(L):
    """Round 5: k_wire_decode turns IdProof::toBufferString() messages (src/ps-encoding.cc:451-467) into verify_id records so that wire batches can take the small /
    mid-size record paths.  Every job of wire_decode_job on the reference's own messages (golden proofs incl. the tampered ones that still parse): the record equals the
    one packed from the decoded message by the model, the mask equals the hidden pattern; truncated messages decode to "invalid"."""
    d = load_golden("bn254_oracle_flows.json")
    n = 0
    for s in d["scenarios"]:
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        A = len(pk.Yi)
        for p in s["proofs"][:2]:
            for c in p["cases"][:6]:
                raw = base64.b64decode(c["proof"])
                try:
                    pr = CD.proof_decode(raw)
                except Exception:
                    continue
                want = pack_verify_id(M, pr)
                rec = ctypes.create_string_buffer(len(want) + 64)
                mask = ctypes.c_uint64(0)
                ok = L.twin_bn254_wire_decode(raw, len(raw), A, 0, rec, ctypes.byref(mask))
                if len(pr.attributes) != A:
                    assert ok == 0
                    continue
                assert ok == 1, (s["name"], c["label"])
                assert rec.raw[:len(want)] == want, (s["name"], c["label"])
                assert mask.value == hidden_mask(pr.attributes)
                n += 1
        raw = base64.b64decode(s["proofs"][0]["cases"][0]["proof"])
        rec = ctypes.create_string_buffer(4096)
        mask = ctypes.c_uint64(0)
        for cut in (0, 1, 33, len(raw) // 2, len(raw) - 1):
            assert L.twin_bn254_wire_decode(raw[:cut], cut, A, 0, rec, ctypes.byref(mask)) == 0
    assert n > 40
    w = load_golden("bn254_oracle_with_retrieval.json")
    for r in w["runs"]:
        pk = CD.pk_decode(base64.b64decode(r["pk"]))
        raw = base64.b64decode(r["proof"])
        pr = CD.proof_decode(raw)
        want = pack_verify_id(M, pr)
        rec = ctypes.create_string_buffer(len(want) + 64)
        mask = ctypes.c_uint64(0)
        assert L.twin_bn254_wire_decode(raw, len(raw), len(pk.Yi), 1, rec, ctypes.byref(mask)) == 1
        assert rec.raw[:len(want)] == want and mask.value == hidden_mask(pr.attributes)
