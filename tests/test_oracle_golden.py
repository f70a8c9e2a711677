"""Pins the oracle: both restatements (oracle/pymodel.py big-int model, oracle/elp_oracle.c) must reproduce every golden
vector captured from the reference's own wasm build (tests/golden/bn254_*.json, generator oracle/gen_fixtures.js), and
must agree with each other on the primitives."""
import base64
import ctypes
import random

import pytest

from elp_testlib import (BN254, Codec, Mcl, Protocol, fb, g1b, g1u, g2b, g2u, hidden_mask, ib, load_golden, oracle, oracle_key,
                         pack_provide_id, pack_ps_verify, pack_verify_id, scalar_stream)

M = Mcl(BN254)
CD = Codec(M)
PR = Protocol(M)
G = M.G
FLOWS = load_golden("bn254_oracle_flows.json")
RETR = load_golden("bn254_oracle_with_retrieval.json")


def test_codec_round_trips_every_golden_message():
    for s in FLOWS["scenarios"]:
        raw = base64.b64decode(s["pk"])
        assert CD.pk_encode(CD.pk_decode(raw)) == raw
        for rq in s["requests"]:
            for key, dec, enc in (("request", CD.req_decode, CD.req_encode), ("credential", CD.cred_decode, CD.cred_encode),
                                  ("unblinded", CD.cred_decode, CD.cred_encode)):
                raw = base64.b64decode(rq[key])
                assert enc(dec(raw)) == raw
        for p in s["proofs"]:
            raw = base64.b64decode(p["cases"][0]["proof"])
            assert CD.proof_encode(CD.proof_decode(raw)) == raw
    for r in RETR["runs"]:
        raw = base64.b64decode(r["proof"])
        pr = CD.proof_decode(raw)
        assert pr.has_E and CD.proof_encode(pr) == raw


def test_wire_sizes_match_reference():
    # pk 464 B (A=3) / 954 B (A=8) / 1738 B (A=16); credential 68 B
    sizes = {s["A"]: len(base64.b64decode(s["pk"])) for s in FLOWS["scenarios"]}
    assert sizes[3] == 464 and sizes[8] == 954 and sizes[16] == 1738
    assert len(base64.b64decode(FLOWS["scenarios"][0]["requests"][0]["credential"])) == 68


def test_get_user_name_string():
    for s in FLOWS["scenarios"][:2]:
        for p in s["proofs"]:
            pr = CD.proof_decode(base64.b64decode(p["cases"][0]["proof"]))
            assert M.g1_getstr(pr.phi) == p["username"]


def test_pymodel_reproduces_reference_verdicts_light():
    """NIZK-only (no pairing) on everything; the full check with pairings on the first two scenarios."""
    for si, s in enumerate(FLOWS["scenarios"]):
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        for rq in s["requests"]:
            q = CD.req_decode(base64.b64decode(rq["request"]))
            assert PR.nizk_verify_request(pk, q, s["ad"]) == rq["accept"]
            assert PR.nizk_verify_request(pk, q, s["ad"] + "x") == rq["wrong_ad_accept"]
            assert PR.nizk_verify_request(pk, CD.req_decode(base64.b64decode(rq["request_flip_c"])), s["ad"]) == rq["flip_c_accept"]
        for p in s["proofs"][:1 if si >= 2 else 2]:
            for c in p["cases"]:
                pr = CD.proof_decode(base64.b64decode(c["proof"]))
                full = si < 2
                got = PR.verify_id_noretr(pk, pr, c["ad"], c["svc"], pairing=full)
                if full or not c["expect"]:
                    # without the pairing the model can only over-accept on signature tampering
                    if full:
                        assert got == c["expect"], (s["name"], c["label"])
                if c["expect"]:
                    assert got


def test_hash_to_g1_branches_pinned_by_reference_proofs():
    hh = FLOWS["hash_to_g1"]
    pk = CD.pk_decode(base64.b64decode(hh["pk"]))
    L = oracle()
    seen = set()
    o = ctypes.create_string_buffer(64)
    for c in hh["cases"]:
        pr = CD.proof_decode(base64.b64decode(c["proof"]))
        assert PR.verify_id_noretr(pk, pr, hh["ad"], c["svc"], pairing=False) == c["expect"]
        seen.add(M._last_branch)
        L.elpo_hash_to_g1(c["svc"].encode(), len(c["svc"]), o)
        assert g1u(o.raw) == M.hash_to_g1(c["svc"])
    assert len(seen) == 6          # all three x-candidates, both signs


def test_c_oracle_primitives_match_model():
    L = oracle()
    rnd = random.Random(11)
    pk = CD.pk_decode(base64.b64decode(FLOWS["scenarios"][0]["pk"]))
    g, gg = pk.g, pk.gg
    o, o2, og = ctypes.create_string_buffer(64), ctypes.create_string_buffer(128), ctypes.create_string_buffer(384)
    for k in [0, 1, 2, 3, 31, 32, M.r - 1, M.r, 2**256 - 1] + [rnd.randrange(M.r) for _ in range(10)]:
        assert L.elpo_g1_mul(g1b(g), fb(k), o) and g1u(o.raw) == G.g1_mul(g, k)
    for k in [0, 1, 2, M.r - 1] + [rnd.randrange(M.r) for _ in range(4)]:
        assert L.elpo_g2_mul(g2b(gg), fb(k), o2) and g2u(o2.raw) == G.g2_mul(gg, k)
    P = G.g1_mul(g, 5)
    for a, b in [(P, g), (P, P), (P, G.g1_neg(P)), (P, None), (None, None)]:
        assert L.elpo_g1_add(g1b(a), g1b(b), o) and g1u(o.raw) == G.g1_add(a, b)
    for Q in (pk.XX, G.g2_neg(pk.XX)):
        w = ctypes.create_string_buffer(64)
        assert L.elpo_g2_compress(g2b(Q), w) and w.raw == M.g2_ser(Q)
        assert L.elpo_g2_decompress(M.g2_ser(Q), o2) and g2u(o2.raw) == Q
    w = ctypes.create_string_buffer(32)
    assert L.elpo_g1_compress(g1b(P), w) and w.raw == M.g1_ser(P)
    assert L.elpo_g1_decompress(M.g1_ser(G.g1_neg(P)), o) and g1u(o.raw) == G.g1_neg(P)
    for msg in (b"", b"abc", b"x" * 200):
        L.elpo_fr_set_hash_of(msg, len(msg), w)
        assert ib(w.raw) == M.fr_hash(msg)
    Pp, Q = G.g1_mul(g, 12345), G.g2_mul(gg, 6789)
    e = G.pairing(Pp, Q)
    assert L.elpo_pairing(g1b(Pp), g2b(Q), og)
    assert og.raw == b"".join(fb(e[k][0]) + fb(e[k][1]) for k in [0, 2, 4, 1, 3, 5])


def test_c_oracle_reproduces_every_reference_verdict():
    L = oracle()
    n = 0
    for s in FLOWS["scenarios"]:
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        keys = {}
        for p in s["proofs"]:
            for c in p["cases"]:
                if c["svc"] not in keys:
                    keys[c["svc"]] = oracle_key(M, pk, svc=c["svc"])
                pr = CD.proof_decode(base64.b64decode(c["proof"]))
                ad = c["ad"].encode()
                got = L.elpo_verify_id(keys[c["svc"]], pack_verify_id(M, pr), hidden_mask(pr.attributes), 0, ad, len(ad))
                assert bool(got) == c["expect"], (s["name"], c["svc"], c["label"])
                n += 1
        # IdP side: NIZK verdicts of the reference-generated requests (the signature itself needs the wasm's secret key)
        key = oracle_key(M, pk, skX=pk.g)
        out = ctypes.create_string_buffer(128)
        for rq in s["requests"]:
            q = CD.req_decode(base64.b64decode(rq["request"]))
            qf = CD.req_decode(base64.b64decode(rq["request_flip_c"]))
            mask = hidden_mask(q.attributes)
            for qq, ad, exp in ((q, s["ad"], rq["accept"]), (q, s["ad"] + "x", rq["wrong_ad_accept"]), (qf, s["ad"], rq["flip_c_accept"])):
                got = L.elpo_provide_id(key, pack_provide_id(M, qq, 7), mask, ad.encode(), len(ad), out)
                assert bool(got) == exp
            # the unblinded credential the reference produced is a valid PS signature on all attributes
            ub = CD.cred_decode(base64.b64decode(rq["unblinded"]))
            assert L.elpo_ps_verify(key, pack_ps_verify(M, ub, s["attr_values"]), s["A"]) == 1
            bl = CD.cred_decode(base64.b64decode(rq["credential"]))
            assert L.elpo_ps_verify(key, pack_ps_verify(M, bl, s["attr_values"]), s["A"]) == 0
    assert n > 200
    g, apk, h = M.hash_to_g1("abc"), M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    for r in RETR["runs"]:
        pk = CD.pk_decode(base64.b64decode(r["pk"]))
        pr = CD.proof_decode(base64.b64decode(r["proof"]))
        key = oracle_key(M, pk, svc=r["svc"], g_eg=g, apk=apk, h=h)
        rec = pack_verify_id(M, pr)
        assert not r["verify_failed_line"]
        assert L.elpo_verify_id(key, rec, hidden_mask(pr.attributes), 1, b"hello", 5) == 1
        assert L.elpo_verify_id(key, rec, hidden_mask(pr.attributes), 1, b"hellO", 5) == 0
        assert PR.verify_id(pk, pr, "hello", "service", apk, g, h)


def test_c_oracle_issuance_matches_model_bit_exact():
    L = oracle()
    seed, A, H = 20211, 4, 2
    gg = CD.pk_decode(base64.b64decode(FLOWS["scenarios"][0]["pk"])).gg
    g = M.hash_to_g1("abc")
    pk, skX = PR.key_gen(g, gg, scalar_stream(seed, 0, M.r), [scalar_stream(seed, 1 + i, M.r) for i in range(A)])
    key = oracle_key(M, pk, skX=skX)
    out = ctypes.create_string_buffer(128)
    for n in range(3):
        attrs = [(("a%d-%d" % (i, n)).encode(), i < H) for i in range(A)]
        rq, t1 = PR.request_id(pk, attrs, b"ad", [scalar_stream(seed, 100 + 10 * n + j, M.r) for j in range(2 + H)])
        u = scalar_stream(seed, 1000 + n, M.r)
        want = PR.provide_id(pk, skX, rq, b"ad", u)
        assert L.elpo_provide_id(key, pack_provide_id(M, rq, u), (1 << H) - 1, b"ad", 2, out) == 1
        assert out.raw == g1b(want.sig1) + g1b(want.sig2)
