"""BLS12-381 instantiation of the device headers, checked on the CPU (host twin) against the big-int model.
First half: equality with the big-int model plus algebraic known-answer properties (bilinearity, group order).  Second half: the vectors
the reference's own wasm produced when run on this curve (tests/golden/bls12_381_*.json, oracle/wasm_curve.js)."""
import ctypes
import os
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
    # the item of aggregated verification on the lane pair (k_verify_id_agg_paired, round 6) == on one lane (k_verify_id_agg): verdict of the NIZK half, the Miller
    # value f_K([d]sig1) coefficient for coefficient, the multiplier and the copy of sig2 -- honest proof, wrong associated data (f stays 1), no retrieval, sig1 = infinity
    # under the lenient rule
    inf1 = copy.copy(pr2)
    inf1.sig1, inf1.sig2 = None, None
    seed32 = bytes(range(7, 39))
    for rec, retr, adv, want_ok in ((pack_verify_id(M, pr), 1, b"sess", 1), (pack_verify_id(M, pr), 1, b"sesS", 0), (pack_verify_id(M, pr2), 0, b"sess", 1),
                                    (pack_verify_id(M, inf1), 0, b"sess", 1)):
        outs = []
        for fn in (L.twin_bls_agg_item, L.twin_blsp_agg_item):
            fo, de, s2 = ctypes.create_string_buffer(12 * N), ctypes.create_string_buffer(32), ctypes.create_string_buffer(2 * N)
            L.twin_bls_ctx_set_flags(ctx, 0)                 # lenient signature rule: (O, O) reaches the Miller loop
            r = fn(ctx, rec, mask, retr, adv, len(adv), seed32, ctypes.c_uint64(5), fo, de, s2)
            outs.append((r, fo.raw, de.raw, s2.raw))
        assert outs[0][0] == want_ok and outs[0] == outs[1], (retr, adv)
        if not want_ok:
            assert outs[0][1] == fb(1, N) + bytes(11 * N) and outs[0][2] == bytes(32)
        else:
            assert outs[0][2] != bytes(32)


# ------------------------------------------------------------------------------------------------------------------------------------------
# Reference-run vectors for this curve (tests/golden/bls12_381_*.json: the reference's wasm with mcl's BLS12-381 CurveParam, oracle/wasm_curve.js)
# ------------------------------------------------------------------------------------------------------------------------------------------
import base64  # noqa: E402

from elp_testlib import load_golden  # noqa: E402

CD = Codec(M)
KEY_STRICT_SIG, KEY_NO_SUBGROUP_CHECK = 1, 2


def test_hash_to_g1_is_mcls_on_every_reference_service_name(L):
    """SHA-512 setHashOf + SvdW (b = 4) + cofactor, in the device headers: equal to the model on the 32 names whose reference-made proofs pin the map."""
    o2 = ctypes.create_string_buffer(2 * N)
    for c in load_golden("bls12_381_oracle_flows.json")["hash_to_g1"]["cases"]:
        s = c["svc"]
        L.twin_bls_hash_to_g1(s.encode(), len(s), o2)
        assert g1u(o2.raw, N) == M.hash_to_g1(s), s
    for s in ("", "x" * 111, "x" * 112, "x" * 127, "x" * 128, "x" * 300):          # SHA-512 padding boundaries
        L.twin_bls_hash_to_g1(s.encode(), len(s), o2)
        assert g1u(o2.raw, N) == M.hash_to_g1(s), len(s)


def test_device_formulas_reproduce_reference_verdicts_record_and_wire(L):
    """Every el_passo_verify_id_without_id_retrieval verdict of the reference's wasm on BLS12-381 (flows + with-retrieval run), through the one-lane,
    two-lane, job-split and wire forms of the device code.  Options as the reference behaves: lenient signature rule; the subgroup test stays ON
    (no flow case leaves G1)."""
    flows = load_golden("bls12_381_oracle_flows.json")
    n = seen = 0
    sample = int(os.environ.get("ELP_TWIN_SAMPLE", "1"))      # the sanitizer run (tests/test_sanitized_arithmetic.py) takes every eighth case: the instrumented build is ~6 x slower
    for si, s in enumerate(flows["scenarios"]):
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        ctx, cur_svc = None, None                 # one context per key; the relying party's base H1(service) is swapped in (twin_bls_ctx_set_g1_base: elp_set_rp on the device)
        for p in s["proofs"][:2 if si < 2 else 1]:
            for c in p["cases"]:
                seen += 1
                if seen % sample:
                    continue
                if ctx is None:
                    ctx = ctypes.c_void_p(L.twin_bls_ctx_new(len(pk.Yi), W_TEST, g1_bases(M, pk, svc=c["svc"].encode()), g2_bases(M, pk)))
                    assert ctx.value
                    cur_svc = c["svc"]
                elif c["svc"] != cur_svc:
                    assert L.twin_bls_ctx_set_g1_base(ctx, len(pk.Yi) + 1, g1b(M.hash_to_g1(c["svc"].encode()), N)) == 1
                    cur_svc = c["svc"]
                raw = base64.b64decode(c["proof"])
                ad = c["ad"].encode()
                want = int(c["expect"])
                try:
                    pr = CD.proof_decode(raw)
                except ValueError:
                    pr = None                      # a flipped y-flag can leave the curve: only the wire form sees such a message
                if pr is not None:
                    rec, mask = pack_verify_id(M, pr), ctypes.c_uint64(hidden_mask(pr.attributes))
                    fns = [L.twin_bls_verify_id, L.twin_blsp_verify_id] + ([L.twin_bls_verify_id_jobs4, L.twin_blsp_verify_id_g1split] if c["label"] in ("original", "sig_both_zero", "wrong_svc", "flip_r0_bit0") else [])
                    for fn in fns:
                        assert fn(ctx, rec, mask, 0, ad, len(ad)) == want, (s["name"], c["label"], fn)
                else:
                    assert want == 0
                assert L.twin_bls_verify_id_wire(ctx, raw, len(raw), 0, ad, len(ad)) == want, ("wire", s["name"], c["label"])
                n += 1
        if ctx is not None:
            L.twin_bls_ctx_free(ctx)
    assert n >= 80 // sample
    g, apk, h = M.hash_to_g1("abc"), M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    for r in load_golden("bls12_381_oracle_with_retrieval.json")["runs"]:
        pk = CD.pk_decode(base64.b64decode(r["pk"]))
        raw = base64.b64decode(r["proof"])
        pr = CD.proof_decode(raw)
        ctx = ctypes.c_void_p(L.twin_bls_ctx_new(3, W_TEST, g1_bases(M, pk, svc=r["svc"].encode(), g_eg=g, apk=apk, h=h), g2_bases(M, pk)))
        rec, mask = pack_verify_id(M, pr), ctypes.c_uint64(hidden_mask(pr.attributes))
        for fn in (L.twin_bls_verify_id, L.twin_blsp_verify_id, L.twin_bls_verify_id_jobs4):
            assert fn(ctx, rec, mask, 1, b"hello", 5) == 1 and fn(ctx, rec, mask, 1, b"hellO", 5) == 0
        assert L.twin_bls_verify_id_wire(ctx, raw, len(raw), 1, b"hello", 5) == 1
        L.twin_bls_ctx_free(ctx)


def test_device_formulas_on_reference_edge_vectors(L):
    """tests/golden/bls12_381_oracle_edge.json.  k outside G2, alternative length forms, model-made proofs: the reference's verdict under any
    option.  Points of E(Fp) outside G1: the library's default (subgroup test on) rejects every phi outside G1 -- the reference rejects them too
    except crafted ones that survive mcl's GLV split (deliberate divergence, stated in tests/test_oracle_bls_golden.py); sig1 outside G1 is
    accepted exactly as the reference does with both rules off, and rejected under the strict rule."""
    edge = load_golden("bls12_381_oracle_edge.json")
    L.twin_bls_ctx_set_flags.argtypes = [ctypes.c_void_p, ctypes.c_int]
    ctxs = {}
    seen = set()
    sampled = int(os.environ.get("ELP_TWIN_SAMPLE", "1")) > 1         # the sanitizer run: two of the four forms per case, in turn
    for ci, c in enumerate(edge["cases"]):
        lb = c["label"]
        pk = CD.pk_decode(base64.b64decode(c["pk"]))
        kk = c["pk"]
        if kk not in ctxs:
            ctxs[kk] = [ctypes.c_void_p(L.twin_bls_ctx_new(len(pk.Yi), W_TEST, g1_bases(M, pk, svc=c["svc"].encode()), g2_bases(M, pk))), c["svc"]]
        elif ctxs[kk][1] != c["svc"]:
            assert L.twin_bls_ctx_set_g1_base(ctxs[kk][0], len(pk.Yi) + 1, g1b(M.hash_to_g1(c["svc"].encode()), N)) == 1
            ctxs[kk][1] = c["svc"]
        ctx = ctxs[kk][0]
        raw = base64.b64decode(c["proof"])
        pr = CD.proof_decode(raw)
        rec, mask, ad = pack_verify_id(M, pr), ctypes.c_uint64(hidden_mask(pr.attributes)), c["ad"].encode()
        forms = [("record", lambda: L.twin_bls_verify_id(ctx, rec, mask, 0, ad, len(ad))), ("paired", lambda: L.twin_blsp_verify_id(ctx, rec, mask, 0, ad, len(ad))),
                 ("jobs4", lambda: L.twin_bls_verify_id_jobs4(ctx, rec, mask, 0, ad, len(ad))), ("wire", lambda: L.twin_bls_verify_id_wire(ctx, raw, len(raw), 0, ad, len(ad)))]
        if sampled:
            forms = forms[ci % 2::2]
        if lb.startswith("phi_") or lb == "crafted_phi_c_mod_3":
            L.twin_bls_ctx_set_flags(ctx, 0)                                        # default policy: rejected whatever the reference said
            for name, f in forms:
                assert f() == 0, (name, lb)
            if not lb.startswith("crafted"):
                assert c["expect"] is False
        elif lb in ("sig1_plus_T3", "sig1_plus_Tbig", "sig_T3_O"):
            assert c["expect"] is True
            L.twin_bls_ctx_set_flags(ctx, KEY_NO_SUBGROUP_CHECK)                    # the reference's behaviour
            for name, f in forms:
                assert f() == 1, (name, lb)
            L.twin_bls_ctx_set_flags(ctx, KEY_STRICT_SIG)                           # the library's default
            for name, f in forms:
                assert f() == 0, (name, lb)
        else:
            for flags in (0, KEY_STRICT_SIG):
                L.twin_bls_ctx_set_flags(ctx, flags)
                for name, f in forms:
                    assert f() == int(c["expect"]), (name, lb, flags)
        L.twin_bls_ctx_set_flags(ctx, 0)
        seen.add(lb)
    assert {"k_plus_T13", "crafted_c_mod_13", "frlist_fd_len", "strlist_fd_len", "model_made_proof", "sig2_plus_T3", "sig_T3_O", "phi_plus_T11"} <= seen
    for ctx, _ in ctxs.values():
        L.twin_bls_ctx_free(ctx)


def test_device_issuance_on_reference_idp_vectors(L):
    """tests/golden/bls12_381_oracle_requests.json: el_passo_provide_id verdicts of the reference's IdP on model-made requests; commitments outside
    G1 are never signed under the library's default."""
    out = ctypes.create_string_buffer(4 * N)
    for s in load_golden("bls12_381_oracle_requests.json")["scenarios"]:
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        ctx = ctypes.c_void_p(L.twin_bls_ctx_new(s["A"], W_TEST, g1_bases(M, pk, skX=pk.g), g2_bases(M, pk)))
        for c in s["cases"]:
            try:
                q = CD.req_decode(base64.b64decode(c["request"]))
            except ValueError:
                continue
            mask = ctypes.c_uint64(hidden_mask(q.attributes))
            got = L.twin_bls_provide_id(ctx, pack_provide_id(M, q, 7), mask, s["ad"].encode(), len(s["ad"]), out)
            if c["label"] == "model_request":
                assert got == 1 and c["accept"] is True
                assert L.twin_bls_provide_id(ctx, pack_provide_id(M, q, 7), mask, (s["ad"] + "x").encode(), len(s["ad"]) + 1, out) == 0
            else:
                assert got == 0, c["label"]
        L.twin_bls_ctx_free(ctx)
