"""GPU parity on BLS12-381 -- the curve north_star names -- against vectors produced by the REFERENCE'S OWN wasm run on that curve
(tests/golden/bls12_381_*.json; generators oracle/gen_fixtures.js --curve bls12_381, oracle/gen_edge_fixtures.py --curve bls12_381,
oracle/gen_request_fixtures.js --curve bls12_381; the curve switch of the wasm: oracle/wasm_curve.js).  Everything goes through the C-ABI.
Record and wire entry points; the small-batch (cooperative / four-lane job), mid-size (four lanes per pairing) and full-size (two lanes per
item) kernels are reached by tiling the reference's cases to the batch sizes that select them."""
import base64

import numpy as np
import pytest

from elp_testlib import (BLS12_381, Codec, Mcl, Protocol, g1b, g1u, g2b, hidden_mask, load_golden, pack_provide_id, pack_ps_verify, pack_verify_id)

pytestmark = pytest.mark.gpu
M = Mcl(BLS12_381)
CD = Codec(M)
PR = Protocol(M)
G = M.G
N = 48
FLOWS = load_golden("bls12_381_oracle_flows.json")
RETR = load_golden("bls12_381_oracle_with_retrieval.json")
EDGE = load_golden("bls12_381_oracle_edge.json")
REQS = load_golden("bls12_381_oracle_requests.json")


@pytest.fixture(scope="module")
def ctx(elp):
    c = elp.Context(elp.CURVE_BLS12_381, 0)
    c.set_strict_signature(False)          # the reference's behaviour on sig1 = sig2 = O (golden case "sig_both_zero"); the library's default rejects it
    yield c
    c.close()


def _set_key(ctx, pk, svc=None, g_eg=None, apk=None, h=None, skX=None, W=6):
    ctx.set_pubkey(g1b(pk.g, N), g2b(pk.gg, N), g2b(pk.XX, N), b"".join(g1b(P, N) for P in pk.Yi), b"".join(g2b(P, N) for P in pk.YYi), W)
    if svc is not None:
        ctx.set_rp(svc.encode() if isinstance(svc, str) else svc, g1b(apk, N) if apk else None, g1b(g_eg, N) if g_eg else None, g1b(h, N) if h else None)
    if skX is not None:
        ctx.set_signer_secret(g1b(skX, N))


def test_hash_to_g1_is_mcls_map(ctx):
    """hashAndMapToG1 on the device = SHA-512 setHashOf + SvdW (b = 4) + cofactor: H1(svc) for the 32 service names whose reference-made proofs
    pin the map (all six branch / sign cases), plus SHA-512 padding boundaries."""
    names = [c["svc"].encode() for c in FLOWS["hash_to_g1"]["cases"]] + [b"", b"x" * 111, b"x" * 112, b"x" * 127, b"x" * 128, b"x" * 300]
    out = ctx.hash_to_g1(names)
    for i, s in enumerate(names):
        assert g1u(out[96 * i:96 * i + 96], N) == M.hash_to_g1(s), s
    # and through the verifier: each proof verifies only under its own service name
    hh = FLOWS["hash_to_g1"]
    pk = CD.pk_decode(base64.b64decode(hh["pk"]))
    _set_key(ctx, pk)
    for c in hh["cases"][:12]:
        P = CD.proof_decode(base64.b64decode(c["proof"]))
        ctx.set_rp(c["svc"].encode())
        flags, _ = ctx.verify_id_batch(pack_verify_id(M, P), hidden_mask(P.attributes), False, hh["ad"].encode())
        assert bool(flags[0]) is c["expect"] is True
        ctx.set_rp((c["svc"] + "!").encode())
        flags, _ = ctx.verify_id_batch(pack_verify_id(M, P), hidden_mask(P.attributes), False, hh["ad"].encode())
        assert not flags[0]


def _groups(s):
    groups = {}
    for p in s["proofs"]:
        for c in p["cases"]:
            raw = base64.b64decode(c["proof"])
            try:
                P = CD.proof_decode(raw)
            except ValueError:
                continue                          # a message that does not decode to points has no record form (wire test below)
            groups.setdefault((c["svc"], hidden_mask(P.attributes)), []).append((pack_verify_id(M, P), c["ad"].encode(), c["expect"], c["label"]))
    return groups


def test_verify_id_every_reference_verdict_record_path(ctx):
    total = 0
    for s in FLOWS["scenarios"]:
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        _set_key(ctx, pk)
        for (svc, mask), items in _groups(s).items():
            ctx.set_rp(svc.encode())
            flags, cnt = ctx.verify_id_batch(b"".join(i[0] for i in items), mask, False, [i[1] for i in items])
            for f, it in zip(flags, items):
                assert bool(f) == it[2], (s["name"], svc, it[3])
            assert cnt == sum(1 for it in items if it[2])
            total += len(items)
    assert total > 200


def test_verify_id_every_reference_verdict_wire_path(ctx):
    """The reference's messages as they travel (IdProof::toBufferString): T-L-V parse, decompression and attribute hashing on the device."""
    total = 0
    for s in FLOWS["scenarios"]:
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        _set_key(ctx, pk)
        by_svc = {}
        for p in s["proofs"]:
            for c in p["cases"]:
                by_svc.setdefault(c["svc"], []).append(c)
        for svc, cases in by_svc.items():
            ctx.set_rp(svc.encode())
            for decode in (1, 0):                      # decoded-record path (small batches) and the fused wire kernels
                ctx.set_wire_decode(decode)
                flags, cnt = ctx.verify_id_wire_batch([base64.b64decode(c["proof"]) for c in cases], False, [c["ad"].encode() for c in cases])
                for f, c in zip(flags, cases):
                    assert bool(f) == c["expect"], (s["name"], svc, c["label"], decode)
            total += len(cases)
    ctx.set_wire_decode(1)
    assert total >= 200


@pytest.mark.parametrize("n", [700, 2500, 6000, 20000])
def test_reference_verdicts_at_the_batch_sizes_that_select_each_kernel(ctx, n):
    """700: small-batch kernels (four-lane G2 jobs + cooperative pairing); 2500: cooperative range above the four-lane job limit; 6000: four lanes per
    pairing (k_vid_mid); 20000: the two-lanes-per-item kernel.  The reference's A8H4 cases for one service name, tiled; verdicts must tile too."""
    s = next(x for x in FLOWS["scenarios"] if x["name"] == "A8H4")
    pk = CD.pk_decode(base64.b64decode(s["pk"]))
    _set_key(ctx, pk)
    (svc, mask), items = max(_groups(s).items(), key=lambda kv: len(kv[1]))
    ctx.set_rp(svc.encode())
    reps = (n + len(items) - 1) // len(items)
    recs = (b"".join(i[0] for i in items) * reps)[:n * len(items[0][0])]
    ads = ([i[1] for i in items] * reps)[:n]
    want = np.array(([int(i[2]) for i in items] * reps)[:n], dtype=np.uint8)
    flags, cnt = ctx.verify_id_batch(recs, mask, False, ads)
    assert (np.asarray(flags) == want).all() and cnt == int(want.sum())
    assert 0 < int(want.sum()) < n


def test_with_retrieval_run_of_the_reference(ctx):
    """tests.wasm run_tests on BLS12-381: the full flow with the ElGamal token (wasm-src/tests.cc:22-86)."""
    import copy
    g, apk, h = M.hash_to_g1("abc"), M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    for r in RETR["runs"]:
        assert not r["verify_failed_line"]
        pk = CD.pk_decode(base64.b64decode(r["pk"]))
        raw = base64.b64decode(r["proof"])
        P = CD.proof_decode(raw)
        _set_key(ctx, pk, svc=r["svc"], g_eg=g, apk=apk, h=h)
        variants = [(P, r["ad"])]
        for fld in ("E1", "E2", "phi"):
            Q = copy.copy(P)
            setattr(Q, fld, G.g1_add(getattr(P, fld), pk.g))
            variants.append((Q, r["ad"]))
        variants.append((P, r["ad"] + "x"))
        flags, cnt = ctx.verify_id_batch(b"".join(pack_verify_id(M, v) for v, _ in variants), hidden_mask(P.attributes), True, [a.encode() for _, a in variants])
        assert [int(f) for f in flags] == [1, 0, 0, 0, 0]
        flags, _ = ctx.verify_id_wire_batch([raw, raw], True, [r["ad"].encode(), b"other"])
        assert [int(f) for f in flags] == [1, 0]
        # the credential and request of the same run: IdP verdict, PS verification of the unblinded credential is covered by the flows below
        rq = CD.req_decode(base64.b64decode(r["request"]))
        ctx.set_signer_secret(g1b(pk.g, N))
        sigs, fl, _ = ctx.provide_id_batch(pack_provide_id(M, rq, 7) * 2, hidden_mask(rq.attributes), [r["ad"].encode(), b"other"])
        assert [int(f) for f in fl] == [1, 0]


def test_ps_verify_and_idp_verdicts_of_the_reference(ctx):
    from oracle.pymodel import Credential
    for s in FLOWS["scenarios"][:3]:
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        _set_key(ctx, pk, skX=pk.g)
        vals = s["attr_values"]
        items = []
        for rq in s["requests"]:
            ub = CD.cred_decode(base64.b64decode(rq["unblinded"]))
            bl = CD.cred_decode(base64.b64decode(rq["credential"]))
            items += [(ub, vals, True), (bl, vals, False), (ub, vals[:-1] + [vals[-1] + "x"], False), (Credential(None, ub.sig2), vals, False),
                      (PR.randomize(ub, 123456789), vals, True)]
        flags, cnt = ctx.ps_verify_batch(b"".join(pack_ps_verify(M, c, a) for c, a, _ in items), s["A"])
        assert [bool(f) for f in flags] == [e for _, _, e in items], s["name"]
        recs, ads, want = [], [], []
        for rq in s["requests"]:
            q = CD.req_decode(base64.b64decode(rq["request"]))
            qf = CD.req_decode(base64.b64decode(rq["request_flip_c"]))
            mask = hidden_mask(q.attributes)
            for qq, ad, exp in ((q, s["ad"], rq["accept"]), (q, s["ad"] + "x", rq["wrong_ad_accept"]), (qf, s["ad"], rq["flip_c_accept"])):
                recs.append(pack_provide_id(M, qq, 7)); ads.append(ad.encode()); want.append(exp)
        sigs, fl, cnt = ctx.provide_id_batch(b"".join(recs), mask, ads)
        assert [bool(f) for f in fl] == want
    # requests made by the model and judged by the reference's IdP module; commitments outside G1 are never signed (library default)
    for s in REQS["scenarios"]:
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        _set_key(ctx, pk, skX=pk.g)
        for c in s["cases"]:
            try:
                q = CD.req_decode(base64.b64decode(c["request"]))
            except ValueError:
                continue
            sigs, fl, _ = ctx.provide_id_batch(pack_provide_id(M, q, 7), hidden_mask(q.attributes), s["ad"].encode())
            if c["label"] == "model_request":
                assert fl[0] == 1 and c["accept"] is True
            else:
                assert fl[0] == 0, c["label"]


def test_edge_vectors_against_reference_and_policy(ctx):
    """tests/golden/bls12_381_oracle_edge.json: k outside G2, the 3-byte length forms, model-made proofs -> the reference's verdict; points of E(Fp)
    outside G1 -> the library's documented policy next to the reference's verdict (include/elpasso.h ELP_OPT_SUBGROUP_CHECK / ELP_OPT_STRICT_SIGNATURE):
      phi outside G1           default: rejected (reference: rejected, except crafted ones that survive mcl's GLV split -- deliberate divergence)
      sig1 outside G1          both rules off: accepted like the reference; strict rule (library default): rejected
    Record and wire entry points."""
    by_key = {}
    for c in EDGE["cases"]:
        by_key.setdefault((c["pk"], c["svc"]), []).append(c)
    try:
        for (pkb, svc), cases in by_key.items():
            pk = CD.pk_decode(base64.b64decode(pkb))
            _set_key(ctx, pk, svc=svc)
            plain = [c for c in cases if not (c["label"].startswith("phi_") or c["label"].startswith("sig1_") or c["label"] in ("sig_T3_O", "crafted_phi_c_mod_3"))]
            phis = [c for c in cases if c["label"].startswith("phi_") or c["label"] == "crafted_phi_c_mod_3"]
            sig1s = [c for c in cases if c["label"].startswith("sig1_") or c["label"] == "sig_T3_O"]
            assert plain and phis and sig1s

            def run(cs):
                Ps = [CD.proof_decode(base64.b64decode(c["proof"])) for c in cs]
                mask = hidden_mask(Ps[0].attributes)
                assert all(hidden_mask(P.attributes) == mask for P in Ps)
                f1, _ = ctx.verify_id_batch(b"".join(pack_verify_id(M, P) for P in Ps), mask, False, [c["ad"].encode() for c in cs])
                f2, _ = ctx.verify_id_wire_batch([base64.b64decode(c["proof"]) for c in cs], False, [c["ad"].encode() for c in cs])
                assert list(f1) == list(f2)
                return [bool(f) for f in f1]

            for strict in (False, True):
                ctx.set_strict_signature(strict)
                assert run(plain) == [c["expect"] for c in plain], [c["label"] for c in plain]
            ctx.set_strict_signature(False)
            assert run(phis) == [False] * len(phis)                         # library default: never accepted
            assert all(c["expect"] is True for c in sig1s)
            ctx.set_subgroup_check(False)
            assert run(sig1s) == [True] * len(sig1s)                        # the reference's behaviour
            ctx.set_subgroup_check(True)
            ctx.set_strict_signature(True)
            assert run(sig1s) == [False] * len(sig1s)                       # the library's default
            ctx.set_strict_signature(False)
    finally:
        ctx.set_subgroup_check(True)
        ctx.set_strict_signature(False)
