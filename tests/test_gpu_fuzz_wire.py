"""Wire ingest ON THE GPU against mutated messages ("an out-of-bounds read on the GPU is silent").

tests/test_fuzz_wire.py runs the device parser `WireSrc` on the CPU under sanitizers; this is the other half: the same mutated corpus (bit flips, truncation at
every offset, 3-byte length forms, oversize counts, splices, garbage, range-edge field values) goes through elp_verify_id_wire_batch -- parser, decompression,
attribute hashing AND the verification -- and every verdict must equal the documented format's (test_fuzz_wire.wire_parse) followed by the C oracle's verdict on
the record the message decodes to.  Messages are packed back to back in one buffer, so a parser that reads past its message reads its neighbour and decides wrongly."""
import base64
import random

import pytest

import test_fuzz_wire as FW
from elp_testlib import BN254, Codec, Mcl, g1b, g2b, hidden_mask, load_golden, oracle, oracle_key, pack_verify_id
from oracle.pymodel import IdProof

pytestmark = pytest.mark.gpu

M = Mcl(BN254)
CD = Codec(M)


def _set_key(ctx, pk, svc, g_eg=None, apk=None, h=None):
    ctx.set_pubkey(g1b(pk.g), g2b(pk.gg), g2b(pk.XX), b"".join(g1b(P) for P in pk.Yi), b"".join(g2b(P) for P in pk.YYi), 8)
    ctx.set_rp(svc.encode(), g1b(apk) if apk else None, g1b(g_eg) if g_eg else None, g1b(h) if h else None)


def _expected(L, key, msg, A, retr, ad):
    f = FW.wire_parse(msg, A, retr)
    if f is None:
        return 0
    s1, s2, kk, phi, c, rs, attrs, e1, e2, _ = f
    pr = IdProof(M.g1_de(s1), M.g1_de(s2), M.g2_de(kk), M.g1_de(phi), c, rs, attrs)
    if retr:
        pr.E1, pr.E2, pr.has_E = M.g1_de(e1), M.g1_de(e2), True
    return L.elpo_verify_id(key, pack_verify_id(M, pr), hidden_mask(attrs), retr, ad, len(ad))


def _mutations(rnd, base, pool, n, A, retr):
    """n mutated messages from `base` (A and the retrieval flag are the batch's: mutations that would change them keep the bytes only)"""
    out = []
    while len(out) < n:
        out.append(FW.mutate(rnd, (A, retr, base), rnd.choice(pool))[2])
    return out


def test_wire_ingest_on_the_gpu_matches_format_and_oracle_on_mutated_messages(gpu_ctx):
    L = oracle()
    rnd = random.Random(4242)
    total = accepted = parsed_ok = 0
    d = load_golden("bn254_oracle_flows.json")
    for s in d["scenarios"]:
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        A = len(pk.Yi)
        groups = {}
        for p in s["proofs"]:
            for c in p["cases"]:
                groups.setdefault(c["svc"], []).append((base64.b64decode(c["proof"]), c["ad"].encode(), c["expect"]))
        for svc, items in groups.items():
            _set_key(gpu_ctx, pk, svc)
            key = oracle_key(M, pk, svc=svc)
            pool = [(A, 0, it[0]) for it in items]
            msgs, ads = [it[0] for it in items], [it[1] for it in items]
            want = [int(it[2]) for it in items]                                  # the reference's own verdicts for the untouched messages
            base, ad = items[0][0], items[0][1]
            extra = [base[:k] for k in range(0, len(base), 3)]                   # truncations
            extra += [base[:o] + bytes([253, 0, base[o]]) + base[o + 1:] for o in FW.length_bytes(base, A, 0)]          # value-preserving 3-byte length forms
            for it in items[:4]:
                extra += _mutations(rnd, it[0], pool, 60, A, 0)
            for m in extra:
                msgs.append(m)
                ads.append(ad)
                want.append(_expected(L, key, m, A, 0, ad))
                parsed_ok += FW.wire_parse(m, A, 0) is not None
            flags, cnt = gpu_ctx.verify_id_wire_batch(msgs, False, ads)
            bad = [i for i in range(len(msgs)) if int(flags[i]) != want[i]]
            assert not bad, (s["name"], svc, bad[:5], msgs[bad[0]].hex())
            assert cnt == sum(want)
            total += len(msgs)
            accepted += sum(want)
            L.elpo_key_free(key)
    w = load_golden("bn254_oracle_with_retrieval.json")
    for r in w["runs"]:
        pk = CD.pk_decode(base64.b64decode(r["pk"]))
        A = len(pk.Yi)
        g, apk, h = M.hash_to_g1(r["g_seed"]), M.hash_to_g1(r["authority_pk_seed"]), M.hash_to_g1(r["h_seed"])
        _set_key(gpu_ctx, pk, r["svc"], g_eg=g, apk=apk, h=h)
        key = oracle_key(M, pk, svc=r["svc"], g_eg=g, apk=apk, h=h)
        raw = base64.b64decode(r["proof"])
        msgs = [raw] + [raw[:k] for k in range(0, len(raw), 5)] + _mutations(rnd, raw, [(A, 1, raw)], 150, A, 1)
        msgs += [raw[:o] + bytes([253, 0, raw[o]]) + raw[o + 1:] for o in FW.length_bytes(raw, A, 1)]
        want = [_expected(L, key, m, A, 1, b"hello") for m in msgs]
        assert want[0] == 1
        flags, cnt = gpu_ctx.verify_id_wire_batch(msgs, True, b"hello")
        bad = [i for i in range(len(msgs)) if int(flags[i]) != want[i]]
        assert not bad, (r.get("name"), bad[:5], msgs[bad[0]].hex())
        total += len(msgs)
        accepted += sum(want)
        L.elpo_key_free(key)
    assert total >= 4000 and accepted >= 100 and parsed_ok >= 200, (total, accepted, parsed_ok)
