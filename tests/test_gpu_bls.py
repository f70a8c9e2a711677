"""GPU tests of the BLS12-381 instantiation (the north-star curve): results are compared with the big-int model and the C oracle's BLS12-381
build and checked through algebraic known-answer properties; the synthetic batches are checked against the generator's expectation.  Both
checkers are pinned to the reference's own wasm run on this curve by tests/test_oracle_bls_golden.py; the reference-made vectors themselves
go through the GPU in tests/test_gpu_bls_golden.py."""
import copy
import importlib
import random

import numpy as np
import pytest

from elp_testlib import (BLS12_381, BLS_G1, BLS_G2, Mcl, Protocol, fb, g1b, g1u, g2b, g2u, hidden_mask, oracle_bls, pack_provide_id, pack_prove_id, pack_ps_verify,
                         pack_request_id, pack_verify_id, scalar_stream)

pytestmark = pytest.mark.gpu
M = Mcl(BLS12_381)
PR = Protocol(M)
G = M.G
N = 48


@pytest.fixture(scope="module")
def bls_ctx(elp):
    ctx = elp.Context(elp.CURVE_BLS12_381, 0)
    yield ctx
    ctx.close()


def test_primitives_vs_model(bls_ctx):
    rnd = random.Random(9)
    ks = [0, 1, 2, 15, 16, M.r - 1, M.r, 2**256 - 1] + [rnd.randrange(M.r) for _ in range(8)]
    out = bls_ctx.g1_mul(g1b(BLS_G1, N) * len(ks), b"".join(fb(k) for k in ks))
    for i, k in enumerate(ks):
        assert g1u(out[96 * i:96 * i + 96], N) == G.g1_mul(BLS_G1, k % M.r)
    ks2 = ks[:10]
    out = bls_ctx.g2_mul(g2b(BLS_G2, N) * len(ks2), b"".join(fb(k) for k in ks2))
    for i, k in enumerate(ks2):
        assert g2u(out[192 * i:192 * i + 192], N) == G.g2_mul(BLS_G2, k % M.r)
    P, Q = G.g1_mul(BLS_G1, 5), G.g2_mul(BLS_G2, 7)
    cases = [(P, BLS_G1), (P, P), (P, G.g1_neg(P)), (P, None), (None, None)]
    out = bls_ctx.g1_add(b"".join(g1b(a, N) for a, _ in cases), b"".join(g1b(b, N) for _, b in cases))
    for i, (a, b) in enumerate(cases):
        assert g1u(out[96 * i:96 * i + 96], N) == G.g1_add(a, b)
    out, ok = bls_ctx.g1_decompress(M.g1_ser(P) + M.g1_ser(G.g1_neg(P)) + bytes(48))
    assert ok.all() and g1u(out[:96], N) == P and g1u(out[96:192], N) == G.g1_neg(P) and g1u(out[192:], N) is None
    out, ok = bls_ctx.g2_decompress(M.g2_ser(Q) + M.g2_ser(G.g2_neg(Q)))
    assert ok.all() and g2u(out[:192], N) == Q and g2u(out[192:], N) == G.g2_neg(Q)
    msgs = [b"abc", b"ghi", b"jkl", b"service", b""]
    out = bls_ctx.hash_to_g1(msgs)
    for i, m in enumerate(msgs):
        assert g1u(out[96 * i:96 * i + 96], N) == M.hash_to_g1(m)


def test_pairing_value_and_bilinearity(bls_ctx):
    a, b = 1234567, 7654321
    P, Q = G.g1_mul(BLS_G1, a), G.g2_mul(BLS_G2, b)
    out = bls_ctx.pairing(g1b(P, N) + g1b(BLS_G1, N), g2b(Q, N) + g2b(G.g2_mul(BLS_G2, a * b % M.r), N))
    e = G.pairing(P, Q)
    want = b"".join(fb(e[k][0], N) + fb(e[k][1], N) for k in [0, 2, 4, 1, 3, 5])
    assert out[:576] == want
    assert out[576:] == want                      # e(aG1, bG2) == e(G1, ab G2)
    items = [([P, G.g1_neg(G.g1_mul(BLS_G1, a * b % M.r))], [Q, BLS_G2], 1), ([P, BLS_G1], [Q, BLS_G2], 0), ([None, None], [Q, Q], 1)]
    ok = bls_ctx.pairing_check(2, b"".join(g1b(p, N) for it in items for p in it[0]), b"".join(g2b(q, N) for it in items for q in it[1]))
    assert list(ok) == [it[2] for it in items]


def test_protocol_flows_vs_model(bls_ctx):
    seed, A, H = 4242, 4, 2
    g, gg = M.hash_to_g1("abc"), BLS_G2
    pk, skX = PR.key_gen(g, gg, scalar_stream(seed, 0, M.r), [scalar_stream(seed, 1 + i, M.r) for i in range(A)])
    apk, h = M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    bls_ctx.set_pubkey(g1b(pk.g, N), g2b(pk.gg, N), g2b(pk.XX, N), b"".join(g1b(P, N) for P in pk.Yi), b"".join(g2b(P, N) for P in pk.YYi), 6)
    bls_ctx.set_rp(b"service", g1b(apk, N), g1b(g, N), g1b(h, N))
    bls_ctx.set_signer_secret(g1b(skX, N))
    attrs = [(b"s-value", True), (b"gamma-value", True), (b"tp", False), (b"other", False)]
    rq, t1 = PR.request_id(pk, attrs, b"ad", [scalar_stream(seed, 50 + j, M.r) for j in range(2 + H)])
    u = scalar_stream(seed, 99, M.r)
    want = PR.provide_id(pk, skX, rq, b"ad", u)
    sigs, flags, cnt = bls_ctx.provide_id_batch(pack_provide_id(M, rq, u) * 2, 3, [b"ad", b"ae"])
    assert list(flags) == [1, 0] and sigs[:192] == g1b(want.sig1, N) + g1b(want.sig2, N) and sigs[192:] == bytes(192)
    cred = PR.unblind(want, t1)
    vals = [a for a, _ in attrs]
    flags, cnt = bls_ctx.ps_verify_batch(pack_ps_verify(M, cred, vals) + pack_ps_verify(M, want, vals), A)
    assert list(flags) == [1, 0]
    rnd = [scalar_stream(seed, 200 + j, M.r) for j in range(3 + H + 2)]
    pr = PR.prove_id(pk, cred, attrs, b"sess", b"service", apk, g, h, rnd)
    bad = copy.copy(pr)
    bad.sig2 = G.g1_add(pr.sig2, g)
    bad2 = copy.copy(pr)
    bad2.rs = list(pr.rs)
    bad2.rs[0] = (bad2.rs[0] + 1) % M.r
    recs = pack_verify_id(M, pr) + pack_verify_id(M, bad) + pack_verify_id(M, bad2) + pack_verify_id(M, pr)
    flags, cnt = bls_ctx.verify_id_batch(recs, hidden_mask(pr.attributes), True, [b"sess", b"sess", b"sess", b"sesS"])
    assert list(flags) == [1, 0, 0, 0]
    assert PR.verify_id(pk, pr, b"sess", b"service", apk, g, h) and not PR.verify_id(pk, bad, b"sess", b"service", apk, g, h)
    # user side on the device: same randomness -> the model's request and proof, byte for byte
    rq_rnd = [scalar_stream(seed, 50 + j, M.r) for j in range(2 + H)]
    got = bls_ctx.request_id_batch(pack_request_id(M, attrs, rq_rnd), 3, b"ad")
    assert got == g1b(rq.A, N) + fb(rq.c) + b"".join(fb(x) for x in rq.rs)
    proofs, pfl, pc = bls_ctx.prove_id_batch(pack_prove_id(M, cred, attrs, rnd), 3, True, b"sess")
    assert pc == 1 and proofs == pack_verify_id(M, pr)
    pr2 = PR.prove_id(pk, cred, attrs, b"sess", b"service", None, None, None, rnd[:2] + rnd[3:3 + H + 1], with_retrieval=False)
    flags, cnt = bls_ctx.verify_id_batch(pack_verify_id(M, pr2), hidden_mask(pr2.attributes), False, b"sess")
    assert list(flags) == [1] and PR.verify_id_noretr(pk, pr2, b"sess", b"service")
    proofs, pfl, pc = bls_ctx.prove_id_batch(pack_prove_id(M, cred, attrs, rnd[:2] + rnd[3:3 + H + 1]), 3, False, b"sess")
    assert pc == 1 and proofs == pack_verify_id(M, pr2)


def test_synthetic_batch_expectation(bls_ctx):
    synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
    A, H, n = 8, 4, 300
    wl = synth.Workload(bls_ctx, A)
    recs, mask, expect = wl.verify_id_batch(n, H, degenerate_items=(3, 150), window_bits=8)
    flags, cnt = bls_ctx.verify_id_batch(recs, mask, True, b"hello")
    assert list(flags) == list(expect) and cnt == int(expect.sum()) and flags[3] == 1 and flags[13] == 0
    # spot-check three items of the synthetic batch with the big-int model (full verification incl. pairings)
    from oracle.pymodel import IdProof, PubKey
    from elp_testlib import ib
    pk = PubKey(g1u(wl.g, N), g2u(wl.gg, N), g2u(wl.XX, N), [g1u(wl.Yi[96 * i:96 * i + 96], N) for i in range(A)],
                [g2u(wl.YYi[192 * i:192 * i + 192], N) for i in range(A)])
    rsz = len(recs) // n
    for i in (0, 13, 150):
        r = recs[i * rsz:(i + 1) * rsz]
        pts = [g1u(r[96 * j:96 * j + 96], N) for j in range(5)]
        k = g2u(r[480:672], N)
        sc = [ib(r[672 + 32 * j:704 + 32 * j]) for j in range(1 + H + 2 + (A - H))]
        attrs = [b""] * H + [a for a in wl.attributes(i)[H:]]
        pr = IdProof(pts[0], pts[1], k, pts[2], sc[0], sc[1:1 + H + 2], attrs, pts[3], pts[4], True)
        assert [M.fr_hash(a) for a in attrs[H:]] == sc[1 + H + 2:]
        assert PR.verify_id(pk, pr, b"hello", b"service", g1u(wl.apk, N), g1u(wl.g, N), g1u(wl.h, N)) == bool(flags[i])


def test_msm_wire_and_aggregated_paths_on_bls(bls_ctx):
    """The wider entry points on the BLS12-381 instantiation: Pippenger MSM vs the model's sum of products, wire ingest of a
    model-encoded proof, aggregated verification vs the per-item verdicts."""
    synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
    from oracle.pymodel import Codec
    rnd = random.Random(4)
    n = 70
    ks = [rnd.randrange(M.r) for _ in range(n)]
    bs = [rnd.randrange(1, 1 << 40) for _ in range(n)]
    pts = bls_ctx.g1_mul(g1b(BLS_G1, N) * n, b"".join(fb(b) for b in bs))
    total = sum(k * b for k, b in zip(ks, bs)) % M.r
    assert g1u(bls_ctx.g1_msm(pts, b"".join(fb(k) for k in ks)), N) == G.g1_mul(BLS_G1, total)
    pts2 = bls_ctx.g2_mul(g2b(BLS_G2, N) * 20, b"".join(fb(b) for b in bs[:20]))
    total2 = sum(k * b for k, b in zip(ks[:20], bs[:20])) % M.r
    assert g2u(bls_ctx.g2_msm(pts2, b"".join(fb(k) for k in ks[:20])), N) == G.g2_mul(BLS_G2, total2)
    A, H, m = 4, 2, 200
    wl = synth.Workload(bls_ctx, A)
    recs, mask, expect = wl.verify_id_batch(m, H, corrupt_every=11, corrupt_at=4)
    flags, cnt = bls_ctx.verify_id_batch(recs, mask, True, b"hello")
    assert (flags == expect).all()
    fl2, cnt2, held = bls_ctx.verify_id_batch_aggregated(recs, mask, True, b"hello", bytes(32))
    assert held and (fl2 == expect).all() and cnt2 == cnt
    rsz = len(recs) // m
    r = bytearray(recs)
    r[3 * rsz:3 * rsz + 96] = r[5 * rsz:5 * rsz + 96]        # foreign sig1: passes the NIZK, fails the pairing
    fl3, cnt3, held = bls_ctx.verify_id_batch_aggregated(bytes(r), mask, True, b"hello", bytes(32))
    ref, _ = bls_ctx.verify_id_batch(bytes(r), mask, True, b"hello")
    assert not held and (fl3 == ref).all() and ref[3] == 0
    # wire path: re-encode record 0 as the reference's T-L-V message through the model's codec
    from oracle.pymodel import IdProof
    from elp_testlib import ib
    rec = recs[:rsz]
    p = [g1u(rec[96 * j:96 * j + 96], N) for j in range(5)]
    k = g2u(rec[480:672], N)
    sc = [ib(rec[672 + 32 * j:704 + 32 * j]) for j in range(1 + H + 2 + (A - H))]
    attrs = [b""] * H + wl.attributes(0)[H:]
    pr = IdProof(p[0], p[1], k, p[2], sc[0], sc[1:1 + H + 2], attrs, p[3], p[4], True)
    wire = Codec(M).proof_encode(pr)
    fl4, _ = bls_ctx.verify_id_wire_batch([wire, wire[:-1]], True, b"hello")
    assert list(fl4) == [1, 0]


def test_batch_against_the_c_oracle_bls_build(bls_ctx):
    """The second independent implementation for this curve (oracle/elp_oracle.c -DELPO_BLS12_381; pinned to the reference's wasm by tests/test_oracle_bls_golden.py; tests/test_oracle_bls.py ties it to the
    model): synthetic el_passo_verify_id and PS / issuance batches, every GPU verdict and every issued signature equal to the C oracle's, GT bytes equal."""
    import ctypes
    import importlib
    synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
    L = oracle_bls()
    A, H, n = 8, 4, 192
    wl = synth.Workload(bls_ctx, A, seed=31, window_bits=8)
    g1 = wl.g + wl.Yi + bls_ctx.hash_to_g1([wl.service]) + wl.g + wl.apk + wl.h + wl.X
    key = ctypes.c_void_p(L.elpo_key_new(A, g1, wl.gg + wl.XX + wl.YYi))
    assert key.value
    for retr in (True, False):
        recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=retr, corrupt_every=5, corrupt_at=2)
        flags, cnt = bls_ctx.verify_id_batch(recs, mask, retr, wl.ad)
        rsz = len(recs) // n
        ofl = np.zeros(n, dtype=np.uint8)
        L.elpo_verify_id_batch(key, n, recs, rsz, mask, 1 if retr else 0, wl.ad, len(wl.ad), ofl.ctypes.data, 8)
        assert (flags == expect).all() and (ofl == flags).all() and cnt == int(ofl.sum())
    precs, pmask, pexpect = wl.provide_id_batch(64, H)
    sigs, pflags, _ = bls_ctx.provide_id_batch(precs, pmask, wl.ad)
    prsz = len(precs) // 64
    out = ctypes.create_string_buffer(4 * N)
    for i in range(0, 64, 7):
        assert L.elpo_provide_id(key, precs[i * prsz:(i + 1) * prsz], pmask, wl.ad, len(wl.ad), out) == int(pflags[i])
        assert out.raw == sigs[4 * N * i:4 * N * (i + 1)]
    P, Q = G.g1_mul(BLS_G1, 424242), G.g2_mul(BLS_G2, 171717)
    gt = ctypes.create_string_buffer(12 * N)
    assert L.elpo_pairing(g1b(P, N), g2b(Q, N), gt) == 1 and gt.raw == bls_ctx.pairing(g1b(P, N), g2b(Q, N))


def test_off_subgroup_g1_inputs_rejected_on_gpu(bls_ctx):
    """ELP_OPT_SUBGROUP_CHECK (default on): a valid proof whose phi / E1 / E2 got a component of order 3 or 11 of the G1 cofactor added is rejected, through the
    record path and the wire path (two lanes per item: the layout this curve always uses); the same for the commitment of a credential request.  The model and the
    C oracle reject the same inputs (tests/test_oracle_bls.py).  With the option off nothing crashes and the verdict still comes back per item."""
    from test_oracle_bls import _small_order_point
    seed, A, H = 4242, 4, 2
    g, gg = M.hash_to_g1("abc"), BLS_G2
    pk, skX = PR.key_gen(g, gg, scalar_stream(seed, 0, M.r), [scalar_stream(seed, 1 + i, M.r) for i in range(A)])
    apk, h = M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    bls_ctx.set_pubkey(g1b(pk.g, N), g2b(pk.gg, N), g2b(pk.XX, N), b"".join(g1b(P, N) for P in pk.Yi), b"".join(g2b(P, N) for P in pk.YYi), 6)
    bls_ctx.set_rp(b"service", g1b(apk, N), g1b(g, N), g1b(h, N))
    bls_ctx.set_signer_secret(g1b(skX, N))
    attrs = [(b"s-value", True), (b"gamma-value", True), (b"tp", False), (b"other", False)]
    rq, t1 = PR.request_id(pk, attrs, b"ad", [scalar_stream(seed, 50 + j, M.r) for j in range(2 + H)])
    u = scalar_stream(seed, 99, M.r)
    cred = PR.unblind(PR.provide_id(pk, skX, rq, b"ad", u), t1)
    rnd = [scalar_stream(seed, 200 + j, M.r) for j in range(3 + H + 2)]
    pr = PR.prove_id(pk, cred, attrs, b"sess", b"service", apk, g, h, rnd)
    t3, t11 = _small_order_point(3), _small_order_point(11, seed=7)
    items = [pr]
    for field, t in (("phi", t3), ("phi", t11), ("E1", t3), ("E2", t11), ("E2", t3)):
        bad = copy.copy(pr)
        setattr(bad, field, G.g1_add(getattr(pr, field), t))
        items.append(bad)
    items.append(pr)
    recs = b"".join(pack_verify_id(M, x) for x in items)
    mask = hidden_mask(pr.attributes)
    flags, cnt = bls_ctx.verify_id_batch(recs, mask, True, b"sess")
    assert list(flags) == [1, 0, 0, 0, 0, 0, 1] and cnt == 2
    assert [PR.verify_id(pk, x, b"sess", b"service", apk, g, h) for x in items] == [True] + [False] * 5 + [True]
    cd = importlib.import_module("elp_testlib").Codec(M)
    wflags, wcnt = bls_ctx.verify_id_wire_batch([cd.proof_encode(x) for x in items], True, b"sess")
    assert list(wflags) == list(flags)
    badrq = copy.copy(rq)
    badrq.A = G.g1_add(rq.A, t3)
    sigs, pflags, _ = bls_ctx.provide_id_batch(pack_provide_id(M, rq, u) + pack_provide_id(M, badrq, u), 3, b"ad")
    assert list(pflags) == [1, 0] and sigs[192:] == bytes(192)
    bls_ctx.set_subgroup_check(False)
    try:
        f2, _ = bls_ctx.verify_id_batch(recs, mask, True, b"sess")
        assert f2[0] == 1 and f2[6] == 1 and len(f2) == 7
    finally:
        bls_ctx.set_subgroup_check(True)


def test_cofactor_sig1_forgery_rejected_on_gpu(bls_ctx):
    """(sig1, sig2) = (T, O) with T of order 3 -- and an honest sig1 with T added -- under the library's default ELP_OPT_STRICT_SIGNATURE: rejected through the
    record path and the wire path; with the option off the lenient verdicts of the model come back (tests/test_oracle_bls.py has the CPU side: model, C oracle,
    host twin).  PSVerifier::verify's sig1 test is asked of the order-r component whatever the option says."""
    from oracle.pymodel import Credential
    from test_oracle_bls import _small_order_point
    seed, A, H = 4242, 4, 2
    g, gg = M.hash_to_g1("abc"), BLS_G2
    pk, skX = PR.key_gen(g, gg, scalar_stream(seed, 0, M.r), [scalar_stream(seed, 1 + i, M.r) for i in range(A)])
    apk, h = M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    bls_ctx.set_pubkey(g1b(pk.g, N), g2b(pk.gg, N), g2b(pk.XX, N), b"".join(g1b(P, N) for P in pk.Yi), b"".join(g2b(P, N) for P in pk.YYi), 6)
    bls_ctx.set_rp(b"service", g1b(apk, N), g1b(g, N), g1b(h, N))
    attrs = [(b"s-value", True), (b"gamma-value", True), (b"tp", False), (b"other", False)]
    rq, t1 = PR.request_id(pk, attrs, b"ad", [scalar_stream(seed, 50 + j, M.r) for j in range(2 + H)])
    cred = PR.unblind(PR.provide_id(pk, skX, rq, b"ad", scalar_stream(seed, 99, M.r)), t1)
    pr = PR.prove_id(pk, cred, attrs, b"sess", b"service", apk, g, h, [scalar_stream(seed, 200 + j, M.r) for j in range(3 + H + 2)])
    t3 = _small_order_point(3)
    forged, mixed = copy.copy(pr), copy.copy(pr)
    forged.sig1, forged.sig2 = t3, None
    mixed.sig1 = G.g1_add(pr.sig1, t3)
    items = [pr, forged, mixed, pr]
    recs = b"".join(pack_verify_id(M, x) for x in items)
    mask = hidden_mask(pr.attributes)
    cd = importlib.import_module("elp_testlib").Codec(M)
    wires = [cd.proof_encode(x) for x in items]
    for strict, want in ((True, [1, 0, 0, 1]), (False, [1, 1, 1, 1])):
        bls_ctx.set_strict_signature(strict)
        PR.strict = strict
        try:
            assert [int(PR.verify_id(pk, x, b"sess", b"service", apk, g, h)) for x in items] == want
            flags, cnt = bls_ctx.verify_id_batch(recs, mask, True, b"sess")
            assert list(flags) == want and cnt == sum(want)
            wflags, _ = bls_ctx.verify_id_wire_batch(wires, True, b"sess")
            assert list(wflags) == want
            aflags, acnt, _ = bls_ctx.verify_id_batch_aggregated(recs, mask, True, b"sess", seed=bytes(range(32)))
            assert list(aflags) == want and acnt == sum(want)
        finally:
            PR.strict = False
            bls_ctx.set_strict_signature(True)
    allattrs = [a for a, _ in attrs]
    pflags, pcnt = bls_ctx.ps_verify_batch(pack_ps_verify(M, cred, allattrs) + pack_ps_verify(M, Credential(t3, None), allattrs) +
                                           pack_ps_verify(M, Credential(G.g1_add(cred.sig1, t3), cred.sig2), allattrs), A)
    assert list(pflags) == [1, 0, 0] and pcnt == 1


def test_headline_size_batch_w20_against_the_c_oracle(bls_ctx):
    """BLS12-381 at its benchmarked configuration: 65 536 el_passo_verify_id proofs, 8 attributes with 4 hidden, id-retrieval, W = 20 tables,
    the two-lanes-per-item kernel this curve always uses -- every verdict against the generator's expectation and 1 024+ of them (a stride + EVERY corrupted item)
    against the C oracle's BLS12-381 build.  The oracle is pinned to the reference's wasm run on this curve (tests/test_oracle_bls_golden.py)."""
    import ctypes
    synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
    L = oracle_bls()
    A, H, n = 8, 4, 65536
    wl = synth.Workload(bls_ctx, A, seed=20211, window_bits=20)
    recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=True)
    flags, cnt = bls_ctx.verify_id_batch(recs, mask, True, wl.ad)
    bad = [i for i in range(n) if i % 97 == 13]
    assert (flags == expect).all() and cnt == int(expect.sum()) == n - len(bad)
    g1 = wl.g + wl.Yi + bls_ctx.hash_to_g1([wl.service]) + wl.g + wl.apk + wl.h + wl.X
    key = ctypes.c_void_p(L.elpo_key_new(A, g1, wl.gg + wl.XX + wl.YYi))
    assert key.value
    rsz = len(recs) // n
    idx = sorted(set(list(range(0, n, 149)) + bad))
    assert len(idx) >= 1024
    samp = b"".join(recs[i * rsz:(i + 1) * rsz] for i in idx)
    ofl = np.zeros(len(idx), dtype=np.uint8)
    import os
    L.elpo_verify_id_batch(key, len(idx), samp, rsz, mask, 1, wl.ad, len(wl.ad), ofl.ctypes.data, max(1, min(16, len(os.sched_getaffinity(0)))))
    assert (ofl == flags[idx]).all() and int((ofl == 0).sum()) == len(bad)
    L.elpo_key_free(key)
    # leave a small key behind for the tests that follow (22 GiB of W = 20 tables are released with it)
    synth.Workload(bls_ctx, 3, seed=1, window_bits=4)


def test_small_batches_cooperative_on_bls(bls_ctx):
    """Round 4: the cooperative small-batch path on BLS12-381 (programs generated by tools/gen_coop.py for the M-type twist and the Hayashida-Hayasaka-Teruya chain;
    k_vid_prep -> k_vid_small / k_vid_nizk4 + k_pair_coop -> k_vid_combine, k_ps_k_coop + k_pair_coop for PS verification, k_agg_final_coop as the tail of aggregated
    verification).  el_passo_verify_id and PS verification at n = 1, 63, 64, 65, 600 and 1100 (both program widths): verdicts equal the two-lanes-per-item kernels'
    (ELP_OPT_COOP_PAIRING off), the generator's expectation and the C oracle's BLS12-381 build; tampered signatures, sig2 = infinity and the cofactor forgery included.
    The oracle is pinned to the reference's wasm run on this curve (tests/test_oracle_bls_golden.py)."""
    import ctypes
    import os
    from test_oracle_bls import _small_order_point
    synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
    L = oracle_bls()
    L.elpo_set_strict.argtypes = [ctypes.c_int]
    NT = max(1, min(16, len(os.sched_getaffinity(0))))
    A, H = 8, 4
    wl = synth.Workload(bls_ctx, A, seed=5, window_bits=8)
    g1 = wl.g + wl.Yi + bls_ctx.hash_to_g1([wl.service]) + wl.g + wl.apk + wl.h + wl.X
    key = ctypes.c_void_p(L.elpo_key_new(A, g1, wl.gg + wl.XX + wl.YYi))
    t3 = g1b(_small_order_point(3), N)
    L.elpo_set_strict(1)                   # the context runs with the library's default ELP_OPT_STRICT_SIGNATURE
    try:
        for n in (1, 63, 64, 65, 600, 1100):
            recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=True, corrupt_every=7, corrupt_at=3)
            rsz = len(recs) // n
            recs = bytearray(recs)
            if n > 10:
                recs[9 * rsz + 96:9 * rsz + 192] = recs[8 * rsz + 96:8 * rsz + 192]        # item 9 gets item 8's sig2: NIZK holds, pairing check fails
                recs[4 * rsz + 96:4 * rsz + 192] = bytes(96)                                # item 4: sig2 = infinity (left to the per-lane kernel)
                recs[5 * rsz:5 * rsz + 96] = t3                                             # item 5: the cofactor forgery (sig1 of order 3, sig2 = O)
                recs[5 * rsz + 96:5 * rsz + 192] = bytes(96)
            recs = bytes(recs)
            bls_ctx.set_coop_pairing(0)
            f0, c0 = bls_ctx.verify_id_batch(recs, mask, True, wl.ad)
            bls_ctx.set_coop_pairing(1)
            f1, c1 = bls_ctx.verify_id_batch(recs, mask, True, wl.ad)
            assert (f0 == f1).all() and c0 == c1 == int(f1.sum()), n
            if n > 10:
                assert f1[9] == 0 and f1[4] == 0 and f1[5] == 0 and f1[8] == int(expect[8])
            ofl = np.zeros(n, dtype=np.uint8)
            L.elpo_verify_id_batch(key, n, recs, rsz, mask, 1, wl.ad, len(wl.ad), ofl.ctypes.data, NT)
            assert (ofl == f1).all(), n
            if n in (64, 600):          # the tail of aggregated verification on the cooperative program
                fa, ca, held = bls_ctx.verify_id_batch_aggregated(recs, mask, True, wl.ad, seed=bytes(range(32)))
                assert (fa == f1).all() and ca == c1
        wl3 = synth.Workload(bls_ctx, 3, seed=6, window_bits=8)
        g13 = wl3.g + wl3.Yi + bls_ctx.hash_to_g1([wl3.service]) + wl3.g + wl3.apk + wl3.h + wl3.X
        key3 = ctypes.c_void_p(L.elpo_key_new(3, g13, wl3.gg + wl3.XX + wl3.YYi))
        for n in (1, 63, 64, 65, 600, 1100):
            precs, pexpect = wl3.ps_verify_batch(n)
            bls_ctx.set_coop_pairing(0)
            p0, pc0 = bls_ctx.ps_verify_batch(precs, 3)
            bls_ctx.set_coop_pairing(1)         # K on 8 lanes per item (k_ps_k_coop), pairing check on 32 / 64 lanes (k_pair_coop)
            p1, pc1 = bls_ctx.ps_verify_batch(precs, 3)
            assert (p0 == pexpect).all() and (p1 == pexpect).all() and pc0 == pc1 == int(pexpect.sum()), n
            prsz = len(precs) // n
            for i in range(0, n, 1 if n < 100 else 37):
                assert L.elpo_ps_verify(key3, precs[i * prsz:(i + 1) * prsz], 3) == int(p1[i])
        L.elpo_key_free(key3)
    finally:
        L.elpo_set_strict(0)
        bls_ctx.set_coop_pairing(1)
        L.elpo_key_free(key)


def test_phase_mix_experiment_keeps_the_verdicts(elp, bls_ctx):
    """ELP_PHASE_MIX (experiment, off by default; profiles/r04_bls_ceiling.md): the second half of a two-lane launch's workgroups checks the pairing BEFORE the NIZK half.
    Same verdicts as the default order on a batch with corrupted NIZK responses, a swapped sig2, sig2 = infinity and the cofactor forgery -- items of both halves of the
    launch."""
    import os
    from test_oracle_bls import _small_order_point
    synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
    A, H, n = 8, 4, 520
    wl = synth.Workload(bls_ctx, A, seed=11, window_bits=8)
    recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=True, corrupt_every=5, corrupt_at=2)
    rsz = len(recs) // n
    recs = bytearray(recs)
    t3 = g1b(_small_order_point(3), N)
    for base in (8, 400):                                                                   # one group of special items in each half of the launch
        recs[(base + 1) * rsz + 96:(base + 1) * rsz + 192] = recs[base * rsz + 96:base * rsz + 192]          # foreign sig2: NIZK holds, pairing check fails
        recs[(base + 3) * rsz + 96:(base + 3) * rsz + 192] = bytes(96)                                         # sig2 = infinity
        recs[(base + 5) * rsz:(base + 5) * rsz + 96] = t3                                                      # the cofactor forgery
        recs[(base + 5) * rsz + 96:(base + 5) * rsz + 192] = bytes(96)
    recs = bytes(recs)
    bls_ctx.set_coop_pairing(0)                       # the two-lane kernel, not the cooperative path
    f0, c0 = bls_ctx.verify_id_batch(recs, mask, True, wl.ad)
    bls_ctx.set_coop_pairing(1)
    os.environ["ELP_PHASE_MIX"] = "1"
    try:
        ctx = elp.Context(elp.CURVE_BLS12_381, 0)
    finally:
        del os.environ["ELP_PHASE_MIX"]
    try:
        wl2 = synth.Workload(ctx, A, seed=11, window_bits=8)           # the same seed: the same key and proofs
        ctx.set_coop_pairing(0)
        f1, c1 = ctx.verify_id_batch(recs, mask, True, wl2.ad)
    finally:
        ctx.close()
    assert (f0 == f1).all() and c0 == c1
    assert f0[9] == 0 and f0[11] == 0 and f0[13] == 0 and f0[401] == 0 and f0[405] == 0 and int(f0.sum()) > n // 2


def test_four_lane_path_on_bls12_381(bls_ctx):
    """Round 5: el_passo_verify_id of 3 073 ... 16 384 items and PS verifications of 4 097 ... 16 384 run the pairing check on FOUR lanes per item (k_vid_mid / k_pair4,
    ELP_OPT_PAIR4; elp/quad.h, pair4.h with the M-type line product and the Hayashida-Hayasaka-Teruya chain cubed).  n = 3 100 and 8 200: verdicts equal the option-off
    paths', the generator's expectation and -- on a stride plus every corrupted / crafted item -- the C oracle's BLS12-381 build; tampered signature, sig2 = infinity and
    the cofactor forgery (sig1 of order 3, sig2 = O) included.  (What the reference's wasm answers on these inputs: tests/test_oracle_bls_golden.py.)"""
    import ctypes
    import os
    from test_oracle_bls import _small_order_point
    synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
    L = oracle_bls()
    L.elpo_set_strict.argtypes = [ctypes.c_int]
    NT = max(1, min(16, len(os.sched_getaffinity(0))))
    A, H = 8, 4
    wl = synth.Workload(bls_ctx, A, seed=55, window_bits=8)
    g1 = wl.g + wl.Yi + bls_ctx.hash_to_g1([wl.service]) + wl.g + wl.apk + wl.h + wl.X
    key = ctypes.c_void_p(L.elpo_key_new(A, g1, wl.gg + wl.XX + wl.YYi))
    t3 = g1b(_small_order_point(3), N)
    L.elpo_set_strict(1)
    try:
        for n in (3100, 8200):
            recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=True, corrupt_every=41, corrupt_at=3)
            rsz = len(recs) // n
            recs = bytearray(recs)
            recs[9 * rsz + 96:9 * rsz + 192] = recs[8 * rsz + 96:8 * rsz + 192]        # item 9 gets item 8's sig2: NIZK holds, pairing check fails
            recs[4 * rsz + 96:4 * rsz + 192] = bytes(96)                                # item 4: sig2 = infinity
            recs[5 * rsz:5 * rsz + 96] = t3                                             # item 5: the cofactor forgery
            recs[5 * rsz + 96:5 * rsz + 192] = bytes(96)
            recs = bytes(recs)
            bls_ctx.set_pair4(1)
            f4, c4 = bls_ctx.verify_id_batch(recs, mask, True, wl.ad)
            bls_ctx.set_pair4(0)
            f0, c0 = bls_ctx.verify_id_batch(recs, mask, True, wl.ad)
            bls_ctx.set_pair4(1)
            assert (f4 == f0).all() and c4 == c0 == int(f4.sum()), n
            assert f4[9] == 0 and f4[4] == 0 and f4[5] == 0 and f4[8] == int(expect[8])
            idx = sorted(set(list(range(0, n, n // 300)) + [i for i in range(n) if not expect[i]] + [4, 5, 8, 9]))
            samp = b"".join(recs[i * rsz:(i + 1) * rsz] for i in idx)
            ofl = np.zeros(len(idx), dtype=np.uint8)
            L.elpo_verify_id_batch(key, len(idx), samp, rsz, mask, 1, wl.ad, len(wl.ad), ofl.ctypes.data, NT)
            assert (ofl == f4[idx]).all(), n
        wl3 = synth.Workload(bls_ctx, 3, seed=56, window_bits=8)
        g13 = wl3.g + wl3.Yi + bls_ctx.hash_to_g1([wl3.service]) + wl3.g + wl3.apk + wl3.h + wl3.X
        key3 = ctypes.c_void_p(L.elpo_key_new(3, g13, wl3.gg + wl3.XX + wl3.YYi))
        n = 5000
        precs, pexpect = wl3.ps_verify_batch(n, corrupt_every=13, corrupt_at=2)
        prsz = len(precs) // n
        precs = bytearray(precs)
        precs[5 * prsz:5 * prsz + 96] = t3                                              # the forgery against PSVerifier::verify
        precs[5 * prsz + 96:5 * prsz + 192] = bytes(96)
        precs = bytes(precs)
        pexpect = pexpect.copy()
        pexpect[5] = 0
        p4, pc4 = bls_ctx.ps_verify_batch(precs, 3)
        bls_ctx.set_pair4(0)
        p0, pc0 = bls_ctx.ps_verify_batch(precs, 3)
        bls_ctx.set_pair4(1)
        assert (p4 == pexpect).all() and (p0 == pexpect).all() and pc4 == pc0 == int(pexpect.sum())
        for i in list(range(0, n, 50)) + [5]:
            assert L.elpo_ps_verify(key3, precs[i * prsz:(i + 1) * prsz], 3) == int(p4[i]), i
        L.elpo_key_free(key3)
    finally:
        L.elpo_set_strict(0)
        bls_ctx.set_pair4(1)
        L.elpo_key_free(key)
