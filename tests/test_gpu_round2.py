"""GPU parity tests added in round 2 (run on the MI355X box: pytest -m gpu), all through the C-ABI:
  * the HEADLINE table configuration (16-bit fixed-base windows, what bench.py measures) against the C oracle and the golden verdicts;
  * BASELINE.json configuration 5 at its real per-rank size (131 072 proofs, 16 attributes, 4 hidden);
  * the edge cases pinned by the reference's wasm (off-subgroup k, 3-byte lengths), record path and wire path;
  * ELP_OPT_STRICT_SIGNATURE and the key-state checks of the fused entry points."""
import base64
import ctypes
import importlib

import numpy as np
import pytest

from elp_testlib import BN254, Codec, Mcl, Protocol, g1b, g2b, hidden_mask, load_golden, oracle, pack_verify_id

pytestmark = pytest.mark.gpu

M = Mcl(BN254)
CD, PR = Codec(M), Protocol(M)
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
try:
    NT = max(1, min(32, len(__import__("os").sched_getaffinity(0))))      # OpenMP threads of the C oracle's batch entry
except Exception:
    NT = 4


def _set_key(ctx, pk, W, svc=None):
    ctx.set_pubkey(g1b(pk.g), g2b(pk.gg), g2b(pk.XX), b"".join(g1b(P) for P in pk.Yi), b"".join(g2b(P) for P in pk.YYi), W)
    if svc is not None:
        ctx.set_rp(svc.encode() if isinstance(svc, str) else svc)


def _oracle_key(L, wl, ctx, A):
    g1 = wl.g + wl.Yi + ctx.hash_to_g1([wl.service]) + wl.g + wl.apk + wl.h + wl.X
    return ctypes.c_void_p(L.elpo_key_new(A, g1, wl.gg + wl.XX + wl.YYi))


@pytest.mark.parametrize("W", [20, 16])
def test_headline_window_config4_vs_oracle(gpu_ctx, W):
    """bench.py's configuration: 8 attributes, 4 hidden, id-retrieval, W = 20 tables (15.5 GiB of signed-digit entries, 13 windows of 20-bit digits; W = 16 was round 1's:
    2.5 GiB, its own k_table_fill chunking).  2 048 proofs incl. every-97th corrupted and valid proofs that send the group law through P + P at
    this window width, every verdict compared with the C oracle (reference structure, src/ps-verifier.cc:37-138), in both kernel layouts."""
    L = oracle()
    A, H, n = 8, 4, 2048
    wl = synth.Workload(gpu_ctx, A, seed=20211, window_bits=W)
    degenerate = (5, 77, 150, 1000, 2047)
    recs, mask, expect = wl.verify_id_batch(n, H, degenerate_items=degenerate, window_bits=W)
    gpu_ctx.set_paired_layout(0)
    flags0, cnt0 = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
    gpu_ctx.set_paired_layout(2)
    assert (flags0 == expect).all() and cnt0 == int(expect.sum())
    flags, cnt = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
    assert (flags == expect).all() and cnt == int(expect.sum())
    assert all(flags[i] == 1 for i in degenerate) and flags[13] == 0 and flags[110] == 0
    key = _oracle_key(L, wl, gpu_ctx, A)
    rsz = len(recs) // n
    ofl = np.zeros(n, dtype=np.uint8)
    L.elpo_verify_id_batch(key, n, recs, rsz, mask, 1, wl.ad, len(wl.ad), ofl.ctypes.data, NT)
    assert (ofl == flags).all()
    # the same records as wire messages (T-L-V parse, decompression, attribute hashing in the kernel) and through the aggregated path
    msgs, moff = wl.wire_messages(recs, n, H, with_retrieval=True)
    m = [msgs[moff[i]:moff[i + 1]] for i in range(n)]
    wflags, wcnt = gpu_ctx.verify_id_wire_batch(m, True, wl.ad)
    assert (wflags == flags).all() and wcnt == cnt
    aflags, acnt, held = gpu_ctx.verify_id_batch_aggregated(recs, mask, True, wl.ad)      # library-drawn seed
    assert held and (aflags == flags).all() and acnt == cnt
    # issuance and PS verification on the W = 16 tables vs the oracle
    precs, pmask, pexpect = wl.provide_id_batch(300, H)
    sigs, pflags, pcnt = gpu_ctx.provide_id_batch(precs, pmask, wl.ad)
    assert (pflags == pexpect).all()
    prsz = len(precs) // 300
    out = ctypes.create_string_buffer(128)
    for i in (0, 13, 64, 110, 299):
        assert L.elpo_provide_id(key, precs[i * prsz:(i + 1) * prsz], pmask, wl.ad, len(wl.ad), out) == int(pflags[i])
        assert out.raw == sigs[128 * i:128 * i + 128]


def test_golden_no_retrieval_window16(gpu_ctx):
    """Every verdict of the reference's wasm on the A = 3 and A = 8 keys, W = 16 tables."""
    d = load_golden("bn254_oracle_flows.json")
    total = 0
    for s in d["scenarios"][:2]:
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        groups = {}
        for p in s["proofs"]:
            for c in p["cases"]:
                P = CD.proof_decode(base64.b64decode(c["proof"]))
                groups.setdefault((c["svc"], hidden_mask(P.attributes)), []).append((pack_verify_id(M, P), c["ad"].encode(), c["expect"], c["label"]))
        _set_key(gpu_ctx, pk, 16)
        for (svc, mask), items in groups.items():
            gpu_ctx.set_rp(svc.encode())
            flags, cnt = gpu_ctx.verify_id_batch(b"".join(i[0] for i in items), mask, False, [i[1] for i in items])
            for f, it in zip(flags, items):
                assert bool(f) == it[2], (s["name"], svc, it[3])
            total += len(items)
    assert total > 100


def test_config5_per_rank_share_full_size(gpu_ctx):
    """BASELINE.json configuration 5, one rank's real share: 131 072 el_passo_verify_id proofs, 16 attributes (4 hidden), W = 16 tables.
    Size-independent properties (accept pattern = the generator's every-97th corruption, counter = number of ones, idempotent re-run,
    nothing verifies under a different session id) plus 512 oracle samples spread over the shard."""
    L = oracle()
    A, H, n = 16, 4, 131072
    rank = 5                                               # the shard rank 5 of 8 would take: items [5n, 6n)
    wl = synth.Workload(gpu_ctx, A, seed=20211, window_bits=16)
    recs, mask, expect = wl.verify_id_batch(n, H, first_item=rank * n)
    rsz = 5 * 64 + 128 + 32 * (A + 3)
    assert len(recs) == n * rsz == n * 1056                # + 4-byte verdict = 1060 B
    flags, cnt = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
    bad = [i for i in range(n) if (rank * n + i) % 97 == 13]
    assert (flags == expect).all() and cnt == int(expect.sum()) == n - len(bad)
    flags2, cnt2 = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
    assert (flags2 == flags).all() and cnt2 == cnt
    sub = recs[:4096 * rsz]
    f3, c3 = gpu_ctx.verify_id_batch(sub, mask, True, b"hellO")
    assert c3 == 0 and not f3.any()
    key = _oracle_key(L, wl, gpu_ctx, A)
    idx = sorted(set(list(range(0, n, 257)) + bad[:24]))[:512]
    samp = b"".join(recs[i * rsz:(i + 1) * rsz] for i in idx)
    ofl = np.zeros(len(idx), dtype=np.uint8)
    L.elpo_verify_id_batch(key, len(idx), samp, rsz, mask, 1, wl.ad, len(wl.ad), ofl.ctypes.data, NT)
    assert (ofl == flags[idx]).all() and (ofl == 0).sum() >= 24


def test_edge_cases_pinned_by_the_reference(gpu_ctx):
    """tests/golden/bn254_oracle_edge.json: off-subgroup k (rejected by the reference, also the proof crafted to pass a plain-multiplication
    Schnorr check) and the 3-byte length forms (accepted).  Record path and wire path, W = 8 and W = 16."""
    d = load_golden("bn254_oracle_edge.json")
    for W in (8, 16):
        by_key = {}
        for c in d["cases"]:
            by_key.setdefault((c["pk"], c["svc"]), []).append(c)
        for (pkb, svc), cases in by_key.items():
            pk = CD.pk_decode(base64.b64decode(pkb))
            _set_key(gpu_ctx, pk, W, svc)
            raws = [base64.b64decode(c["proof"]) for c in cases]
            Ps = [CD.proof_decode(r) for r in raws]
            masks = {hidden_mask(P.attributes) for P in Ps}
            assert len(masks) == 1
            flags, _ = gpu_ctx.verify_id_batch(b"".join(pack_verify_id(M, P) for P in Ps), masks.pop(), False, [c["ad"].encode() for c in cases])
            wflags, _ = gpu_ctx.verify_id_wire_batch(raws, False, [c["ad"].encode() for c in cases])
            for c, f, wf in zip(cases, flags, wflags):
                assert bool(f) == c["expect"], ("record", W, c["scenario"], c["label"])
                assert bool(wf) == c["expect"], ("wire", W, c["scenario"], c["label"])


def test_strict_signature_default_and_state_checks(elp):
    """A fresh context (library defaults): sig1 == infinity is rejected on every verify_id path while all other golden verdicts stay
    the reference's; fused entry points refuse to run on missing key material instead of using infinity bases."""
    ctx = elp.Context(elp.CURVE_BN254, 0)
    try:
        d = load_golden("bn254_oracle_flows.json")
        s = d["scenarios"][0]
        pk = CD.pk_decode(base64.b64decode(s["pk"]))
        rec0 = pack_verify_id(M, CD.proof_decode(base64.b64decode(s["proofs"][0]["cases"][0]["proof"])))
        fl, cnt = np.zeros(1, dtype=np.uint8), ctypes.c_uint64(0)          # no public key yet: ELP_ERR_STATE straight from the C-ABI
        assert ctx.lib.elp_verify_id_batch(ctx.h, 1, rec0, 0b011, 0, b"x", None, 1, fl.ctypes.data, ctypes.byref(cnt)) == -3
        _set_key(ctx, pk, 8)
        with pytest.raises(elp.ElpassoError):                         # public key but no elp_set_rp: H1(service) would be infinity
            ctx.verify_id_batch(rec0, 0b011, False, b"x")
        with pytest.raises(elp.ElpassoError):
            ctx.verify_id_wire_batch([b"\x01"], False, b"x")
        with pytest.raises(elp.ElpassoError):                         # issuance without the signer secret
            ctx.provide_id_batch(bytes(ctx.lib.elp_provide_id_record_size(0, 3, 2)), 0b011, b"x")
        ctx.set_rp(b"svc")
        with pytest.raises(elp.ElpassoError):                         # id-retrieval without authority_pk / g / h
            ctx.verify_id_batch(bytes(ctx.lib.elp_verify_id_record_size(0, 3, 2, 1)), 0b011, True, b"x")
        seen_zero = 0
        for p in s["proofs"]:
            ctx.set_rp(p["svc"].encode())
            items = [c for c in p["cases"] if c["svc"] == p["svc"]]
            Ps = [CD.proof_decode(base64.b64decode(c["proof"])) for c in items]
            mask = hidden_mask(Ps[0].attributes)
            recs = b"".join(pack_verify_id(M, P) for P in Ps)
            ads = [c["ad"].encode() for c in items]
            want = [c["expect"] and c["label"] != "sig_both_zero" for c in items]
            seen_zero += sum(1 for c in items if c["label"] == "sig_both_zero" and c["expect"])
            flags, cnt = ctx.verify_id_batch(recs, mask, False, ads)
            assert [bool(f) for f in flags] == want and cnt == sum(want)
            wflags, _ = ctx.verify_id_wire_batch([base64.b64decode(c["proof"]) for c in items], False, ads)
            assert [bool(f) for f in wflags] == want
            aflags, acnt, _ = ctx.verify_id_batch_aggregated(recs, mask, False, ads)
            assert [bool(f) for f in aflags] == want and acnt == sum(want)
            ctx.set_strict_signature(False)
            flags, _ = ctx.verify_id_batch(recs, mask, False, ads)
            assert [bool(f) for f in flags] == [c["expect"] for c in items]
            ctx.set_strict_signature(True)
        assert seen_zero >= 2
        with pytest.raises(ValueError):
            ctx.verify_id_batch_aggregated(recs, mask, False, ads, seed=b"short")
    finally:
        ctx.close()


def test_paired_and_plain_layouts_agree(gpu_ctx):
    """ELP_OPT_PAIRED_LAYOUT: the two-lanes-per-item kernels (default) and the one-lane-per-item kernels give identical verdicts on
    valid, NIZK-corrupted, signature-tampered, infinity and garbage records, for ragged batch sizes (pairs straddling nothing: 32 items
    per wave), with and without id-retrieval, and for plain PS verification."""
    A, H = 8, 4
    wl = synth.Workload(gpu_ctx, A, seed=77, window_bits=8)
    for retr in (True, False):
        n = 333
        recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=retr, degenerate_items=(5, 64))
        rsz = len(recs) // n
        r = bytearray(recs)
        r[7 * rsz + 64:7 * rsz + 128], r[8 * rsz + 64:8 * rsz + 128] = r[8 * rsz + 64:8 * rsz + 128], r[7 * rsz + 64:7 * rsz + 128]   # swap sig2
        r[100 * rsz:100 * rsz + 64] = r[101 * rsz:101 * rsz + 64]                      # foreign sig1
        r[200 * rsz:200 * rsz + 128] = bytes(128)                                      # (inf, inf)
        r[201 * rsz + 128:201 * rsz + 192] = bytes(64)                                 # phi = inf
        r[202 * rsz + 3] ^= 0x40                                                       # sig1.x off the curve
        koff = (5 if retr else 3) * 64
        r[203 * rsz + koff + 40] ^= 1                                                  # k off the curve
        r[204 * rsz + koff:204 * rsz + koff + 128] = bytes(128)                        # k = inf
        rnd = np.random.RandomState(3)
        r[300 * rsz:301 * rsz] = rnd.randint(0, 256, size=rsz, dtype=np.uint8).tobytes()
        bad = bytes(r)
        for cut in (1, 31, 32, 33, 65, n):
            gpu_ctx.set_paired_layout(True)
            fp, cp = gpu_ctx.verify_id_batch(bad[:cut * rsz], mask, retr, wl.ad)
            gpu_ctx.set_paired_layout(False)
            fq, cq = gpu_ctx.verify_id_batch(bad[:cut * rsz], mask, retr, wl.ad)
            gpu_ctx.set_paired_layout(2)
            assert (fp == fq).all() and cp == cq == int(fp.sum()), (retr, cut)
            # ELP_OPT_TABLE_WORKSPACE off: the tables of 1P .. 8P in private memory instead of the launch workspace, both layouts
            try:
                gpu_ctx.set_table_workspace(False)
                gpu_ctx.set_paired_layout(True)
                fp2, cp2 = gpu_ctx.verify_id_batch(bad[:cut * rsz], mask, retr, wl.ad)
                gpu_ctx.set_paired_layout(False)
                fq2, cq2 = gpu_ctx.verify_id_batch(bad[:cut * rsz], mask, retr, wl.ad)
            finally:
                gpu_ctx.set_table_workspace(True)
                gpu_ctx.set_paired_layout(2)
            assert (fp2 == fp).all() and (fq2 == fp).all() and cp2 == cq2 == cp, (retr, cut)
        assert fp[5] == 1 and fp[64] == 1 and fp[7] == 0 and fp[8] == 0 and fp[100] == 0 and fp[13] == 0 and fp[202] == 0 and fp[203] == 0
        assert fp[200] == 1 and fp[9] == 1       # the fixture context runs in reference-compatible mode: (inf, inf) is accepted
        # the same (valid and corrupted) proofs as wire messages through both layouts' wire-ingest kernels
        msgs, moff = wl.wire_messages(recs, n, H, with_retrieval=retr)
        m = [msgs[moff[i]:moff[i + 1]] for i in range(n)]
        m[40] = m[40][:-5]                       # truncated
        m[41] = m[41] + b"\x00"                  # trailing byte
        gpu_ctx.set_paired_layout(True)
        wp, wcp = gpu_ctx.verify_id_wire_batch(m, retr, wl.ad)
        gpu_ctx.set_paired_layout(False)
        wq, wcq = gpu_ctx.verify_id_wire_batch(m, retr, wl.ad)
        gpu_ctx.set_paired_layout(2)
        assert (wp == wq).all() and wcp == wcq and wp[40] == 0 and wp[13] == 0 and wp[9] == 1 and wp[5] == 1
    recs, expect = wl.ps_verify_batch(130)
    gpu_ctx.set_paired_layout(True)
    fp, cp = gpu_ctx.ps_verify_batch(recs, A)
    gpu_ctx.set_paired_layout(False)
    fq, cq = gpu_ctx.ps_verify_batch(recs, A)
    gpu_ctx.set_paired_layout(2)
    assert (fp == fq).all() and (fp == expect).all() and cp == cq


def test_layout_by_batch_size_splits_a_batch(gpu_ctx):
    """Default policy on a batch of one full round + a small remainder (65 536 + 4 196 items, per-item session ids): verdicts, counter and
    per-item associated data must line up whatever kernel(s) the library picks."""
    A, H = 3, 2
    wl = synth.Workload(gpu_ctx, A, seed=5, window_bits=8)
    n0 = 500
    recs, mask, expect = wl.verify_id_batch(n0, H, with_retrieval=True)
    rsz = len(recs) // n0
    n = 65536 + 4096 + 100                       # one full round (plain) + a remainder (paired) on a 256-CU part
    reps = (n + n0 - 1) // n0
    big = (recs * reps)[:n * rsz]
    exp = np.tile(expect, reps)[:n]
    gpu_ctx.set_paired_layout(2)
    flags, cnt = gpu_ctx.verify_id_batch(big, mask, True, wl.ad)
    assert (flags == exp).all() and cnt == int(exp.sum())
    ads = [wl.ad if i % 1000 else b"other" for i in range(n)]     # per-item associated data: every 1000th item gets a wrong one
    flags2, cnt2 = gpu_ctx.verify_id_batch(big, mask, True, ads)
    exp2 = exp.copy()
    exp2[::1000] = 0
    assert (flags2 == exp2).all() and cnt2 == int(exp2.sum())
