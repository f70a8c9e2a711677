"""GPU parity tests added in round 3 (pytest -m gpu on the MI355X box), all through the C-ABI:
  * the two-phase el_passo_verify_id (k_vid_nizk with two job lanes per item + k_vid_pair) against the fused kernel, the C oracle and the golden verdicts."""
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
    NT = max(1, min(32, len(__import__("os").sched_getaffinity(0))))
except Exception:
    NT = 4


def _oracle_key(L, wl, ctx, A):
    g1 = wl.g + wl.Yi + ctx.hash_to_g1([wl.service]) + wl.g + wl.apk + wl.h + wl.X
    return ctypes.c_void_p(L.elpo_key_new(A, g1, wl.gg + wl.XX + wl.YYi))


@pytest.mark.parametrize("retr", [True, False])
def test_two_phase_equals_fused_and_oracle(gpu_ctx, retr):
    """k_vid_nizk + k_vid_pair vs the fused k_verify_id on the same records (with and without id-retrieval, ragged sizes that leave the last
    workgroup partly empty, corrupted items, group-law degenerate items), every verdict also compared with the C oracle."""
    L = oracle()
    A, H = 8, 4
    wl = synth.Workload(gpu_ctx, A, seed=7, window_bits=8)
    key = _oracle_key(L, wl, gpu_ctx, A)
    gpu_ctx.set_paired_layout(0)
    try:
        for n in (1, 63, 64, 65, 1000):
            recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=retr, corrupt_every=7, corrupt_at=3,
                                                   degenerate_items=(5,) if n > 5 else (), window_bits=8)
            gpu_ctx.set_split_phases(False)
            f0, c0 = gpu_ctx.verify_id_batch(recs, mask, retr, wl.ad)
            gpu_ctx.set_split_phases(2)         # the two jobs as concurrent kernels on two streams
            f2_, c2_ = gpu_ctx.verify_id_batch(recs, mask, retr, wl.ad)
            assert (f2_ == expect).all() and c2_ == int(expect.sum())
            gpu_ctx.set_split_phases(1)         # the two jobs as two waves of one workgroup
            f1, c1 = gpu_ctx.verify_id_batch(recs, mask, retr, wl.ad)
            assert (f0 == expect).all() and (f1 == expect).all() and c0 == c1 == int(expect.sum())
            rsz = len(recs) // n
            ofl = np.zeros(n, dtype=np.uint8)
            L.elpo_verify_id_batch(key, n, recs, rsz, mask, 1 if retr else 0, wl.ad, len(wl.ad), ofl.ctypes.data, NT)
            assert (ofl == f1).all()
            # a different session id verifies nothing, per-item associated data works through the offsets
            f2, c2 = gpu_ctx.verify_id_batch(recs, mask, retr, [wl.ad if i % 2 else b"other" for i in range(n)])
            assert all(bool(f2[i]) == (bool(expect[i]) and i % 2 == 1) for i in range(n))
    finally:
        gpu_ctx.set_paired_layout(2)
        gpu_ctx.set_split_phases(True)


def test_two_phase_golden_verdicts(gpu_ctx):
    """Every verdict of the reference's wasm on the A = 3 and A = 8 keys through the two-phase kernels (one lane per item forced)."""
    d = load_golden("bn254_oracle_flows.json")
    total = 0
    gpu_ctx.set_paired_layout(0)
    gpu_ctx.set_split_phases(True)
    try:
        for s in d["scenarios"][:2]:
            pk = CD.pk_decode(base64.b64decode(s["pk"]))
            groups = {}
            for p in s["proofs"]:
                for c in p["cases"]:
                    P = CD.proof_decode(base64.b64decode(c["proof"]))
                    groups.setdefault((c["svc"], hidden_mask(P.attributes)), []).append((pack_verify_id(M, P), c["ad"].encode(), c["expect"], c["label"]))
            gpu_ctx.set_pubkey(g1b(pk.g), g2b(pk.gg), g2b(pk.XX), b"".join(g1b(P) for P in pk.Yi), b"".join(g2b(P) for P in pk.YYi), 8)
            for (svc, mask), items in groups.items():
                gpu_ctx.set_rp(svc.encode())
                flags, cnt = gpu_ctx.verify_id_batch(b"".join(i[0] for i in items), mask, False, [i[1] for i in items])
                for f, it in zip(flags, items):
                    assert bool(f) == it[2], (s["name"], svc, it[3])
                total += len(items)
    finally:
        gpu_ctx.set_paired_layout(2)
    assert total > 100


def test_cpp_verifier_shards_over_several_contexts(gpu_ctx):
    """The C++ PSVerifier with a device list (three contexts on GPU 0: one host thread + one stream each, contiguous shards, counts summed):
    IdProof objects, wire messages and the packed wire buffer give the generator's verdicts, ragged shard sizes included."""
    import os
    b = importlib.import_module("ps-signature-and-el-passo_amd.build")
    L = ctypes.CDLL(b.HOST_LIB)
    L.elph_last_error.restype = ctypes.c_char_p
    A, H = 8, 4
    wl = synth.Workload(gpu_ctx, A, seed=11, window_bits=8)
    for n, nctx in ((301, 3), (64, 1), (5, 4), (1000, 1)):
        recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=True, corrupt_every=7, corrupt_at=2)
        msgs, moff = wl.wire_messages(recs, n, H, with_retrieval=True)
        moff = np.ascontiguousarray(moff, dtype=np.uint32)
        outv = (ctypes.c_double * 10)()
        acc = (ctypes.c_uint64 * 3)()
        flags = np.zeros(n, dtype=np.uint8)
        # _pipelined: the same three paths + el_passo_verify_id_submit / _collect with two batches in flight (one context: the overlapped path over
        # elp_verify_id_batch_submit / _wait; several contexts: its synchronous fallback), verdicts compared with the batch call's inside
        rc = L.elph_bench_verify_id_pipelined(ctypes.c_int(A), ctypes.c_int(H), wl.g, wl.gg, wl.XX, wl.Yi, wl.YYi, wl.apk, wl.g, wl.h, wl.service, wl.ad,
                                    recs, ctypes.c_size_t(n), ctypes.c_uint64(0), msgs, moff.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(8),
                                    ctypes.c_int(nctx), ctypes.c_int(0), ctypes.c_int(1), outv, acc, flags.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, L.elph_last_error()
        assert (flags == expect).all() and int(acc[0]) == int(acc[1]) == int(acc[2]) == int(expect.sum())


def test_cpp_verifier_over_all_visible_devices(elp):
    """Multi-GPU readiness: PSVerifier(pk, devices, W) built from the devices the process sees (elp_device_count), one context each, a batch sharded
    over them by the C++ dispatcher -- real hipSetDevice switching.  Skips on a one-GPU box (where test_cpp_verifier_shards_over_several_contexts covers the
    dispatcher with several contexts on device 0); the first run on a node exercises it."""
    lib = elp.load_library()
    nd = lib.elp_device_count()
    if nd < 2:
        pytest.skip("one GPU visible (%d): the several-contexts-on-one-device test covers the dispatcher here" % nd)
    ctx = elp.Context(elp.CURVE_BN254, 0)
    L = ctypes.CDLL(importlib.import_module("ps-signature-and-el-passo_amd.build").HOST_LIB)
    L.elph_last_error.restype = ctypes.c_char_p
    A, H, n = 8, 4, 4096 * nd
    wl = synth.Workload(ctx, A, seed=777, window_bits=8)
    recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=True)
    msgs, moff = wl.wire_messages(recs, n, H, with_retrieval=True)
    moff = np.ascontiguousarray(moff, dtype=np.uint32)
    outv, acc, flags = (ctypes.c_double * 8)(), (ctypes.c_uint64 * 3)(), np.zeros(n, dtype=np.uint8)
    rc = L.elph_bench_verify_id(ctypes.c_int(A), ctypes.c_int(H), wl.g, wl.gg, wl.XX, wl.Yi, wl.YYi, wl.apk, wl.g, wl.h, wl.service, wl.ad, recs,
                                ctypes.c_size_t(n), ctypes.c_uint64(0), msgs, moff.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(8), ctypes.c_int(1),
                                ctypes.c_int(-2), ctypes.c_int(1), outv, acc, flags.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, L.elph_last_error()
    assert (flags == expect).all() and int(acc[0]) == int(acc[1]) == int(acc[2]) == int(expect.sum())
    ctx.close()


def test_headline_batch_full_size_w20_with_4096_oracle_samples(gpu_ctx):
    """The batch bench.py sells: 65 536 el_passo_verify_id proofs, 8 attributes with 4 hidden, id-retrieval, W = 20 tables (15.5 GiB), one-lane-per-item
    kernel -- every verdict against the generator's expectation, and 4 096+ of them (a stride over the batch + EVERY corrupted item) against the C oracle
    (reference structure, src/ps-verifier.cc:37-138).  Also the two-phase kernels on the same batch."""
    L = oracle()
    A, H, n = 8, 4, 65536
    wl = synth.Workload(gpu_ctx, A, seed=20211, window_bits=20)
    recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=True)
    gpu_ctx.set_paired_layout(0)
    try:
        flags, cnt = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
        bad = [i for i in range(n) if i % 97 == 13]
        assert (flags == expect).all() and cnt == int(expect.sum()) == n - len(bad)
        for mode in (1, 2):
            gpu_ctx.set_split_phases(mode)
            f2, c2 = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
            assert (f2 == flags).all() and c2 == cnt
    finally:
        gpu_ctx.set_split_phases(0)
        gpu_ctx.set_paired_layout(2)
    key = _oracle_key(L, wl, gpu_ctx, A)
    rsz = len(recs) // n
    idx = sorted(set(list(range(0, n, 17)) + bad))
    assert len(idx) >= 4096
    samp = b"".join(recs[i * rsz:(i + 1) * rsz] for i in idx)
    ofl = np.zeros(len(idx), dtype=np.uint8)
    L.elpo_verify_id_batch(key, len(idx), samp, rsz, mask, 1, wl.ad, len(wl.ad), ofl.ctypes.data, NT)
    assert (ofl == flags[idx]).all() and int((ofl == 0).sum()) == len(bad)


def test_config5_end_to_end_eight_virtual_ranks(gpu_ctx):
    """BASELINE.json configuration 5 end to end on ONE GPU: 2^20 el_passo_verify_id proofs (16 attributes, 4 hidden, id-retrieval) cut by shard_range into the
    eight contiguous shards eight ranks would take, each shard verified in turn, per-shard accepted counts summed exactly as the count all-reduce would:
    total == 2^20 - #corrupted; per shard every verdict == the generator's, plus oracle samples in every shard."""
    shard = importlib.import_module("ps-signature-and-el-passo_amd.shard")
    L = oracle()
    A, H, N, R = 16, 4, 1 << 20, 8
    wl = synth.Workload(gpu_ctx, A, seed=20211, window_bits=16)
    key = _oracle_key(L, wl, gpu_ctx, A)
    total, covered = 0, 0
    for r in range(R):
        first, count = shard.shard_range(N, r, R)
        assert first == covered and count == N // R
        covered += count
        recs, mask, expect = wl.verify_id_batch(count, H, first_item=first)
        flags, cnt = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
        assert (flags == expect).all() and cnt == int(expect.sum())
        assert cnt == count - len([i for i in range(first, first + count) if i % 97 == 13])
        total += cnt
        rsz = len(recs) // count
        idx = list(range(0, count, 2048)) + [i for i in range(count) if (first + i) % 97 == 13][:8]
        samp = b"".join(recs[i * rsz:(i + 1) * rsz] for i in idx)
        ofl = np.zeros(len(idx), dtype=np.uint8)
        L.elpo_verify_id_batch(key, len(idx), samp, rsz, mask, 1, wl.ad, len(wl.ad), ofl.ctypes.data, NT)
        assert (ofl == flags[idx]).all()
    assert covered == N
    assert total == N - len([i for i in range(N) if i % 97 == 13])


def test_cooperative_pairing_small_batches(gpu_ctx):
    """ELP_OPT_COOP_PAIRING: batches of <= 8192 items run the pairing check on 32 lanes per item (k_pair_coop, level-scheduled program).  Plain PS verification and
    el_passo_verify_id at n = 1 ... 4096 (every switch of the small-batch path: 512 / 513, 1 792 / 1 793): verdicts equal the per-lane kernels' (option off), the generator's expectation and the C oracle's; items with
    sig2 = infinity / tampered signatures / corrupted NIZK take the same verdicts in both modes."""
    L = oracle()
    A, H = 8, 4
    wl = synth.Workload(gpu_ctx, A, seed=5, window_bits=8)
    key = _oracle_key(L, wl, gpu_ctx, A)
    gpu_ctx.set_pair4(0)        # the four-lane path of round 5 (default from 3 073 items) off: this test is about the interpreter at every size it can serve
    try:
        # sizes on both sides of every switch of the path: k_vid_small on 32 lane pairs per item (4 items per pairing workgroup: 3, 5 leave one partly filled) up to 512,
        # on 16 lane pairs (8 items per workgroup, all four waves interpreting) up to 1 792, its two-waves-per-SIMD build k_vid_small2 up to 4 096; the two-launch
        # form (k_vid_nizk4, k_pair_coop) runs under ELP_OPT_STREAM_OVERLAP below
        for n in (1, 3, 5, 63, 64, 65, 511, 513, 1027, 1792, 1793, 4096, 9216, 9217):      # 9 216 / 9 217: last cooperative batch / first two-lane one at the default limit
            recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=True, corrupt_every=7, corrupt_at=3)
            rsz = len(recs) // n
            recs = bytearray(recs)
            if n > 10:
                recs[9 * rsz + 64:9 * rsz + 128] = recs[8 * rsz + 64:8 * rsz + 128]        # item 9 gets item 8's sig2: NIZK holds, pairing check fails
                recs[4 * rsz + 64:4 * rsz + 128] = bytes(64)                                # item 4: sig2 = infinity (left to the per-lane kernel)
            recs = bytes(recs)
            gpu_ctx.set_coop_pairing(0)
            f0, c0 = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
            gpu_ctx.set_coop_pairing(1)
            f1, c1 = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
            assert (f0 == f1).all() and c0 == c1 == int(f1.sum())
            gpu_ctx.set_stream_overlap(1)          # ELP_OPT_STREAM_OVERLAP: pairing check beside the NIZK half on the context's second stream
            f2, c2 = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
            f3, c3 = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
            gpu_ctx.set_stream_overlap(0)
            assert (f2 == f1).all() and (f3 == f1).all() and c2 == c3 == c1
            if n > 10:
                assert f1[9] == 0 and f1[4] == 0 and f1[8] == int(expect[8])
            ofl = np.zeros(n, dtype=np.uint8)
            L.elpo_verify_id_batch(key, n, recs, rsz, mask, 1, wl.ad, len(wl.ad), ofl.ctypes.data, NT)
            assert (ofl == f1).all()
        wl3 = synth.Workload(gpu_ctx, 3, seed=6, window_bits=8)
        key3 = _oracle_key(L, wl3, gpu_ctx, 3)
        try:
            for n in (1, 63, 64, 65, 4096):
                precs, pexpect = wl3.ps_verify_batch(n)
                gpu_ctx.set_coop_pairing(0)
                p0, pc0 = gpu_ctx.ps_verify_batch(precs, 3)
                gpu_ctx.set_coop_pairing(1)         # K on 8 lanes per item (k_ps_k_coop), pairing check on 32 (k_pair_coop)
                p1, pc1 = gpu_ctx.ps_verify_batch(precs, 3)
                assert (p0 == pexpect).all() and (p1 == pexpect).all() and pc0 == pc1 == int(pexpect.sum())
                prsz = len(precs) // n
                for i in range(0, n, 1 if n < 100 else 37):
                    assert L.elpo_ps_verify(key3, precs[i * prsz:(i + 1) * prsz], 3) == int(p1[i])
        finally:
            L.elpo_key_free(key3)
        # aggregated verification with the cooperative closing step: fast path and fallback
        wl = synth.Workload(gpu_ctx, A, seed=5, window_bits=8)
        recs, mask, expect = wl.verify_id_batch(700, H, with_retrieval=True)
        fl, cnt, held = gpu_ctx.verify_id_batch_aggregated(recs, mask, True, wl.ad)
        assert held and (fl == expect).all() and cnt == int(expect.sum())
        gpu_ctx.set_stream_overlap(1)              # Fp12 product beside the Pippenger sum
        fl, cnt, held = gpu_ctx.verify_id_batch_aggregated(recs, mask, True, wl.ad)
        gpu_ctx.set_stream_overlap(0)
        assert held and (fl == expect).all() and cnt == int(expect.sum())
        rsz = len(recs) // 700
        bad = bytearray(recs)
        bad[9 * rsz + 64:9 * rsz + 128] = bad[8 * rsz + 64:8 * rsz + 128]
        fl2, cnt2, held2 = gpu_ctx.verify_id_batch_aggregated(bytes(bad), mask, True, wl.ad)
        exp2 = expect.copy()
        exp2[9] = 0
        assert not held2 and (fl2 == exp2).all()
    finally:
        gpu_ctx.set_pair4(1)
        gpu_ctx.set_coop_pairing(1)


@pytest.mark.gpu
def test_coalesced_record_loads_equal_in_place_reads(gpu_ctx):
    """ELP_OPT_COALESCED_RECORDS (default on): k_verify_id_staged fetches the 64 records of a workgroup as one contiguous block through LDS into a private copy;
    verdicts equal the in-place kernel's, the generator's expectation and the oracle's at ragged sizes (last workgroup partly empty, a lone item), with corrupted
    and degenerate items; records larger than the private copy (A = 24: 1312 bytes) silently take the in-place kernel."""
    L = oracle()
    A, H = 8, 4
    wl = synth.Workload(gpu_ctx, A, seed=11, window_bits=8)
    key = _oracle_key(L, wl, gpu_ctx, A)
    gpu_ctx.set_paired_layout(0)
    gpu_ctx.set_coop_pairing(0)            # small batches through the fused kernel too
    try:
        for n in (1, 63, 65, 777, 5000):
            recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=True, corrupt_every=5, corrupt_at=2, degenerate_items=(3,) if n > 10 else ())
            gpu_ctx.set_coalesced_records(0)
            f0, c0 = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
            gpu_ctx.set_coalesced_records(1)
            f1, c1 = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
            assert (f0 == f1).all() and (f1 == expect).all() and c0 == c1 == int(expect.sum())
            rsz = len(recs) // n
            ofl = np.zeros(n, dtype=np.uint8)
            m = min(n, 400)
            L.elpo_verify_id_batch(key, m, recs[:m * rsz], rsz, mask, 1, wl.ad, len(wl.ad), ofl.ctypes.data, NT)
            assert (ofl[:m] == f1[:m]).all()
        # without id-retrieval (672-byte records at A = 8, H = 4)
        recs, mask, expect = wl.verify_id_batch(300, H, with_retrieval=False, corrupt_every=9, corrupt_at=1)
        f1, c1 = gpu_ctx.verify_id_batch(recs, mask, False, wl.ad)
        assert (f1 == expect).all()
        # 1056-byte records (A = 16): four passes of 16 records through the LDS area; 1312-byte records (A = 24) exceed the private copy: in-place kernel
        for A2, seed in ((16, 12), (24, 13)):
            wl2 = synth.Workload(gpu_ctx, A2, seed=seed, window_bits=8)
            recs, mask, expect = wl2.verify_id_batch(130, H, with_retrieval=True, corrupt_every=6, corrupt_at=1)
            assert len(recs) // 130 == 672 + 32 * (A2 - H)
            f2, c2 = gpu_ctx.verify_id_batch(recs, mask, True, wl2.ad)
            assert (f2 == expect).all() and c2 == int(expect.sum())
    finally:
        L.elpo_key_free(key)
        gpu_ctx.set_paired_layout(2)
        gpu_ctx.set_coop_pairing(1)
