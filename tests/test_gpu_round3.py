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
