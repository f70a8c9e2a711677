"""GPU parity tests added in round 5 (pytest -m gpu on the MI355X box), all through the C-ABI.  VERDICT r4 "thicken parity where it is thin":
  * BASELINE config 3 at its full size -- 65 536 x el_passo_provide_id (src/ps-signer.cc:63-146): 1 024+ issued signatures (a stride over the batch + EVERY
    rejected slot) byte for byte against the C oracle's elpo_provide_id, not through properties only;
  * BASELINE config 4 through the WIRE path at its full size -- 65 536 IdProof::toBufferString() messages (src/ps-encoding.cc:451-467) parsed, decompressed and
    hashed on the GPU: 1 024+ verdicts against the C oracle (src/ps-verifier.cc:37-138) on the same items' records, every corrupted item included."""
import ctypes
import importlib

import numpy as np
import pytest

from elp_testlib import oracle

pytestmark = pytest.mark.gpu
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
try:
    NT = max(1, min(32, len(__import__("os").sched_getaffinity(0))))
except Exception:
    NT = 4


def _oracle_key(L, wl, ctx, A):
    g1 = wl.g + wl.Yi + ctx.hash_to_g1([wl.service]) + wl.g + wl.apk + wl.h + wl.X
    return ctypes.c_void_p(L.elpo_key_new(A, g1, wl.gg + wl.XX + wl.YYi))


def test_config3_full_size_signatures_byte_for_byte_vs_oracle(gpu_ctx):
    """65 536 issuances (8 attributes, 4 hidden, injected nonce u): flags == the generator's expectation for every item, and for a stride over the batch plus every
    rejected slot the 128 signature bytes (sig1 | sig2) equal the C oracle's -- rejected slots are all-zero on both sides."""
    L = oracle()
    A, H, n = 8, 4, 65536
    wl = synth.Workload(gpu_ctx, A, seed=20211, window_bits=16)
    recs, mask, expect = wl.provide_id_batch(n, H)
    sigs, flags, cnt = gpu_ctx.provide_id_batch(recs, mask, b"hello")
    assert (flags == expect).all() and cnt == int(expect.sum())
    bad = [i for i in range(n) if not expect[i]]
    assert len(bad) == len([i for i in range(n) if i % 97 == 13])
    idx = sorted(set(list(range(0, n, 61)) + bad))
    assert len(idx) >= 1024 + len(bad) // 2
    key = _oracle_key(L, wl, gpu_ctx, A)
    rsz = len(recs) // n
    out = ctypes.create_string_buffer(128)
    for i in idx:
        ok = L.elpo_provide_id(key, recs[i * rsz:(i + 1) * rsz], mask, b"hello", 5, out)
        assert ok == int(flags[i]), i
        assert sigs[128 * i:128 * i + 128] == out.raw, i
        if not ok:
            assert out.raw == bytes(128)
    L.elpo_key_free(key)


def test_config4_wire_path_full_size_vs_oracle(gpu_ctx):
    """65 536 wire messages (511 bytes each at A = 8, H = 4 with id-retrieval) through elp_verify_id_wire_batch: every verdict equals the generator's expectation and
    the record path's, and 1 024+ of them (stride + every corrupted item) equal the C oracle's verdict on the item's record."""
    L = oracle()
    A, H, n = 8, 4, 65536
    wl = synth.Workload(gpu_ctx, A, seed=20211, window_bits=16)
    recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=True)
    msgs, moff = wl.wire_messages(recs, n, H, with_retrieval=True)
    moff = np.asarray(moff, dtype=np.int64)
    mlist = [msgs[int(moff[i]):int(moff[i + 1])] for i in range(n)]
    wflags, wcnt = gpu_ctx.verify_id_wire_batch(mlist, True, wl.ad)
    bad = [i for i in range(n) if i % 97 == 13]
    assert (wflags == expect).all() and wcnt == int(expect.sum()) == n - len(bad)
    rflags, rcnt = gpu_ctx.verify_id_batch(recs, mask, True, wl.ad)
    assert (rflags == wflags).all() and rcnt == wcnt
    key = _oracle_key(L, wl, gpu_ctx, A)
    rsz = len(recs) // n
    idx = sorted(set(list(range(0, n, 61)) + bad))
    assert len(idx) >= 1024 + len(bad) // 2
    samp = b"".join(recs[i * rsz:(i + 1) * rsz] for i in idx)
    ofl = np.zeros(len(idx), dtype=np.uint8)
    L.elpo_verify_id_batch(key, len(idx), samp, rsz, mask, 1, wl.ad, len(wl.ad), ofl.ctypes.data, NT)
    assert (ofl == wflags[idx]).all() and int((ofl == 0).sum()) == len(bad)
    L.elpo_key_free(key)
