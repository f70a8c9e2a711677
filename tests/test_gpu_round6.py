"""GPU parity tests added in round 6 (pytest -m gpu on the MI355X box), all through the C-ABI.
  * the ROW-OF-16 pairing check (k_pair16, ELP_OPT_PAIR16; csrc/elpasso_pair16.h, tools/gen_row16.py) behind PSVerifier::verify (src/ps-verifier.cc:13-35) for small
    batches: every verdict against the C oracle, against the cooperative interpreter it replaces, on valid, tampered, infinite and undecodable signatures;
  * records handed over in parts (elp_verify_id_batch_stage) == the one-piece submit."""
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


def test_row_of_16_pairing_check_vs_oracle(gpu_ctx):
    """PS verification, A = 3 (BASELINE config 2's shape) at n = 1 ... 4 096 and A = 8: the row-of-16 kernel (default for at most 4 096 items) must give the C oracle's
    verdict (elpo_ps_verify) on EVERY item -- honest signatures, randomised ones, a wrong attribute, sig1 = infinity (rejected, src/ps-verifier.cc:16-18), sig2 = infinity,
    (sig1, sig2) swapped, a coordinate off the curve, a coordinate >= p -- and the cooperative interpreter's (ELP_OPT_PAIR16 = 0)."""
    L = oracle()
    for A, sizes in ((3, (1, 2, 3, 4, 5, 17, 63, 64, 65, 257, 1000, 4096)), (8, (7, 300))):
        wl = synth.Workload(gpu_ctx, A, seed=606 + A, window_bits=8)
        nmax = max(sizes)
        recs, expect = wl.ps_verify_batch(nmax)
        rsz = len(recs) // nmax
        r = bytearray(recs)
        step = 64 if nmax >= 64 else 1                                # the tampered items sit in the first 16 of every 64: every size above sees them
        for base in range(0, nmax - 15, max(step, 16)):
            r[(base + 1) * rsz:(base + 1) * rsz + 64] = bytes(64)                                   # sig1 = infinity
            r[(base + 2) * rsz + 64:(base + 2) * rsz + 128] = bytes(64)                              # sig2 = infinity
            a = bytes(r[(base + 3) * rsz:(base + 3) * rsz + 64])
            r[(base + 3) * rsz:(base + 3) * rsz + 64] = r[(base + 3) * rsz + 64:(base + 3) * rsz + 128]
            r[(base + 3) * rsz + 64:(base + 3) * rsz + 128] = a                                      # sig1 <-> sig2
            r[(base + 4) * rsz + 5] ^= 1                                                             # sig1.x: off the curve
            r[(base + 5) * rsz + 32:(base + 5) * rsz + 64] = b"\xff" * 32                            # sig1.y >= p
            r[(base + 6) * rsz + 128 + 3] ^= 1                                                       # a different attribute hash
            r[(base + 7) * rsz + 64 + 40] ^= 4                                                       # sig2.y: off the curve
        recs = bytes(r)
        key = _oracle_key(L, wl, gpu_ctx, A)
        want = np.array([L.elpo_ps_verify(key, recs[i * rsz:(i + 1) * rsz], A) for i in range(nmax)], dtype=np.uint8)
        assert want[0] == 1 and (nmax < 16 or (not want[1:8].any() and want[8:13].all()))
        try:
            for n in sizes:
                gpu_ctx.set_pair16(1)
                fl, cnt = gpu_ctx.ps_verify_batch(recs[:n * rsz], A)
                assert (fl == want[:n]).all() and cnt == int(want[:n].sum()), (A, n, np.nonzero(fl != want[:n])[0][:8])
                gpu_ctx.set_pair16(0)
                fl0, cnt0 = gpu_ctx.ps_verify_batch(recs[:n * rsz], A)
                assert (fl0 == fl).all() and cnt0 == cnt, (A, n)
        finally:
            gpu_ctx.set_pair16(1)
            L.elpo_key_free(key)


def test_row_of_16_pairing_check_bls12_381_vs_oracle(elp):
    """The same kernel generated for BLS12-381 (M-type lines, loop over |z|, the cubed Hayashida-Hayasaka-Teruya chain): PS verification at n = 4 ... 4 096 against the C
    oracle's BLS12-381 build on every item and against the interpreter, with the tamperings of the BN254 test plus a sig1 outside G1 (rejected: subgroup check on)."""
    import os
    from elp_testlib import oracle_bls
    L = oracle_bls()
    os.environ["ELP_PAIR16_MIN"] = "1"            # the rows at every size (the library's default on this curve starts them at 2 049 items, where they win)
    try:
        ctx = elp.Context(elp.CURVE_BLS12_381, 0)
    finally:
        del os.environ["ELP_PAIR16_MIN"]
    try:
        A = 3
        wl = synth.Workload(ctx, A, seed=707, window_bits=8)
        nmax = 4096
        recs, expect = wl.ps_verify_batch(nmax)
        rsz = len(recs) // nmax
        r = bytearray(recs)
        G1B = 96
        for base in range(0, nmax - 15, 64):
            r[(base + 1) * rsz:(base + 1) * rsz + G1B] = bytes(G1B)                                   # sig1 = infinity
            r[(base + 2) * rsz + G1B:(base + 2) * rsz + 2 * G1B] = bytes(G1B)                         # sig2 = infinity
            a = bytes(r[(base + 3) * rsz:(base + 3) * rsz + G1B])
            r[(base + 3) * rsz:(base + 3) * rsz + G1B] = r[(base + 3) * rsz + G1B:(base + 3) * rsz + 2 * G1B]
            r[(base + 3) * rsz + G1B:(base + 3) * rsz + 2 * G1B] = a                                  # sig1 <-> sig2
            r[(base + 4) * rsz + 5] ^= 1                                                              # sig1.x: off the curve
            r[(base + 6) * rsz + 2 * G1B + 3] ^= 1                                                    # a different attribute hash
        # sig1 = (0, p - 2): on the curve, order 3 (outside G1)
        p381 = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
        r[7 * rsz:7 * rsz + G1B] = bytes(48) + (p381 - 2).to_bytes(48, "little")
        recs = bytes(r)
        g1 = wl.g + wl.Yi + ctx.hash_to_g1([wl.service]) + wl.g + wl.apk + wl.h + wl.X
        key = ctypes.c_void_p(L.elpo_key_new(A, g1, wl.gg + wl.XX + wl.YYi))
        idx = sorted(set(list(range(0, 80)) + list(range(80, nmax, 13))))
        want = {i: L.elpo_ps_verify(key, recs[i * rsz:(i + 1) * rsz], A) for i in idx}
        assert want[0] == 1 and not any(want[i] for i in (1, 2, 3, 4, 6, 7))
        for n in (4, 5, 64, 65, 1000, 4096):
            ctx.set_pair16(1)
            fl, cnt = ctx.ps_verify_batch(recs[:n * rsz], A)
            for i in idx:
                if i < n:
                    assert fl[i] == want[i], (n, i)
            ctx.set_pair16(0)
            fl0, cnt0 = ctx.ps_verify_batch(recs[:n * rsz], A)
            assert (fl0 == fl).all() and cnt0 == cnt, n
        L.elpo_key_free(key)
    finally:
        ctx.close()


def test_records_staged_in_parts_equal_one_piece_submit(gpu_ctx):
    """elp_verify_id_batch_stage (round 6): the records of a batch delivered in ragged parts, out of order, then submitted with records == NULL, give the verdicts of
    the one-piece submit and of the generator; a submit after an incomplete staging fails without leaving the slot busy."""
    A, H, n = 8, 4, 5000
    wl = synth.Workload(gpu_ctx, A, seed=99, window_bits=8)
    recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=True, corrupt_every=11, corrupt_at=3)
    rsz = len(recs) // n
    lib, h = gpu_ctx.lib, gpu_ctx.h
    buf = np.frombuffer(recs, dtype=np.uint8).copy()
    ad = np.frombuffer(wl.ad, dtype=np.uint8).copy()
    flags = np.zeros(n, dtype=np.uint8)
    acc = ctypes.c_uint64(0)
    parts = [(0, 1), (4000, 1000), (1, 1999), (2000, 2000)]
    for first, count in parts:
        gpu_ctx._chk(lib.elp_verify_id_batch_stage(h, 0, n, rsz, first, count, buf.ctypes.data + first * rsz))
    gpu_ctx._chk(lib.elp_verify_id_batch_submit(h, 0, n, None, mask, 1, ad.ctypes.data, None, len(wl.ad), flags.ctypes.data))
    gpu_ctx._chk(lib.elp_verify_id_batch_wait(h, 0, ctypes.byref(acc)))
    assert (flags == expect).all() and acc.value == int(expect.sum())
    # incomplete staging: refused, and the slot is free afterwards
    gpu_ctx._chk(lib.elp_verify_id_batch_stage(h, 0, n, rsz, 0, 10, buf.ctypes.data))
    assert lib.elp_verify_id_batch_submit(h, 0, n, None, mask, 1, ad.ctypes.data, None, len(wl.ad), flags.ctypes.data) != 0
    flags[:] = 0
    gpu_ctx._chk(lib.elp_verify_id_batch_submit(h, 0, n, buf.ctypes.data, mask, 1, ad.ctypes.data, None, len(wl.ad), flags.ctypes.data))
    gpu_ctx._chk(lib.elp_verify_id_batch_wait(h, 0, ctypes.byref(acc)))
    assert (flags == expect).all()
    assert lib.elp_verify_id_batch_stage(h, 0, n, rsz, n - 1, 2, buf.ctypes.data) != 0              # a part beyond the batch


def test_aggregated_product_tree_on_rows_vs_oracle(gpu_ctx, elp):
    """The product of the per-wave Miller values of aggregated verification on 16-lane rows (k_fp12_reduce16, 32 values per wave; csrc/elpasso_pair16.h): batches whose
    wave counts exercise a lone value (1 wave), one row (8), two rows (9), a full workgroup plus two values (34), plus a row and one (41), and a second level of five
    values (129) -- (a) bad items fail the NIZK half only: the batch equation must HOLD, which it does only if the product is exact; (b) two swapped sig2: it must FAIL
    and the per-item fallback decide.  Every verdict against the C oracle (elpo_verify_id_batch, src/ps-verifier.cc:37-138) and against the one-lane product tree
    (ELP_OPT_PAIR16 = 0).  The BLS12-381 build of the same kernel at 34 and 41 waves, under both main kernels of that curve: on lane pairs (k_verify_id_agg_paired, the
    default: 32 items per wave, so twice the waves) and on one lane per item (ELP_AGG_PAIRED=0)."""
    import os
    from elp_testlib import oracle_bls
    seed = bytes(range(32, 64))
    for curve in ("bn254", "bls12_381", "bls12_381_one_lane"):
        L = oracle() if curve == "bn254" else oracle_bls()
        if curve == "bls12_381_one_lane":
            os.environ["ELP_AGG_PAIRED"] = "0"
        try:
            ctx = gpu_ctx if curve == "bn254" else elp.Context(elp.CURVE_BLS12_381, 0)
        finally:
            os.environ.pop("ELP_AGG_PAIRED", None)
        G1B = 64 if curve == "bn254" else 96
        A, H = 4, 2
        wl = synth.Workload(ctx, A, seed=808, window_bits=8)
        key = _oracle_key(L, wl, ctx, A)
        try:
            for n in ((1, 64 * 8, 64 * 8 + 1, 64 * 33 + 5, 64 * 41, 64 * 128 + 3) if curve == "bn254" else (64 * 33 + 5, 64 * 41)):
                recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=True, corrupt_every=11, corrupt_at=min(3, n - 1))
                rsz = len(recs) // n
                ofl = np.zeros(n, dtype=np.uint8)
                L.elpo_verify_id_batch(key, n, recs, rsz, mask, 1, wl.ad, len(wl.ad), ofl.ctypes.data, NT)
                assert (ofl == expect).all(), (curve, n)
                for rows in (1, 0):
                    ctx.set_pair16(rows)
                    fl, cnt, held = ctx.verify_id_batch_aggregated(recs, mask, True, wl.ad, seed)
                    assert held and (fl == ofl).all() and cnt == int(ofl.sum()), (curve, n, rows)
                if n >= 512:
                    a, b = n - 2, n // 2                                 # two accepted items of different waves: swap their sig2
                    while not (expect[a] and expect[b]):
                        a, b = a - 1, b + 1
                    r = bytearray(recs)
                    r[a * rsz + G1B:a * rsz + 2 * G1B], r[b * rsz + G1B:b * rsz + 2 * G1B] = r[b * rsz + G1B:b * rsz + 2 * G1B], r[a * rsz + G1B:a * rsz + 2 * G1B]
                    bad = bytes(r)
                    L.elpo_verify_id_batch(key, n, bad, rsz, mask, 1, wl.ad, len(wl.ad), ofl.ctypes.data, NT)
                    assert ofl[a] == 0 and ofl[b] == 0 and int(ofl.sum()) == int(expect.sum()) - 2
                    for rows in (1, 0):
                        ctx.set_pair16(rows)
                        fl, cnt, held = ctx.verify_id_batch_aggregated(bad, mask, True, wl.ad, seed)
                        assert not held and (fl == ofl).all() and cnt == int(ofl.sum()), (curve, n, rows)
        finally:
            ctx.set_pair16(1)
            L.elpo_key_free(key)
            if curve != "bn254":
                ctx.close()


def test_aggregated_verification_on_lane_pairs_bls12_381_vs_oracle(elp):
    """k_verify_id_agg_paired (round 6): the main kernel of aggregated verification on BLS12-381 with two lanes per item.  Against the C oracle's BLS12-381 build
    (elpo_verify_id / elpo_verify_id_batch, src/ps-verifier.cc:37-138) on every item: with and without id-retrieval, batches of 1 ... 1 061 items (one pair, a ragged
    last wave, more than one level of the product tree), items failing the NIZK half (equation holds), a foreign sig1 and a swapped pair of sig2 (equation fails, the
    per-item fallback decides), per-item associated data, sig1 = sig2 = infinity under the lenient and the strict rule, sig1 outside G1 (subgroup test on and off)."""
    from elp_testlib import oracle_bls
    L = oracle_bls()
    ctx = elp.Context(elp.CURVE_BLS12_381, 0)
    ctx.set_strict_signature(False)
    seed = bytes(range(64, 96))
    G1B = 96
    try:
        for A, H, retr in ((4, 2, True), (5, 3, False)):
            wl = synth.Workload(ctx, A, seed=900 + A, window_bits=8)
            key = _oracle_key(L, wl, ctx, A)
            try:
                for n in (1, 2, 31, 33, 64, 1061):
                    recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=retr, corrupt_every=7, corrupt_at=min(2, n - 1))
                    rsz = len(recs) // n
                    ofl = np.zeros(n, dtype=np.uint8)
                    L.elpo_verify_id_batch(key, n, recs, rsz, mask, int(retr), wl.ad, len(wl.ad), ofl.ctypes.data, NT)
                    assert (ofl == expect).all(), (A, n)
                    fl, cnt, held = ctx.verify_id_batch_aggregated(recs, mask, retr, wl.ad, seed)
                    assert held and (fl == ofl).all() and cnt == int(ofl.sum()), (A, n)
                    if n >= 31:
                        r = bytearray(recs)
                        a, b = 5, n - 2
                        while not expect[b]:
                            b -= 1
                        assert expect[a] and expect[b] and expect[10] and b > 12
                        r[a * rsz + G1B:a * rsz + 2 * G1B], r[b * rsz + G1B:b * rsz + 2 * G1B] = r[b * rsz + G1B:b * rsz + 2 * G1B], r[a * rsz + G1B:a * rsz + 2 * G1B]
                        r[10 * rsz:10 * rsz + G1B] = r[11 * rsz:11 * rsz + G1B]                  # a foreign sig1
                        r[12 * rsz:12 * rsz + 2 * G1B] = bytes(2 * G1B)                        # (O, O): accepted by the reference's VerifyID
                        bad = bytes(r)
                        L.elpo_verify_id_batch(key, n, bad, rsz, mask, int(retr), wl.ad, len(wl.ad), ofl.ctypes.data, NT)
                        assert ofl[a] == 0 and ofl[b] == 0 and ofl[10] == 0 and ofl[12] == 1
                        fl, cnt, held = ctx.verify_id_batch_aggregated(bad, mask, retr, wl.ad, seed)
                        assert not held and (fl == ofl).all() and cnt == int(ofl.sum()), (A, n)
                        # only the (O, O) item: the equation still holds (its pair contributes 1) under the lenient rule; the strict rule rejects the item up front
                        r = bytearray(recs)
                        r[12 * rsz:12 * rsz + 2 * G1B] = bytes(2 * G1B)
                        only = bytes(r)
                        want = expect.copy()
                        want[12] = 1
                        fl, cnt, held = ctx.verify_id_batch_aggregated(only, mask, retr, wl.ad, seed)
                        assert held and (fl == want).all()
                        ctx.set_strict_signature(True)
                        want[12] = 0
                        fl, cnt, held = ctx.verify_id_batch_aggregated(only, mask, retr, wl.ad, seed)
                        assert held and (fl == want).all()
                        ctx.set_strict_signature(False)
                # per-item associated data
                n = 67
                recs, mask, expect = wl.verify_id_batch(n, H, with_retrieval=retr, corrupt_every=9, corrupt_at=1)
                rsz = len(recs) // n
                ads = [wl.ad] * n
                ads[6] = b"another session"
                want = expect.copy()
                want[6] = 0
                fl, cnt, held = ctx.verify_id_batch_aggregated(recs, mask, retr, ads, seed)
                assert held and (fl == want).all()
                for i_ in (5, 6, 7):
                    assert L.elpo_verify_id(key, recs[i_ * rsz:(i_ + 1) * rsz], mask, int(retr), ads[i_], len(ads[i_])) == want[i_]
                # sig1 + T3 (T3 = (0, p - 2): order 3, outside G1): the pairing value is unchanged, so the reference accepts; the library's strict rule rejects
                p381 = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
                s1 = recs[8 * rsz:8 * rsz + G1B]
                t3 = bytes(48) + (p381 - 2).to_bytes(48, "little")
                r = bytearray(recs)
                r[8 * rsz:8 * rsz + G1B] = ctx.g1_add(s1, t3)
                off = bytes(r)
                assert expect[8]
                ctx.set_subgroup_check(False)
                L.elpo_set_subgroup_check(0)
                try:
                    assert L.elpo_verify_id(key, off[8 * rsz:9 * rsz], mask, int(retr), wl.ad, len(wl.ad)) == 1
                    fl, cnt, held = ctx.verify_id_batch_aggregated(off, mask, retr, wl.ad, seed)
                    assert held and (fl == expect).all()
                finally:
                    ctx.set_subgroup_check(True)
                    L.elpo_set_subgroup_check(1)
                ctx.set_strict_signature(True)
                want = expect.copy()
                want[8] = 0
                fl, cnt, held = ctx.verify_id_batch_aggregated(off, mask, retr, wl.ad, seed)
                assert held and (fl == want).all()
                ctx.set_strict_signature(False)
            finally:
                L.elpo_key_free(key)
    finally:
        ctx.close()


def test_g1_msm_with_split_scalars_on_exceptional_values(gpu_ctx, elp):
    """elp_g1_msm after round 6 splits every scalar with the GLV lattice (k_msm_split_scalars) and carries the signs of the halves into the bucket kernel.  Scalars that
    stress the split -- 0, 1, r - 1, r, r + 1, 2^256 - 1 (reduced modulo r first), lam and -lam (one half is 0 or +-1), 2^127, 2^128, values just around the lattice's
    rounding points -- on distinct points, a repeated point and the point at infinity, both curves: the sum must be the big-int model's (oracle/pymodel.py Groups.g1_mul / g1_add)."""
    import random
    from elp_testlib import BLS12_381, BLS_G1, BN254, Mcl, fb, g1b, g1u
    for curve, ctx, close in (("bn254", gpu_ctx, False), ("bls12_381", elp.Context(elp.CURVE_BLS12_381, 0), True)):
        try:
            M_ = Mcl(BN254 if curve == "bn254" else BLS12_381)
            G_ = M_.G
            N_ = 32 if curve == "bn254" else 48
            base = M_.hash_to_g1("msm-edge") if curve == "bn254" else BLS_G1
            lam = next(v for v in (pow(x, (M_.r - 1) // 3, M_.r) for x in range(2, 50)) if v != 1)      # a primitive cube root of unity modulo r: lam or lam^2 of the lattice
            rnd = random.Random(606)
            ks = [0, 1, M_.r - 1, M_.r, M_.r + 1, (1 << 256) - 1, 1 << 127, 1 << 128, (1 << 128) - 1, (M_.r + 1) // 2, M_.r // 3, 2 * M_.r // 3]
            if lam:
                ks += [lam, M_.r - lam, lam + 1, lam - 1, (lam * lam) % M_.r, (1 << 64) * lam % M_.r]
            ks += [rnd.randrange(1 << 256) for _ in range(14)]
            pts = [G_.g1_mul(base, rnd.randrange(1, M_.r)) for _ in ks]
            pts[3] = pts[2]                                            # a repeated point with r - 1 and r
            pts[7] = None                                              # the point at infinity with a full-size scalar
            want = None
            for P, k in zip(pts, ks):
                want = G_.g1_add(want, G_.g1_mul(P, k % M_.r))
            got = ctx.g1_msm(b"".join(g1b(P, N_) for P in pts), b"".join((k % (1 << 256)).to_bytes(32, "little") for k in ks))
            assert g1u(got, N_) == want, curve
            # the same scalars over one point: (sum k_i) P
            got1 = ctx.g1_msm(g1b(base, N_) * len(ks), b"".join((k % (1 << 256)).to_bytes(32, "little") for k in ks))
            assert g1u(got1, N_) == G_.g1_mul(base, sum(ks) % M_.r), curve
        finally:
            if close:
                ctx.close()
