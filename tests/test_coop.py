"""Cooperative pairing check (csrc/elp/coop.h + the level-scheduled programs of tools/gen_coop.py) on the CPU: the host twin interprets the SAME program tables
the kernels k_pair_coop / k_agg_final_coop carry, slot by slot, and the resulting GT element must equal the model's  e(sig1, K) e(-sig2, gg)  bit for bit
(the model computes pairings with affine formulas and a plain final exponentiation); verdicts on PS signatures issued through the reference's protocol."""
import base64
import ctypes

import pytest

from elp_testlib import BN254, Codec, Mcl, Protocol, fb, g1_bases, g1b, g2_bases, g2b, load_golden, scalar_stream, twin

M = Mcl(BN254)
G, CD, PR = M.G, Codec(M), Protocol(M)


def make_env(L):
    d = load_golden("bn254_oracle_flows.json")
    pk = CD.pk_decode(base64.b64decode(d["scenarios"][0]["pk"]))
    L.twin_bn254_ctx_new.restype = ctypes.c_void_p
    h = L.twin_bn254_ctx_new(len(pk.Yi), 4, g1_bases(M, pk, svc=b"svc"), g2_bases(M, pk))
    assert h
    return L, ctypes.c_void_p(h), pk


@pytest.fixture(scope="module")
def env():
    return make_env(twin())


def _gt_bytes(e):
    return b"".join(fb(e[k][0]) + fb(e[k][1]) for k in [0, 2, 4, 1, 3, 5])


def test_program_value_equals_model_pairing_product(env):
    L, ctx, pk = env
    g1 = M.hash_to_g1("abc")
    out = ctypes.create_string_buffer(384)
    for a, b, k in ((424242, 171717, 987654321), (1, 1, 1), (M.r - 1, 2, M.r - 2)):
        P1, P2, K = G.g1_mul(g1, a), G.g1_mul(g1, b), G.g2_mul(pk.gg, k)
        want = _gt_bytes(G.F.f12_mul(G.pairing(P1, K), G.pairing(G.g1_neg(P2), pk.gg)))
        for mode in (0, 2, 1):       # 0 / 2: check program for 16 / 32 lane pairs (variable + fixed pair), 1: tail program (F given, fixed pair; 32 lane pairs)
            r = L.twin_bn254_pair_coop(ctx, g1b(P1), g1b(P2), g2b(K), mode, out)
            assert out.raw == want and r == int(a * k % M.r == b % M.r)


def test_verdicts_on_ps_signatures(env):
    """e(sig1, K) == e(sig2, gg) for a credential issued by the model's signer (K = XX prod YY_i^{m_i}), and not for a tampered one."""
    L, ctx, pk0 = env
    seed, A = 99, 3
    g = M.hash_to_g1("abc")
    x, ys = scalar_stream(seed, 0, M.r), [scalar_stream(seed, 1 + i, M.r) for i in range(A)]
    pk, skX = PR.key_gen(g, pk0.gg, x, ys)
    h = L.twin_bn254_ctx_new(A, 4, g1_bases(M, pk, svc=b"svc", skX=skX), g2_bases(M, pk))
    ctx2 = ctypes.c_void_p(h)
    ms = [M.fr_hash(b"attr-%d" % i) for i in range(A)]
    u = scalar_stream(seed, 50, M.r)
    sig1 = G.g1_mul(g, u)
    sig2 = G.g1_mul(g, u * (x + sum(y * m for y, m in zip(ys, ms))) % M.r)
    K = pk.XX
    for i in range(A):
        K = G.g2_add(K, G.g2_mul(pk.YYi[i], ms[i]))
    out = ctypes.create_string_buffer(384)
    one = fb(1) + bytes(352)
    for mode in (0, 2):
        assert L.twin_bn254_pair_coop(ctx2, g1b(sig1), g1b(sig2), g2b(K), mode, out) == 1 and out.raw == one
        assert L.twin_bn254_pair_coop(ctx2, g1b(sig1), g1b(G.g1_add(sig2, g)), g2b(K), mode, out) == 0
        assert L.twin_bn254_pair_coop(ctx2, g1b(G.g1_mul(sig1, 2)), g1b(sig2), g2b(K), mode, out) == 0


def test_committed_program_header_is_what_the_generator_emits(tmp_path):
    """csrc/elp/coop_prog_bn254.h (3.4 MB of generated tables) is committed: re-running tools/gen_coop.py must reproduce it byte for byte, so the header cannot
    drift from its generator (and the generator's own big-integer validation of the program has run on exactly these bytes)."""
    import os
    import subprocess
    import sys
    from elp_testlib import ROOT
    out = tmp_path / "coop_prog_bn254.h"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_coop.py"), "bn254", str(out)])
    committed = os.path.join(ROOT, "ps-signature-and-el-passo_amd", "csrc", "elp", "coop_prog_bn254.h")
    assert out.read_bytes() == open(committed, "rb").read()


def test_bls12_381_programs_on_the_host_twin():
    """The BLS12-381 programs (tools/gen_coop.py: M-type line placement, Hayashida-Hayasaka-Teruya chain taken to the third power) interpreted by the host twin: the
    value equals the CUBE of the model's  e(sig1, K) e(-sig2, gg)  bit for bit for all three programs, and is 1 exactly for a valid PS signature.
    (The model is pinned to the reference's wasm run on this curve: tests/test_oracle_bls_golden.py.)"""
    from elp_testlib import BLS12_381, BLS_G2
    Mb = Mcl(BLS12_381)
    Gb, PRb = Mb.G, Protocol(Mb)
    L = twin()
    L.twin_bls_ctx_new.restype = ctypes.c_void_p
    seed, A, N = 77, 3, 48
    g = Mb.hash_to_g1("abc")
    pk, skX = PRb.key_gen(g, BLS_G2, scalar_stream(seed, 0, Mb.r), [scalar_stream(seed, 1 + i, Mb.r) for i in range(A)])
    ctx = ctypes.c_void_p(L.twin_bls_ctx_new(A, 4, g1_bases(Mb, pk, svc=b"svc"), g2_bases(Mb, pk)))
    assert ctx.value
    m = [scalar_stream(seed, 10 + i, Mb.r) for i in range(A)]
    u = scalar_stream(seed, 20, Mb.r)
    full = (scalar_stream(seed, 0, Mb.r) + sum(scalar_stream(seed, 1 + i, Mb.r) * m[i] for i in range(A))) % Mb.r
    sig1, sig2 = Gb.g1_mul(g, u), Gb.g1_mul(g, u * full % Mb.r)
    K = pk.XX
    for i in range(A):
        K = Gb.g2_add(K, Gb.g2_mul(pk.YYi[i], m[i]))
    gt = ctypes.create_string_buffer(12 * N)
    for s2, valid in ((sig2, True), (Gb.g1_add(sig2, g), False)):
        e = Mb.G.F.f12_mul(Gb.pairing(sig1, K), Gb.pairing(Gb.g1_neg(s2), pk.gg))
        e3 = Mb.G.F.f12_mul(Mb.G.F.f12_mul(e, e), e)
        want = b"".join(fb(e3[k][0], N) + fb(e3[k][1], N) for k in [0, 2, 4, 1, 3, 5])
        for mode in (0, 2, 1):
            rc = L.twin_bls_pair_coop(ctx, g1b(sig1, N), g1b(s2, N), g2b(K, N), mode, gt)
            assert rc == (1 if valid else 0), (mode, valid)
            assert gt.raw == want, mode
    L.twin_bls_ctx_free(ctx)


def test_committed_bls12_381_program_header_is_what_the_generator_emits(tmp_path):
    import os
    import subprocess
    import sys
    from elp_testlib import ROOT
    out = tmp_path / "coop_prog_bls12_381.h"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_coop.py"), "bls12_381", str(out)])
    assert out.read_bytes() == open(os.path.join(ROOT, "ps-signature-and-el-passo_amd", "csrc", "elp", "coop_prog_bls12_381.h"), "rb").read()


def test_register_allocation_keeps_the_lds_banks_apart():
    """tools/coop_bank_model.py on the BN254 programs: the bank-aware allocation of tools/gen_coop.py (round 5) leaves under 5 % of the modelled LDS cycles as bank
    conflicts in every program (33.9 % with first-free allocation; the model matched the hardware counter to 1.5 points, profiles/r05_coop_lds_banks.md)."""
    import os
    import subprocess
    import sys
    from elp_testlib import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "coop_bank_model.py"), "bn254"], check=True, capture_output=True, text=True).stdout
    rows = [ln for ln in out.splitlines() if "interleaved" in ln]
    assert len(rows) == 3
    for ln in rows:
        share = float(ln.split("conflicts")[1].split("%")[0])
        assert share < 5.0, ln
