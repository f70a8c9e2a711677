"""Edge cases pinned by the reference's wasm (tests/golden/bn254_oracle_edge.json, generator oracle/gen_edge_fixtures.py):
a commitment k OUTSIDE the order-r subgroup of the twist (src/ps-verifier.cc:73 multiplies an attacker-supplied k; mcl does no
order check there) and the 3-byte length form inside T-L-V lists (src/ps-encoding.cc:149-162).  The reference rejects every
off-subgroup k -- also a proof crafted so that its Schnorr half passes under plain scalar multiplication -- and accepts the
alternative length encodings.  Both oracle restatements and the host twin of the HIP formulas (GLS multiplication, which is only
valid inside the subgroup) must give the same verdicts; the GPU leg is tests/test_gpu_round2.py."""
import base64
import ctypes

from elp_testlib import BN254, Codec, Mcl, Protocol, g1_bases, g2_bases, hidden_mask, load_golden, oracle, oracle_key, pack_verify_id, twin

M = Mcl(BN254)
CD, PR = Codec(M), Protocol(M)
EDGE = load_golden("bn254_oracle_edge.json")


def test_fixture_shape():
    labels = {c["label"] for c in EDGE["cases"]}
    assert {"original", "k_plus_T13", "k_plus_Tbig", "k_random_twist", "crafted_c_mod_13", "frlist_fd_len", "strlist_fd_len"} <= labels
    for c in EDGE["cases"]:
        want = c["label"] in ("original", "frlist_fd_len", "strlist_fd_len")
        assert c["expect"] is want, c["label"]     # what the reference's wasm answered
    T = M.g2_de(base64.b64decode(EDGE["T13"]))
    assert M.G.g2_on_curve(T)


def test_model_and_c_oracle_reproduce_edge_verdicts():
    L = oracle()
    keys = {}
    for c in EDGE["cases"]:
        pk = CD.pk_decode(base64.b64decode(c["pk"]))
        P = CD.proof_decode(base64.b64decode(c["proof"]))
        full = c["scenario"] == "A3H2" or c["label"] == "crafted_c_mod_13"
        got = PR.verify_id_noretr(pk, P, c["ad"], c["svc"], pairing=full)
        if full or c["expect"]:
            assert got == c["expect"], (c["scenario"], c["label"])
        kk = (c["pk"], c["svc"])
        if kk not in keys:
            keys[kk] = oracle_key(M, pk, svc=c["svc"].encode())
        ad = c["ad"].encode()
        assert bool(L.elpo_verify_id(keys[kk], pack_verify_id(M, P), hidden_mask(P.attributes), 0, ad, len(ad))) == c["expect"], c["label"]


def test_host_twin_reproduces_edge_verdicts():
    """The HIP formulas (host build): record path and wire path (T-L-V parse + decompression in the kernel code)."""
    L = twin()
    ctxs = {}
    for c in EDGE["cases"]:
        pk = CD.pk_decode(base64.b64decode(c["pk"]))
        kk = (c["pk"], c["svc"])
        if kk not in ctxs:
            h = L.twin_bn254_ctx_new(len(pk.Yi), 4, g1_bases(M, pk, svc=c["svc"].encode()), g2_bases(M, pk))
            assert h
            ctxs[kk] = ctypes.c_void_p(h)
        raw = base64.b64decode(c["proof"])
        P = CD.proof_decode(raw)
        ad = c["ad"].encode()
        got = L.twin_bn254_verify_id(ctxs[kk], pack_verify_id(M, P), ctypes.c_uint64(hidden_mask(P.attributes)), 0, ad, len(ad))
        assert bool(got) == c["expect"], ("record", c["scenario"], c["label"])
        got = L.twin_bn254_verify_id_split(ctxs[kk], pack_verify_id(M, P), ctypes.c_uint64(hidden_mask(P.attributes)), 0, ad, len(ad))
        assert bool(got) == c["expect"], ("two-phase record", c["scenario"], c["label"])
        got = L.twin_bn254_verify_id_jobs4(ctxs[kk], pack_verify_id(M, P), ctypes.c_uint64(hidden_mask(P.attributes)), 0, ad, len(ad))
        assert got == int(c["expect"]), ("four-job / five-job record", got, c["scenario"], c["label"])
        got = L.twin_bn254_verify_id_wire(ctxs[kk], raw, len(raw), 0, ad, len(ad))
        assert bool(got) == c["expect"], ("wire", c["scenario"], c["label"])
        # the two-lanes-per-item layout (GLS on a lane pair, decompression split between the lanes)
        got = L.twin_bn254p_verify_id(ctxs[kk], pack_verify_id(M, P), ctypes.c_uint64(hidden_mask(P.attributes)), 0, ad, len(ad))
        assert bool(got) == c["expect"], ("paired record", c["scenario"], c["label"])
        got = L.twin_bn254p_verify_id_wire(ctxs[kk], raw, len(raw), 0, ad, len(ad))
        assert bool(got) == c["expect"], ("paired wire", c["scenario"], c["label"])
