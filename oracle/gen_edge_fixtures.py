#!/usr/bin/env python3
"""Edge-case golden vectors (test infrastructure; runs ONLY in the build container, needs node + /root/reference/wasm-build).

Builds inputs the reference's own prover never emits -- a commitment k outside the order-r subgroup of the twist, alternative
(3-byte) length encodings inside the T-L-V message -- with the big-int model, asks the reference's prebuilt wasm (protocol
layer + mcl) for its verdict on each (oracle/wasm_verify.js), and records inputs + verdicts in
tests/golden/<curve>_oracle_edge.json.  Only data is stored (base64 messages, strings, booleans).

Usage: python oracle/gen_edge_fixtures.py [--curve bn254|bls12_381]      (bls12_381: the reference wasm with mcl's BLS12-381 CurveParam,
oracle/wasm_curve.js; needs tests/golden/bls12_381_oracle_flows.json from `node oracle/gen_fixtures.js --curve bls12_381` first)

Cases per base proof (all el_passo_verify_id_without_id_retrieval, src/ps-verifier.cc:140-212):
  k_plus_T13        k' = k + T, T a point of order 13 on the twist (13 | #E'(Fp2)/r): on the curve, outside the subgroup
  k_plus_Tbig       k' = k + T, T = r * (random twist point): order = cofactor part, outside the subgroup
  k_random_twist    k' = a random twist point (not multiplied by the cofactor)
  crafted_c_mod_13  a prover who knows the credential commits with k' = k + T13 and grinds its nonce until 13 | c, so that a
                    verifier computing a plain [c]k' recomputes the committed V_k (the Schnorr half then passes under plain
                    multiplication; the pairing half sees K + T13)
  frlist_fd_len     the FrList entries carry the 3-byte length form FD 00 20 (parseVar accepts it, src/ps-encoding.cc:149-162)
  strlist_fd_len    the revealed attribute strings carry FD 00 len
BLS12-381 only (E(Fp) has the cofactor (z-1)^2/3 = 3 * 11^2 * ...; mcl's default does not test the order of a deserialised G1 point):
  phi_plus_T3 / phi_plus_T11 / phi_plus_Tbig   the pseudonym phi moved out of G1 by a point of order 3 / 11 / of the full cofactor part
  phi_random_curve  phi = a random point of E(Fp)
  sig1_plus_T3 / sig2_plus_T3 / sig1_plus_Tbig  the randomised signature moved out of G1 (a small-order component pairs to 1 with everything)
  sig_T3_O          (sig1, sig2) = (T3, O): "sig1 is not the point at infinity" alone admits it for any K
  crafted_phi_c_mod_3  a prover who knows the credential publishes phi' = phi + T3 and grinds until 3 | c (under plain multiplication by c
                    the verifier recomputes the committed V_phi)
"""
import base64
import json
import os
import random
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle.pymodel import BLS12_381, BN254, Codec, Credential, IdProof, Mcl, Protocol  # noqa: E402

CURVE = "bn254"
if "--curve" in sys.argv:
    CURVE = sys.argv[sys.argv.index("--curve") + 1].lower()
assert CURVE in ("bn254", "bls12_381")
IS_BLS = CURVE == "bls12_381"
M = Mcl(BLS12_381 if IS_BLS else BN254)
G, F = M.G, M.F
CD, PR = Codec(M), Protocol(M)
PR.subgroup_check = False          # the constructions below leave G1 on purpose; the verdicts come from the wasm
GOLD = os.path.join(ROOT, "tests", "golden")
FBY = M.fb


def twist_point(rnd):
    while True:
        x = (rnd.randrange(M.p), rnd.randrange(M.p))
        y = F.f2_sqrt(F.f2_add(F.f2_mul(F.f2_sqr(x), x), F.b2))
        if y is not None:
            return (x, y)


def curve_point(rnd):
    while True:
        x = rnd.randrange(M.p)
        y = F.sqrt((x * x * x + M.cv.b) % M.p)
        if y is not None:
            return (x, y)


def g2_mul_int(P, k):
    """plain double-and-add by an arbitrary integer (pymodel's g2_mul reduces mod r)"""
    R = None
    for bit in bin(k)[2:] if k else "":
        R = G.g2_add(R, R)
        if bit == "1":
            R = G.g2_add(R, P)
    return R


def reencode_fd(msg, frlist, strlist):
    """Re-emit an IdProof message with 3-byte (FD hi lo) lengths inside the FrList and / or on the non-empty strings."""
    b = bytes(msg)
    out = bytearray()
    o = 0
    for _ in range(2):          # sig1, sig2
        out += b[o:o + FBY + 2]; o += FBY + 2
    out += b[o:o + 2 * FBY + 2]; o += 2 * FBY + 2   # k
    out += b[o:o + FBY + 2]; o += FBY + 2   # phi
    out += b[o:o + 34]; o += 34   # c
    assert b[o] == 6
    n = b[o + 1]
    out += b[o:o + 2]; o += 2
    for _ in range(n):
        assert b[o] == 32
        out += (bytes([253, 0, 32]) if frlist else bytes([32])) + b[o + 1:o + 33]
        o += 33
    assert b[o] == 7
    na = b[o + 1]
    out += b[o:o + 2]; o += 2
    for _ in range(na):
        l = b[o]
        assert l < 253
        out += (bytes([253, 0, l]) if (strlist and l) else bytes([l])) + b[o + 1:o + 1 + l]
        o += 1 + l
    out += b[o:]
    return bytes(out)


def main():
    rnd = random.Random(20212)
    flows = json.load(open(os.path.join(GOLD, CURVE + "_oracle_flows.json")))
    if IS_BLS:
        z = M.cv.z                           # #E'(Fp2) = r * h2, h2 = 13^2 * 23^2 * 2713 * ...
        h2 = (z**8 - 4 * z**7 + 5 * z**6 - 4 * z**4 + 6 * z**3 - 4 * z**2 - 4 * z + 13) // 9
        assert g2_mul_int(twist_point(random.Random(3)), M.r * h2) is None
    else:
        h2 = 2 * M.p - M.r                   # #E'(Fp2) = r * h2, h2 = 13 * 96757 * (233-bit)
    assert h2 % 13 == 0
    # (on BLS12-381 13^2 | h2 and the 13-part of E'(Fp2) is not cyclic: clear everything but the whole 13-part, then descend to order 13)
    e13 = M.r * h2
    while e13 % 13 == 0:
        e13 //= 13
    T13 = None
    while T13 is None:
        T13 = g2_mul_int(twist_point(rnd), e13)
        while T13 is not None and g2_mul_int(T13, 13) is not None:
            T13 = g2_mul_int(T13, 13)
    assert g2_mul_int(T13, 13) is None and G.g2_on_curve(T13)
    Tbig = g2_mul_int(twist_point(rnd), M.r)
    assert Tbig is not None and G.g2_on_curve(Tbig)
    if IS_BLS:
        h1 = M.g1_cofactor                   # #E(Fp) = r * h1, h1 = 3 * 11^2 * 10177^2 * 859267^2 * 52437899^2
        assert h1 % 3 == 0 and h1 % 11 == 0
        S3 = (0, M.p - 2)                    # the points of order 3 on y^2 = x^3 + 4 are (0, +-2); (0, 2) itself serialises to the all-zero string = infinity
        assert G.g1_on_curve(S3) and G.g1_mul_plain(S3, 3) is None
        S11 = None                           # E(Fp) = Z_(z-1)/3 x Z_r(z-1): its 11-part is Z_11 x Z_11
        while S11 is None:
            S11 = G.g1_mul_plain(curve_point(rnd), (M.r * h1) // 121)
        assert G.g1_mul_plain(S11, 11) is None
        Sbig = G.g1_mul_plain(curve_point(rnd), M.r)
        assert Sbig is not None and G.g1_mul_plain(Sbig, h1) is None
    cases = []
    for sc in flows["scenarios"][:2]:          # A3H2, A8H4
        pk = CD.pk_decode(base64.b64decode(sc["pk"]))
        for p in sc["proofs"][:2]:
            base = next(c for c in p["cases"] if c["label"] == "original")
            raw = base64.b64decode(base["proof"])
            P = CD.proof_decode(raw)

            def emit(label, msg):
                cases.append({"scenario": sc["name"], "label": label, "pk": sc["pk"], "proof": base64.b64encode(msg).decode(),
                              "ad": base["ad"], "svc": base["svc"]})

            emit("original", raw)
            for label, T in (("k_plus_T13", T13), ("k_plus_Tbig", Tbig)):
                Q = IdProof(P.sig1, P.sig2, G.g2_add(P.k, T), P.phi, P.c, list(P.rs), list(P.attributes))
                emit(label, CD.proof_encode(Q))
            Q = IdProof(P.sig1, P.sig2, twist_point(rnd), P.phi, P.c, list(P.rs), list(P.attributes))
            emit("k_random_twist", CD.proof_encode(Q))
            emit("frlist_fd_len", reencode_fd(raw, True, False))
            emit("strlist_fd_len", reencode_fd(raw, False, True))
            if IS_BLS:
                for label, T in (("phi_plus_T3", S3), ("phi_plus_T11", S11), ("phi_plus_Tbig", Sbig)):
                    emit(label, CD.proof_encode(IdProof(P.sig1, P.sig2, P.k, G.g1_add(P.phi, T), P.c, list(P.rs), list(P.attributes))))
                emit("phi_random_curve", CD.proof_encode(IdProof(P.sig1, P.sig2, P.k, curve_point(rnd), P.c, list(P.rs), list(P.attributes))))
                emit("sig1_plus_T3", CD.proof_encode(IdProof(G.g1_add(P.sig1, S3), P.sig2, P.k, P.phi, P.c, list(P.rs), list(P.attributes))))
                emit("sig2_plus_T3", CD.proof_encode(IdProof(P.sig1, G.g1_add(P.sig2, S3), P.k, P.phi, P.c, list(P.rs), list(P.attributes))))
                emit("sig1_plus_Tbig", CD.proof_encode(IdProof(G.g1_add(P.sig1, Sbig), P.sig2, P.k, P.phi, P.c, list(P.rs), list(P.attributes))))
                emit("sig_T3_O", CD.proof_encode(IdProof(S3, None, P.k, P.phi, P.c, list(P.rs), list(P.attributes))))
        # crafted proof from the scenario's unblinded credential: commitments made honestly, k' = k + T13 in the transcript,
        # nonce ground until 13 | c
        cred = CD.cred_decode(base64.b64decode(sc["requests"][-1]["unblinded"]))
        attrs = [(v.encode(), i < sc["H"]) for i, v in enumerate(sc["attr_values"])]
        svc, ad = "svc", sc["ad"]
        assert PR.ps_verify(pk, cred, [a for a, _ in attrs])
        t, rr = rnd.randrange(M.r), rnd.randrange(M.r)
        sig1 = G.g1_mul(cred.sig1, rr)
        sig2 = G.g1_mul(G.g1_add(G.g1_mul(cred.sig1, t), cred.sig2), rr)
        Hs = M.hash_to_g1(svc.encode())
        hs = [M.fr_hash(a) for a, hide in attrs if hide]
        hid = [i for i, (_, hide) in enumerate(attrs) if hide]
        phi = G.g1_mul(Hs, hs[0])
        k = pk.XX
        for i, m_ in zip(hid, hs):
            k = G.g2_add(k, G.g2_mul(pk.YYi[i], m_))
        k = G.g2_add(k, G.g2_mul(pk.gg, t))
        kp = G.g2_add(k, T13)
        rhos = [rnd.randrange(M.r) for _ in hid]
        Vk0 = pk.XX
        for i, rho in zip(hid, rhos):
            Vk0 = G.g2_add(Vk0, G.g2_mul(pk.YYi[i], rho))
        Vphi = G.g1_mul(Hs, rhos[0])
        rho_t = rnd.randrange(M.r)
        Vk = G.g2_add(Vk0, G.g2_mul(pk.gg, rho_t))
        while True:
            c = M.challenge([M.g2_hex(kp), M.g1_hex(phi), M.g2_hex(Vk), M.g1_hex(Vphi)], ad)
            if c % 13 == 0:
                break
            rho_t = (rho_t + 1) % M.r
            Vk = G.g2_add(Vk, pk.gg)
        rs = [(rho - m_ * c) % M.r for rho, m_ in zip(rhos, hs)] + [(rho_t - t * c) % M.r]
        Q = IdProof(sig1, sig2, kp, phi, c, rs, [b"" if hide else a for a, hide in attrs])
        # sanity: with the honest k the same construction is a valid proof under the model; with k' the Schnorr half passes under
        # plain multiplication by c (the model's g2_mul) because [c]T13 = O
        assert PR.verify_id_noretr(pk, Q, ad, svc, pairing=False)
        cases.append({"scenario": sc["name"], "label": "crafted_c_mod_13", "pk": sc["pk"], "proof": base64.b64encode(CD.proof_encode(Q)).decode(),
                      "ad": ad, "svc": svc})
        if IS_BLS:
            # the same prover publishes phi' = phi + T3 (honest k) and grinds until 3 | c -- several times: whether the reference accepts such a proof
            # depends on how mcl's G1::mul splits c (tests/test_oracle_bls.py pins the split by these verdicts)
            phip = G.g1_add(phi, S3)
            for rep in range(4):
                rho_t = rnd.randrange(M.r)
                Vk = G.g2_add(Vk0, G.g2_mul(pk.gg, rho_t))
                while True:
                    c = M.challenge([M.g2_hex(k), M.g1_hex(phip), M.g2_hex(Vk), M.g1_hex(Vphi)], ad)
                    if c % 3 == 0:
                        break
                    rho_t = (rho_t + 1) % M.r
                    Vk = G.g2_add(Vk, pk.gg)
                rs = [(rho - m_ * c) % M.r for rho, m_ in zip(rhos, hs)] + [(rho_t - t * c) % M.r]
                Q = IdProof(sig1, sig2, k, phip, c, rs, [b"" if hide else a for a, hide in attrs])
                cases.append({"scenario": sc["name"], "label": "crafted_phi_c_mod_3", "pk": sc["pk"],
                              "proof": base64.b64encode(CD.proof_encode(Q)).decode(), "ad": ad, "svc": svc})
            # a proof made end to end by the model's el_passo_prove_id_without_id_retrieval (src/ps-requester.cc:312-432): the reference must accept it
            H = sum(1 for _, hide in attrs if hide)
            Q = PR.prove_id(pk, cred, attrs, ad.encode(), svc.encode(), None, None, None, [rnd.randrange(M.r) for _ in range(3 + H)], with_retrieval=False)
            cases.append({"scenario": sc["name"], "label": "model_made_proof", "pk": sc["pk"],
                          "proof": base64.b64encode(CD.proof_encode(Q)).decode(), "ad": ad, "svc": svc})
    res = subprocess.run(["node", os.path.join(ROOT, "oracle", "wasm_verify.js"), "--curve", CURVE], input=json.dumps(cases), capture_output=True,
                         text=True, check=True)
    verdicts = json.loads(res.stdout.strip().splitlines()[-1])
    assert len(verdicts) == len(cases)
    for c, v in zip(cases, verdicts):
        c["expect"] = v
        print("%-8s %-18s %s" % (c["scenario"], c["label"], v))
    doc = {"curve": flows["curve"], "generator": "oracle/gen_edge_fixtures.py --curve %s + oracle/wasm_verify.js" % CURVE,
           "T13": base64.b64encode(M.g2_ser(T13)).decode(), "cases": cases}
    if not IS_BLS:
        doc["generator"] = "oracle/gen_edge_fixtures.py + oracle/wasm_verify.js"
    else:
        doc["S3"], doc["S11"] = base64.b64encode(M.g1_ser(S3)).decode(), base64.b64encode(M.g1_ser(S11)).decode()
    json.dump(doc, open(os.path.join(GOLD, CURVE + "_oracle_edge.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
