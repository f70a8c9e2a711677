#!/usr/bin/env python3
"""Credential requests made by the big-int model for the reference's IdP to judge (test infrastructure; build container only).

Called by oracle/gen_request_fixtures.js, which owns the IdP's secret key inside the reference's wasm module: reads
{"pk": base64, "attrs": [[value, hidden], ...], "ad": str, "seed": int} on stdin and prints a JSON list of
{"label", "request" (base64 of PSCredRequest::toBufferString, src/ps-encoding.cc:423-449), "t1" (hex; the blinding scalar the
requester keeps for unblind_credential, src/ps-requester.cc:101-112)}.

  model_request        PSRequester::el_passo_request_id as restated in oracle/pymodel.py (src/ps-requester.cc:19-99)
  A_flip_ysign         the same message with the y-flag of A flipped
  A_plus_T3            (BLS12-381) the commitment moved out of G1 by the point (0, 2) of order 3; transcript recomputed honestly over A'
  crafted_A_c_mod_3    (BLS12-381) A' = A + T3 with the nonce ground until 3 | c: passes under plain multiplication by c
  A_infinity           A = O with a NIZK made for it (t1 = 0, no hidden attribute contributes: only possible when H = 0; otherwise skipped)
"""
import base64
import json
import os
import random
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle.pymodel import BLS12_381, BN254, Codec, CredRequest, Mcl, Protocol  # noqa: E402


def main():
    curve = sys.argv[sys.argv.index("--curve") + 1].lower() if "--curve" in sys.argv else "bn254"
    M = Mcl(BLS12_381 if curve == "bls12_381" else BN254)
    G, CD, PR = M.G, Codec(M), Protocol(M)
    PR.subgroup_check = False
    q = json.load(sys.stdin)
    pk = CD.pk_decode(base64.b64decode(q["pk"]))
    attrs = [(v.encode(), bool(h)) for v, h in q["attrs"]]
    ad = q["ad"]
    rnd = random.Random(q["seed"])
    H = sum(1 for _, h in attrs if h)
    out = []

    def emit(label, rq, t1):
        out.append({"label": label, "request": base64.b64encode(CD.req_encode(rq)).decode(), "t1": "%x" % t1})

    for rep in range(2):
        rq, t1 = PR.request_id(pk, attrs, ad.encode(), [rnd.randrange(M.r) for _ in range(2 + H)])
        assert PR.nizk_verify_request(pk, rq, ad)
        emit("model_request", rq, t1)
    raw = bytearray(CD.req_encode(rq))
    raw[2 + M.fb - 1] ^= 0x80
    out.append({"label": "A_flip_ysign", "request": base64.b64encode(bytes(raw)).decode(), "t1": "%x" % t1})
    if curve == "bls12_381":
        S3 = (0, M.p - 2)          # (0, 2) itself would serialise to the all-zero string, i.e. infinity
        # honest nonces, commitment A' = A + T3, challenge over A'; then the nonce ground until 3 | c -- several times: whether the reference accepts
        # depends on how mcl's G1::mul splits c (tests/test_oracle_bls.py pins the split by these verdicts)
        hid = [(i, M.fr_hash(a)) for i, (a, h) in enumerate(attrs) if h]
        for rep in range(4):
            t1, rho0 = rnd.randrange(M.r), rnd.randrange(M.r)
            rhos = [rnd.randrange(M.r) for _ in hid]
            A = G.g1_mul(pk.g, t1)
            V = G.g1_mul(pk.g, rho0)
            for (i, m_), rho in zip(hid, rhos):
                A = G.g1_add(A, G.g1_mul(pk.Yi[i], m_))
                V = G.g1_add(V, G.g1_mul(pk.Yi[i], rho))
            Ap = G.g1_add(A, S3)
            first = rep == 0
            while True:
                c = M.challenge([M.g1_hex(Ap), M.g1_hex(V)], ad)
                if first:
                    rs = [(rho0 - t1 * c) % M.r] + [(rho - m_ * c) % M.r for (_, m_), rho in zip(hid, rhos)]
                    emit("A_plus_T3", CredRequest(Ap, c, rs, [b"" if h else a for a, h in attrs]), t1)
                    first = False
                if c % 3 == 0:
                    break
                rho0 = (rho0 + 1) % M.r
                V = G.g1_add(V, pk.g)
            rs = [(rho0 - t1 * c) % M.r] + [(rho - m_ * c) % M.r for (_, m_), rho in zip(hid, rhos)]
            emit("crafted_A_c_mod_3", CredRequest(Ap, c, rs, [b"" if h else a for a, h in attrs]), t1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
