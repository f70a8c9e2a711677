"""Scratch GPU probe: k_pairing kernel time (Miller loop + final exponentiation only) at a batch that allows 2 waves/SIMD."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
elp = importlib.import_module("ps-signature-and-el-passo_amd")
ctx = elp.Context()
n = int(os.environ.get("N", "131072"))
P = ctx.hash_to_g1([b"abc"])
from elp_testlib import g2b, load_golden, Codec, Mcl, BN254
import base64
M = Mcl(BN254)
pk = Codec(M).pk_decode(base64.b64decode(load_golden("bn254_oracle_flows.json")["scenarios"][0]["pk"]))
Q = g2b(pk.gg)
for rep in range(3):
    t0 = time.perf_counter()
    out = ctx.pairing(P * n, Q * n)
    dt = time.perf_counter() - t0
    print("n=%d pairing call %.1f ms (%.3f M/s incl. copies)" % (n, dt * 1e3, n / dt / 1e6), flush=True)
assert out[:384] == out[-384:]
