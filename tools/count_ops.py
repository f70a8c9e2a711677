#!/usr/bin/env python3
"""Count Montgomery products per item of the fused kernels (BN254 headline configuration and friends).

Builds the host twin of the device headers (tests/host_twin/twin.cpp, the same template code the GPU runs) with
-DELP_COUNT_OPS, pushes one synthetic item of each workload through it and records fp_mul / fp_sqr calls.  The fixed-base
tables cannot be built at W=16 on the host in reasonable time, so the count is taken at W=8 and W=4 and extrapolated
linearly in the number of table windows (the only W-dependent term: one mixed addition per window per scalar).
Output: profiles/op_counts.json, read by bench.py to state the integer-VALU roofline.  Build/measurement tooling only.
"""
import ctypes
import importlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from elp_testlib import OracleBackedCtx  # noqa: E402

synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")


def build():
    so = os.path.join(tempfile.mkdtemp(), "libtwin_count.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-DELP_COUNT_OPS", "-I",
                           os.path.join(ROOT, "ps-signature-and-el-passo_amd", "csrc"), "-o", so,
                           os.path.join(ROOT, "tests", "host_twin", "twin.cpp")])
    L = ctypes.CDLL(so)
    L.twin_bn254_ctx_new.restype = ctypes.c_void_p
    return L


def counts(L):
    c = (ctypes.c_ulonglong * 4)()
    L.twin_op_counts(c, 1)
    return int(c[0]), int(c[1]), int(c[2]), int(c[3])


def main():
    L = build()
    A, H = 8, 4
    ob = OracleBackedCtx()
    wl = synth.Workload(ob, A)
    recs, mask, expect = wl.verify_id_batch(1, H, with_retrieval=True, corrupt_every=0)
    precs, pmask = wl.prove_id_batch(1, H, with_retrieval=True)
    res = {}
    b1 = b"".join(ob.b1.get(i, bytes(64)) for i in range(A + 6))
    b2 = b"".join(ob.b2[i] for i in range(A + 2))
    per_w = {}
    for W in (8, 4):
        ctx = ctypes.c_void_p(L.twin_bn254_ctx_new(A, W, b1, b2))
        counts(L)
        assert L.twin_bn254_verify_id(ctx, recs, ctypes.c_uint64(mask), 1, b"hello", 5) == 1
        v = counts(L)
        out = ctypes.create_string_buffer(4096)
        assert L.twin_bn254_prove_id(ctx, precs, ctypes.c_uint64(pmask), 1, b"hello", 5, out) == 1
        p = counts(L)
        per_w[W] = (v, p)
        L.twin_bn254_ctx_free(ctx)

    def extrapolate(i, W):
        # windows: ceil(256 / W); linear through the two measured points (32 and 64 windows)
        n8, n4 = 32, 64
        out = []
        for j in range(4):
            y8, y4 = per_w[8][i][j], per_w[4][i][j]
            slope = (y4 - y8) / (n4 - n8)
            out.append(y8 + slope * ((256 + W - 1) // W - n8))
        return out

    for name, i in (("verify_id", 0), ("prove_id", 1)):
        def entry(m, s, pr, qd, extrapolated=False):
            # multiply-adds (v_mad_i64_i32) of the 9-limb routines: product 81 + reduction 81; square 45 + 81; pair 2 x 81 + 81; quad 4 x 81 + 81
            e = {"fp_mul": round(m), "fp_sqr": round(s), "fp_mul_pair": round(pr), "fp_mul_quad": round(qd),
                 "multiply_adds": round(162 * m + 126 * s + 243 * pr + 405 * qd), "fp_mul_equivalents": round(m + 126 / 162 * s + 1.5 * pr + 2.5 * qd)}
            if extrapolated:
                e["extrapolated"] = True
            return e
        res[name] = {"config": "BN254, 8 attributes, 4 hidden, id-retrieval", "W8": entry(*per_w[8][i]), "W4": entry(*per_w[4][i])}
        for W in (12, 16):
            res[name]["W%d" % W] = entry(*extrapolate(i, W), extrapolated=True)
    res["note"] = ("Calls per item of the three Montgomery routines, counted on the host twin (same template code as the kernels), and the "
                   "multiply-add instructions they stand for; fp_mul_equivalents = multiply_adds / 162.")
    path = os.path.join(ROOT, "profiles", "op_counts.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
