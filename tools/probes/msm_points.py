"""Writes n distinct BN254 G1 points (std affine, 64 bytes each: P_i = s_i g with random 64-bit s_i, made by the library on the GPU) for tools/ubench_msm.hip.
python tools/probes/msm_points.py <n> <file>"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
n, path = int(sys.argv[1]), sys.argv[2]
ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 3, seed=1, window_bits=8)
rng = np.random.default_rng(7)
with open(path, "wb") as fh:
    for lo in range(0, n, 1 << 16):
        m = min(1 << 16, n - lo)
        ks = np.zeros((m, 32), dtype=np.uint8)
        ks[:, :8] = rng.integers(1, 256, size=(m, 8), dtype=np.uint8)
        fh.write(ctx.g1_mul(bytes(wl.g) * m, ks.tobytes()))
ctx.close()
