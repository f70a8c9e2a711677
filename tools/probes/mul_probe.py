"""Variable-base multiplications alone (k_mul: GLV on G1, GLS on G2, per-lane tables of 8 affine multiples) at a light and at the full load; run under
rocprofv3 --kernel-trace and read the k_mul durations from the trace."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 3, seed=20211, window_bits=8)           # gives the generators
rnd = np.random.default_rng(1)
ks = rnd.integers(0, 256, size=(65536, 32), dtype=np.uint8)
ks[:, 31] &= 0x1f
g1 = ctx.hash_to_g1([b"abc"])
b1 = ctx.g1_mul(g1 * 4, ks[:4].tobytes())
b2 = ctx.g2_mul(wl.gg * 4, ks[:4].tobytes())
same = np.repeat(ks[:1], 65536, axis=0)                             # every lane the same scalar: table reads with a wave-uniform index
for tag, kk in (("distinct scalars", ks), ("one scalar", same)):
    for n in (2048, 16384, 65536):
        ctx.g1_mul((b1 * (n // 4))[: n * ctx.G1], kk[:n].tobytes())
        ctx.g2_mul((b2 * (n // 4))[: n * ctx.G2], kk[:n].tobytes())
        print("%s n=%d ok" % (tag, n), flush=True)
