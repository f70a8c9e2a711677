"""The reference-API path alone (bench.py's `host_api` object): python tools/probes/host_api_probe.py [window] [batch]"""
import importlib
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module(bench.PKG)
synth = importlib.import_module(bench.PKG + ".synth")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=8)
recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
print(json.dumps(bench.host_api(pkg, wl, recs, B, 8, 4, 0, expect, W, 0), indent=1))
ctx.close()
