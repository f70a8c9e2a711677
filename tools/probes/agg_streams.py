"""Aggregated batches round-robin over S streams of one context (the pathology behind bench.py's sustained_*_three_streams line: 18 ms per batch on 1-2 streams, 80+ ms on 3-4).
Usage: python tools/probes/agg_streams.py [window] [reps] [stream counts, e.g. 1,2,3]        (run under rocprofv3 --kernel-trace for the per-kernel timeline)"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
dev = torch.device("cuda", 0)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 12
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
SS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3, 4]
print("env", {k: v for k, v in os.environ.items() if k.startswith(("HSA_SCRATCH", "HSA_ENABLE_SCRATCH", "GPU_MAX", "ELP_"))}, flush=True)
ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
B = 65536
recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
fls = [torch.zeros(B, dtype=torch.uint8, device=dev) for _ in range(4)]
cnt = torch.zeros(4, dtype=torch.int64, device=dev)
seed = np.frombuffer(bytes(range(32)), dtype=np.uint8).copy()
n = B
if os.environ.get("BIG_FIRST"):
    # what bench.py's process did before its multi-stream phase: ONE call over 2^20 items on the default stream (16 384 workgroups: the runtime provisions that
    # queue's scratch for every wave slot of the chip, several GB, and keeps it)
    tiles = int(os.environ["BIG_FIRST"])
    big = d_rec.repeat(tiles)
    flb = torch.zeros(B * tiles, dtype=torch.uint8, device=dev)
    t0 = time.perf_counter()
    for _ in range(2):
        ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, torch.cuda.current_stream().cuda_stream, B * tiles, big.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), flb.data_ptr(),
                                                 cnt.data_ptr()))
    torch.cuda.synchronize()
    print("big call first: %d items on the default stream, %.1f ms per call" % (B * tiles, (time.perf_counter() - t0) * 500), flush=True)
for S in SS:
    def call(k):
        ctx._chk(ctx.lib.elp_verify_id_batch_aggregated_dev(ctx.h, streams[k].cuda_stream, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), seed.ctypes.data,
                                                            fls[k].data_ptr(), cnt.data_ptr() + 8 * k))
    for k in range(S):
        call(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(REPS):
        t1 = time.perf_counter()
        call(b % S)
        if os.environ.get("AGG_PRINT_HOST"):
            print("  host side of call %d: %.3f ms" % (b, (time.perf_counter() - t1) * 1e3))
    th = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / REPS * 1e3
    ok = all(bool((fls[k][:n].cpu().numpy() == expect[:n]).all()) for k in range(S))
    print("aggregated n=%6d streams=%d  %.2f ms per batch (host enqueue of all %d calls: %.1f ms)  ok=%s" % (n, S, dt, REPS, th, ok), flush=True)
