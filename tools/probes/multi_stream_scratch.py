"""Root cause of the multi-stream stall: kernels with a large private-memory (scratch) frame launched round-robin on S streams of one
process.  k_verify_id_staged needs 14 976 B x 65 536 lanes = 0.98 GB of scratch per hardware queue; bench.py's aggregated batches measured 18 ms per batch on one or two
streams and 80-85 ms on three or four (profiles/r04_scratch_streams.log).  This probe sweeps S = 1..4 for the per-item kernel at several batch sizes (scratch per queue
scales with the grid) under whatever HSA_* environment it is started with.
Usage: python tools/probes/multi_stream_scratch.py [window] [batches-per-stream-count]"""
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
print("env", {k: v for k, v in os.environ.items() if k.startswith(("HSA_SCRATCH", "HSA_ENABLE_SCRATCH", "GPU_MAX"))}, flush=True)
ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
B = 65536
recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
fls = [torch.zeros(B, dtype=torch.uint8, device=dev) for _ in range(4)]
cnt = torch.zeros(4, dtype=torch.int64, device=dev)
for n in (65536, 16384, 4096):
    for S in (1, 2, 3, 4):
        def call(k):
            ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, streams[k].cuda_stream, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), fls[k].data_ptr(),
                                                     cnt.data_ptr() + 8 * k))
        for k in range(S):
            call(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in range(REPS):
            call(b % S)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / REPS * 1e3
        ok = all(bool((fls[k][:n].cpu().numpy() == expect[:n]).all()) for k in range(S))
        print("n=%6d streams=%d  %.2f ms per batch  ok=%s" % (n, S, dt, ok), flush=True)
