"""Root-cause probe for the cross-stream stall of ELP_OPT_STREAM_OVERLAP: bench.py's process saw ~1.2 s per call of the two-stream small-batch
sequence, a process of its own never did.  What bench.py's process has and tools/probes/stall_probe.py lacked: a kernel with a LARGE private-memory (scratch)
frame had run on the launch stream's hardware queue before (k_verify_id_staged, 14 976 B per lane x 65 536 lanes = 0.98 GB).  Hypothesis: the second stream is
a second hardware queue with a scratch allocation of its own; once the first queue holds a large scratch block the runtime has to reclaim it (queue idle +
free + allocate) whenever the other queue's kernel needs scratch, and the two queues then take the block from each other on every call.
Usage: python tools/probes/stall_repro.py <mode>     mode: plain | big_first | big_first_serial
Environment knobs to try around it: HSA_SCRATCH_SINGLE_LIMIT, HSA_ENABLE_SCRATCH_ASYNC_RECLAIM, GPU_MAX_HW_QUEUES."""
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
stream = torch.cuda.current_stream().cuda_stream
mode = sys.argv[1] if len(sys.argv) > 1 else "big_first"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 12


def wall(fn, reps=3):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return out


ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
nbig = 65536
recs, mask, expect = wl.verify_id_batch(nbig, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(nbig, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)


def call(n):
    ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr()))


print("mode", mode, "W", W, "env", {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "GPU_MAX", "AMD_"))}, flush=True)
if mode.startswith("bench_like"):
    # what bench.py's process holds when it reaches the small batches: the headline context (still open: tables, launch workspace, the copy stream and pinned
    # staging of its host-buffer path) and a second context of the same table width that runs the small batches
    print("headline ctx: big kernel:", ["%.2f" % t for t in wall(lambda: call(nbig), 2)], flush=True)
    if "host" in mode:
        t0 = time.perf_counter()
        for _ in range(2):
            fl2, cnt2 = ctx.verify_id_batch(recs, mask, True, wl.ad)
        print("headline ctx: host-buffer path x2: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    if "agg" in mode:
        seed_buf = np.frombuffer(bytes((7 * i + 1) & 0xFF for i in range(32)), dtype=np.uint8).copy()
        fl_t = torch.zeros(nbig, dtype=torch.uint8, device=dev)
        print("headline ctx: aggregated:", ["%.2f" % t for t in wall(lambda: ctx._chk(ctx.lib.elp_verify_id_batch_aggregated_dev(ctx.h, stream, nbig, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), seed_buf.ctypes.data, fl_t.data_ptr(), d_cnt.data_ptr())), 2)], flush=True)
    ctx0, wl0 = ctx, wl
    ctx = pkg.Context(pkg.CURVE_BN254, 0)
    wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
    recs, mask, expect = wl.verify_id_batch(4096, 4, with_retrieval=True)
    d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
if mode.startswith("big_first"):
    print("big kernel (65536 items) on the launch stream:", ["%.2f" % t for t in wall(lambda: call(nbig), 2)], flush=True)
ctx.set_stream_overlap(0 if mode.endswith("serial") else 1)
for n in (4096, 64, 4096, 1024, 4096):
    print("overlap=%d n=%5d ms per call:" % (0 if mode.endswith("serial") else 1, n), ["%.2f" % t for t in wall(lambda: call(n), 4)], flush=True)
if mode.startswith("big_first"):
    print("big kernel again:", ["%.2f" % t for t in wall(lambda: call(nbig), 2)], flush=True)
    print("small again n=4096:", ["%.2f" % t for t in wall(lambda: call(4096), 3)], flush=True)
fl = d_fl.cpu().numpy()
print("verdicts ok:", bool((fl[:4096] == expect[:4096]).all()))
