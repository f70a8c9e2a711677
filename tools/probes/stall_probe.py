"""Looks for the cross-stream stall of ELP_OPT_STREAM_OVERLAP seen inside bench.py's process: the small-batch call sequence with the option on,
alone / beside a second live context / beside a second context that used its own second stream.  None of the three reproduces it.
Usage: python tools/probes/stall_probe.py alone|second_ctx|second_ctx_used"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
mode = sys.argv[1]
others = []
if mode in ("second_ctx", "second_ctx_used"):
    c2 = pkg.Context(pkg.CURVE_BN254, 0)
    w2 = synth.Workload(c2, 8, seed=1, window_bits=8)
    others.append((c2, w2))
    if mode == "second_ctx_used":
        r2, m2, e2 = w2.verify_id_batch(64, 4, with_retrieval=True)
        c2.set_stream_overlap(1)
        c2.verify_id_batch(r2, m2, True, w2.ad)          # creates its second stream, host-buffer path on its own stream
ctx = pkg.Context(pkg.CURVE_BN254, 0)
ctx.set_stream_overlap(1)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=16)
nl = 4096
vrecs, vmask, vexpect = wl.verify_id_batch(nl, 4, with_retrieval=True)
d_vrec = torch.from_numpy(np.frombuffer(vrecs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(nl, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for m in (4096, 1, 64, 1024, 4096, 4096):
    ms = timed(lambda: ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, m, d_vrec.data_ptr(), vmask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr())))
    print(mode, "n=%5d  %.3f ms" % (m, ms), flush=True)
