"""Why is a lone el_passo_verify_id (n = 1) slower than 64 of them in the same code path?  Phases of 12 calls each at n = 1, 64, 1, 64 (run under
rocprofv3 --kernel-trace [--pmc GRBM_GUI_ACTIVE]); tools/kernel_phase_stats.py groups the trace by phase.
Usage: python tools/probes/lone_call_probe.py [window]"""
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
W = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
nl = 64
vrecs, vmask, vexpect = wl.verify_id_batch(nl, 4, with_retrieval=True)
d_vrec = torch.from_numpy(np.frombuffer(vrecs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(nl, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
marker = torch.zeros(1024, device=dev)
for phase, m in enumerate((64, 1, 64, 1, 64)):
    marker.add_(1.0)                    # a torch kernel between phases: the phase separator in the trace
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12):
        ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, m, d_vrec.data_ptr(), vmask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr()))
    torch.cuda.synchronize()
    print("phase %d n=%2d  %.3f ms per call (wall, 12 calls back to back)" % (phase, m, (time.perf_counter() - t0) / 12 * 1e3), flush=True)
ctx.close()
