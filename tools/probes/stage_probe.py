"""A/B of el_passo_verify_id with records read in place (k_verify_id) against coalesced record loads through LDS (k_verify_id_staged) on one GPU: kernel time per batch size,
verdicts checked against the generator's expectation.  Usage: python tools/probes/stage_probe.py [window] [sizes,comma,separated]"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1000, 16384, 65536, 131072]
dev = torch.device("cuda", 0)
ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
B = max(sizes)
recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
d_flags = torch.zeros(B, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
ms = ctypes.c_float()
ctx.set_paired_layout(0)
for stage in (0, 1, 0, 1, 0, 1):
    ctx.set_coalesced_records(stage)
    for n in sizes:
        d_flags.zero_()
        for reps in (1, 4):
            ctx._chk(ctx.lib.elp_time_verify_id_dev(ctx.h, stream, reps, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad),
                                                    d_flags.data_ptr(), d_cnt.data_ptr(), ctypes.byref(ms)))
        ok = bool((d_flags[:n].cpu().numpy() == expect[:n]).all())
        print("coalesced=%d n=%6d  %.3f ms  %.3f M/s  ok=%s" % (stage, n, ms.value, n / ms.value / 1e3, ok), flush=True)
ctx.close()
