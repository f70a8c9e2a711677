"""Where does the full-chip slowdown come from?  Times the pairing half alone (k_ps_verify: K accumulation + two-pair Miller loop + final exponentiation)
and the whole verification at a light load (one wave per CU or less) and at the headline load (one wave per SIMD), plain layout."""
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
W = int(sys.argv[1]) if len(sys.argv) > 1 else 20
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


ctx = pkg.Context(pkg.CURVE_BN254, 0)
ctx.set_paired_layout(0)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
B = 65536
recs, expect = wl.ps_verify_batch(B)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(B, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for n in (2048, 16384, 32768, 65536):
    ms = timed(lambda: ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), 8, d_fl.data_ptr(), d_cnt.data_ptr())))
    print("ps_verify (pairing half, A=8)  n=%6d  %.3f ms  ok=%s" % (n, ms, bool((d_fl[:n].cpu().numpy() == expect[:n]).all())), flush=True)
recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
msf = ctypes.c_float()
for n in (2048, 16384, 32768, 65536):
    for reps in (1, 3):
        ctx._chk(ctx.lib.elp_time_verify_id_dev(ctx.h, stream, reps, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr(),
                                                ctypes.byref(msf)))
    print("verify_id (whole)              n=%6d  %.3f ms" % (n, msf.value), flush=True)
