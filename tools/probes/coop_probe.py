"""Small-batch latency with and without the cooperative pairing kernels: python tools/probes/coop_probe.py [window]"""
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
dev = torch.device("cuda", 0)
ctx = pkg.Context(pkg.CURVE_BN254, 0)
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
B = 16384
recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(B, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for coop in (0, 1):
    ctx.set_coop_pairing(16384 if coop else 0)
    for n in (1, 64, 1024, 2048, 4096, 8192, 16384):
        ms = timed(lambda: ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr())))
        ok = bool((d_fl[:n].cpu().numpy() == expect[:n]).all())
        print("verify_id coop=%d n=%6d  %.3f ms  %.3f M/s ok=%s" % (coop, n, ms, n / ms / 1e3, ok), flush=True)
wl3 = synth.Workload(ctx, 3, seed=20211, window_bits=W)
precs, pexpect = wl3.ps_verify_batch(B)
d_prec = torch.from_numpy(np.frombuffer(precs, dtype=np.uint8).copy()).to(dev)
for coop in (0, 1):
    ctx.set_coop_pairing(16384 if coop else 0)
    for n in (1, 1024, 4096, 8192, 16384):
        ms = timed(lambda: ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_prec.data_ptr(), 3, d_fl.data_ptr(), d_cnt.data_ptr())))
        ok = bool((d_fl[:n].cpu().numpy() == pexpect[:n]).all())
        print("ps_verify coop=%d n=%6d  %.3f ms  %.3f M/s ok=%s" % (coop, n, ms, n / ms / 1e3, ok), flush=True)
ctx.close()
