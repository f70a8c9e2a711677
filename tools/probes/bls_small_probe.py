"""BLS12-381 small batches: el_passo_verify_id and PS verification at n = 1 .. 4096 with the cooperative path on / off (ELP_OPT_COOP_PAIRING).
Usage: python tools/probes/bls_small_probe.py [window]"""
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


ctx = pkg.Context(pkg.CURVE_BLS12_381, 0)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
nl = 4096
vrecs, vmask, vexpect = wl.verify_id_batch(nl, 4, with_retrieval=True)
d_vrec = torch.from_numpy(np.frombuffer(vrecs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(nl, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for coop in (1, 0):
    ctx.set_coop_pairing(8192 if coop else 0)
    for m in (4096, 1, 64, 256, 512, 1024, 2048, 4096):
        ms = timed(lambda: ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, m, d_vrec.data_ptr(), vmask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr())))
        ok = bool((d_fl[:m].cpu().numpy() == vexpect[:m]).all())
        print("verify_id coop=%d n=%5d  %.3f ms  ok=%s" % (coop, m, ms, ok), flush=True)
wl3 = synth.Workload(ctx, 3, seed=20211, window_bits=W)
precs, pexpect = wl3.ps_verify_batch(nl)
d_prec = torch.from_numpy(np.frombuffer(precs, dtype=np.uint8).copy()).to(dev)
for coop in (1, 0):
    ctx.set_coop_pairing(8192 if coop else 0)
    for m in (4096, 1, 64, 512, 1024, 2048, 4096):
        ms = timed(lambda: ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, m, d_prec.data_ptr(), 3, d_fl.data_ptr(), d_cnt.data_ptr())))
        ok = bool((d_fl[:m].cpu().numpy() == pexpect[:m]).all())
        print("ps_verify coop=%d n=%5d  %.3f ms  ok=%s" % (coop, m, ms, ok), flush=True)
ctx.close()
