"""A/B probe of the row-of-16 pairing check (ELP_OPT_PAIR16) for small PS-verification batches: verdicts against the default path and the generator's expectation,
kernel time by HIP events on the launch stream.  python tools/probes/pair16_probe.py"""
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
dev = torch.device("cuda:0")
ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 3, seed=20211, window_bits=16)
stream = torch.cuda.current_stream().cuda_stream
nmax = 16384
recs, expect = wl.ps_verify_batch(nmax)
rsz = len(recs) // nmax
r = bytearray(recs)
r[5 * rsz:5 * rsz + 64] = bytes(64)                      # sig1 = infinity: rejected (src/ps-verifier.cc:16-18)
r[9 * rsz + 64:9 * rsz + 128] = bytes(64)                # sig2 = infinity
r[11 * rsz + 3] ^= 1                                     # sig1 off the curve
recs = bytes(r)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(nmax, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)


def run(n):
    d_fl.zero_()
    ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), 3, d_fl.data_ptr(), d_cnt.data_ptr()))
    torch.cuda.synchronize()
    return d_fl.cpu().numpy()[:n].copy()


def timed(n, reps=5):
    run(n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), 3, d_fl.data_ptr(), d_cnt.data_ptr()))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n in (1, 3, 4, 5, 64, 1000, 2048, 4096, 6144, 8192):
    ctx.set_pair16(0)
    ref = run(n)
    t0 = timed(n)
    ctx.set_pair16(16384)
    got = run(n)
    t1 = timed(n)
    same = bool((ref == got).all())
    print("n=%5d  default %.3f ms   row16 %.3f ms   verdicts %s   accepted %d / %d   (expected pattern ok: %s)" % (
        n, t0, t1, "EQUAL" if same else "DIFFER at %s" % np.nonzero(ref != got)[0][:8], int(got.sum()), n,
        bool((got[12:] == expect[12:n]).all()) if n > 12 else "-"), flush=True)
ctx.close()
