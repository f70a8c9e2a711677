"""A/B of the four-lanes-per-item pairing check (ELP_OPT_PAIR4; round 5): el_passo_verify_id (A = 8, H = 4, id-retrieval) and PS verification (A = 3) at a range of batch
sizes with the option off (0: the round-4 paths), at its default (1) and forced (2).  Usage: [CURVE=bls] python tools/probes/pair4_probe.py [window] [n ...]"""
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
NS = [int(a) for a in sys.argv[2:]] or [1024, 4096, 8192, 9217, 12288, 16384, 24576, 32768, 65536]
MODES = [int(x) for x in os.environ.get("MODES", "0,2").split(",")]
dev = torch.device("cuda", 0)
bls = os.environ.get("CURVE", "bn254").startswith("bls")
ctx = pkg.Context(pkg.CURVE_BLS12_381 if bls else pkg.CURVE_BN254, 0)
stream = torch.cuda.current_stream().cuda_stream
B = max(NS)
d_fl = torch.zeros(B, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)


def timed(call, reps=4):
    call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if os.environ.get("WHAT", "both") in ("both", "verify"):
    wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
    recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
    d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
    d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
    for n in NS:
        row = []
        for mode in MODES:
            ctx.set_pair4(mode)
            ms = timed(lambda: ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr())))
            ok = bool((d_fl[:n].cpu().numpy() == expect[:n]).all())
            row.append("pair4=%d %8.3f ms %5.2f M/s ok=%s" % (mode, ms, n / ms / 1e3, ok))
        print("verify_id n=%6d  " % n + "   ".join(row), flush=True)
if os.environ.get("WHAT", "both") in ("both", "ps"):
    wl3 = synth.Workload(ctx, 3, seed=20211, window_bits=W)
    precs, pexpect = wl3.ps_verify_batch(B)
    d_prec = torch.from_numpy(np.frombuffer(precs, dtype=np.uint8).copy()).to(dev)
    for n in NS:
        row = []
        for mode in MODES:
            ctx.set_pair4(mode)
            ms = timed(lambda: ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_prec.data_ptr(), 3, d_fl.data_ptr(), d_cnt.data_ptr())))
            ok = bool((d_fl[:n].cpu().numpy() == pexpect[:n]).all())
            row.append("pair4=%d %8.3f ms %5.2f M/s ok=%s" % (mode, ms, n / ms / 1e3, ok))
        print("ps_verify n=%6d  " % n + "   ".join(row), flush=True)
ctx.close()
