"""el_passo_verify_id at a few mid-size batch lengths through the default path, a few calls each (kernel-trace / counter passes run over this).
Usage: [CURVE=bls] [COOP_MAX=..] [ELP_SMALL_ONE_MAX=..] [ELP_SMALL_DENSE_FROM=..] python tools/probes/vid_mid_probe.py [window] [n ...]"""
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
NS = [int(a) for a in sys.argv[2:]] or [4096, 1024]
dev = torch.device("cuda", 0)
ctx = pkg.Context(pkg.CURVE_BLS12_381 if os.environ.get("CURVE", "bn254").startswith("bls") else pkg.CURVE_BN254, 0)
if os.environ.get("COOP_MAX"):          # A/B: 0 = per-lane kernels, > 1 = upper limit of the cooperative path
    ctx.set_coop_pairing(int(os.environ["COOP_MAX"]))
stream = torch.cuda.current_stream().cuda_stream
wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
B = max(NS)
recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(B, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for n in NS:
    call = lambda: ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr()))
    call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        call()
    e1.record()
    torch.cuda.synchronize()
    ok = bool((d_fl[:n].cpu().numpy() == expect[:n]).all())
    print("verify_id n=%6d  %.3f ms per call  ok=%s" % (n, e0.elapsed_time(e1) / 4, ok), flush=True)
ctx.close()
