"""Aggregated verification at the headline size, a few calls (run under rocprofv3 --kernel-trace --stats to see its kernels).
Usage: [CURVE=bls] [AGG_TWO=0|1|2] python tools/probes/agg_probe.py [batch] [window] [ELP_COOP value]      (AGG_TWO: ELP_OPT_AGG_TWO_PER_LANE)"""
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
ctx = pkg.Context(pkg.CURVE_BLS12_381 if os.environ.get("CURVE", "bn254").startswith("bls") else pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=int(sys.argv[2]) if len(sys.argv) > 2 else 16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
if len(sys.argv) > 3:
    os.environ["ELP_COOP"] = sys.argv[3]
recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
d_flags = torch.zeros(B, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
if os.environ.get("AGG_TWO"):
    ctx.set_agg_two_per_lane(int(os.environ["AGG_TWO"]))
for it in range(4):
    d_cnt.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx._chk(ctx.lib.elp_verify_id_batch_aggregated_dev(ctx.h, stream, B, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), None,
                                                        d_flags.data_ptr(), d_cnt.data_ptr()))
    e1.record()
    torch.cuda.synchronize()
    print("agg call %d: %.3f ms  ok=%s" % (it, e0.elapsed_time(e1), bool((d_flags.cpu().numpy() == expect).all())), flush=True)
for it in range(4):      # per-item calls beside them (same lease: clock and cycles per instruction compared in one rocprofv3 pass)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, B, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), d_flags.data_ptr(), d_cnt.data_ptr()))
    e1.record()
    torch.cuda.synchronize()
    print("per-item call %d: %.3f ms" % (it, e0.elapsed_time(e1)), flush=True)
ctx.close()
