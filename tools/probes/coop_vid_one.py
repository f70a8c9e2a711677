"""One small el_passo_verify_id batch through the cooperative path (for rocprofv3 passes): python tools/probes/coop_vid_one.py [n] [window]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
recs, mask, expect = wl.verify_id_batch(n, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(n, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr()))
torch.cuda.synchronize()
print("ok", bool((d_fl.cpu().numpy() == expect).all()))
ctx.close()
