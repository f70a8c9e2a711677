"""Scratch GPU probe: k_verify_id time vs batch size (does a second resident wave per SIMD help?), plus bench_op at 1/2 waves."""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
elp = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
W = int(os.environ.get("W", "8"))
ctx = elp.Context()
wl = synth.Workload(ctx, 8, window_bits=W)
recs, mask, expect = wl.verify_id_batch(16384, 4, window_bits=W if W <= 16 else 8)
dev = torch.device("cuda:0")
host = np.frombuffer(recs, dtype=np.uint8)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
stream = torch.cuda.current_stream().cuda_stream
for mult in (1, 2, 3, 4, 6, 8, 16):
    B = 16384 * mult
    d_rec = torch.from_numpy(np.tile(host, mult)).to(dev)
    d_flags = torch.zeros(B, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    ms = ctypes.c_float()
    ctx._chk(ctx.lib.elp_time_verify_id_dev(ctx.h, stream, 3, B, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad),
                                            d_flags.data_ptr(), d_cnt.data_ptr(), ctypes.byref(ms)))
    print("B=%7d  waves/SIMD=%.2f  kernel %.2f ms  %.3f M/s" % (B, B / 65536, ms.value, B / ms.value / 1e3), flush=True)
