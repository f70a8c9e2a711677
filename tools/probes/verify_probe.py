"""Headline-shaped batch alone on either curve (CURVE=bn254|bls; ELP_LAYOUT, ELP_SPLIT, ELP_STAGE select the kernel variant); BLS12-381 by default (for rocprofv3 passes and A/B of ELP_OPT_SUBGROUP_CHECK): python tools/probes/verify_probe.py [window] [batch] [reps]"""
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
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
ctx = pkg.Context((pkg.CURVE_BN254 if os.environ.get("CURVE", "bls") == "bn254" else pkg.CURVE_BLS12_381), 0)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
d_flags = torch.zeros(B, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
ms = ctypes.c_float()
for check in ((1, 0, 1) if os.environ.get("BLS_AB") else (1,)):
    ctx.set_subgroup_check(check)
    for reps in (1, REPS):
        ctx._chk(ctx.lib.elp_time_verify_id_dev(ctx.h, stream, reps, B, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad),
                                                d_flags.data_ptr(), d_cnt.data_ptr(), ctypes.byref(ms)))
    ok = bool((d_flags.cpu().numpy() == expect).all())
    print("subgroup_check=%d n=%d  %.3f ms  %.3f M/s  ok=%s  table_GiB=%.2f" % (check, B, ms.value, B / ms.value / 1e3, ok, ctx.key_table_bytes() / 2**30), flush=True)
ctx.close()
