"""Key set-up time and footprint: elp_set_pubkey + elp_set_rp / elp_set_signer_secret at W = 16 and W = 20 in ONE fresh process, in the order
W16, W16 (second key, both resident), W20 (three keys resident), W16 again -- to tell the cost of a table width from the cost of being the first / a later context of the
process -- then the headline batch on the first W = 16 key while the others stay resident.  python tools/probes/key_setup_probe.py [batch]"""
import ctypes
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
torch.cuda.init()
ctxs, wls = [], []
for W, seed in ((16, 20211), (16, 777), (20, 20211), (16, 4242)):
    t0 = time.perf_counter()
    ctx = pkg.Context(pkg.CURVE_BN254, 0)
    t1 = time.perf_counter()
    wl = synth.Workload(ctx, 8, seed=seed, window_bits=W)
    free, total = torch.cuda.mem_get_info()
    print("W=%d  context %.0f ms  set_pubkey %.0f ms  set_rp+signer %.0f ms  table_GiB %.2f  device memory in use %.1f GiB" %
          (W, (t1 - t0) * 1e3, wl.t_set_pubkey_ms, wl.t_set_params_ms, ctx.key_table_bytes() / 2**30, (total - free) / 2**30), flush=True)
    ctxs.append(ctx)
    wls.append(wl)
stream = torch.cuda.current_stream().cuda_stream
d_fl = torch.zeros(B, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
ms = ctypes.c_float()
for which in (0, 1, 2):
    ctx, wl = ctxs[which], wls[which]
    recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
    d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
    d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
    for reps in (1, 4):
        ctx._chk(ctx.lib.elp_time_verify_id_dev(ctx.h, stream, reps, B, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr(), ctypes.byref(ms)))
    ok = bool((d_fl.cpu().numpy() == expect).all())
    print("key %d (W as above), %d proofs with %d keys resident: %.3f ms  %.3f M/s  ok=%s" % (which, B, len(ctxs), ms.value, B / ms.value / 1e3, ok), flush=True)
for c in ctxs:
    c.close()
