"""Scratch GPU probe: aggregated verification pipelined over two streams / two contexts (tail of batch i overlaps the per-item
kernel of batch i+1)."""
import ctypes
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

elp = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
B, A, H = 65536, 8, 4
W = int(os.environ.get("ELP_W", "16"))
ctxs = [elp.Context(), elp.Context()]
wls = [synth.Workload(c, A, window_bits=W) for c in ctxs]
recs, mask, expect = wls[0].verify_id_batch(B, H)
dev = torch.device("cuda", 0)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(b"hello", dtype=np.uint8).copy()).to(dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
d_flags = [torch.zeros(B, dtype=torch.uint8, device=dev) for _ in range(2)]
d_cnt = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2)]
seed = np.frombuffer(bytes(range(32)), dtype=np.uint8).copy()


def launch(k, agg):
    s = k % 2
    c = ctxs[s]
    with torch.cuda.stream(streams[s]):
        if agg:
            c._chk(c.lib.elp_verify_id_batch_aggregated_dev(c.h, streams[s].cuda_stream, B, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, 5,
                                                            seed.ctypes.data, d_flags[s].data_ptr(), d_cnt[s].data_ptr()))
        else:
            c._chk(c.lib.elp_verify_id_batch_dev(c.h, streams[s].cuda_stream, B, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, 5,
                                                 d_flags[s].data_ptr(), d_cnt[s].data_ptr()))


for agg in (False, True):
    for k in range(2):
        launch(k, agg)
    torch.cuda.synchronize()
    K = 8
    t0 = time.perf_counter()
    for k in range(K):
        launch(k, agg)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(bool((f.cpu().numpy() == expect).all()) for f in d_flags)
    print("aggregated=%s two-stream pipeline: %.2f ms per batch -> %.0f verif/s  parity=%s" % (agg, dt / K * 1e3, B * K / dt, ok))
