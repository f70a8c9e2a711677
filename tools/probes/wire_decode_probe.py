"""A/B of the wire -> record decode in front of the record paths (ELP_OPT_WIRE_DECODE; round 5): el_passo_verify_id on undecoded IdProof messages (A = 8, H = 4,
id-retrieval) at a range of batch sizes with the option off (the fused wire kernels) and on (k_wire_decode + the record path of that size + k_wire_combine), the record
path on the same proofs beside it.  Usage: [CURVE=bls] python tools/probes/wire_decode_probe.py [window] [n ...]"""
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
NS = [int(a) for a in sys.argv[2:]] or [1, 64, 1024, 4096, 8192, 16384]
dev = torch.device("cuda", 0)
bls = os.environ.get("CURVE", "bn254").startswith("bls")
ctx = pkg.Context(pkg.CURVE_BLS12_381 if bls else pkg.CURVE_BN254, 0)
stream = torch.cuda.current_stream().cuda_stream
B = max(NS)
d_fl = torch.zeros(B, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)


def timed(call, reps=8):
    call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
msgs, moff = wl.wire_messages(recs, B, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_msg = torch.from_numpy(np.frombuffer(msgs, dtype=np.uint8).copy()).to(dev)
d_off = torch.from_numpy(np.asarray(moff, dtype=np.uint32).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
for n in NS:
    row = []
    ms = timed(lambda: ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr())))
    row.append("records %8.3f ms" % ms)
    for mode in (0, 1):
        ctx.set_wire_decode(mode)
        d_fl.zero_()
        ms = timed(lambda: ctx._chk(ctx.lib.elp_verify_id_wire_batch_dev(ctx.h, stream, n, d_msg.data_ptr(), d_off.data_ptr(), 1, d_ad.data_ptr(), None, len(wl.ad),
                                                                              d_fl.data_ptr(), d_cnt.data_ptr())))
        ok = bool((d_fl[:n].cpu().numpy() == expect[:n]).all())
        row.append("wire, decode=%d %8.3f ms %5.2f M/s ok=%s" % (mode, ms, n / ms / 1e3, ok))
    print("verify_id n=%6d  " % n + "   ".join(row), flush=True)
ctx.close()
