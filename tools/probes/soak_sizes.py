"""Soak over the batch-size dispatch of el_passo_verify_id (WHAT=records, the default), of the same proofs as wire messages (WHAT=wire: decode kernel + record paths) and of
PS verification (WHAT=ps): random batch lengths between 1 and 20 000 (every kernel of the cooperative range, the four-lane range up to 16 384 and the first sizes of the
two-lane kernels, with all the thresholds and their neighbours), corrupted items sprinkled in, back-to-back calls; every verdict compared with the generator's expectation,
slowest call reported.  Usage: [CURVE=bls] [WHAT=records|wire|ps] python tools/probes/soak_sizes.py [calls] [window]"""
import importlib
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
W = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
bls = os.environ.get("CURVE", "bn254").startswith("bls")
ctx = pkg.Context(pkg.CURVE_BLS12_381 if bls else pkg.CURVE_BN254, 0)
WHAT = os.environ.get("WHAT", "records")
N = int(os.environ.get("NMAX", "20000"))
if WHAT == "ps":
    wl = synth.Workload(ctx, 3, seed=20211, window_bits=W)
    recs, expect = wl.ps_verify_batch(N)
    mask = 0
else:
    wl = synth.Workload(ctx, 8, seed=20211, window_bits=W)
    recs, mask, expect = wl.verify_id_batch(N, 4, with_retrieval=True, corrupt_every=11, corrupt_at=5)
rsz = len(recs) // N
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
if WHAT == "wire":
    msgs, moff = wl.wire_messages(recs, N, 4, with_retrieval=True)
    d_msg = torch.from_numpy(np.frombuffer(msgs, dtype=np.uint8).copy()).to(dev)
    d_off = torch.from_numpy(np.asarray(moff, dtype=np.uint32).view(np.int32).copy()).to(dev)
d_fl = torch.zeros(N, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
rnd = random.Random(99)
edges = [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1791, 1792, 1793, 2047, 2048, 2049, 3071, 3072, 3073, 4095, 4096, 4097, 8191, 8192, 8193, 9215, 9216, 9217, 12000,
         16383, 16384, 16385, N]
edges = [e for e in edges if e <= N]
slow, bad, t_all = (0.0, 0), 0, time.perf_counter()
for it in range(CALLS):
    n = edges[it] if it < len(edges) else (rnd.choice(edges) if rnd.random() < 0.2 else rnd.randrange(1, N + 1))
    off = rnd.randrange(0, N - n + 1)                      # a window of the prepared records: different items, different corrupted positions every call
    d_fl.zero_()
    d_cnt.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if WHAT == "ps":
        ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_rec.data_ptr() + off * rsz, 3, d_fl.data_ptr(), d_cnt.data_ptr()))
    elif WHAT == "wire":        # the offsets are absolute: a window of the offset array over the whole message buffer
        ctx._chk(ctx.lib.elp_verify_id_wire_batch_dev(ctx.h, stream, n, d_msg.data_ptr(), d_off.data_ptr() + 4 * off, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(),
                                                      d_cnt.data_ptr()))
    else:
        ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, n, d_rec.data_ptr() + off * rsz, mask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr()))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    if ms > slow[0]:
        slow = (ms, n)
    got = d_fl[:n].cpu().numpy()
    if not (got == expect[off:off + n]).all() or int(d_cnt.item()) != int(expect[off:off + n].sum()):
        bad += 1
        print("MISMATCH call %d n=%d off=%d" % (it, n, off), flush=True)
print("%s %s: %d calls, %d mismatches, slowest call %.2f ms (n = %d), %.1f s in all" % ("bls12_381" if bls else "bn254", WHAT, CALLS, bad, slow[0], slow[1], time.perf_counter() - t_all), flush=True)
ctx.close()
sys.exit(1 if bad else 0)
