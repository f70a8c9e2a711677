"""Soak of the row-of-16 pairing check (k_pair16 runs without barriers: one wave per workgroup, LDS operations in program order): many calls of PS verification at mixed
sizes on both curves, every verdict compared with the generator's expectation (tampered items included); the aggregated path (k_agg_final16 on BN254) beside it.
python tools/probes/soak_pair16.py [calls]"""
import importlib
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
dev = torch.device("cuda:0")
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rnd = random.Random(6)
stream = torch.cuda.current_stream().cuda_stream
bad_total = 0
for curve, name in ((pkg.CURVE_BN254, "BN254"), (pkg.CURVE_BLS12_381, "BLS12-381")):
    os.environ["ELP_PAIR16_MIN"] = "1"
    ctx = pkg.Context(curve, 0)
    del os.environ["ELP_PAIR16_MIN"]
    wl = synth.Workload(ctx, 3, seed=31337, window_bits=16)
    nmax = 4096
    recs, expect = wl.ps_verify_batch(nmax)
    d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
    d_fl = torch.zeros(nmax, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    bad = 0
    ncalls = calls if curve == pkg.CURVE_BN254 else calls // 3
    for it in range(ncalls):
        n = rnd.choice((1, 3, 4, 5, 63, 64, 65, 1000, 2048, 4095, 4096, rnd.randrange(1, 4097)))
        d_fl.zero_()
        d_cnt.zero_()
        ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), 3, d_fl.data_ptr(), d_cnt.data_ptr()))
        torch.cuda.synchronize()
        got = d_fl.cpu().numpy()[:n]
        if not (got == expect[:n]).all() or int(d_cnt.item()) != int(expect[:n].sum()):
            bad += 1
            print("%s MISMATCH call %d n=%d at %s" % (name, it, n, np.nonzero(got != expect[:n])[0][:8]), flush=True)
    print("%s: %d PS-verification calls on the row-of-16 kernel, %d mismatching" % (name, ncalls, bad), flush=True)
    bad_total += bad
    ctx.close()
# aggregated verification with the closing step on one row (BN254)
ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 8, seed=4141, window_bits=16)
B = 8192
recs, mask, expect = wl.verify_id_batch(B, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(B, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
bad = 0
for it in range(max(20, calls // 10)):
    n = rnd.choice((64, 1000, 4097, 8192))
    d_fl.zero_()
    ctx._chk(ctx.lib.elp_verify_id_batch_aggregated_dev(ctx.h, stream, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), None, d_fl.data_ptr(), d_cnt.data_ptr()))
    torch.cuda.synchronize()
    if not (d_fl.cpu().numpy()[:n] == expect[:n]).all():
        bad += 1
        print("aggregated MISMATCH call %d n=%d" % (it, n), flush=True)
print("aggregated (k_agg_final16): %d calls, %d mismatching" % (max(20, calls // 10), bad), flush=True)
bad_total += bad
# ... and through the host-buffer entry point, which also tells whether the batch equation HELD: a wrong product of the Miller values (k_fp12_reduce16) or a wrong
# Pippenger sum (k_msm_combine) would send the batch to the per-item fallback with the verdicts still right
rsz = len(recs) // B
bad = 0
for it in range(max(20, calls // 10)):
    n = rnd.choice((1, 64, 513, 2117, 2624, 4097, 8192))
    fl, cnt, held = ctx.verify_id_batch_aggregated(recs[:n * rsz], mask, True, wl.ad, None)
    if not held or not (fl == expect[:n]).all() or cnt != int(expect[:n].sum()):
        bad += 1
        print("aggregated (host entry) MISMATCH call %d n=%d held=%s" % (it, n, held), flush=True)
print("aggregated, batch equation held and verdicts equal: %d calls, %d failing" % (max(20, calls // 10), bad), flush=True)
bad_total += bad
ctx.close()
# the same on BLS12-381: main kernel on lane pairs (k_verify_id_agg_paired), product tree and closing step on rows
ctx = pkg.Context(pkg.CURVE_BLS12_381, 0)
wl = synth.Workload(ctx, 6, seed=5151, window_bits=12)
B = 4096
recs, mask, expect = wl.verify_id_batch(B, 3, with_retrieval=True, corrupt_every=13, corrupt_at=5)
rsz = len(recs) // B
bad = 0
for it in range(max(10, calls // 20)):
    n = rnd.choice((1, 31, 64, 513, 2117, 4096))
    fl, cnt, held = ctx.verify_id_batch_aggregated(recs[:n * rsz], mask, True, wl.ad, None)
    if not held or not (fl == expect[:n]).all() or cnt != int(expect[:n].sum()):
        bad += 1
        print("BLS12-381 aggregated MISMATCH call %d n=%d held=%s" % (it, n, held), flush=True)
print("BLS12-381 aggregated (lane pairs + rows), batch equation held and verdicts equal: %d calls, %d failing" % (max(10, calls // 20), bad), flush=True)
bad_total += bad
ctx.close()
sys.exit(1 if bad_total else 0)
