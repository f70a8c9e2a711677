import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
dev = torch.device("cuda", 0)
ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 8, seed=20211, window_bits=16)
N = 4096
recs, mask, expect = wl.verify_id_batch(N, 4, with_retrieval=True)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_ad = torch.from_numpy(np.frombuffer(wl.ad, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(N, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
worst = {}
for rnd in range(6):
    for n in (1, 64, 1024, 4096, 64, 1):
        ts = []
        for it in range(40):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx._chk(ctx.lib.elp_verify_id_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(wl.ad), d_fl.data_ptr(), d_cnt.data_ptr()))
            if it % 4 == 3:
                torch.cuda.synchronize()     # back-to-back calls in groups of four, then a drain
                ts.append((time.perf_counter() - t0) * 1e3)
        assert bool((d_fl[:n].cpu().numpy() == expect[:n]).all())
        worst[n] = max(worst.get(n, 0), max(ts))
        print("round %d n=%5d  median %.2f ms  max %.2f ms" % (rnd, n, sorted(ts)[len(ts)//2], max(ts)), flush=True)
print("worst", worst)
ctx.close()
