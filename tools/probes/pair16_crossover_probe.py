import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
pkg = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
dev = torch.device("cuda:0")
bls = os.environ.get("CURVE", "bn254").startswith("bls")
ctx = pkg.Context(pkg.CURVE_BLS12_381 if bls else pkg.CURVE_BN254, 0)
wl = synth.Workload(ctx, 3, seed=20211, window_bits=16)
stream = torch.cuda.current_stream().cuda_stream
recs, expect = wl.ps_verify_batch(4096)
d_rec = torch.from_numpy(np.frombuffer(recs, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(4096, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
def timed(n, reps=4):
    f = lambda: ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_rec.data_ptr(), 3, d_fl.data_ptr(), d_cnt.data_ptr()))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for n in (1024, 1536, 2048, 2049, 2560, 3072, 4096):
    ctx.set_pair16(0); a = timed(n)
    ctx.set_pair16(1); b = timed(n)
    print("n=%d interpreter %.3f ms  row16 %.3f ms" % (n, a, b), flush=True)
