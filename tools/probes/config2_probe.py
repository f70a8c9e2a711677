"""BASELINE config 2 (4 096 PS verifications, A = 3) and a lone PS verification through the cooperative kernels, a few calls each: the command the per-kernel
counter passes of tools/pmc_kernels.sh run over.  Usage: python tools/probes/config2_probe.py [window] [calls]"""
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
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
ctx = pkg.Context(pkg.CURVE_BN254, 0)
stream = torch.cuda.current_stream().cuda_stream
wl3 = synth.Workload(ctx, 3, seed=20211, window_bits=W)
B = 4096
precs, pexpect = wl3.ps_verify_batch(B)
d_prec = torch.from_numpy(np.frombuffer(precs, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(B, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for n in (B, 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_prec.data_ptr(), 3, d_fl.data_ptr(), d_cnt.data_ptr()))
    e0.record()
    for _ in range(CALLS):
        ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_prec.data_ptr(), 3, d_fl.data_ptr(), d_cnt.data_ptr()))
    e1.record()
    torch.cuda.synchronize()
    ok = bool((d_fl[:n].cpu().numpy() == pexpect[:n]).all())
    print("ps_verify n=%5d  %.3f ms per call  ok=%s" % (n, e0.elapsed_time(e1) / CALLS, ok), flush=True)
ctx.close()
