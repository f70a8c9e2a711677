"""One small PS-verification batch through the cooperative kernels (for rocprofv3 passes): python tools/probes/coop_one.py [n]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
ctx = pkg.Context(pkg.CURVE_BN254, 0)
wl3 = synth.Workload(ctx, 3, seed=20211, window_bits=8)
precs, pexpect = wl3.ps_verify_batch(n)
d_prec = torch.from_numpy(np.frombuffer(precs, dtype=np.uint8).copy()).to(dev)
d_fl = torch.zeros(n, dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    ctx._chk(ctx.lib.elp_ps_verify_batch_dev(ctx.h, stream, n, d_prec.data_ptr(), 3, d_fl.data_ptr(), d_cnt.data_ptr()))
torch.cuda.synchronize()
print("ok", bool((d_fl.cpu().numpy() == pexpect).all()))
ctx.close()
