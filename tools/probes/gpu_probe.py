"""Scratch GPU probe (test infrastructure): timing of the fused verify kernel on tiled oracle-generated proofs, plus the
fp_mul micro-benchmark.  Usage on the GPU box: python tests/gpu_probe.py [N]"""
import ctypes
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from elp_testlib import *  # noqa

import numpy as np
import torch

elp = importlib.import_module("ps-signature-and-el-passo_amd")
M = Mcl(BN254)
PR = Protocol(M)
G = M.G


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    A, H, seed = 8, 4, 20211
    ctx = elp.Context()
    ms = ctypes.c_float()
    for lanes, iters in ((256 * 4 * 64, 2000), (256 * 4 * 64 * 2, 2000), (256 * 4 * 64 * 4, 2000), (256 * 4 * 64 * 8, 2000)):
        ctx._chk(ctx.lib.elp_bench_fp_mul(ctx.h, lanes, iters, ctypes.byref(ms)))
        print("fp_mul bench: lanes=%d iters=%d  %.3f ms  -> %.3e modmul/s" % (lanes, iters, ms.value, lanes * iters * 2 / (ms.value * 1e-3)))
    d = load_golden("bn254_oracle_flows.json")
    gg = Codec(M).pk_decode(base64.b64decode(d["scenarios"][0]["pk"])).gg
    g = M.hash_to_g1("abc")
    x = scalar_stream(seed, 0, M.r)
    ys = [scalar_stream(seed, 1 + i, M.r) for i in range(A)]
    pk, skX = PR.key_gen(g, gg, x, ys)
    apk, h = M.hash_to_g1("ghi"), M.hash_to_g1("jkl")
    t0 = time.time()
    ctx.set_pubkey(g1b(pk.g), g2b(pk.gg), g2b(pk.XX), b"".join(g1b(P) for P in pk.Yi), b"".join(g2b(P) for P in pk.YYi), int(os.environ.get("ELP_W", "8")))
    ctx.set_rp(b"service", g1b(apk), g1b(g), g1b(h))
    print("key setup %.2fs" % (time.time() - t0))
    recs = []
    ND = 4
    for n in range(ND):
        attrs = [(("a%d-%d" % (i, n)).encode(), i < H) for i in range(A)]
        m = [M.fr_hash(a) for a, _ in attrs]
        u = scalar_stream(seed, 50 + n, M.r)
        cred = Credential(G.g1_mul(g, u), G.g1_mul(g, u * (x + sum(y * mi for y, mi in zip(ys, m))) % M.r))
        rnd = [scalar_stream(seed, 100 + 20 * n + j, M.r) for j in range(3 + H + 2)]
        pr = PR.prove_id(pk, cred, attrs, b"hello", b"service", apk, g, h, rnd)
        if n == 3:
            pr.c ^= 1
        recs.append(pack_verify_id(M, pr))
    print("proofs built %.1fs" % (time.time() - t0))
    flags, cnt = ctx.verify_id_batch(b"".join(recs), (1 << H) - 1, True, b"hello")
    print("small batch flags", flags, cnt)
    rsz = len(recs[0])
    host = np.frombuffer(b"".join(recs[i % ND] for i in range(N)), dtype=np.uint8)
    drec = torch.from_numpy(host.copy()).cuda()
    dad = torch.from_numpy(np.frombuffer(b"hello", dtype=np.uint8).copy()).cuda()
    dfl = torch.zeros(N, dtype=torch.uint8, device="cuda")
    dcnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for reps in (1, 3):
        dcnt.zero_()
        ctx._chk(ctx.lib.elp_time_verify_id_dev(ctx.h, st, reps, N, drec.data_ptr(), (1 << H) - 1, 1, dad.data_ptr(), None, 5,
                                                dfl.data_ptr(), dcnt.data_ptr(), ctypes.byref(ms)))
        torch.cuda.synchronize()
        print("verify_id N=%d: %.2f ms/launch -> %.0f verif/s ; accepted=%d (expect %d per launch)" %
              (N, ms.value, N / (ms.value * 1e-3), int(dcnt.item()), sum(1 for i in range(N) if i % ND != 3)))


if __name__ == "__main__":
    from oracle.pymodel import Credential
    main()
