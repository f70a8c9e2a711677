"""One-off scale check (BASELINE.json config 5 shape on one GPU): 1 048 576 proofs for a 16-attribute key (8 hidden) made by the batch
prover in 16 launches and checked by the batch verifier - every proof must be accepted under its session id and rejected under another.
Exercises the group-law exceptional-case handling and the lazy-limb arithmetic on ~10^9 group operations."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
elp = importlib.import_module("ps-signature-and-el-passo_amd")
synth = importlib.import_module("ps-signature-and-el-passo_amd.synth")
A, H, B, CH = 16, 8, 65536, int(os.environ.get("CHUNKS", "16"))
ctx = elp.Context()
wl = synth.Workload(ctx, A, window_bits=12)
tot_ok = tot_bad = 0
t0 = time.time()
for ch in range(CH):
    recs, mask = wl.prove_id_batch(B, H, first_item=ch * B, with_retrieval=True)
    proofs, flags, cnt = ctx.prove_id_batch(recs, mask, True, b"hello")
    vf, vc = ctx.verify_id_batch(proofs, mask, True, b"hello")
    wf, wc = ctx.verify_id_batch(proofs, mask, True, b"hellp")
    tot_ok += int(vc)
    tot_bad += int(wc)
    print("chunk %2d: produced %d accepted %d accepted-under-wrong-session %d  (%.0f s)" % (ch, cnt, vc, wc, time.time() - t0), flush=True)
print("TOTAL accepted %d of %d, wrongly accepted %d" % (tot_ok, CH * B, tot_bad))
assert tot_ok == CH * B and tot_bad == 0
