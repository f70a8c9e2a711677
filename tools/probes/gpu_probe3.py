"""Scratch GPU probe: cycles per device routine (elp_bench_op) at 1 and 2 waves per SIMD."""
import ctypes
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
elp = importlib.import_module("ps-signature-and-el-passo_amd")
OPS = ["fp_mul", "fp_sqr", "fp2_mul", "fp2_sqr", "fp6_mul", "fp12_mul", "fp12_sqr", "fp12_cyc_sqr", "mul_by_line", "jac_dbl<G1>",
       "jac_madd<G1>", "jac_add<G1>", "jac_dbl<G2>", "jac_madd<G2>", "jac_add<G2>", "ml_dbl_step", "ml_add_step", "fp_inv", "fp_add+sub",
       "fp2_add+sub", "jac_mul_var<G1>", "jac_mul_var<G2>", "miller_loop 1 pair", "final_exp"]
MULS = [1, 1, 3, 2, 18, 54, 36, 18, 39, 7, 11, 16, 16, 29, 43, 25, 35, 380, 0, 0, 2970, 7200, 9000, 6000]   # Fp products per application (model)
ctx = elp.Context()
ms = ctypes.c_float()
import os
SEL = [int(x) for x in os.environ.get('OPS', ','.join(str(i) for i in range(24))).split(',')]
for waves in [int(x) for x in os.environ.get('WAVES','1').split(',')]:
    lanes = 256 * 4 * 64 * waves
    print("== %d wave(s) per SIMD" % waves)
    for op, name in enumerate(OPS):
        if op not in SEL:
            continue
        iters = 2 if MULS[op] >= 2000 else (20 if name == "fp_inv" else (2000 if MULS[op] <= 3 else 200))
        ctx._chk(ctx.lib.elp_bench_op(ctx.h, op, lanes, iters, ctypes.byref(ms)))
        ns = ms.value * 1e6 / iters          # per application per wave (all SIMDs in parallel)
        cyc = ns * 2.4 / waves               # SIMD cycles per application at 2.4 GHz
        print("%-14s %9.1f ns/app  ~%8.0f SIMD-cycles  %s" % (name, ns, cyc, ("%.0f cyc per Fp product" % (cyc / MULS[op])) if MULS[op] else ""))
