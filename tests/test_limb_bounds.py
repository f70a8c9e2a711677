"""The unsaturated-limb arithmetic relies on magnitude bounds (int32 limbs in lazy sums, signed 64-bit column accumulators).
A second host build of the device headers with -DELP_BOUND_CHECK redoes every limb operation in 64 bits and asserts the bounds
while the pairing / group-law / protocol flows of both curves run through it."""
import ctypes
import os
import subprocess

import elp_testlib
from elp_testlib import ROOT


def test_flows_stay_within_limb_bounds(tmp_path):
    # built beside the ordinary twin and rebuilt only when a header or the twin's source is newer (as elp_testlib.twin() does)
    so = os.path.join(ROOT, "tests", "host_twin", "libtwin.chk.so")
    inc = os.path.join(ROOT, "ps-signature-and-el-passo_amd", "csrc")
    newest = max([os.path.getmtime(os.path.join(dp, f)) for dp, _, fs in os.walk(inc) for f in fs if f.endswith(".h")] +
                 [os.path.getmtime(os.path.join(ROOT, "tests", "host_twin", "twin.cpp")), os.path.getmtime(os.path.join(ROOT, "tests", "elp_testlib.py"))])
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        elp_testlib.build_twin(so, ["-O1", "-DELP_BOUND_CHECK"])
    L = ctypes.CDLL(so)
    L.twin_bn254_ctx_new.restype = ctypes.c_void_p
    L.twin_bls_ctx_new.restype = ctypes.c_void_p
    import test_host_twin as T
    import test_host_twin_bls as TB
    import test_coop as TC
    saved = elp_testlib._twin
    elp_testlib._twin = L
    try:
        T.test_fp_arith(L)
        T.test_group_ops(L)
        T.test_pairing_value_and_cyclotomic(L)
        T.test_two_miller_loops_on_one_accumulator(L)
        T.test_glv_gls_scalar_multiplication(L)
        T.test_ps_verify_and_provide_id(L)
        T.test_verify_id_with_retrieval_golden(L)
        T.test_paired_layout_primitives(L)             # the two-lanes-per-item formulas under the same bound checks
        T.test_paired_layout_verify_id_golden(L)
        TB.test_group_ops(L)
        TB.test_pairing_equals_model_and_is_bilinear(L)
        TB.test_protocol_flows(L)
        # the cooperative programs: the light finish of a linear combination leaves magnitudes to the generator's analysis (tools/gen_coop.py analyse);
        # here every product operand, every lazy sum and every top limb of those programs is checked on real values
        TC.test_program_value_equals_model_pairing_product(TC.make_env(L))
        TC.test_bls12_381_programs_on_the_host_twin()
    finally:
        elp_testlib._twin = saved
