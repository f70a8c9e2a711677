"""Regression test for the root cause of the BLS12-381 "-DELP_FP6_INLINE=1" fault (profiles/r05_bls_fault.md): a LEAF device function larger than the
+-128 KB reach of s_cbranch whose far branches were expanded through s[30:31] -- its own, never saved, return address -- by ROCm 7.2.0's hipcc.  The shipped library is
disassembled (every gfx950 code object of the fat binary) and must not contain such a function; the scanner itself is checked on a synthetic listing of both shapes."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_scanner_flags_the_hazard_and_only_it(tmp_path):
    import check_long_branch as clb
    body = "\tv_add_u32_e32 v0, v1, v2\n" * 4
    far = "\ts_getpc_b64 s[30:31]\n.Lpost_getpc1:\n\ts_add_u32 s30, s30, 12\n\ts_addc_u32 s31, s31, 0\n\ts_setpc_b64 s[30:31]\n"
    listing = ("leaf_clobbered:\n" + body + far + body + "\ts_setpc_b64 s[30:31]\n"                                   # the hazard
               "nonleaf_saved:\n\tv_writelane_b32 v40, s30, 0\n\tv_writelane_b32 v40, s31, 1\n" + body + far + "\tv_readlane_b32 s30, v40, 0\n\ts_setpc_b64 s[30:31]\n"
               "leaf_other_pair:\n" + body + far.replace("s[30:31]", "s[4:5]").replace("s30", "s4").replace("s31", "s5") + "\ts_setpc_b64 s[30:31]\n"
               "a_kernel:\n" + body + far + "\ts_endpgm\n"
               # a save AFTER the clobbering s_getpc_b64 stores the clobbered value: still the hazard (round-5 advisor finding)
               "leaf_saved_too_late:\n" + body + far + "\tv_writelane_b32 v40, s30, 0\n\tv_writelane_b32 v40, s31, 1\n" + body + "\ts_setpc_b64 s[30:31]\n")
    p = tmp_path / "listing.s"
    p.write_text(listing)
    bad = clb.scan(str(p))
    assert [b[0] for b in bad] == ["leaf_clobbered", "leaf_saved_too_late"] and bad[0][1] == 1


def test_shipped_library_has_no_function_that_clobbers_its_return_address():
    import check_long_branch as clb
    b = importlib.import_module("ps-signature-and-el-passo_amd.build")
    lib = b.build_hip()
    bad, n = clb.scan_library(lib)
    assert n >= 10, "no gfx950 code objects found in %s" % lib
    assert bad == [], bad
