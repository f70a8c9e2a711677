"""The reference's own test programs (test/ps-tests.cc, test/encoding-test.cc) and protocol layer (src/*.cc), compiled
UNCHANGED in the build container against elp_mcl_compat.h (oracle/Makefile `dropin` -> oracle/_ref/), executed here on the
HIP arithmetic.  Every flow must end "without errors" (the reference's own pass criterion)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REFDIR = os.path.join(ROOT, "oracle", "_ref")


def _run(name):
    exe = os.path.join(REFDIR, name)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/%s was not built (needs /root/reference at build time)" % name)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_reference_ps_tests_run_on_the_hip_path():
    out = _run("ref_ps_tests")
    # test/ps-tests.cc:139-145: test_ps_sign_verify + test_el_passo(3)
    assert out.count("ends without errors") == 2, out
    assert "fail" not in out.lower(), out


def test_reference_encoding_test_runs_on_the_hip_path():
    out = _run("ref_encoding_test")
    # test/encoding-test.cc:273-281: pk sizes, sign/verify, el_passo with 3 and 4 attributes (buffer test prints only on failure)
    assert out.count("ends without errors") == 4, out
    assert "fail" not in out.lower(), out
