"""Lower drop-in boundary: the reference's OWN sources must compile, unchanged, against this
project's stand-in for <mcl/bls12_381.hpp> and <cybozu/sha2.hpp>.  Runs only where
/root/reference exists (the build container); the GPU box runs the binaries built here (test_gpu_reference_dropin.py)."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = "/root/reference"
HOST = os.path.join(ROOT, "ps-signature-and-el-passo_amd", "csrc", "host")

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present")


@needs_ref
def test_reference_sources_compile_unchanged():
    srcs = sorted(glob.glob(os.path.join(REF, "src", "*.cc")) + glob.glob(os.path.join(REF, "test", "*.cc")))
    assert len(srcs) == 6, srcs   # ps-encoding, ps-signer, ps-requester, ps-verifier + ps-tests, encoding-test
    for s in srcs:
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", HOST, "-I", os.path.join(REF, "src"), s],
                           capture_output=True, text=True)
        assert r.returncode == 0, "%s does not compile against the stand-in headers:\n%s" % (s, r.stderr[:2000])


@needs_ref
def test_reference_programs_link_against_the_c_abi():
    """oracle/Makefile `dropin`: reference protocol layer + reference tests + elp_mcl_compat.cc + libelpasso_hip.so."""
    lib = os.path.join(ROOT, "ps-signature-and-el-passo_amd", "csrc", "libelpasso_hip.so")
    if not os.path.exists(lib):
        pytest.skip("HIP library not built yet")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "dropin"])
    for b in ("ref_ps_tests", "ref_encoding_test"):
        assert os.path.exists(os.path.join(ROOT, "oracle", "_ref", b))
