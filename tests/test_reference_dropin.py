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
        # the reference's own headers first (csrc/host holds this project's same-named ps-*.h), then the mcl / cybozu stand-ins
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(REF, "src"), "-I", HOST, s],
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


@needs_ref
def test_reference_programs_run_on_the_oracle_backed_shim():
    """The reference's test programs end "without errors" when the C-ABI calls of the stand-in layer are answered by the CPU oracle
    (tests/cpu_shim): value semantics, aliasing (G1::mul(x, x, k)), encodings and sizes of the stand-in types are what the
    reference's protocol code expects.  (The GPU leg runs the same programs on the HIP library.)"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "dropin-cpu"])
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_ps_tests_cpu")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.count("ends without errors") == 2 and "fail" not in out.stdout.lower(), out.stdout[-2000:]
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_encoding_test_cpu")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.count("ends without errors") == 4 and "fail" not in out.stdout.lower(), out.stdout[-2000:]
    # sizes the reference prints
    for needle in ("3 total attributes. Public Key size: 464", "20 total attributes. Public Key size: 2130", "Setup Request payload size: 176",
                   "Setup Response payload size: 68", "Sign-on Request payload size: 411"):
        assert needle in out.stdout, needle
