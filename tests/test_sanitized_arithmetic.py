"""The arithmetic under AddressSanitizer + UndefinedBehaviorSanitizer (VERDICT r4 #2 / "what's missing" #6): the host twins of the device headers -- field, tower, curve,
pairing, the protocol bodies in the one-lane and two-lane layouts, the cooperative interpreter, the four-lane layer -- are built with
`clang++ -fsanitize=address,undefined -fno-sanitize-recover=all` (ROCm's clang; -DELP_NO_FORCE_INLINE keeps the instrumented build made of small functions, which is what
lets it compile in a minute where the always-inlined form did not finish in half an hour) and the existing twin tests run against that build in a child interpreter
with the ASan runtime preloaded: BN254 protocol flows on golden vectors of the reference's wasm (src/ps-verifier.cc:37-138, src/ps-signer.cc:63-146), the BLS12-381 flows
against the big-int model, the paired layout, the level-scheduled programs, the quad.  Any out-of-bounds access, signed overflow, invalid shift or misaligned access in the
limb code aborts the child.  ELP_SAN_FULL=1 runs the three long golden sweeps too (+1 minute)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SAN = ["-std=c++17", "-O1", "-g", "-fPIC", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-shared-libsan", "-DELP_NO_FORCE_INLINE"]
HT = os.path.join(ROOT, "tests", "host_twin")
INC = os.path.join(ROOT, "ps-signature-and-el-passo_amd", "csrc")


def _newest():
    return max([os.path.getmtime(os.path.join(dp, f)) for dp, _, fs in os.walk(INC) for f in fs if f.endswith(".h")] +
               [os.path.getmtime(os.path.join(HT, f)) for f in ("twin.cpp", "twin_quad.cpp")])


def _build(so, src, part_flag, parts):
    if os.path.exists(so) and os.path.getmtime(so) >= _newest():
        return
    objs, procs = [], []
    for part in parts:
        obj = "%s.p%d.o" % (so, part)
        objs.append(obj)
        procs.append(subprocess.Popen([CLANG] + SAN + ["-D%s=%d" % (part_flag, part), "-I", INC, "-c", "-o", obj, src]))
    for p in procs:
        assert p.wait() == 0, "sanitizer build failed"
    subprocess.check_call([CLANG, "-shared", "-fsanitize=address,undefined", "-shared-libsan", "-o", so] + objs + ["-lpthread"])
    for o in objs:
        os.remove(o)


def test_twin_tests_pass_under_asan_and_ubsan():
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not os.path.exists(CLANG) or not rt:
        pytest.skip("ROCm's clang / its ASan runtime are not in this image")
    twin_so, quad_so = os.path.join(HT, "libtwin.san.so"), os.path.join(HT, "libtwin_quad.san.so")
    _build(twin_so, os.path.join(HT, "twin.cpp"), "TWIN_PART", (1, 2))
    _build(quad_so, os.path.join(HT, "twin_quad.cpp"), "TWINQ_CURVE", (0, 1))
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               ELP_TWIN_LIB=twin_so, ELP_TWINQ_LIB=quad_so, ELP_TWIN_SAMPLE="8")
    long_sweeps = "test_verify_id_from_wire_messages_golden or test_verify_id_golden or test_paired_layout_verify_id_golden"
    # pure-Python generator checks of test_coop.py never enter the instrumented library: they run once, in the plain suite
    python_only = "program_header_is_what_the_generator_emits or test_register_allocation_keeps_the_lds_banks_apart"
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_host_twin.py"),
           os.path.join(ROOT, "tests", "test_host_twin_bls.py"), os.path.join(ROOT, "tests", "test_coop.py"), os.path.join(ROOT, "tests", "test_quad_twin.py")]
    if not os.environ.get("ELP_SAN_FULL"):
        cmd += ["-k", "not (%s or %s)" % (long_sweeps, python_only)]
    else:
        env["ELP_TWIN_SAMPLE"] = "1"
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
