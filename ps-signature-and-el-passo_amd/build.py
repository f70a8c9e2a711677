"""Build the HIP shared library in-tree (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libelpasso_hip.so")
HOST_LIB = os.path.join(CSRC, "libelpasso_host.so")


def _newest(paths):
    t = 0.0
    for p in paths:
        if os.path.isdir(p):
            for dp, _, fs in os.walk(p):
                for f in fs:
                    if f.endswith((".h", ".hip", ".cc", ".cpp")):
                        t = max(t, os.path.getmtime(os.path.join(dp, f)))
        elif os.path.exists(p):
            t = max(t, os.path.getmtime(p))
    return t


# translation units of libelpasso_hip.so: the two curves compile in parallel (the per-curve units hold the explicit template
# instantiations of every kernel); per-unit flags: BN254 additionally inlines the Fp6-level routines into Fp12-level leaf
# functions (+13 % on the verify kernel; on the 14-limb BLS12-381 field it only doubles the compile time).
HIP_UNITS = [("elpasso_capi.hip", []), ("elpasso_bn254.hip", ["-DELP_FP6_INLINE=1"]), ("elpasso_bn254_pair.hip", ["-DELP_FP6_INLINE=1"]),
             ("elpasso_bn254_nizk.hip", ["-DELP_FP6_INLINE=1"]),
             ("elpasso_bls12_381.hip", []), ("elpasso_bls12_381_pair.hip", [])]


def build_hip(force=False, verbose=False):
    deps = [os.path.join(CSRC, u) for u, _ in HIP_UNITS] + [os.path.join(CSRC, "elpasso_impl.h"), os.path.join(CSRC, "elp"),
                                                           os.path.join(HERE, "..", "include")]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest(deps):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objdir = os.path.abspath(os.path.join(HERE, "..", "build", "obj"))
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for unit, flags in HIP_UNITS:
        obj = os.path.join(objdir, unit.replace(".hip", ".o"))
        objs.append(obj)
        extra = os.environ.get("ELP_EXTRA_FLAGS_" + unit.split(".")[0].upper(), "").split()     # experiments only
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + flags + extra + ["-c", "-o", obj, os.path.join(CSRC, unit)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


HOST_SOURCES = ["elp_mcl_compat.cc", "ps-encoding.cc", "elp_key.cc", "ps-signer.cc", "ps-requester.cc", "ps-verifier.cc", "host_capi.cc"]


def build_host(force=False, verbose=False):
    """Host C++ protocol layer (PSSigner / PSRequester / PSVerifier over the C-ABI) -> libelpasso_host.so."""
    build_hip(force=False, verbose=verbose)
    hd = os.path.join(CSRC, "host")
    if not force and os.path.exists(HOST_LIB) and os.path.getmtime(HOST_LIB) >= max(_newest([hd, os.path.join(HERE, "..", "include")]), os.path.getmtime(LIB)):
        return HOST_LIB
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-I", hd, "-o", HOST_LIB] + [os.path.join(hd, f) for f in HOST_SOURCES] + \
          ["-L", CSRC, "-lelpasso_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return HOST_LIB


def build_cpp_tests(verbose=False):
    """tests/cpp/ps_tests.cc -> build/ps_tests (asserting counterpart of the reference's test/ps-tests.cc)."""
    build_host(verbose=verbose)
    root = os.path.abspath(os.path.join(HERE, ".."))
    out = os.path.join(root, "build", "ps_tests")
    src = os.path.join(root, "tests", "cpp", "ps_tests.cc")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(HOST_LIB)):
        return out
    cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(CSRC, "host"), "-o", out, src, "-L", CSRC, "-lelpasso_host", "-lelpasso_hip",
           "-Wl,-rpath," + CSRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
