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


def build_hip(force=False, verbose=False):
    src = os.path.join(CSRC, "elpasso_hip.hip")
    deps = [src, os.path.join(CSRC, "elp"), os.path.join(HERE, "..", "include")]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest(deps):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
