"""Build the HIP shared library in-tree (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libelpasso_hip.so")
HOST_LIB = os.path.join(CSRC, "libelpasso_host.so")


def _digest(paths, extra=()):
    """sha256 over the names and contents of every source file under `paths` (+ the flag strings in `extra`): the rebuild key.  Modification
    times are not trusted: a tree copied to another machine can carry a stale .so that is newer than its sources."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for p in paths:
        if os.path.isdir(p):
            for dp, _, fs in os.walk(p):
                files += [os.path.join(dp, f) for f in fs if f.endswith((".h", ".hip", ".cc", ".cpp"))]
        elif os.path.exists(p):
            files.append(p)
    for f in sorted(os.path.abspath(x) for x in files):
        h.update(os.path.relpath(f, HERE).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


def _stamp_ok(target, digest):
    try:
        return os.path.exists(target) and open(target + ".srchash").read().strip() == digest
    except OSError:
        return False


def _stamp(target, digest):
    with open(target + ".srchash", "w") as f:
        f.write(digest + "\n")


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
# functions (+13 % on the verify kernel).  The two-lane BLS12-381 unit takes the same flag since round 5 (el_passo_verify_id 51.8 -> 49.4 ms, PS verification 33.0 -> 29.9 ms per
# 65 536) together with -DELP_NONLEAF_GUARD=1: its Miller loop becomes one 730 KB function, and a LEAF function of that size trips the branch-relaxation
# bug of profiles/r05_bls_fault.md -- the guard makes it a caller.  tests/test_isa_hazards.py scans the built library for the pattern.  With the inlined routines the
# unit is also better off with ONE wave per SIMD and 512 registers (-DELP_PAIR_WAVES=1: 50.0 -> 48.6 ms, PS 30.3 -> 27.5 ms; round 3 had measured the opposite for the
# build with calls); the BN254 two-lane unit is not (98 304 items 25.9 -> 29.5 ms; profiles/r05_bls_fp6_inline_ab.log).
HIP_UNITS = [("elpasso_capi.hip", []), ("elpasso_bn254.hip", ["-DELP_FP6_INLINE=1"]), ("elpasso_bn254_pair.hip", ["-DELP_FP6_INLINE=1"]),
             ("elpasso_bn254_nizk.hip", ["-DELP_FP6_INLINE=1"]), ("elpasso_bn254_g2job.hip", ["-DELP_FP6_INLINE=1"]),
             ("elpasso_bn254_g1job.hip", ["-DELP_FP6_INLINE=1"]), ("elpasso_bn254_stage.hip", ["-DELP_FP6_INLINE=1"]), ("elpasso_bn254_coop.hip", []),
             ("elpasso_bls12_381.hip", []), ("elpasso_bls12_381_pair.hip", ["-DELP_FP6_INLINE=1", "-DELP_NONLEAF_GUARD=1", "-DELP_PAIR_WAVES=1"]), ("elpasso_bls12_381_g1jobs.hip", []), ("elpasso_bn254_g1jobs.hip", []),
             ("elpasso_bls12_381_coop.hip", []), ("elpasso_bls12_381_nizk.hip", []), ("elpasso_bn254_small2.hip", ["-DELP_WAVES_PER_EU=2"]),
             ("elpasso_bn254_pair4.hip", ["-DELP_FP6_INLINE=1"]), ("elpasso_bls12_381_pair4.hip", ["-DELP_FP6_INLINE=1", "-DELP_NONLEAF_GUARD=1"]),
             ("elpasso_bn254_pair16.hip", []), ("elpasso_bls12_381_pair16.hip", []), ("elpasso_bn254_msm.hip", ["-DELP_FP6_INLINE=1"]), ("elpasso_bls12_381_msm.hip", [])]


def build_hip(force=False, verbose=False):
    deps = [os.path.join(CSRC, u) for u, _ in HIP_UNITS] + [os.path.join(CSRC, "elpasso_impl.h"), os.path.join(CSRC, "elpasso_pair4.h"), os.path.join(CSRC, "elpasso_pair16.h"), os.path.join(CSRC, "elpasso_pair16_prog.h"), os.path.join(CSRC, "elpasso_pair16_prog_bls12_381.h"), os.path.join(CSRC, "elp"),
                                                           os.path.join(HERE, "..", "include")]
    # rebuild key = hash of the sources and flags (per unit: the shared headers + that unit's own file), not modification times
    shared = [os.path.join(CSRC, "elpasso_impl.h"), os.path.join(CSRC, "elp"), os.path.join(HERE, "..", "include")]
    unit_headers = {"elpasso_bn254_pair4.hip": ["elpasso_pair4.h"], "elpasso_bls12_381_pair4.hip": ["elpasso_pair4.h"], "elpasso_bn254_pair16.hip": ["elpasso_pair16.h", "elpasso_pair16_prog.h"],
                    "elpasso_bls12_381_pair16.hip": ["elpasso_pair16.h", "elpasso_pair16_prog_bls12_381.h"]}      # headers only these units include
    unit_extra = {u: os.environ.get("ELP_EXTRA_FLAGS_" + u.split(".")[0].upper(), "").split() for u, _ in HIP_UNITS}     # experiments only
    lib_digest = _digest(deps, [(u, f, unit_extra[u]) for u, f in HIP_UNITS])
    if not force and _stamp_ok(LIB, lib_digest):
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
        extra = unit_extra[unit]
        udig = _digest(shared + [os.path.join(CSRC, unit)] + [os.path.join(CSRC, h) for h in unit_headers.get(unit, [])], [flags, extra])
        if not force and _stamp_ok(obj, udig):
            continue                                   # this unit's sources and flags are unchanged: keep its object
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + flags + extra + ["-c", "-o", obj, os.path.join(CSRC, unit)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd), obj, udig))
    for cmd, p, obj, udig in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        _stamp(obj, udig)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    _stamp(LIB, lib_digest)
    return LIB


HOST_SOURCES = ["elp_mcl_compat.cc", "ps-encoding.cc", "elp_key.cc", "ps-signer.cc", "ps-requester.cc", "ps-verifier.cc", "host_capi.cc"]


def build_host(force=False, verbose=False):
    """Host C++ protocol layer (PSSigner / PSRequester / PSVerifier over the C-ABI) -> libelpasso_host.so."""
    build_hip(force=False, verbose=verbose)
    hd = os.path.join(CSRC, "host")
    host_digest = _digest([hd, os.path.join(HERE, "..", "include")], [open(LIB + ".srchash").read() if os.path.exists(LIB + ".srchash") else ""])
    if not force and _stamp_ok(HOST_LIB, host_digest):
        return HOST_LIB
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-I", hd, "-o", HOST_LIB] + [os.path.join(hd, f) for f in HOST_SOURCES] + \
          ["-L", CSRC, "-lelpasso_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    _stamp(HOST_LIB, host_digest)
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
