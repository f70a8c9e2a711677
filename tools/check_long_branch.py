"""Scanner for the code-generation hazard behind the BLS12-381 "-DELP_FP6_INLINE=1" fault (profiles/r05_bls_fault.md).

A device FUNCTION (not a kernel) returns with `s_setpc_b64 s[30:31]`.  When its body outgrows the +-128 KB reach of s_cbranch, the branch-relaxation pass expands far
branches as `s_getpc_b64 sN ; s_add_u32 ; s_addc_u32 ; s_setpc_b64 sN` with a scavenged SGPR pair -- and under scalar-register pressure ROCm 7.2.0's hipcc picks
s[30:31] itself in a LEAF function, which never saved it: the first far branch taken destroys the return address.  This script reads hipcc assembly listings
(-save-temps) or `llvm-objdump -d` output and reports every function that (a) is not a kernel, (b) writes s[30:31] with s_getpc_b64 and (c) has not saved s30
BEFORE the first such write (v_writelane_b32 ..., s30, ... / s_mov_b64 ..., s[30:31] / a scalar or vector store of s[30:31] to memory): a save after the clobbering
instruction would store the clobbered value.
    python tools/check_long_branch.py <file.s | objdump.txt> [...]      exit code 1 if a hazardous function is found."""
import re
import sys


def scan(path):
    bad = []
    name, lines = None, []

    def flush():
        if name is None:
            return
        body = "\n".join(lines)
        if ("s_endpgm" in body) or not re.search(r"s_setpc_b64 s\[30:31\]", body):
            return                                     # a kernel, or no return through s[30:31]
        far = len(re.findall(r"s_getpc_b64 s\[30:31\]", body))
        first = re.search(r"s_getpc_b64 s\[30:31\]", body)
        # the return address counts as saved only if the save precedes the first clobbering s_getpc_b64 (spills through memory included)
        saved = first and re.search(r"v_writelane_b32 v\d+, s30\b|s_mov_b64 s\[\d+:\d+\], s\[30:31\]|s_store_dwordx2 s\[30:31\]|s_mov_b32 s\d+, s30\b",
                                    body[:first.start()])
        if far and not saved:
            bad.append((name, far, len(body.encode())))

    with open(path, errors="replace") as f:
        for ln in f:
            m = re.match(r"^([A-Za-z_.$][\w.$]*):\s*(;.*)?$", ln) if not ln.startswith((" ", "\t")) else None      # assembly label at column 0
            m2 = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)                                                              # objdump symbol header
            lab = m.group(1) if m else (m2.group(1) if m2 else None)
            if lab and not lab.startswith((".L", "$")):
                flush()
                name, lines = lab, []
            elif name is not None:
                lines.append(ln)
    flush()
    return bad


def llvm_objdump():
    import os
    import shutil
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root and os.path.exists(os.path.join(root, "lib", "llvm", "bin", "llvm-objdump")):
            return os.path.join(root, "lib", "llvm", "bin", "llvm-objdump")
    hipcc = shutil.which("hipcc")
    if hipcc:
        cand = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin", "llvm-objdump")
        if os.path.exists(cand):
            return cand
    return shutil.which("llvm-objdump") or "llvm-objdump"


def extract_code_objects(lib, outdir):
    """The gfx950 code objects of a HIP fat binary (clang offload bundles: magic, entry table of (offset, size, triple)), written to outdir."""
    import os
    import struct
    data = open(lib, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], 0
    while True:
        pos = data.find(magic, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        q = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode(errors="replace")
            q += 24 + tl
            if "gfx950" in triple and size:
                path = os.path.join(outdir, "co_%d.o" % len(out))
                with open(path, "wb") as f:
                    f.write(data[pos + off:pos + off + size])
                out.append(path)
        pos += 24
    return out


def scan_library(lib):
    """Disassembles every gfx950 code object of the library and scans it; returns (hazards, number of code objects)."""
    import os
    import subprocess
    import tempfile
    bad = []
    with tempfile.TemporaryDirectory() as td:
        cos = extract_code_objects(lib, td)
        for co in cos:
            txt = co + ".txt"
            with open(txt, "w") as f:
                subprocess.check_call([llvm_objdump(), "-d", "--no-show-raw-insn", co], stdout=f)
            bad += [(os.path.basename(co),) + b for b in scan(txt)]
            os.remove(txt)
    return bad, len(cos)


def main():
    rc = 0
    if len(sys.argv) > 2 and sys.argv[1] == "--lib":
        bad, n = scan_library(sys.argv[2])
        for co, name, far, size in bad:
            print("%s [%s]: HAZARD in %s: %d far branches through s[30:31], return address never saved" % (sys.argv[2], co, name, far))
        print("%d gfx950 code object(s) scanned, %d hazardous function(s)" % (n, len(bad)))
        return 1 if bad or n == 0 else 0
    for p in sys.argv[1:]:
        for name, far, size in scan(p):
            print("%s: HAZARD in %s: %d far branches through s[30:31], return address never saved" % (p, name, far))
            rc = 1
    if rc == 0:
        print("no function clobbers its return address with a far branch (%d file(s) scanned)" % len(sys.argv[1:]))
    return rc


if __name__ == "__main__":
    sys.exit(main())
