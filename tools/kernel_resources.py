"""Registers, private memory (scratch) and spills of every kernel in a hipcc assembly listing (-save-temps) or a code object's notes:
python tools/kernel_resources.py <file.s> [name filter]"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def main():
    txt = open(sys.argv[1], errors="replace").read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    meta = txt[txt.rfind("amdhsa.kernels:"):]
    rows = []
    for blk in re.split(r"\n  - ", meta)[1:]:
        f = dict(re.findall(r"\.(\w+):\s+(\S+)", blk))
        name = f.get("name", "?")
        if flt and flt not in name:
            continue
        rows.append((demangle(name), f))
    print("%-110s %5s %5s %8s %6s %6s %6s" % ("kernel", "vgpr", "agpr", "scratchB", "vspill", "sspill", "ldsB"))
    for name, f in rows:
        short = re.sub(r"\(.*$", "", name)[:110]
        print("%-110s %5s %5s %8s %6s %6s %6s" % (short, f.get("vgpr_count"), f.get("agpr_count"), f.get("private_segment_fixed_size"),
                                                f.get("vgpr_spill_count"), f.get("sgpr_spill_count"), f.get("group_segment_fixed_size")))


main()
