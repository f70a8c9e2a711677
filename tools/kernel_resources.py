"""Registers, private memory (scratch) and spills of every kernel in a hipcc assembly listing (-save-temps), a gfx950 code object or a hipcc
object file (the code object is taken out of its .hip_fatbin section):
python tools/kernel_resources.py <file.s | file.co | file.o> [name filter]"""
import os
import re
import subprocess
import sys
import tempfile


def llvm_tool(name):
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root and os.path.exists(os.path.join(root, "lib", "llvm", "bin", name)):
            return os.path.join(root, "lib", "llvm", "bin", name)
    return name


def code_object_notes(path):
    """The amdhsa.kernels metadata text of a code object, or of the gfx950 code object bundled in a host object file."""
    with open(path, "rb") as f:
        head = f.read(20)
    with tempfile.TemporaryDirectory() as td:
        co = path
        if head[:4] == b"\x7fELF" and head[18:20] == b"\x3e\x00":        # x86-64 host object: unbundle the device code
            fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
            subprocess.check_call([llvm_tool("llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, path])
            subprocess.check_call([llvm_tool("clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        return subprocess.run([llvm_tool("llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout


def kernel_fields(path, name_filter=""):
    """{mangled kernel name: {field: value}} from the notes of a code object / object file."""
    meta = code_object_notes(path)
    meta = meta[meta.rfind("amdhsa.kernels:"):]
    out = {}
    for blk in re.split(r"\n  - \.", "\n" + meta)[1:]:              # kernel entries sit at two spaces; argument entries deeper
        f = dict(re.findall(r"\.?(\w+):\s+(\S+)", "." + blk))
        if "name" in f and (not name_filter or name_filter in f["name"]):
            out[f["name"]] = f
    return out


def demangle(n):
    try:
        return subprocess.run([llvm_tool("llvm-cxxfilt"), n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def main():
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    if sys.argv[1].endswith((".o", ".co", ".hsaco", ".so")):
        rows = [(demangle(n), f) for n, f in kernel_fields(sys.argv[1], flt).items()]
    else:
        txt = open(sys.argv[1], errors="replace").read()
        meta = txt[txt.rfind("amdhsa.kernels:"):]
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


if __name__ == "__main__":
    main()
