#!/bin/bash
# Per-kernel durations of a command (rocprofv3 --kernel-trace only), grouped by kernel name and grid size, in dispatch order of first appearance.
# usage (through gpurun): tools/kernel_times.sh <tag> python3 <script> [args...]      (environment variables of the caller are inherited by the program)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kt_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o r -- "$@" > $O/probe.log 2>&1
cd $R
python3 - "$O" <<'PY'
import collections, csv, glob, sys
O = sys.argv[1]
fs = glob.glob(O + "/t/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"])) if fs else []
agg, order = collections.defaultdict(list), []
for r in rows:
    n = r["Kernel_Name"]
    if "at::native" in n or "elementwise" in n:
        continue
    key = (n.split("(")[0].replace("void ", "")[:70], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Scratch_Size", r.get("Private_Segment_Size", "?")), r.get("VGPR_Count", "?"))
    if key not in agg:
        order.append(key)
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(O + "/kernel_times.txt", "w") as out:
    for k in order:
        v = agg[k]
        line = "%-70s grid=%9s scratch=%6s vgpr=%4s  n=%3d  avg %10.1f us  min %10.1f  max %10.1f" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), min(v), max(v))
        print(line)
        out.write(line + "\n")
PY
