#!/bin/bash
# Start / end timestamps of the dispatches of a probe (rocprofv3 --kernel-trace only): do kernels of different streams overlap?
# usage (through gpurun): tools/trace_overlap.sh <tag> <python script> [args...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o r -- python3 "$@" > $O/probe.log 2>&1
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_vid" in r["Kernel_Name"] or "k_verify_id" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
with open(O + "/overlap.txt", "w") as out:
    for r in rows[-40:]:
        line = "%-28s q=%s grid=%7s start=%10.3f ms end=%10.3f ms dur=%8.3f" % (r["Kernel_Name"].split("<")[0][:28], r.get("Queue_Id", "?"), r["Grid_Size"], (int(r["Start_Timestamp"]) - t0) / 1e6,
                                                                    (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        print(line)
        out.write(line + "\n")
PY
