#!/bin/bash
# Shader clock under load: GRBM_GUI_ACTIVE (cycles the GPU was busy) over the dispatch duration, per batch size and layout, from one counter pass
# over tools/probes/scale_probe.py (run through gpurun).  Output: gpurun_out/pmc_clock/summary.txt
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_clock
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_clock/a -o r -- python3 $R/tools/probes/scale_probe.py 20 > $R/gpurun_out/pmc_clock/probe.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/pmc_clock/a/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(fs[0])) if "k_verify_id" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    key = (r["Dispatch_Id"], r["Kernel_Name"][:40], r["Grid_Size"])
    d = by.setdefault(key, {"dur_ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
with open("gpurun_out/pmc_clock/summary.txt", "w") as f:
    for (did, k, g), d in by.items():
        gui = d.get("GRBM_GUI_ACTIVE", 0.0)
        line = "%-42s grid=%7s dur=%8.3f ms  GRBM_GUI_ACTIVE=%.4g  -> %.3f GHz  SQ_WAVES=%d" % (k, g, d["dur_ms"], gui, gui / d["dur_ms"] / 1e6, d.get("SQ_WAVES", 0))
        print(line)
        f.write(line + "\n")
PY
