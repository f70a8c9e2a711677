#!/bin/bash
# Instruction-fetch counters per kernel of any probe command (run through gpurun):  tools/pmc_icache_kernels.sh <tag> python3 <abs script> [args...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmci_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ --output-format csv -d $O/a -o r -- "$@" > $O/probe.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $O/b -o r -- "$@" > /dev/null 2>&1
cd $R
python3 - "$O" <<'PY'
import csv, collections, glob, json, sys
O = sys.argv[1]
for sub in "ab":
    fs = glob.glob("%s/%s/**/*counter_collection.csv" % (O, sub), recursive=True)
    if not fs:
        continue
    per = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        if not any(k in n for k in ("k_vid", "k_pair", "k_ps_k", "k_verify")):
            continue
        k = n.split("(")[0].replace("void ", "")[:44] + " grid=" + r["Grid_Size"] + " wg=" + r["Workgroup_Size"]
        d = per.setdefault(k, collections.defaultdict(dict))[r["Dispatch_Id"]]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, disp in per.items():
        keys = sorted(set(c for d in disp.values() for c in d))
        avg = {c: sum(d[c] for d in disp.values() if c in d) / len(disp) for c in keys}
        print(k, json.dumps({c: (round(v, 3) if v < 1e4 else int(v)) for c, v in avg.items()}))
PY
