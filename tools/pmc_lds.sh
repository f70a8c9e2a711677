#!/bin/bash
# LDS / issue counters per kernel of any probe command (run through gpurun):  tools/pmc_lds.sh <tag> <python script> [args...]
# Two --pmc passes next to a --kernel-trace (never combined with other trace domains); output gpurun_out/pmc_<tag>/lds_{a,b}/ + a per-kernel digest on stdout.
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/lds_a -o r -- python3 "$@" > $O/probe_lds.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES --output-format csv -d $O/lds_b -o r -- python3 "$@" > /dev/null 2>&1
cd $R
python3 - "$O" <<'PY'
import csv, collections, glob, json, sys
O = sys.argv[1]
out = collections.OrderedDict()
for sub in ("lds_a", "lds_b"):
    fs = glob.glob("%s/%s/**/*counter_collection.csv" % (O, sub), recursive=True)
    if not fs:
        print("no counters for pass", sub)
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:70] + "|grid=" + r["Grid_Size"]
        d = per[k][r["Dispatch_Id"]]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["dur_ms_" + sub] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, disp in per.items():
        e = out.setdefault(k, {})
        keys = set()
        for d in disp.values():
            keys |= set(d)
        for c in keys:
            vals = [d[c] for d in disp.values() if c in d]
            e[c] = sum(vals) / len(vals)
json.dump(out, open(O + "/lds_summary.json", "w"), indent=1)
for k, e in out.items():
    print(k, json.dumps({a: round(b, 3) for a, b in sorted(e.items())}))
PY
