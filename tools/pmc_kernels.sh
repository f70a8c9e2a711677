#!/bin/bash
# Per-kernel counters of any probe command (run through gpurun):  tools/pmc_kernels.sh <tag> <python script> [args...]
# One --kernel-trace --stats run and separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ counters cannot share a pass on gfx950; never combined
# with other trace domains).  Output: gpurun_out/pmc_<tag>/summary.json (+ the raw csv files), counters averaged per kernel name over its dispatches.
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r -- python3 "$@" > $O/probe.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $O/a -o r -- python3 "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/b -o r -- python3 "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $O/c -o r -- python3 "$@" > /dev/null 2>&1
cd $R
python3 - "$O" <<'PY'
import csv, collections, glob, json, sys
O = sys.argv[1]
out = collections.OrderedDict()
for sub in "abc":
    fs = glob.glob("%s/%s/**/*counter_collection.csv" % (O, sub), recursive=True)
    if not fs:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(dict))      # kernel -> dispatch -> counter -> value
    meta = {}
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:70] + "|grid=" + r["Grid_Size"]
        d = per[k][r["Dispatch_Id"]]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        meta[k] = {"vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "scratch": r["Scratch_Size"], "lds": r.get("LDS_Block_Size"), "wg": r["Workgroup_Size"]}
    for k, disp in per.items():
        e = out.setdefault(k, dict(meta[k]))
        e["dispatches_" + sub] = len(disp)
        keys = set()
        for d in disp.values():
            keys |= set(d)
        for c in keys:
            vals = [d[c] for d in disp.values() if c in d]
            e[(c + "_" + sub) if c == "dur_ms" else c] = sum(vals) / len(vals)
for k, e in out.items():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_corrected"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024       # MI355X_MICROARCH.md: KiB units, FETCH_SIZE halves wide reads on gfx950
    if "GRBM_GUI_ACTIVE" in e and e.get("dur_ms_c"):
        e["clock_ghz"] = e["GRBM_GUI_ACTIVE"] / 8 / e["dur_ms_c"] / 1e6       # the counter is summed over the 8 XCDs of a dispatch
    if "TCC_HIT_sum" in e:
        e["l2_hit"] = e["TCC_HIT_sum"] / max(1.0, e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
stats = glob.glob("%s/trace/**/*kernel_stats.csv" % O, recursive=True)
ks = []
if stats:
    for r in list(csv.DictReader(open(stats[0])))[:14]:
        ks.append({"name": r["Name"][:80], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6})
json.dump({"kernel_stats": ks, "pmc": out}, open(O + "/summary.json", "w"), indent=1)
for s in ks:
    print("%-82s calls=%3d avg=%9.3f min=%9.3f max=%9.3f ms" % (s["name"], s["calls"], s["avg_ms"], s["min_ms"], s["max_ms"]))
for k, e in out.items():
    if "verify" in k or "vid" in k or "agg" in k:
        print(k, json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in e.items()}))
PY
