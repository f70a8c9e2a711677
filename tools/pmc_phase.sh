#!/bin/bash
# Memory-side counters of the pairing half alone (k_ps_verify, A = 8) and of the whole verification (k_verify_id), light and full load, from
# tools/probes/phase_probe.py (run through gpurun).  Two --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share one on gfx950).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_phase
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_FLAT --output-format csv -d $R/gpurun_out/pmc_phase/a -o r -- python3 $R/tools/probes/phase_probe.py 20 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_phase/b -o r -- python3 $R/tools/probes/phase_probe.py 20 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
out = collections.OrderedDict()
for sub in "ab":
    fs = glob.glob("gpurun_out/pmc_phase/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        if "k_ps_verify" not in r["Kernel_Name"] and "k_verify_id" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"][:22], int(r["Grid_Size"]))
        d = out.setdefault(key, collections.defaultdict(list))
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
        d["dur_ms_" + sub].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open("gpurun_out/pmc_phase/summary.txt", "w") as f:
    for key, d in out.items():
        # counters are reported once per dispatch and counter; average over the dispatches of this size
        nd = {}
        for k, v in d.items():
            nd[k] = sum(v) / len(v)
        line = "%-24s grid=%7d  " % key + "  ".join("%s=%.4g" % (k, nd[k]) for k in sorted(nd))
        print(line)
        f.write(line + "\n")
PY
