#!/bin/bash
# A/B counter passes of the headline kernel in both layouts (run through gpurun): ELP_LAYOUT=paired|plain.  Separate --pmc passes
# (FETCH_SIZE / WRITE_SIZE / SQ counters cannot share one on gfx950), kernel-trace only.  Output: gpurun_out/pmc_ab/<layout>_{a,b,c}/ + a summary.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_ab
cd /tmp && export TMPDIR=/tmp
for LAYOUT in paired plain; do
  export ELP_LAYOUT=$LAYOUT
  rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_ab/${LAYOUT}_a -o r -- python3 $R/bench.py --steps 2 --warmup 1 --headline-only > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_ab/${LAYOUT}_b -o r -- python3 $R/bench.py --steps 2 --warmup 1 --headline-only > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_ab/${LAYOUT}_c -o r -- python3 $R/bench.py --steps 2 --warmup 1 --headline-only > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, collections, glob, json
out = {}
for layout in ("paired", "plain"):
    res = {}
    for sub in "abc":
        fs = glob.glob("gpurun_out/pmc_ab/%s_%s/**/*counter_collection.csv" % (layout, sub), recursive=True)
        if not fs:
            continue
        rows = [r for r in csv.DictReader(open(fs[0])) if "k_verify_id" in r["Kernel_Name"]]
        agg = collections.defaultdict(list)
        for r in rows:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            res[k] = sum(v) / len(v)
        if rows:
            r = rows[0]
            res.update({"kernel": r["Kernel_Name"][:60], "vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "scratch": r["Scratch_Size"], "lds": r.get("LDS_Block_Size"),
                        "grid": r["Grid_Size"], "dur_ms_%s" % sub: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
    out[layout] = res
json.dump(out, open("gpurun_out/pmc_ab/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
