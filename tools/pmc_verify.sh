mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmcA -o r -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --headline-only > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmcB -o r -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --headline-only > /dev/null 2>&1
cd $R; python3 - <<PY
import csv,collections
for name in ("pmcA","pmcB"):
    rows=list(csv.DictReader(open(f"gpurun_out/{name}/r_counter_collection.csv")))
    agg=collections.defaultdict(list)
    for r in rows:
        if "k_verify_id" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    kr=[r for r in rows if "k_verify_id" in r["Kernel_Name"]][0]
    print(name,"vgpr",kr["VGPR_Count"],"agpr",kr["Accum_VGPR_Count"],"scratch",kr["Scratch_Size"], "dur_ms", (int(kr["End_Timestamp"])-int(kr["Start_Timestamp"]))/1e6)
    for k,v in agg.items(): print("  ",k,"%.4g"%(sum(v)/len(v)))
PY
