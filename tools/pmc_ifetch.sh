# PMC passes for the instruction-fetch question (k_verify_id): I-cache requests / hits / misses, VALU / SALU instruction counts.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ --output-format csv -d $R/gpurun_out/pmcI -o r -- python3 $R/bench.py --steps 2 --warmup 1 --headline-only > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmcJ -o r -- python3 $R/bench.py --steps 2 --warmup 1 --headline-only > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmcK -o r -- python3 $R/bench.py --steps 2 --warmup 1 --headline-only > /dev/null 2>&1
cd $R; python3 - <<PY
import csv,collections,glob
for name in ("pmcI","pmcJ","pmcK"):
    fs=glob.glob(f"gpurun_out/{name}/**/*counter_collection.csv", recursive=True)
    if not fs: print(name,"no output"); continue
    rows=list(csv.DictReader(open(fs[0])))
    agg=collections.defaultdict(list)
    for r in rows:
        if "k_verify_id<elp::BN254>" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(name)
    for k,v in agg.items(): print("  ",k,"%.5g"%(sum(v)/len(v)), "n=%d"%len(v))
PY
