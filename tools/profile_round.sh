#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun): kernel-trace stats of bench.py and separate
# PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters cannot share a pass on gfx950).  Output under gpurun_out/prof_<tag>/.
TAG=${1:-r02}
WIN=${2:-20}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/trace -o $TAG -- python3 $R/bench.py --steps 5 --warmup 1 --headline-only --window $WIN > $R/gpurun_out/prof_$TAG/bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_a -o $TAG -- python3 $R/bench.py --steps 2 --warmup 1 --headline-only --window $WIN > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_b -o $TAG -- python3 $R/bench.py --steps 2 --warmup 1 --headline-only --window $WIN > /dev/null 2>&1
cd $R
python3 bench.py --steps 10 --warmup 2 --window $WIN > gpurun_out/prof_$TAG/bench_n1.json 2> /dev/null
python3 tools/summarize_profile.py gpurun_out/prof_$TAG $TAG $WIN
