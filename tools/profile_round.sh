#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun), all in ONE lease: the plain headline bench line first, then the
# kernel-trace stats of the same command and separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters / clock cannot share a pass on gfx950; --pmc is
# never combined with another trace domain), then the full bench line.  Output under gpurun_out/prof_<tag>/; tools/summarize_profile.py checks that the
# rocprof average of the dominant kernel agrees with the bench line of the same lease (<= 1.02 x ms_per_step) and exits non-zero otherwise.
TAG=${1:-r03}
WIN=${2:-20}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
python3 bench.py --steps 10 --warmup 2 --headline-only --window $WIN > $O/bench_headline.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o $TAG -- python3 $R/bench.py --steps 10 --warmup 2 --headline-only --window $WIN > $O/bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $O/pmc_a -o $TAG -- python3 $R/bench.py --steps 2 --warmup 1 --headline-only --window $WIN > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_b -o $TAG -- python3 $R/bench.py --steps 2 --warmup 1 --headline-only --window $WIN > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_c -o $TAG -- python3 $R/bench.py --steps 2 --warmup 1 --headline-only --window $WIN > /dev/null 2>&1
cd $R
python3 bench.py --steps 10 --warmup 2 --window $WIN > $O/bench_n1.json 2> /dev/null
python3 tools/summarize_profile.py $O $TAG $WIN
