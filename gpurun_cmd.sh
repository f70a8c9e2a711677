timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py --steps 5 --warmup 1 > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; tail -c 300 gpurun_out/bench_now.err
