timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "aggregated or user_side or golden" 2>&1 | tail -3
python bench.py --steps 5 --warmup 1 > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; tail -c 300 gpurun_out/bench_now.err
