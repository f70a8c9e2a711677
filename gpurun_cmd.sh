W=16 python tools/probes/gpu_probe5.py 2>&1 | grep "B= *16384\|B= *65536\|B= *262144"
python bench.py --curve bls12_381 --batch 32768 --steps 3 --warmup 1 --headline-only 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bls', d['value'], d['roofline']['kernel_ms'], d['parity_ok'])"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -3
