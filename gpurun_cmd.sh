timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 5 --warmup 1 --headline-only 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'], d['parity_ok'])"
python bench.py --curve bls12_381 --batch 32768 --steps 3 --warmup 1 --headline-only 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bls', d['value'], d['roofline']['kernel_ms'], d['parity_ok'])"
