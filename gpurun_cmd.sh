W=16 python tests/gpu_probe5.py 2>&1 | grep "B= *65536\|B= *262144"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bls.py -x -q -m gpu 2>&1 | tail -3
