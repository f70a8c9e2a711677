set -x
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "user_side or prover_to_verifier or provide_id" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_bls.py -x -q -m gpu -k "protocol_flows" 2>&1 | tail -15
