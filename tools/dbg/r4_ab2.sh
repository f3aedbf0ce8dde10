timeout 900 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "joint_top_chains" 2>&1 | tail -15
