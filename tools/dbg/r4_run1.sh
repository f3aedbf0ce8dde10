timeout 900 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "joint_top_chains or focal_plane_ccds or config_several or fft or native_planner" 2>&1 | tail -4
python3 tools/dbg/r4_joint_host.py 24 2>&1 | cut -c1-170 | head -40
