#!/bin/bash
# round 6: the one-wavefront update of sparse listed tiles -- parity, then C5 A/B by the cut
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "joint_top_chains or focal_plane" 2>&1 | tail -15
for K in 0 8 16; do
  IMS_JOINT_SPARSE_MAX=$K R5_CALLS=3 timeout 600 python3 tools/dbg/c5_full.py ${1:-64} 2>&1 | grep "call" 
done
