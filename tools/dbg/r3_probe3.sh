#!/bin/bash
# round 3, probe 3: all kernels of a long chain on ONE XCD (IMS_XCD_PIN = largest number of active objects pinned)
for v in 0 64; do
  export IMS_XCD_PIN=$v
  echo "IMS_XCD_PIN=$v"; python3 tools/dbg/one_star.py 2>&1 | grep "one star"
  bash tools/dbg/one_star.sh > /dev/null 2>&1; head -5 gpurun_out/star_kernel_stats.txt | tail -3 | cut -c1-150
done
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "bf or bright or silicon or lsst_image or pooling or chain or boundar" 2>&1 | tail -2
for v in 0 8 64; do
  IMS_XCD_PIN=$v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cold 2>&1 | tail -1 | cut -c1-250
done
