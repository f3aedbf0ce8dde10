#!/bin/bash
# round 3, probe 1: DPP f64 semantics / rate; update kernel with DPP operand delivery vs the SGPR form (one star, kernel trace)
R=$PWD
mkdir -p $R/gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/dbg/dpp_f64.hip -o /tmp/dpp_f64 2>/dev/null && /tmp/dpp_f64 > $R/gpurun_out/dpp_f64.txt 2>&1
cat $R/gpurun_out/dpp_f64.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "bf or bright or silicon or lsst_image or pooling or chain or boundar" 2>&1 | tail -5
for v in 0 1; do
  export IMS_UPD_DPP=$v
  python3 tools/dbg/one_star.py 2>&1 | grep "one star"
  bash tools/dbg/one_star.sh > /dev/null 2>&1
  cp $R/gpurun_out/star_kernel_stats.txt $R/gpurun_out/star_kernel_stats_dpp$v.txt
  head -8 $R/gpurun_out/star_kernel_stats_dpp$v.txt
done
unset IMS_UPD_DPP
for v in 0 1; do
  IMS_UPD_DPP=$v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cold 2>&1 | tail -1 | cut -c1-400
done
