#!/bin/bash
# round 3, probe 2: per-round time of the brightest star against the size of its private region
for st in 0 64 128 256; do
  if [ $st = 0 ]; then unset STAMP; else export STAMP=$st; fi
  echo "STAMP=$st"; python3 tools/dbg/one_star.py 2>&1 | grep -v "^$" | tail -3
  bash tools/dbg/one_star.sh > /dev/null 2>&1; head -5 gpurun_out/star_kernel_stats.txt | tail -3 | cut -c1-150
done
