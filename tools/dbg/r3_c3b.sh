#!/bin/bash
# round 3: C3b with the phase-screen pre-pass against the in-place gathers (quads), step time and kernel trace
R=$PWD
for v in 2 1 0; do
  IMS_SCREEN_PREPASS=$v python3 bench.py --config c3b --steps 10 --warmup 3 --no-cpu-baseline --no-cold 2>&1 | tail -1 | cut -c1-260
done
for nb in 32 256; do
  IMS_SCREEN_BUCKETS=$nb python3 bench.py --config c3b --steps 10 --warmup 3 --no-cpu-baseline --no-cold 2>&1 | tail -1 | cut -c60-260
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c3b_kt -- python3 $R/bench.py --config c3b --steps 5 --warmup 2 --no-cpu-baseline --no-cold > $R/gpurun_out/c3b_kt.log 2>&1
DB=$(find $R/gpurun_out/c3b_kt -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/c3b_prepass_kernel_stats.txt > /dev/null
head -14 $R/gpurun_out/c3b_prepass_kernel_stats.txt | cut -c1-160
rm -rf $R/gpurun_out/c3b_kt
