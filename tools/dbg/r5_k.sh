#!/bin/bash
# round 5 profiles: C3 (bench line, kernel stats, HBM + SQ counters), C5 (bench line at 189 CCDs, kernel stats at 64), cold start
R=$PWD
mkdir -p $R/gpurun_out
ulimit -c 0
export HSA_ENABLE_COREDUMP=0
bash tools/profile_round.sh round5 c3 > $R/gpurun_out/r5k_c3.log 2>&1; tail -25 $R/gpurun_out/r5k_c3.log | cut -c1-200
python3 bench.py --config c5 --steps 5 --warmup 2 > $R/gpurun_out/round5_c5_bench.json 2> $R/gpurun_out/round5_c5_bench.err; cut -c1-600 $R/gpurun_out/round5_c5_bench.json
cd /tmp && export TMPDIR=/tmp
IMS_C5_CCDS=64 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/round5_c5_kt -- python3 $R/bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-cold > $R/gpurun_out/round5_c5_kt.log 2>&1
cd $R
DB=$(find $R/gpurun_out/round5_c5_kt -name "*.db" | head -1)
python3 tools/rocprof_summary.py $DB $R/gpurun_out/round5_c5_kernel_stats.txt > /dev/null; head -24 $R/gpurun_out/round5_c5_kernel_stats.txt | cut -c1-150
rm -rf $R/gpurun_out/round5_c5_kt
timeout 300 python3 tools/dbg/r5_cold.py 189 2>&1 | grep -v amdgpu.ids | head -8 | tee $R/gpurun_out/round5_c5_cold.log
python3 bench.py --config fft --no-cpu-allcore > $R/gpurun_out/round5_fft_bench.json 2>/dev/null; cut -c1-300 $R/gpurun_out/round5_fft_bench.json
