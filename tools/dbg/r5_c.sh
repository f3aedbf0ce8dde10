#!/bin/bash
# round 5: full GPU suite after the tuning refactor + the fixes, then C5 variants of the pipeline, then the cold profile
R=$PWD
mkdir -p $R/gpurun_out
ulimit -c 0
export HSA_ENABLE_COREDUMP=0
N=${1:-64}
run() {   # tag, env...
  tag=$1; shift
  env "$@" timeout 150 python3 tools/dbg/r5_c5.py $N > $R/gpurun_out/r5c_$tag.log 2>&1
  echo "== $tag: rc $?"
  grep -v "amdgpu.ids" $R/gpurun_out/r5c_$tag.log | cut -c1-330 | tail -6
}
timeout 1500 python3 -m pytest tests -m gpu -x -q > $R/gpurun_out/r5c_tests.log 2>&1; tail -4 $R/gpurun_out/r5c_tests.log
run a1l3 IMS_FOCAL_ARENA=1 IMS_FOCAL_ALIVE=3
run a0l3 IMS_FOCAL_ARENA=0 IMS_FOCAL_ALIVE=3
run a1l2 IMS_FOCAL_ARENA=1 IMS_FOCAL_ALIVE=2
run a1l3j32 IMS_FOCAL_JOINT=32
run a1l3p2 IMS_FOCAL_AHEAD=pre:2
run a1l3j32p2 IMS_FOCAL_JOINT=32 IMS_FOCAL_AHEAD=pre:2
run a1l4j32 IMS_FOCAL_JOINT=32 IMS_FOCAL_ALIVE=4
timeout 200 python3 tools/dbg/r5_cold.py 48 > $R/gpurun_out/r5c_cold.log 2>&1; grep -v amdgpu.ids $R/gpurun_out/r5c_cold.log | cut -c1-200 | head -90
