#!/bin/bash
# round 5, first measurements of the sensor arena and the pipeline: parity tests, malloc cost, C5 on 64 CCDs in several forms
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python3 -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "focal or joint or config_several" > $R/gpurun_out/r5a_tests.log 2>&1; tail -5 $R/gpurun_out/r5a_tests.log
python3 tools/dbg/r5_malloc.py > $R/gpurun_out/r5a_malloc.log 2>&1; cat $R/gpurun_out/r5a_malloc.log
hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_rate tools/dbg/gather_rate.hip 2>/dev/null && /tmp/gather_rate 1.6 > $R/gpurun_out/r5a_gather_rate.log 2>&1; cat $R/gpurun_out/r5a_gather_rate.log
/tmp/gather_rate 6.4 > $R/gpurun_out/r5a_gather_rate_6g.log 2>&1; cat $R/gpurun_out/r5a_gather_rate_6g.log
for V in "IMS_FOCAL_ARENA=0 IMS_FOCAL_ALIVE=2" "IMS_FOCAL_ARENA=1 IMS_FOCAL_ALIVE=2" "IMS_FOCAL_ARENA=1 IMS_FOCAL_ALIVE=3" "IMS_FOCAL_ARENA=1 IMS_FOCAL_ALIVE=3 IMS_FOCAL_JOINT=32" "IMS_FOCAL_ARENA=1 IMS_FOCAL_ALIVE=3 IMS_FOCAL_AHEAD=pre:2" "IMS_FOCAL_ARENA=1 IMS_FOCAL_ALIVE=3 IMS_FOCAL_JOINT=32 IMS_FOCAL_AHEAD=pre:2"; do
  env $V timeout 600 python3 tools/dbg/r5_c5.py 64 2>&1 | grep -v amdgpu.ids | tee -a $R/gpurun_out/r5a_c5.log
done
IMS_FOCAL_ARENA=1 IMS_FOCAL_ALIVE=3 timeout 600 python3 tools/dbg/r5_c5.py 64 profile > $R/gpurun_out/r5a_profile.log 2>&1; head -80 $R/gpurun_out/r5a_profile.log
