#!/bin/bash
R=$PWD
drv() { n=$1; shift; env R4_SKIP_SINGLE=1 R4_CONC=4 "$@" python3 tools/dbg/r4_c5.py $n 2>&1 | grep "concurrent\|Error\|error" | head -3; }
timeout 900 python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "joint_top_chains or focal_plane_ccds" 2>&1 | tail -3
echo "joint 8 lists"; drv 189 IMS_FOCAL_JOINT=8
echo "joint 8 no lists"; drv 189 IMS_FOCAL_JOINT=8 IMS_JOINT_LISTS=0
echo "joint 16 lists"; drv 189 IMS_FOCAL_JOINT=16
echo "joint 16 lists, fraction 0.125"; drv 189 IMS_FOCAL_JOINT=16 IMS_ACTIVE_FRACTION=0.125
echo "joint 16 lists, fraction 0.5"; drv 189 IMS_FOCAL_JOINT=16 IMS_ACTIVE_FRACTION=0.5
echo "joint 16 lists, ahead off"; drv 189 IMS_FOCAL_JOINT=16 IMS_FOCAL_AHEAD=pre:0
