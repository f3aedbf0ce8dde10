#!/bin/bash
R=$PWD
drv() { n=$1; shift; env R4_SKIP_SINGLE=1 R4_CONC=4 "$@" python3 tools/dbg/r4_c5.py $n 2>&1 | grep "concurrent\|Error\|error" | head -3; }
for A in pre:0 pre:1 pre:2 bulk:1 bulk:2 mid:1 mid:2; do echo "joint 8, ahead $A"; drv 189 IMS_FOCAL_JOINT=8 IMS_FOCAL_AHEAD=$A; done
echo "joint 16, ahead pre:1"; drv 189 IMS_FOCAL_JOINT=16 IMS_FOCAL_AHEAD=pre:1
echo "joint 8, sync uploads, ahead off"; drv 189 IMS_FOCAL_JOINT=8 IMS_UPLOAD_SYNC=1 IMS_FOCAL_AHEAD=pre:0
