#!/bin/bash
# a chain class of its own for the very longest chains (IMS_CHAIN_CLASSES), with and without more hardware queues
run() { echo "== $*"; env "$@" python3 bench.py --config c3 --no-cpu-baseline --no-cold 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
run IMS_CHAIN_CLASSES=40,6
run IMS_CHAIN_CLASSES=90,40,6
run IMS_CHAIN_CLASSES=120,40,6
run IMS_CHAIN_CLASSES=90,40,6 GPU_MAX_HW_QUEUES=5
run IMS_CHAIN_CLASSES=90,40,6 GPU_MAX_HW_QUEUES=8
run IMS_CHAIN_CLASSES=90,40,6 IMS_STREAM_PRIORITIES=-1,0,-1,0,0
run IMS_CHAIN_CLASSES=90,6
run IMS_CHAIN_CLASSES=120,20
