#!/bin/bash
# joint top chains of a focal plane: C5 whole, joint batches of 8 / 4 / tile marks against a chain per CCD
R=$PWD; T=r4j; mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  env "$@" timeout 900 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_c5_$name.json 2> gpurun_out/${T}_c5_$name.err
  python3 -c "
import json
d=json.load(open('gpurun_out/${T}_c5_$name.json')); print('$name:', round(d['ms_per_step'],1), 'ms', round(d['ms_per_step']/189,2), 'per CCD', round(d['value']))" || tail -5 gpurun_out/${T}_c5_$name.err
}
run joint8 IMS_FOCAL_JOINT=8
run joint0 IMS_FOCAL_JOINT=0
run joint8_tags IMS_FOCAL_JOINT=8 IMS_BF_TAGS=1
run joint4 IMS_FOCAL_JOINT=4
