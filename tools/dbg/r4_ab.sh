#!/bin/bash
# A/B on one box: C5 whole with the previous commit's tree (_old) and this one (chain per CCD, joint 8), steps 3 warmup 2
R=$PWD; mkdir -p gpurun_out
one() { # dir name env...
  d=$1; name=$2; shift; shift
  (cd $d && env "$@" timeout 900 python3 bench.py --config c5 --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/ab_$name.json 2> $R/gpurun_out/ab_$name.err)
  python3 -c "
import json
d=json.load(open('$R/gpurun_out/ab_$name.json')); print('$name:', round(d['ms_per_step'],1), 'ms', round(d['ms_per_step']/189,2), 'per CCD')" || tail -5 $R/gpurun_out/ab_$name.err
}
one _old old IMS_X=0
one . new_joint0 IMS_FOCAL_JOINT=0
one . new_joint8 IMS_FOCAL_JOINT=8
one _old old_again IMS_X=0
