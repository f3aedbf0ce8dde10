#!/bin/bash
for M in 1000000000 20000 8192 4096; do
  echo "C3 through the joint runner, lists above $M tiles"; env IMS_PLAN_LISTS=1 IMS_JOINT_LIST_MIN=$M python3 bench.py --no-cpu-baseline --no-cold 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['roofline'].get('kernel'))"
done
