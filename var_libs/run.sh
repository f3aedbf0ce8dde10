#!/bin/bash
# usage: run.sh <variants...> ; prints ms/step of the default bench per variant
for v in "$@"; do
  IMSIM_HIP_LIB=$PWD/var_libs/lib_$v.so timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-cold 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],2), round(d['value']))"
done
