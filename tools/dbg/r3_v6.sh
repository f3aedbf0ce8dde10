#!/bin/bash
# spec v6 experiment: the same bench with the shipped library and with a -DIMS_V6 build (no parity legs)
for cfg in c3 c4; do
  for lib in libimsim_hip.so libimsim_hip_v6.so; do
    echo "== $cfg $lib"
    IMSIM_HIP_LIB=$PWD/imsim_amd/lib/$lib python3 bench.py --config $cfg --no-cpu-baseline --no-cold 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
