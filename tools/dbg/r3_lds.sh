#!/bin/bash
# photon kernels at three / two workgroups per CU (IMS_PHOTON_LDS) for the phase-screen PSF (C3b)
for rep in 1 2; do
for lds in 0 41984 56000; do
  for cfg in c3b; do
    echo "== IMS_PHOTON_LDS=$lds $cfg"
    IMS_PHOTON_LDS=$lds python3 bench.py --config $cfg --no-cpu-baseline --no-cold 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
done
