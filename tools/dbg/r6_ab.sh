#!/bin/bash
# C5 step over n CCDs for value combinations of knobs: bash tools/dbg/r6_ab.sh <n_ccd> "<ENV=V ENV2=V2>" "<...>" ...
N=${1:-64}; shift
for SET in "$@"; do
  env $SET R5_CALLS=${R6_CALLS:-4} timeout 900 python3 tools/dbg/c5_full.py $N 2>&1 | grep "call [1-9]" | sed "s/^/{$SET} /" | cut -c1-200
done
