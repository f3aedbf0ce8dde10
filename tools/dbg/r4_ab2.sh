#!/bin/bash
c3() { env "$@" python3 bench.py --no-cpu-baseline --no-cold 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))"; }
drv() { n=$1; shift; env R4_SKIP_SINGLE=1 R4_CONC=4 "$@" python3 tools/dbg/r4_c5.py $n 2>&1 | grep "concurrent\|Error\|error" | head -3; }
for R in 0 8 16 32 64 16:low 32:low; do echo "C3 reserve $R"; c3 IMS_RESERVE_CUS=$R; done
for R in 0 16 32 64; do echo "C5 reserve $R"; drv 189 IMS_RESERVE_CUS=$R; done
