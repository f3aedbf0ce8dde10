#!/bin/bash
# round 4: head start of the top chain class (IMS_HEAD_START, default 1) against the plain order of the pool slices
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('kernel_ms_per_step'))"; }
for rep in 1 2; do for h in 0 1; do echo "== C3 head start $h (run $rep)"; IMS_HEAD_START=$h python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cold 2>/dev/null | ms; done; done
for h in 0 1; do echo "== C3b head start $h"; IMS_HEAD_START=$h python3 bench.py --config c3b --steps 10 --warmup 3 --no-cpu-baseline --no-cold 2>/dev/null | ms; done
for h in 0 1; do echo "== shard replay, head start $h"; IMS_HEAD_START=$h python3 tools/dbg/shard_times.py 2>&1 | grep world; done
export R4_SKIP_SINGLE=1 R4_CONC=4
for h in 0 1; do echo "== C5 24 CCDs head start $h"; IMS_HEAD_START=$h python3 tools/dbg/r4_c5.py 24 2>&1 | grep concurrent; done
