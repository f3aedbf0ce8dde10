#!/bin/bash
# end-of-round passes of the configs other than C3: bash tools/dbg/round_rest.sh <tag>
TAG=${1:-round}
R=$PWD
bash tools/profile_round.sh $TAG c3b > gpurun_out/${TAG}_c3b.log 2>&1
for CFG in c2 c4 c5 fft; do
  S=5; [ $CFG = c4 ] && S=3; [ $CFG = c5 ] && S=1; [ $CFG = fft ] && S=20
  timeout 900 python3 bench.py --config $CFG --steps $S --warmup 1 > gpurun_out/${TAG}_${CFG}_bench.json 2> gpurun_out/${TAG}_${CFG}_bench.err
  tail -c 600 gpurun_out/${TAG}_${CFG}_bench.json | cut -c1-300
done
for CFG in c4 fft; do bash tools/dbg/kt.sh $TAG $CFG > /dev/null 2>&1; done
grep -h ms_per_step gpurun_out/${TAG}_*_bench.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config']['workload'][:40], d['ms_per_step'], d['value'])"
