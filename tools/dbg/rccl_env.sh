#!/bin/bash
# bench.py with and without an RCCL communicator in the process (one rank: IMS_BENCH_RCCL_ONE_RANK=1), every config
run() { echo "== $*"; env "$@" timeout 400 python3 bench.py --no-cpu-baseline --no-cold --steps 3 2>/dev/null | python3 -c "import sys,json; L=sys.stdin.read().strip().splitlines(); d=json.loads(L[-1]); print(len(L), d['config']['workload'][:4], d['ms_per_step'], d.get('rccl_one_rank'))"; }
for c in c2 c3 c3b c4 c5 fft; do
run IMSIM_BENCH_CONFIG=$c
run IMSIM_BENCH_CONFIG=$c IMS_BENCH_RCCL_ONE_RANK=1
done
