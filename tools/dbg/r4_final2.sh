#!/bin/bash
# round 4, state with the joint rounds: full GPU suite, bench lines of C3 (default) and C5, kernel trace of C5 on 32 CCDs
R=$PWD; T=r4f2; mkdir -p gpurun_out
python3 -m pytest tests -m gpu -q 2>&1 | tail -4
python3 bench.py > gpurun_out/${T}_c3_bench.json 2> gpurun_out/${T}_c3_bench.err
python3 bench.py --config c5 --steps 3 --warmup 2 > gpurun_out/${T}_c5_bench.json 2> gpurun_out/${T}_c5_bench.err
IMS_FOCAL_JOINT=0 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_c5_chain_per_ccd.json 2> gpurun_out/${T}_c5_chain_per_ccd.err
for f in gpurun_out/${T}_*.json; do python3 -c "
import json
d=json.load(open('$f')); print('$f', round(d['ms_per_step'],2), round(d['value']), d['roofline'].get('kernel'), d['roofline'].get('frac'), d.get('cpu_baseline',{}).get('parity',{}).get('bit_identical'), d.get('extra',{}).get('end_to_end_ms'))"; done
cd /tmp && export TMPDIR=/tmp
export IMS_C5_CCDS=32
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_c5_kt -- python3 $R/bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${T}_c5_kt.log 2>&1
DB=$(find $R/gpurun_out/${T}_c5_kt -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/${T}_c5_kernel_stats.txt > /dev/null
rm -rf $R/gpurun_out/${T}_c5_kt
head -16 $R/gpurun_out/${T}_c5_kernel_stats.txt
