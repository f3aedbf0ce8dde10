#!/bin/bash
# round 4, final state: bench lines of every config, kernel traces of C3 and C5, counter passes of C3 (profile_round.sh), a TCC
# pass of C3b's fused launch.  Outputs under gpurun_out/r4fin_*; the summaries are copied into profiles/ by hand.
R=$PWD
TAG=r4fin
mkdir -p $R/gpurun_out
bash tools/profile_round.sh $TAG c3 > $R/gpurun_out/${TAG}_c3_profile.log 2>&1
for cfg in c2 c3b fft; do python3 bench.py --config $cfg > $R/gpurun_out/${TAG}_${cfg}_bench.json 2> $R/gpurun_out/${TAG}_${cfg}_bench.err; done
python3 bench.py --config c4 --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_c4_bench.json 2> $R/gpurun_out/${TAG}_c4_bench.err
python3 bench.py --config c5 --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_c5_bench.json 2> $R/gpurun_out/${TAG}_c5_bench.err
cd /tmp && export TMPDIR=/tmp
export IMS_C5_CCDS=24
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_c5_kt -- python3 $R/bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_c5_kt.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_c5_kt -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/${TAG}_c5_kernel_stats.txt > /dev/null
rm -rf $R/gpurun_out/${TAG}_c5_kt
unset IMS_C5_CCDS
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $R/gpurun_out/${TAG}_c3b_pmc_TCC --output-format csv -- python3 $R/bench.py --config c3b --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $R/gpurun_out/${TAG}_c3b_pmc_TCC.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_c3b_tcc_pmc.txt $R/gpurun_out/${TAG}_c3b_tcc_pmc.json $R/gpurun_out/${TAG}_c3b_pmc_TCC > /dev/null 2>&1
find $R/gpurun_out/${TAG}_c3b_pmc_TCC -name "*.csv" -delete
cd $R
for f in gpurun_out/${TAG}_*_bench.json; do python3 -c "
import json,sys
d=json.load(open('$f')); print('$f', round(d['ms_per_step'],3), round(d['value']), d['roofline'].get('kernel'), d.get('cpu_baseline',{}).get('parity',{}).get('bit_identical'), d.get('extra',{}).get('end_to_end_ms'))"; done
head -12 gpurun_out/${TAG}_c5_kernel_stats.txt
head -30 gpurun_out/${TAG}_c3b_tcc_pmc.txt
