#!/bin/bash
# Collect the judged measurements on the GPU box: bench lines, rocprofv3 kernel stats and the HBM
# PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, no other trace domain).
#   gpurun -- 'bash tools/profile_round.sh <tag> [config]'
# Outputs go to gpurun_out/<tag>_*; copy the summaries you keep into profiles/.
TAG=${1:-round}
CFG=${2:-c3}
R=$PWD
mkdir -p $R/gpurun_out
python3 $R/bench.py --config $CFG > $R/gpurun_out/${TAG}_${CFG}_bench.json 2> $R/gpurun_out/${TAG}_${CFG}_bench.err
cat $R/gpurun_out/${TAG}_${CFG}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_${CFG}_kt -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-cold > $R/gpurun_out/${TAG}_${CFG}_kt.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_${CFG}_kt -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/${TAG}_${CFG}_kernel_stats.txt > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/${TAG}_${CFG}_pmc_$C --output-format csv -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $R/gpurun_out/${TAG}_${CFG}_pmc_$C.log 2>&1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_${CFG}_hbm_pmc.txt $R/gpurun_out/${TAG}_${CFG}_hbm_pmc.json $R/gpurun_out/${TAG}_${CFG}_pmc_FETCH_SIZE $R/gpurun_out/${TAG}_${CFG}_pmc_WRITE_SIZE > /dev/null
# SQ counters of the same command (one pass, 8 SQ slots): VALU instructions per wave and issue / wait cycles
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES -d $R/gpurun_out/${TAG}_${CFG}_pmc_SQ --output-format csv -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $R/gpurun_out/${TAG}_${CFG}_pmc_SQ.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_${CFG}_sq_pmc.txt $R/gpurun_out/${TAG}_${CFG}_sq_pmc.json $R/gpurun_out/${TAG}_${CFG}_pmc_SQ > /dev/null
find $R/gpurun_out/${TAG}_${CFG}_pmc_* -name "*.csv" -delete
rm -rf $R/gpurun_out/${TAG}_${CFG}_kt
head -20 $R/gpurun_out/${TAG}_${CFG}_kernel_stats.txt
cat $R/gpurun_out/${TAG}_${CFG}_hbm_pmc.txt | head -20
grep "k_shoot" $R/gpurun_out/${TAG}_${CFG}_sq_pmc.txt | head -20
