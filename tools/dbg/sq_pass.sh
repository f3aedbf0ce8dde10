#!/bin/bash
# one SQ counter pass of a bench config: tools/dbg/sq_pass.sh <tag> <config>   (run under gpurun)
TAG=$1; CFG=$2; R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d $R/gpurun_out/${TAG}_${CFG}_pmc_SQ --output-format csv -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $R/gpurun_out/${TAG}_${CFG}_pmc_SQ.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_${CFG}_sq_pmc.txt $R/gpurun_out/${TAG}_${CFG}_sq_pmc.json $R/gpurun_out/${TAG}_${CFG}_pmc_SQ > /dev/null
find $R/gpurun_out/${TAG}_${CFG}_pmc_SQ -name "*.csv" -delete
grep "k_shoot" $R/gpurun_out/${TAG}_${CFG}_sq_pmc.txt | head -40
