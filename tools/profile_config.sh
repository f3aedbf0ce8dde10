#!/bin/bash
# Every judged measurement of ONE bench config on the GPU box (round 6: every config current, same passes for all of them):
#   gpurun -- 'bash tools/profile_config.sh <tag> <config>'
#   1  the bench line as `python bench.py --config X` prints it                      -> gpurun_out/<tag>_<cfg>_bench.json
#   2  timeout ${PASS_TIMEOUT:-420} rocprofv3 --kernel-trace --stats                                              -> <tag>_<cfg>_kernel_stats.txt
#   3  HBM traffic: FETCH_SIZE and WRITE_SIZE in separate --pmc passes               -> <tag>_<cfg>_hbm_pmc.{txt,json}
#   4  SQ: instructions, waves, wave / wait cycles                                   -> <tag>_<cfg>_sq_pmc.{txt,json}
#   5  SQ: the VALU instructions by class (f64 fma / mul / add / trans, int, cvt)    -> <tag>_<cfg>_mix_pmc.{txt,json}
#   6  GRBM_GUI_ACTIVE with the kernels' durations: the sustained shader clock       -> <tag>_<cfg>_clock.json
# Counter passes carry --kernel-trace only (no other trace domain); every pass runs under `timeout` (a counter pass over the focal plane
# can hang in the profiler's dispatch hook: round 6 lost 47 GPU-minutes to one).  ONLY="sq mix" restricts the passes.  Copy what you keep into profiles/.
TAG=${1:-round}
CFG=${2:-c3}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
S=2; W=1
case $CFG in c2) S=5;; fft) S=10; W=3;; fftx|fftxs) S=1;; c5) S=1; export IMS_C5_CCDS=${IMS_C5_CCDS_PMC:-48};; esac
ARGS="--config $CFG --steps $S --warmup $W --no-cpu-baseline --no-cold --no-extra-configs"
if [ -z "$SKIP_BENCH" ]; then
  ( unset IMS_C5_CCDS; timeout 1500 python3 $R/bench.py --config $CFG --no-extra-configs > $O/${TAG}_${CFG}_bench.json 2> $O/${TAG}_${CFG}_bench.err )
  cut -c1-400 $O/${TAG}_${CFG}_bench.json
fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc_$CFG && mkdir -p /tmp/pc_$CFG
want() { [ -z "$ONLY" ] || [[ " $ONLY " == *" $1 "* ]]; }
if want kt; then
timeout ${PASS_TIMEOUT:-420} rocprofv3 --kernel-trace --stats -d /tmp/pc_$CFG/kt -- python3 $R/bench.py $ARGS > $O/${TAG}_${CFG}_kt.log 2>&1
DB=$(find /tmp/pc_$CFG/kt -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $O/${TAG}_${CFG}_kernel_stats.txt > /dev/null
fi
if want hbm; then
for C in FETCH_SIZE WRITE_SIZE; do
  timeout ${PASS_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc $C -d /tmp/pc_$CFG/pmc_$C --output-format csv -- python3 $R/bench.py $ARGS > $O/${TAG}_${CFG}_pmc_$C.log 2>&1
done
python3 $R/tools/pmc_summary.py $O/${TAG}_${CFG}_hbm_pmc.txt $O/${TAG}_${CFG}_hbm_pmc.json /tmp/pc_$CFG/pmc_FETCH_SIZE /tmp/pc_$CFG/pmc_WRITE_SIZE > /dev/null
fi
if want sq; then
timeout ${PASS_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES -d /tmp/pc_$CFG/pmc_SQ --output-format csv -- python3 $R/bench.py $ARGS > $O/${TAG}_${CFG}_pmc_SQ.log 2>&1
python3 $R/tools/pmc_summary.py $O/${TAG}_${CFG}_sq_pmc.txt $O/${TAG}_${CFG}_sq_pmc.json /tmp/pc_$CFG/pmc_SQ > /dev/null
fi
if want mix; then
timeout ${PASS_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d /tmp/pc_$CFG/pmc_MIX --output-format csv -- python3 $R/bench.py $ARGS > $O/${TAG}_${CFG}_pmc_MIX.log 2>&1
python3 $R/tools/pmc_summary.py $O/${TAG}_${CFG}_mix_pmc.txt $O/${TAG}_${CFG}_mix_pmc.json /tmp/pc_$CFG/pmc_MIX > /dev/null
fi
if want clk; then
timeout ${PASS_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d /tmp/pc_$CFG/pmc_CLK --output-format csv -- python3 $R/bench.py $ARGS > $O/${TAG}_${CFG}_pmc_CLK.log 2>&1
python3 $R/tools/clock_summary.py $O/${TAG}_${CFG}_clock.json /tmp/pc_$CFG/pmc_CLK
fi
rm -rf /tmp/pc_$CFG
[ -f $O/${TAG}_${CFG}_kernel_stats.txt ] && head -12 $O/${TAG}_${CFG}_kernel_stats.txt | cut -c1-150
