#!/bin/bash
# round 5: SQ and HBM counters of C5 on the first N CCDs of the visit (rocprofv3's counter pass segfaults in its own dispatch hook on the full
# visit with the list kernels -- 16 and 48 CCDs, or 189 without the lists, pass; launches from ONE host thread here)
ulimit -c 0
R=$PWD
mkdir -p $R/gpurun_out
export IMS_FOCAL_JOINT_THREAD=0 IMS_FFT_WARM=0
cd /tmp && export TMPDIR=/tmp
for N in 48; do
  export IMS_C5_CCDS=$N
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES -d $R/gpurun_out/round5_c5_pmc_SQ --output-format csv -- python3 $R/bench.py --config c5 --no-extra-configs --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $R/gpurun_out/round5_c5_pmc_SQ.log 2>&1
  rc=$?
  echo "SQ pass, $N CCDs: rc $rc"
  if [ $rc -eq 0 ]; then break; fi
  rm -rf $R/gpurun_out/round5_c5_pmc_SQ
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/round5_c5_sq_pmc.txt $R/gpurun_out/round5_c5_sq_pmc.json $R/gpurun_out/round5_c5_pmc_SQ > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/round5_c5_pmc_$C --output-format csv -- python3 $R/bench.py --config c5 --no-extra-configs --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $R/gpurun_out/round5_c5_pmc_$C.log 2>&1
  echo "$C pass, $IMS_C5_CCDS CCDs: rc $?"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/round5_c5_hbm_pmc.txt $R/gpurun_out/round5_c5_hbm_pmc.json $R/gpurun_out/round5_c5_pmc_FETCH_SIZE $R/gpurun_out/round5_c5_pmc_WRITE_SIZE > /dev/null
find $R/gpurun_out/round5_c5_pmc_* -name "*.csv" -delete
echo "CCDs: $IMS_C5_CCDS, 3 calls (1 warm-up + 2 steps)" > $R/gpurun_out/round5_c5_pmc_ccds.txt
head -40 $R/gpurun_out/round5_c5_sq_pmc.txt | cut -c1-150
head -12 $R/gpurun_out/round5_c5_hbm_pmc.txt | cut -c1-150
