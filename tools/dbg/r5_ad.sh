#!/bin/bash
# round 5: where rocprofv3's counter pass crashes on C5 (small visit; without the list kernels)
ulimit -c 0
R=$PWD
mkdir -p $R/gpurun_out
export IMS_FOCAL_JOINT_THREAD=0 IMS_FFT_WARM=0
cd /tmp && export TMPDIR=/tmp
for v in "IMS_C5_CCDS=16" "IMS_C5_CCDS=48" "IMS_JOINT_LISTS=0"; do
  export $v
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $R/gpurun_out/r5ad_pmc --output-format csv -- python3 $R/bench.py --config c5 --no-extra-configs --steps 1 --warmup 1 --no-cpu-baseline --no-cold > $R/gpurun_out/r5ad.log 2>&1
  echo "$v: rc $?"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/r5ad_sq.txt $R/gpurun_out/r5ad_sq.json $R/gpurun_out/r5ad_pmc > /dev/null 2>&1
  head -8 $R/gpurun_out/r5ad_sq.txt | cut -c1-150
  rm -rf $R/gpurun_out/r5ad_pmc
  unset IMS_C5_CCDS IMS_JOINT_LISTS
done
