R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in base xyv; do
  L=$R/imsim_amd/lib/libimsim_hip.so; [ $v = xyv ] && L=$R/imsim_amd/lib/variants/libimsim_hip_xyv.so
  export IMSIM_HIP_LIB=$L
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmc_$v --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cold > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmc_$v.txt $R/gpurun_out/pmc_$v | grep "k_shoot_photons<true>"
  rm -rf $R/gpurun_out/pmc_$v
done
