#!/bin/bash
# round 3: HBM counters of C3b with the phase-screen pre-pass over every photon (IMS_SCREEN_PREPASS=1)
export IMS_SCREEN_PREPASS=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/r3pp_c3b_pmc_$C --output-format csv -- python3 $R/bench.py --config c3b --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $R/gpurun_out/r3pp_c3b_pmc_$C.log 2>&1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/round3_c3b_prepass_hbm_pmc.txt $R/gpurun_out/round3_c3b_prepass_hbm_pmc.json $R/gpurun_out/r3pp_c3b_pmc_FETCH_SIZE $R/gpurun_out/r3pp_c3b_pmc_WRITE_SIZE > /dev/null
find $R/gpurun_out/r3pp_c3b_pmc_* -name "*.csv" -delete
grep -E "k_shoot|k_screen" $R/gpurun_out/round3_c3b_prepass_hbm_pmc.txt | cut -c1-160
