#!/bin/bash
# round 6, first GPU call: the spike sweep, the atomic-rate probe, the listed-tile histogram of C5
R=$PWD
mkdir -p gpurun_out
timeout 300 ./var_libs/atomic_rate > gpurun_out/r6_atomic_rate.txt 2>&1; echo "atomic_rate rc $?"
cat gpurun_out/r6_atomic_rate.txt
timeout 900 python3 tools/spike_sweep.py > gpurun_out/r6_spike_sweep.log 2> gpurun_out/r6_spike_sweep.err; echo "sweep rc $?"
cat gpurun_out/r6_spike_sweep.log; tail -5 gpurun_out/r6_spike_sweep.err
IMSIM_HIP_LIB=$R/var_libs/hist.so timeout 900 python3 tools/dbg/c5_tile_hist.py 40 > gpurun_out/r6_c5_tile_hist.log 2> gpurun_out/r6_c5_tile_hist.err; echo "hist rc $?"
cat gpurun_out/r6_c5_tile_hist.log; tail -5 gpurun_out/r6_c5_tile_hist.err
