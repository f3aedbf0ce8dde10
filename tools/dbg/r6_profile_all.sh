#!/bin/bash
# round 6: every judged measurement of every bench config (tools/profile_config.sh), one after the other
for CFG in ${@:-c3 c3b c2 c4 fft fftx fftxs c5}; do
  echo "=== $CFG"
  bash tools/profile_config.sh round6 $CFG > gpurun_out/round6_${CFG}_profile.log 2>&1
  tail -4 gpurun_out/round6_${CFG}_profile.log | cut -c1-200
done
