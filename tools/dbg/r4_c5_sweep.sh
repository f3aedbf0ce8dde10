#!/bin/bash
# C5 with long chains: host threads x top-chain streams x hardware queues x CCDs in flight (24 CCDs)
export R4_SKIP_SINGLE=1
for q in 0 8; do for tops in 2 4; do for thr in 1 2 4; do
  conc="3,6"; [ $tops = 4 ] && conc="4,8"
  echo "== queues $q tops $tops threads $thr"
  if [ $q = 0 ]; then IMS_FOCAL_TOPS=$tops IMS_FOCAL_THREADS=$thr R4_CONC=$conc python3 tools/dbg/r4_c5.py 24 2>&1 | grep concurrent
  else GPU_MAX_HW_QUEUES=$q IMS_FOCAL_TOPS=$tops IMS_FOCAL_THREADS=$thr R4_CONC=$conc python3 tools/dbg/r4_c5.py 24 2>&1 | grep concurrent; fi
done; done; done
