#!/bin/bash
# round 5: lists of a joint round from charge marks per 4 x 4 pixels (IMS_JOINT_FINE_MARKS): C5 A/B, kernel trace
ulimit -c 0
R=$PWD
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r5y_fine_marks.log
: > $L
for v in 1 0 1 0 1 0; do
  IMS_JOINT_FINE_MARKS=$v timeout 300 python bench.py --config c5 --no-extra-configs --steps 3 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "IMS_JOINT_FINE_MARKS=$v" <<'PY' >> $L
import json, sys
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 1), d["extra"].get("step_ms"))
PY
done
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export IMS_JOINT_FINE_MARKS=$v IMS_C5_CCDS=64
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5y_kt_$v -- python3 $R/bench.py --config c5 --no-extra-configs --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $R/gpurun_out/r5y_kt_$v.log 2>&1
  DB=$(find $R/gpurun_out/r5y_kt_$v -name "*.db" | head -1)
  python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/r5y_kernel_stats_fine$v.txt > /dev/null
  rm -rf $R/gpurun_out/r5y_kt_$v
  echo "== kernel stats, IMS_JOINT_FINE_MARKS=$v (64 CCDs, 3 calls)" >> $L
  head -14 $R/gpurun_out/r5y_kernel_stats_fine$v.txt | cut -c1-170 >> $L
done
cat $L
