IMS_C5_CCDS=24 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/c5q.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1)); print(json.dumps(d['roofline'], indent=1)[:1500])" || tail -5 gpurun_out/c5q.err
