#!/bin/bash
R=$PWD
timeout 900 python3 -m pytest tests -m gpu -q -x -k "fft or spike or focal" 2>&1 | tail -3
python3 bench.py --config fft 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fft', round(d['ms_per_step'],3), d.get('cpu_baseline',{}).get('parity',{}).get('bit_identical'))"
python3 bench.py --config c5 --steps 3 --warmup 2 > gpurun_out/r4f3_c5_bench.json 2>/dev/null; python3 -c "
import json
d=json.load(open('gpurun_out/r4f3_c5_bench.json')); print('c5', round(d['ms_per_step'],1), round(d['value']), d.get('cpu_baseline',{}).get('parity',{}).get('bit_identical'), d['cpu_baseline']['value'], d['cpu_baseline']['sample'])"
