#!/bin/bash
# the spike convolution with the whole-sum reject: FFT parity tests, the FFT bench line, C5 (24 CCDs and whole), then the full suite
R=$PWD; T=r4sr; mkdir -p gpurun_out
python3 -m pytest tests -m gpu -q -x -k "fft or spike or focal" 2>&1 | tail -3
python3 bench.py --config fft > gpurun_out/${T}_fft_bench.json 2> gpurun_out/${T}_fft_bench.err
IMS_C5_CCDS=24 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_c5_24.json 2> gpurun_out/${T}_c5_24.err
python3 bench.py --config c5 --steps 2 --warmup 1 > gpurun_out/${T}_c5_bench.json 2> gpurun_out/${T}_c5_bench.err
for f in gpurun_out/${T}_*.json; do python3 -c "
import json
d=json.load(open('$f')); print('$f', round(d['ms_per_step'],3), round(d['value']), d['roofline'].get('kernel'), d.get('cpu_baseline',{}).get('parity',{}))"; done
python3 -m pytest tests -m gpu -q 2>&1 | tail -4
python3 bench.py > gpurun_out/${T}_c3_bench.json 2> gpurun_out/${T}_c3_bench.err; python3 -c "
import json
d=json.load(open('gpurun_out/${T}_c3_bench.json')); print(d['ms_per_step'], d['value'], d['extra'].get('end_to_end_ms'), d['extra'].get('end_to_end_parts_last'))"
