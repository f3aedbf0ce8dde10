#!/bin/bash
# round 4: one-wavefront workgroups for the pixel search of the rounds (IMS_ROUND_WG=64, default) against 256-thread ones
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('kernel_ms_per_step'), d['roofline'].get('mean_launch_ms'))"; }
echo "== parity"; python3 -m pytest tests/test_parity_gpu.py -q -x -k "native or c3_lsst or edge or several or focal or slot_pairs or specialised" 2>&1 | tail -2
for rep in 1 2; do for w in 256 64; do echo "== C3 round workgroup $w (run $rep)"; IMS_ROUND_WG=$w python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cold 2>/dev/null | ms; done; done
for w in 256 64; do echo "== one star, workgroup $w"; IMS_ROUND_WG=$w python3 tools/dbg/one_star.py 2>&1 | tail -1; done
for w in 256 64; do echo "== C3b workgroup $w"; IMS_ROUND_WG=$w python3 bench.py --config c3b --steps 10 --warmup 3 --no-cpu-baseline --no-cold 2>/dev/null | ms; done
for w in 256 64; do echo "== shard replay, workgroup $w"; IMS_ROUND_WG=$w python3 tools/dbg/shard_times.py 2>&1 | grep world; done
export R4_SKIP_SINGLE=1 R4_CONC=4
for w in 256 64; do echo "== C5 24 CCDs workgroup $w"; IMS_ROUND_WG=$w python3 tools/dbg/r4_c5.py 24 2>&1 | grep concurrent; done
echo "== timeline with 64"; IMS_ROUND_WG=64 bash tools/dbg/r3_timeline.sh 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | head -70
