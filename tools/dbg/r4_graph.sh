#!/bin/bash
# round 4: hipGraph capture of a plan's enqueue (IMS_PLAN_GRAPH=1) against the plain enqueue; tile marks for C5's long chains
for g in 0 1; do echo "== one star, graph $g"; IMS_PLAN_GRAPH=$g python3 tools/dbg/one_star.py 2>&1 | grep -v amdgpu.ids | tail -2; done
for g in 0 1; do echo "== C3 bench, graph $g"; IMS_PLAN_GRAPH=$g python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cold 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
echo "== parity under the graph"; IMS_PLAN_GRAPH=1 python3 -m pytest tests/test_parity_gpu.py -x -q -k "native_planner or c3_lsst_image or edge_cases or several_brighter" 2>&1 | tail -2
export R4_SKIP_SINGLE=1 R4_CONC=3
echo "== C5 24 CCDs plain"; python3 tools/dbg/r4_c5.py 24 2>&1 | grep concurrent
echo "== C5 24 CCDs graph"; IMS_PLAN_GRAPH=1 python3 tools/dbg/r4_c5.py 24 2>&1 | grep concurrent
echo "== C5 24 CCDs tile marks"; IMS_BF_TAGS=1 python3 tools/dbg/r4_c5.py 24 2>&1 | grep concurrent
echo "== C5 24 CCDs tile marks + graph"; IMS_BF_TAGS=1 IMS_PLAN_GRAPH=1 python3 tools/dbg/r4_c5.py 24 2>&1 | grep concurrent
echo "== C5 24 CCDs DPP update for every launch"; IMS_UPD_DPP_MAX=100000 python3 tools/dbg/r4_c5.py 24 2>&1 | grep concurrent
