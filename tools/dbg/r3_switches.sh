#!/bin/bash
# every switch of DESIGN.md 4 "Switches" must give the same results: the GPU parity tests under each alternative
run() { echo "== $*"; env "$@" python3 -m pytest tests -m gpu -x -q -k "not focal_plane_bench" 2>&1 | tail -1; }
run IMS_CHAIN_KERNELS=0
run IMS_LAYOUT_KERNELS=0
run IMS_UPD_DPP=0
run IMS_UPD_DPP_MAX=100000
run IMS_PSF_SCREENS_KERNEL=0 IMS_PHOTON_LDS=0
run IMS_SCREEN_PREPASS=1
run IMS_SCREEN_PREPASS=2
run IMS_FOCAL_STREAMS=0
run IMS_FOCAL_THREADS=2
run IMS_PLAN_ROUNDS=0
run IMS_BF_TAGS=1
