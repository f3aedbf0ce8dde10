#!/bin/bash
# C5: which of the four shared streams carries a CCD's static-state initialisation and its image copy
for v in "bulk mid" "mid mid" "mid bulk" "top mid" "top top" "mid top" "bulk top"; do set -- $v; for c in 3 4; do echo "== init $1 copy $2 concurrent $c"; IMS_FOCAL_INIT=$1 IMS_FOCAL_COPY=$2 C5_ONLY=$c python3 tools/dbg/c5_profile.py 36 2>&1 | grep concurrent | tail -1; done; done
