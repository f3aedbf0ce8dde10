#!/bin/bash
# C5: plan streams by role for all CCDs of a device (engine._focal_streams); host threads
python3 -m pytest tests -m gpu -x -q -k "focal" 2>&1 | tail -3
for t in 1 2 3; do for c in 3 4; do echo "== threads $t concurrent $c"; IMS_FOCAL_THREADS=$t C5_ONLY=$c python3 tools/dbg/c5_profile.py 24 2>&1 | grep concurrent | tail -1; done; done
