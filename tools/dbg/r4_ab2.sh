for J in 16 8 4 0; do python3 tools/dbg/r4_cold.py 64 $J 2>&1 | grep "^joint"; done
