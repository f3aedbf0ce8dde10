timeout 900 python3 -m pytest tests/test_device_table.py -m gpu -q -x -k "settled_objects" 2>&1 | tail -3
for S in 1 0 1 0; do echo "end to end, early $S"; IMS_EARLY_TOP=$S python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']; print(round(d['ms_per_step'],2), e.get('end_to_end_ms'), e.get('end_to_end_first_ms'), e.get('end_to_end_parts_last'))"; done
