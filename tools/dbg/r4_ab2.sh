timeout 900 python3 -m pytest tests/test_device_table.py -m gpu -q -x 2>&1 | tail -5
for S in 1 0; do echo "end to end, split $S"; IMS_E2E_SPLIT=$S python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']; print(round(d['ms_per_step'],2), e.get('end_to_end_ms'), e.get('end_to_end_first_ms'), e.get('end_to_end_parts_last'), e.get('cold_render_ms'))"; done
