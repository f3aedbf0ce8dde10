python3 -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
"
c3() { env "$@" python3 bench.py --no-cpu-baseline --no-cold 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))"; }
for P in "-1,0,0,0,0" "-1,0,1,1,1" "-1,1,0,0,0" "-1,1,1,1,1" "-1,0,-1,0,0" "-1,-1,0,0,0" "0,0,0,0,0" "-1,1,0,1,1"; do echo "C3 priorities $P"; c3 IMS_STREAM_PRIORITIES=$P; done
