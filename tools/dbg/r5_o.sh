#!/bin/bash
# round 5: device memory by route (one block / many blocks / one virtual range over chunks / torch expandable segments)
ulimit -c 0
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 -o /tmp/vmm_alloc tools/dbg/vmm_alloc.hip 2>/dev/null
L=gpurun_out/round5_vmm_alloc.log
: > $L
for r in 3 2 1; do timeout 300 /tmp/vmm_alloc 48 $r >> $L 2>&1; done
cat > /tmp/exp.py <<'PY'
import time, torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for gib in (4, 16, 48):
    t0 = time.perf_counter()
    t = torch.empty(gib << 30, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    t.zero_(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"torch.empty {gib} GiB: {1e3 * (t1 - t0):.1f} ms, first fill {1e3 * (t2 - t1):.1f} ms", flush=True)
    del t
    torch.cuda.empty_cache()
PY
echo "torch default allocator:" >> $L
timeout 300 python /tmp/exp.py 2>&1 | grep -v amdgpu.ids >> $L
echo "torch expandable_segments:True:" >> $L
PYTORCH_ALLOC_CONF=expandable_segments:True PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout 300 python /tmp/exp.py 2>&1 | grep -v amdgpu.ids >> $L
cat $L
timeout 300 python bench.py --config c5 --no-extra-configs --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5o_c5_s3w1.log 2>&1
timeout 300 python bench.py --config c5 --no-extra-configs --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r5o_c5_s4w2.log 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/r5o_c5_s3w1.log", "gpurun_out/r5o_c5_s4w2.log"):
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line); print(f, d["ms_per_step"], d["extra"].get("step_ms"))
PY
