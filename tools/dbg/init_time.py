"""k_init_tiles on the static state of one 4k x 4k CCD, alone on the GPU (run under gpurun)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from imsim_amd import configs
from imsim_amd.engine import Renderer
scene = configs.BENCH_CONFIGS["c3"]["scene"]()
r = Renderer(scene)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    for _ in range(10):
        r.init_boundaries(0, 1)
    e1.record()
    torch.cuda.synchronize()
    cells = r.bound.static_cells
    ms = e0.elapsed_time(e1) / 10
    print(f"init of {cells} static cells: {ms:.3f} ms = {cells * 232 / ms / 1e9:.2f} TB/s of the 232 B it writes per cell")
