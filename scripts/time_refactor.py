"""Setup cost on the GPU box: structure (host) + numeric factorisation (device) vs the numpy multifrontal."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # as bench.py does: library load / page-in is not what is timed here
torch.zeros(1).cuda()
import bench
t0 = time.time()
fs = bench.build_solver(0)
print(f"build_solver {time.time()-t0:.2f}s", flush=True)
t0 = time.time()
fs.step(np.zeros(2))
print(f"first step (assemble + factorise both systems) {time.time()-t0:.2f}s", flush=True)
dev = fs.th.device()
from flowcontrol_amd._lib import SLOT_BDF2
for i in range(3):
    t0 = time.time()
    ms = dev.refactor(SLOT_BDF2)
    print(f"refactor: device {ms:.1f} ms, wall {1e3*(time.time()-t0):.1f} ms", flush=True)
y = fs.step(np.zeros(2))
print("residual after refactor", fs.solve_info if hasattr(fs, "solve_info") else None, y)
