"""Where does a synchronous FlowSolver.step() spend host time? (run on the GPU box)"""
import sys, time, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench
fs = bench.build_solver(0)
u0 = np.zeros(2)
for _ in range(50): fs.step(u0)
dev = fs.th.device()
from flowcontrol_amd._lib import SLOT_BDF2
import ctypes as C
n = 2000
t0 = time.perf_counter()
for _ in range(n): fs.step(u0)
t_fs = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n): dev.step(SLOT_BDF2, u0, True)
t_dev = (time.perf_counter() - t0) / n
lib, h = dev.lib, dev._h
y = np.empty(3); info = np.empty(4); dE = C.c_double()
pu, py, pi = u0.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), info.ctypes.data_as(C.c_void_p)
t0 = time.perf_counter()
for _ in range(n): lib.fc_step(h, SLOT_BDF2, pu, None, py, C.byref(dE), 1, pi)
t_raw = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
fs.run(n, u0)
t_run = (time.perf_counter() - t0) / n
print(f"fs.step {t_fs*1e6:.1f} us | DeviceSolver.step {t_dev*1e6:.1f} us | raw fc_step {t_raw*1e6:.1f} us | fc_run per step {t_run*1e6:.1f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(1000): fs.step(u0)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
