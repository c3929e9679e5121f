"""Where the host's time goes inside FlowSolver.step (wall clock of the three library calls and of the Python between them):
    FC_EARLY_NT=0|1 python scripts/host_split.py refined1 [steps]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401,E402

import bench  # noqa: E402
from flowcontrol_amd.comm import SingleComm  # noqa: E402

KEYS = {"pinball": "config5", "cavity_fine": "config3", "refined1": "config4", "O1": None}
name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
if KEYS[name] is None:
    fs = bench.build_solver(0)
    ctrl = lambda: np.zeros(2)  # noqa: E731
else:
    case = bench.CASES[KEYS[name]]
    fs = case.make(0, case.prepare(SingleComm(), 0))
    ctrl = case.controller(fs)
for _ in range(20):
    fs.step(ctrl())
dev = fs.th.device()
acc = {"begin": 0.0, "flush": 0.0, "end": 0.0}
marks = []
ob, oe, of = dev.step_begin, dev.step_end, fs._flush_log


def wrap(fn, key):
    def inner(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[key] += time.perf_counter() - t
        return r

    return inner


dev.step_begin, dev.step_end, fs._flush_log = wrap(ob, "begin"), wrap(oe, "end"), wrap(of, "flush")
t0 = time.perf_counter()
for _ in range(steps):
    fs.step(ctrl())
tot = time.perf_counter() - t0
other = tot - sum(acc.values())
print(f"{name}: {steps / tot:.1f} steps/s; per step [us]: total {1e6 * tot / steps:.1f} = step_begin {1e6 * acc['begin'] / steps:.1f} + flush_log(collect) "
      f"{1e6 * acc['flush'] / steps:.1f} + step_end {1e6 * acc['end'] / steps:.1f} + python {1e6 * other / steps:.1f}")
