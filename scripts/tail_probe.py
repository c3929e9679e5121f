"""fc_tail under a profiler: a few dozen closed-loop steps of one case with the energy cells on, then off, so that
`rocprofv3 --kernel-trace --stats` (scripts/tail_probe.sh) separates the row / shift workgroups from the cell workgroups.

    python scripts/tail_probe.py cavity_fine|cavity_coarse|pinball|O1|refined1 [--steps 60]
"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401,E402
import bench  # noqa: E402
from flowcontrol_amd.comm import SingleComm  # noqa: E402


def build(case):
    if case in ("O1", "refined1"):
        fs = bench.build_solver(0, distributed=False, refine=0 if case == "O1" else 1)
        return fs, lambda: np.zeros(2)
    key = {"cavity_fine": "config3", "pinball": "config5"}.get(case)
    if key:
        c = bench.CASES[key]
        fs = c.make(0, c.prepare(SingleComm(), 0))
        fs.distributed = False
        return fs, c.controller(fs)
    raise SystemExit(f"unknown case {case}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("--steps", type=int, default=60)
    a = ap.parse_args()
    fs, ctrl = build(a.case)
    for _ in range(a.steps):
        fs.step(ctrl())
    dev = fs.th.device()
    dev.set_phase_timing(True)
    for _ in range(a.steps):
        fs.step(ctrl())
    ph = dev.get_phase_timing()
    dev.set_phase_timing(False)
    print("phase_us", {k: round(v, 1) for k, v in ph.items()}, "N", fs.th.N, "nnz", dev.nnz)
    fs.params_save.energy_every = 0  # the energy cells leave fc_tail's grid
    for _ in range(a.steps):
        fs.step(ctrl())
    fs.th.release_device()


if __name__ == "__main__":
    main()
