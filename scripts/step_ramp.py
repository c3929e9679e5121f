"""Development probe: per-step wall time of the first steps after ``initialize_time_stepping`` (does a 20-step sample after 5 warm-up
steps -- the driver's default -- see the steady rate?).  Prints the mean step time of consecutive windows.
    python scripts/step_ramp.py [--windows 12] [--window 20]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=12)
    ap.add_argument("--window", type=int, default=20)
    ap.add_argument("--sleep", type=float, default=0.0, help="idle seconds between setup and the first step")
    a = ap.parse_args()
    fs = bench.build_solver(0)
    u0 = np.zeros(2)
    if a.sleep:
        time.sleep(a.sleep)
    for _ in range(5):
        fs.step(u0)
    torch.cuda.synchronize()
    out = []
    for w in range(a.windows):
        t0 = time.perf_counter()
        for _ in range(a.window):
            fs.step(u0)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / a.window * 1e6)
    print("us/step per window of", a.window, ":", " ".join(f"{v:.1f}" for v in out), flush=True)


if __name__ == "__main__":
    main()
