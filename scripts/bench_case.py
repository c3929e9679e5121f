"""One of BASELINE configs 3 / 4 / 5 on one MI355X, through bench.py's own leg driver (``bench.run_case``): closed-loop steps/s
of the synchronous public loop, the factor sweeps' roofline from an instrumented replay, phase split, fc_refactor ms.

    python scripts/bench_case.py pinball --steps 10000      # config 5
    python scripts/bench_case.py cavity_fine --steps 1000   # config 3
    python scripts/bench_case.py refined1 --steps 2000      # config 4

(The default ``python bench.py`` run carries the same three entries, at fewer steps, in ``other_configs``; this is the tool for
long runs and for the profiler passes of scripts/profile_case.sh.)
"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401,E402  (one HIP runtime, loaded first)

import bench  # noqa: E402
from flowcontrol_amd.comm import SingleComm  # noqa: E402

KEYS = {"pinball": "config5", "cavity_fine": "config3", "refined1": "config4"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case", choices=sorted(KEYS))
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=20)
    args = ap.parse_args()
    case = bench.CASES[KEYS[args.case]]
    case.steps_cap, case.warm = max(args.steps, 10), args.warmup
    out = bench.run_case(case, SingleComm(), 0, args.steps, with_roofline=True)
    out.update(metric="timesteps/s (closed loop, synchronous public FlowSolver.step)", value=out["steps_per_s"], unit="timesteps/s", dtype="f64")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
