"""Do kernels of two HIP streams overlap on this part?  Two handles (two streams) of the same case: the factor sweeps on one, the CSR SpMV
(the residual monitor's access pattern) on the other -- each timed alone, then both at once from two host threads.

    python scripts/overlap_probe.py [O1|refined1|pinball|cavity_fine]
"""
import sys
import threading
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "scripts"))
import torch  # noqa: F401,E402
from flowcontrol_amd._lib import SLOT_BDF2  # noqa: E402
from tail_probe import build  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "O1"
fa, ca = build(case)
fb, cb = build(case)
for fs, c in ((fa, ca), (fb, cb)):
    for _ in range(5):
        fs.step(c())
da, db = fa.th.device(), fb.th.device()
reps_sweep = 300 if case == "O1" else 60
ms_spmv_alone = db.bench_spmv(SLOT_BDF2, 200)
n_spmv = max(50, int(1.5 * reps_sweep * da.bench_sweeps(SLOT_BDF2, 20)[0] / ms_spmv_alone))
ms_sweep_alone, nl = da.bench_sweeps(SLOT_BDF2, reps_sweep)
ms_spmv_alone = db.bench_spmv(SLOT_BDF2, n_spmv)
out = {}


def run_a():
    out["sweep"] = da.bench_sweeps(SLOT_BDF2, reps_sweep)[0]


def run_b():
    out["spmv"] = db.bench_spmv(SLOT_BDF2, n_spmv)


ta, tb = threading.Thread(target=run_a), threading.Thread(target=run_b)
ta.start(), tb.start()
ta.join(), tb.join()
serial = reps_sweep * ms_sweep_alone + n_spmv * ms_spmv_alone
both = max(reps_sweep * out["sweep"], n_spmv * out["spmv"])
print(f"{case}: apply alone {1e3 * ms_sweep_alone:.1f} us ({nl} launches), SpMV alone {1e3 * ms_spmv_alone:.1f} us")
print(f"  together: apply {1e3 * out['sweep']:.1f} us, SpMV {1e3 * out['spmv']:.1f} us; {reps_sweep} applies + {n_spmv} SpMVs: serial {serial:.2f} ms, "
      f"concurrent {both:.2f} ms ({100 * (1 - both / serial):.0f} % saved)")
fa.th.release_device(), fb.th.release_device()
