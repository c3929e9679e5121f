"""Partitioned (multi-GPU) path vs the golden open-loop series.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 scripts/check_partitioned.py [--same-device]

One process per GPU (``--same-device``: every rank on GPU 0, for a 1-GPU box if RCCL allows it).
"""
import argparse
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    dev = 0 if args.same_device else local
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo" if args.same_device else "nccl")
    from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC

    g = np.load(ROOT / "tests" / "golden" / "cylinder_O1.npz")
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tempfile.mkdtemp(), num_steps=args.steps)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    fs.th.device(dev)
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    for _ in range(args.steps):
        fs.step([0.0, 0.0])
    ts = fs.timeseries
    y = ts[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy()
    n = args.steps + 1
    ey = np.linalg.norm(y - g["ol_y"][:n]) / np.linalg.norm(g["ol_y"][:n])
    eE = np.linalg.norm(ts["dE"].to_numpy() - g["ol_dE"][:n]) / np.linalg.norm(g["ol_dE"][:n])
    u = fs.fields.u_.vector().get_local()
    print(f"[rank {rank}/{world}] cells {fs.th.device().part.local_cells.size if fs.th.device().part else fs.th.nc} "
          f"rel-L2 y {ey:.2e} dE {eE:.2e} residual {fs.solve_info[1]:.1e} |u| {np.linalg.norm(u):.12g}", flush=True)
    assert ey < 1e-8 and eE < 1e-8, "partitioned run deviates from the golden series"
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
