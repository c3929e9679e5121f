"""Initial-condition sweep of the lid-driven cavity — the reference's ``src/examples/lidcavity/batch_run_lidcavity.py`` workload.

The reference builds one ``LidCavityFlowSolver`` per initial condition (a 3 × 3 grid of vortex centres), steps each of them
100 times and writes, per run, the snapshot matrices ``U_field_alldata.npy`` / ``P_field_alldata.npy`` /
``UP_field_alldata.npy`` ([ndof, nsnapshots, 1], full fields = perturbation + base flow, one column per checkpoint) and the base
flow ``U0_field_data.npy`` / ``P0_field_data.npy`` / ``UP0_field_data.npy`` into ``run<count>/``.  All those runs share mesh, base
flow, Δt and boundary conditions, hence operators and factors: here they are ONE ``BatchedFlowSolver`` whose k runs advance in lock
step (``fc_step_batch``: the factors are read once per step for all of them), and the same files come out.

    python -m flowcontrol_amd.examples.lidcavity.batch_run_lidcavity [out_dir] [Re]
"""

from __future__ import annotations

import logging
import sys
import time
from pathlib import Path

import numpy as np

from flowcontrol_amd.batch import BatchedFlowSolver
from flowcontrol_amd.examples.lidcavity.lidcavityflowsolver import LidCavityFlowSolver
from flowcontrol_amd.flowsolverparameters import ParamIC

logger = logging.getLogger(__name__)


def run_lidcavity_with_ics(Re: float, ics: list[ParamIC], save_dirs: list[Path], num_steps: int = 100, save_every: int = 20,
                           picard_iterations: int = 40, meshpath=None) -> BatchedFlowSolver:
    """All initial conditions of ``ics`` at once (at most 16 per batch).  Returns the batched solver (time series of run i:
    ``bfs.timeseries(i)``)."""
    if len(ics) != len(save_dirs) or not 1 <= len(ics) <= 16:
        raise ValueError("one output directory per initial condition, at most 16 of them per batch")
    fs = LidCavityFlowSolver.make_default(Re=Re, path_out=save_dirs[0], num_steps=num_steps, save_every=0, meshpath=meshpath)
    # the reference loads a base flow computed by continuation in Re (compute_steady_state_increasing_Re.py); Picard from rest
    # converges for the moderate Reynolds numbers this script is run at
    fs.compute_steady_state(method="picard", max_iter=picard_iterations, tol=1e-8, u_ctrl=[0.0])
    k = len(ics)
    bfs = BatchedFlowSolver(fs, k)
    bfs.initialize_time_stepping(ics=ics)
    U0 = fs.fields.U0.vector().get_local()
    P0 = fs.fields.P0.vector().get_local()
    nsnap = num_steps // save_every
    U = np.empty((k, U0.size, nsnap))
    P = np.empty((k, P0.size, nsnap))
    u_ctrl = np.zeros((k, fs.params_control.actuator_number))  # unforced runs, as in the reference script
    for n in range(1, num_steps + 1):
        bfs.step(u_ctrl)
        if n % save_every == 0:
            u_n, _, p_n = bfs.state()  # one download for all runs
            U[:, :, n // save_every - 1] = u_n + U0  # checkpoints hold the full field (exporter.py:126-132)
            P[:, :, n // save_every - 1] = p_n + P0
    UP0 = np.r_[U0, P0]
    for i, d in enumerate(save_dirs):
        d = Path(d)
        d.mkdir(parents=True, exist_ok=True)
        np.save(d / "U_field_alldata.npy", U[i][:, :, None])
        np.save(d / "P_field_alldata.npy", P[i][:, :, None])
        np.save(d / "UP_field_alldata.npy", np.concatenate([U[i], P[i]], axis=0)[:, :, None])
        np.save(d / "U0_field_data.npy", U0)
        np.save(d / "P0_field_data.npy", P0)
        np.save(d / "UP0_field_data.npy", UP0)
        bfs.timeseries(i).to_csv(d / "timeseries1D.csv", sep=",", index=False)
    return bfs


def main(base_dir: Path | None = None, Re: float = 1000.0, num_steps: int = 100) -> None:
    logging.basicConfig(level=logging.INFO)
    parent = (Path(base_dir) if base_dir else Path.cwd() / "data_output" / "lidcavity_batch") / f"Re{Re:g}"
    x_vals, y_vals = np.linspace(0.2, 0.8, 3), np.linspace(0.2, 0.8, 3)
    radius = amplitude = 0.1
    ics = [ParamIC(xloc=float(x), yloc=float(y), radius=radius, amplitude=amplitude) for x in x_vals for y in y_vals]
    dirs = [parent / f"run{c + 1}" for c in range(len(ics))]
    t0 = time.perf_counter()
    run_lidcavity_with_ics(Re, ics, dirs, num_steps=num_steps)
    print(f"{len(ics)} runs x {num_steps} steps in {time.perf_counter() - t0:.2f} s (base flow included); results under {parent}")


if __name__ == "__main__":
    main(Path(sys.argv[1]) if len(sys.argv) > 1 else None, float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0)
