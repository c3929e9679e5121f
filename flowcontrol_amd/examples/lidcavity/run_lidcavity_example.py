"""Lid-driven cavity close to its Hopf bifurcation (Re_c ≈ 7 700; Re = 8 000), on an MI355X.

Same user-visible sequence as the reference's ``src/examples/lidcavity/run_lidcavity_example.py``: base flow by 40 Picard sweeps
(no Newton that close to the bifurcation), then an unactuated simulation from a small vortex at (0.1, 0.1), checkpoints every
20 steps.  An enclosed flow: the pressure is defined up to a constant, pinned inside the factorisation.

    python -m flowcontrol_amd.examples.lidcavity.run_lidcavity_example [num_steps]
"""
import logging
import sys
import time
from pathlib import Path

from flowcontrol_amd import utils as flu
from flowcontrol_amd.examples.lidcavity.lidcavityflowsolver import LidCavityFlowSolver
from flowcontrol_amd.flowsolverparameters import ParamIC


def main(num_steps: int = 100, path_out: Path | None = None, Re: float = 8000):
    logging.basicConfig(level=logging.INFO)
    out = Path(path_out) if path_out else Path.cwd() / "data_output"
    fs = LidCavityFlowSolver.make_default(Re=Re, path_out=out, num_steps=num_steps, save_every=20, verbose=10)
    fs.params_ic = ParamIC(xloc=0.1, yloc=0.1, radius=0.1, amplitude=0.1)
    flu.export_subdomains(fs.mesh, fs.boundaries.subdomain, out / "subdomains.xdmf")
    t0 = time.perf_counter()
    fs.compute_steady_state(method="picard", max_iter=40, tol=1e-7, u_ctrl=[0.0])
    print(f"base flow on {fs.th.N} dofs: {time.perf_counter() - t0:.2f} s")
    fs.initialize_time_stepping(ic=None)
    t0 = time.perf_counter()
    for _ in range(fs.params_time.num_steps):
        y_meas = flu.MpiUtils.mpi_broadcast(fs.y_meas)
        fs.step(u_ctrl=[0.0 * y_meas[0]])
    seconds = time.perf_counter() - t0
    print(f"{num_steps} steps: {seconds:.2f} s = {num_steps / seconds:.0f} steps/s; y = {fs.y_meas}")
    fs.write_timeseries()
    return fs


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100)
