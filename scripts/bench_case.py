"""Closed-loop throughput runs of the larger BASELINE configs on one MI355X (not the headline: that is bench.py).

    python scripts/bench_case.py pinball --steps 10000     # config 5: pinball Re=100, ROTATION, 3 actuators, 3-in/3-out LTI controller
    python scripts/bench_case.py cavity_fine --steps 2000  # config 3: open cavity Re=7500 on cavity_fine, FORCE actuator + wall-shear
                                                           #           sensor through a first-order low-pass controller

One JSON line: timesteps/s of the synchronous public loop  y -> Controller.step -> FlowSolver.step  (every step's
measurement goes back to the host controller, as in the reference's closed-loop scripts), the factor sweeps' roofline from
an instrumented replay, the worst per-step residual.  The pinball starts from the golden base flow (oracle, Picard x15 ->
Newton); cavity_fine from a few Picard sweeps on the device (a throughput run, not a converged base flow).
"""
import argparse
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "golden"))
import torch  # noqa: F401,E402  (one HIP runtime, loaded first)

from flowcontrol_amd._lib import SLOT_BDF2  # noqa: E402
from flowcontrol_amd.controller import Controller  # noqa: E402

HBM_PEAK_GBS = 8000.0


def build(case):
    from flowcontrol_amd.flowsolverparameters import ParamIC

    if case == "pinball":
        from make_config45_fixtures import PINBALL_K

        from flowcontrol_amd.actuator import CYLINDER_ACTUATION_MODE
        from flowcontrol_amd.examples.pinball.pinballflowsolver import PinballFlowSolver
        from flowcontrol_amd.fem.spaces import Function

        g = np.load(ROOT / "tests" / "golden" / "pinball_re100_rotation.npz")
        fs = PinballFlowSolver.make_default(Re=100, mode_actuation=CYLINDER_ACTUATION_MODE.ROTATION, path_out=tempfile.mkdtemp(), num_steps=0)
        fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
        U0, P0 = Function(fs.W, g["UP0"]).split()
        fs._assign_steady_state(U0, P0)
        # the fixture's gain (x 2e4) is sized for the first 50 steps (sensor readings of 1e-4); over 10 000 steps the wake
        # saturates at O(1) readings, so the long run closes the loop with |u| ~ 0.4 |y|
        K = Controller(A=PINBALL_K["A"], B=PINBALL_K["B"], C=PINBALL_K["C"] / 2.0e6, D=PINBALL_K["D"])
        n_in = 3
        what = "fluidic pinball Re=100, ROTATION, 3 actuators / 3 sensors, synthetic stable LTI controller"
    else:
        from flowcontrol_amd.examples.cavity.cavityflowsolver import CavityFlowSolver

        fs = CavityFlowSolver.make_default(Re=7500, path_out=tempfile.mkdtemp(), num_steps=0, meshpath=ROOT / "tests" / "golden" / "meshes" / "cavity_fine.npz")
        fs.compute_steady_state(method="picard", max_iter=4, tol=1e-7, u_ctrl=[0.0])
        K = Controller(A=[[-100.0]], B=[[1.0]], C=[[0.5]], D=[[0.0]])
        n_in = 1  # the controller reads the wall-shear sensor (y_meas_1) only
        what = "open cavity Re=7500 on cavity_fine, Gaussian FORCE actuator, wall-shear sensor, first-order low-pass controller"
    fs.params_save.save_every = 0
    fs.initialize_time_stepping(ic=None)
    return fs, K, what, n_in


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case", choices=["pinball", "cavity_fine"])
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=20)
    args = ap.parse_args()
    t0 = time.time()
    fs, K, what, n_in = build(args.case)
    dt = fs.params_time.dt
    y0 = fs.y_meas.copy()  # the controllers act on the deviation from the initial reading

    def step():
        u = np.asarray(K.step(y=(fs.y_meas - y0)[:n_in], dt=dt)).reshape(-1)
        fs.step(u_ctrl=u)
        return u

    step()  # assembles and factorises both operators
    t_setup = time.time() - t0
    for _ in range(args.warmup):
        step()
    worst, umax = 0.0, 0.0
    t1 = time.perf_counter()
    for _ in range(args.steps):
        u = step()
        worst = max(worst, float(fs.solve_info[1]))
        umax = max(umax, float(np.abs(u).max()))
    elapsed = time.perf_counter() - t1
    dev = fs.th.device()
    dev.set_timing(True)
    n_rep = min(args.steps, 200)
    for _ in range(n_rep):
        step()
    tim = dev.get_timing()
    dev.set_timing(False)
    sweep_bytes, _ = dev.algorithmic_bytes(SLOT_BDF2)
    achieved = sweep_bytes * n_rep / tim["sweep_ms"] / 1e6  # GB/s over the sweep launches of the replay
    print(json.dumps({
        "metric": "timesteps/s (closed loop, synchronous public FlowSolver.step)", "value": args.steps / elapsed, "unit": "timesteps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "dtype": "f64",
        "config": {"workload": f"{what}; {fs.th.nc} cells, {fs.th.N} dofs, dt={dt}"},
        "roofline": {"bound": "hbm", "kernel": "fc_nd_sweep + fc_nd_down_block (factor sweeps)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "bytes_per_apply": sweep_bytes, "launches_per_apply": tim["sweep_launches"] / n_rep,
                     "apply_us": 1e3 * tim["sweep_ms"] / n_rep},
        "worst_relative_residual": worst, "max_abs_u_ctrl": umax, "y_last": np.asarray(fs.y_meas).tolist(), "dE_last": float(fs.timeseries["dE"].iloc[-1]),
        "setup_s": t_setup, "refactor_ms": dev.refactor_ms, "factor_values": int(dev._n_factor_values),
    }))
    fs.th.release_device()


if __name__ == "__main__":
    main()
