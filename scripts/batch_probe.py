"""Development probe of the batched stepping path (prints, asserts nothing): parity of the batched factor apply and of
batched trajectories against single runs, then throughput for k = 1, 4, 8, 16, 32.

    python scripts/batch_probe.py [--mesh O1] [--steps 400]
"""
import argparse
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--ks", default="1,4,8,16,32")
    ap.add_argument("--refine", type=int, default=0, help="red-refinements of the cylinder mesh (1: BASELINE config 4 mesh)")
    ap.add_argument("--skip-parity", action="store_true")
    ap.add_argument("--energy-every", type=int, default=1, help="params_save.energy_every of the throughput runs (0: the tail has no cell workgroups)")
    ap.add_argument("--reps", type=int, default=1, help="repeat every throughput measurement (box noise)")
    a = ap.parse_args()
    from flowcontrol_amd._lib import SLOT_BDF2
    from flowcontrol_amd.batch import BatchedFlowSolver
    from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver, refined_cylinder_mesh
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC

    kw = {"meshpath": refined_cylinder_mesh(a.refine)} if a.refine else {}
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tempfile.mkdtemp(), num_steps=10, **kw)
    fs.params_save.energy_every = a.energy_every
    g = np.load(ROOT / "tests" / "golden" / ("cylinder_O1_refined1.npz" if a.refine else "cylinder_O1.npz"))
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    fs.initialize_time_stepping(ic=None)
    y1 = [fs.step(u_ctrl=[0.0, 0.0]).copy() for _ in range(5)]
    dev = fs.th.device()
    N = dev.N
    rng = np.random.default_rng(0)
    if not a.skip_parity:
        for k in (3, 8, 16, 24):
            dev.set_batch(k)
            B = rng.standard_normal((k, N))
            t0 = time.time()
            X = dev.solve_batch(SLOT_BDF2, B)
            errs = []
            for s in range(k):
                xs, _ = dev.solve(SLOT_BDF2, B[s])
                errs.append(rel(X[s], xs))
            print(f"solve_batch k={k}: max rel diff vs single solves {max(errs):.3e}  info {dev.batch_info()}", flush=True)
        # trajectories: 5 steps of k runs vs the single run above (all the same IC) and distinct ICs vs their own single runs
        k = 8
        bfs = BatchedFlowSolver(fs, k)
        ics = [ParamIC(xloc=2.0 + 0.3 * i, yloc=0.1 * i, radius=0.5 + 0.05 * i, amplitude=1.0 + 0.1 * i) for i in range(k)]
        bfs.initialize_time_stepping(ics=ics)
        us = [np.array([[0.02 * i * np.sin(0.3 * n), -0.01 * i * np.cos(0.2 * n)] for i in range(k)]) for n in range(12)]
        for n in range(12):
            bfs.step(us[n])
        for i in range(k):
            fs.params_ic = ics[i]
            fs.initialize_time_stepping(ic=None)
            for n in range(12):
                fs.step(us[n][i])
            ts1, tsb = fs.timeseries, bfs.timeseries(i)
            yc = [c for c in ts1.columns if c.startswith("y_meas_")]
            print(f"run {i}: y rel {rel(tsb[yc].to_numpy(), ts1[yc].to_numpy()):.3e}  dE rel {rel(tsb['dE'].to_numpy(), ts1['dE'].to_numpy()):.3e}  "
                  f"resid {bfs.solve_info[i, 1]:.2e}", flush=True)
        bfs.close()
    # throughput
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    fs.initialize_time_stepping(ic=None)
    for _ in range(50):
        fs.step(u_ctrl=[0.0, 0.0])
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fs.step(u_ctrl=[0.0, 0.0])
    single = a.steps / (time.perf_counter() - t0)
    print(f"single FlowSolver.step: {single:.0f} steps/s", flush=True)
    for k in [int(v) for v in a.ks.split(",")]:
        bfs = BatchedFlowSolver(fs, k)
        bfs.initialize_time_stepping(ics=[ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)] * k)
        u = np.zeros((k, 2))
        for _ in range(20):
            bfs.step(u)
        dts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            for _ in range(a.steps):
                bfs.step(u)
            dts.append((time.perf_counter() - t0) / a.steps)
        dt = min(dts)
        if a.reps > 1:
            print("   reps [sim-steps/s]:", " ".join(f"{k / d:.0f}" for d in dts), flush=True)
        ms = dev.bench_batch_apply(SLOT_BDF2, 100)
        info = dev.batch_info()
        gb = (info["factor_bytes"] + info["vector_bytes"]) / 1e9
        print(f"k={k:2d} KB={info['KB']:2d}: {k / dt:9.0f} sim-steps/s ({dt * 1e3:.3f} ms per batched step, x{k / dt / single:.2f} of single); "
              f"apply {ms * 1e3:.1f} us, {gb / ms:.2f} TB/s algorithmic ({info['block_launches']}+{info['fold_launches']} launches)", flush=True)
        bfs.close()


if __name__ == "__main__":
    main()
