#!/usr/bin/env python3
"""Headline benchmark: timesteps/s of ``FlowSolver.step`` on the cylinder Re=100 case (MI355X).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): shipped coarse mesh O1 (12 284 P2/P1 triangles, 56 203
DoFs), Re=100, dt=0.005, BDF1→BDF2, IC ParamIC(xloc=2, yloc=0, radius=0.5, amplitude=1), open loop
u_ctrl=[0,0], energy every step.  A "step" is one public ``FlowSolver.step()`` call: RHS element
loop with BC lifting → LU-grade solve (ND selected-inverse sweeps, residual monitored every step) → shift →
sensors → energy, synchronised back to the host every step as the reference's loop is.

One JSON line on rank 0 with the driver's keys plus
  roofline      dominant kernel (fc_nd_sweep): algorithmic bytes per launch ÷ mean launch duration
                measured with HIP events on the solver's stream (one pair around the back-to-back
                sweep launches of each apply) during an instrumented replay of the same K steps
  cpu_baseline  compiled single-thread restatement of the reference's per-step work (oracle/cpu_step.cpp element
                loop + SuperLU triangular solves, 1 core) timed on a bounded sample of the same workload; the numpy
                oracle's figure is reported next to it
  replicas      shared-operator batched stepping at N = 1: k = 1, 4, 8, 16, 32 lock-step replicas on one handle
                (replicas_steps_per_s, bytes per simulated step, roofline of the batched factor sweeps)
  spmv          CSR SpMV probe on the assembled BDF2 matrices of the five shipped meshes (O1: the run's own matrix; the
                others with a synthetic uniform base flow; cavity_fine is the one beyond the Infinity Cache), % of 8 TB/s
  krylov_factor_free  N = 1: the same public loop with NO factorisation (fc_setup_krylov: GMRES + SIMPLE / AMG preconditioner):
                iterations per step, steps/s, device bytes held
  other_configs N = 1: BASELINE configs 4 / 5 / 3 (refined cylinder, pinball closed loop, cavity_fine closed loop) on this GPU —
                steps/s of the synchronous public loop, the factor sweeps' roofline from their own bytes and HIP-event time (cavity_fine
                streams 4.7 GB of factors per step from HBM: the true HBM-streaming evidence), phase split, fc_refactor ms
N > 1 (torchrun, one rank per GPU): the SAME mesh is row-partitioned over the ranks (one sub-tree of the
elimination tree and its cells per GPU, the root's rows split over the ranks; three small RCCL all-reduces per step) —
total work fixed, "scaling": "strong", value = steps ÷ max time.  The shipped mesh is tiny (56 k
DoFs), so this is latency-bound; ``replicas_steps_per_s`` (N × the single-GPU rate measured on rank 0
in the same run) is reported next to it, and the meshes the partition is meant for run as further legs of the same protocol:
``strong_scaling_config4`` / ``_config5`` / ``_config3`` (steps/s over the N GPUs, the single-GPU rate of the same loop, and
``phase_us``: every rank's mean microseconds per step in {rhs, up-sweeps, exchange 1, root, exchange 2, down-sweeps, tail,
exchange 3, publish} from HIP-event marks inside fc_step).  ``--replicas`` times N independent simulations instead.
Rehearsals of the N > 1 path on ONE GPU (not measurements): ``FC_BENCH_THREAD_RANKS=8 python bench.py --gpus 8`` (ranks =
threads of one process) or ``FC_BENCH_SAME_DEVICE=1 torchrun --nproc-per-node 4 bench.py --gpus 4`` (process ranks over gloo).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
from flowcontrol_amd.examples.data import mesh_file  # noqa: E402
GOLDEN = ROOT / "tests" / "golden"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


REFINE = 0  # --refine K: BASELINE config 4 (O1 red-refined K times, cylinder midpoints projected to r = 0.5)


def refined_mesh_file(levels: int) -> Path:
    from flowcontrol_amd.examples.cylinder.cylinderflowsolver import refined_cylinder_mesh

    return refined_cylinder_mesh(levels)


def build_solver(device: int, distributed: bool = False, refine: int | None = None):
    from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC

    refine = REFINE if refine is None else refine
    meshpath = refined_mesh_file(refine) if refine else None
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tempfile.mkdtemp(prefix="fc_bench_"), num_steps=0, save_every=0, meshpath=meshpath)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    fs.distributed = distributed
    fs.th.device(device)
    golden = {0: "cylinder_O1.npz", 1: "cylinder_O1_refined1.npz"}.get(refine)
    if golden is None:
        # no golden base flow for this mesh: compute it as the reference's scripts do (setup, untimed)
        fs.compute_steady_state(method="picard", max_iter=3, tol=1e-7, u_ctrl=[0.0, 0.0])
        fs.compute_steady_state(method="newton", max_iter=25, u_ctrl=[0.0, 0.0], initial_guess=fs.fields.UP0)
    else:
        up0 = np.load(GOLDEN / golden)["UP0"]  # the oracle's base flow (tests/golden/make_*_fixtures.py)
        U0, P0 = Function(fs.W, up0).split()
        fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    return fs


def cpu_baseline(fs, n_steps: int = 700, warm: int = 3) -> dict:
    """Oracle leg: same mesh/dt/IC; element-loop RHS (numpy) + SuperLU triangular solves with the
    nested-dissection ordering (the best CPU ordering found) + sensors + energy;
    1 thread; mean over steps ≥ 3 so both factorisations are excluded (utils/fem.py:94-96)."""
    from threadpoolctl import threadpool_limits

    from oracle import ns_oracle as O

    with threadpool_limits(limits=1):  # "cores": 1 — no hidden BLAS / OpenMP threads
        return _cpu_baseline_1core(fs, O, n_steps, warm)


def _cpu_baseline_1core(fs, O, n_steps, warm) -> dict:
    """Two legs on one core: the COMPILED restatement (oracle/cpu_step.cpp: scalar C++ element loop + lifting, sensors,
    element-wise energy; SuperLU triangular solves) is the baseline value — it is what FFC-generated kernels + a sparse
    direct solver cost the reference per step; the numpy einsum oracle is timed next to it on a shorter sample."""
    from oracle import cpu_step

    th = fs.th
    d = O.Disc.from_taylor_hood(th)
    U0 = fs.fields.U0.vector().array()
    dofs, prof = fs._bc_tables()
    perm = th.device().perm
    ts = O.TimeStepper(d, fs.params_flow.Re, fs.params_time.dt, U0, dofs, prof, perm=perm)
    rows = [s.row(fs) for s in fs.params_control.sensor_list]
    cs = cpu_step.CompiledStepper(ts, rows)
    M = O.velocity_mass(d)

    def run(rhs, measure, energy, n):
        u_n = fs.fields.ic.u.vector().array().copy()
        u_nn = u_n.copy()
        order, times, t_asm, t_sol = 1, [], [], []
        y = dE = None
        for _ in range(n + warm):
            t0 = time.perf_counter()
            b = rhs(order, u_n, u_nn, np.zeros(2))
            t1 = time.perf_counter()
            up = ts.solve(order, b)
            t2 = time.perf_counter()
            t_asm.append(t1 - t0), t_sol.append(t2 - t1)
            order = 2
            u_nn, u_n = u_n, up[: 2 * th.nn].copy()
            y = measure(up)
            dE = energy(u_n)
            times.append(time.perf_counter() - t0)
        return float(np.mean(times[warm:])), float(np.mean(t_asm[warm:])), float(np.mean(t_sol[warm:])), y, dE

    mean_c, asm_c, sol_c, y_c, dE_c = run(cs.rhs, cs.sensors, cs.energy, n_steps)
    n_np = max(10, n_steps // 20)  # ~10 s of compiled stepping + ~3 s of the numpy oracle on one core
    mean_n, asm_n, sol_n, y_n, dE_n = run(ts.rhs, lambda up: np.array([w @ up[i] for i, w in rows]),
                                          lambda u: 0.5 * u @ (M @ u), n_np)
    return {
        "value": 1.0 / mean_c,
        "unit": "timesteps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n_steps} steps of the same workload after {warm} warm-up steps (factorisations excluded), compiled C++ restatement "
        f"(oracle/cpu_step.cpp, g++ -O3 -march=x86-64-v3 (portable to the GPU box's host CPU)): {mean_c * 1e3:.2f} ms/step = element-loop RHS + lifting {asm_c * 1e3:.2f} ms + SuperLU "
        f"triangular solves (ND ordering) {sol_c * 1e3:.2f} ms + sensors/energy; 1 thread (threadpool_limits); host has {os.cpu_count()} logical cores",
        "numpy_oracle": {"value": 1.0 / mean_n, "ms_per_step": mean_n * 1e3, "rhs_ms": asm_n * 1e3, "solve_ms": sol_n * 1e3, "steps": n_np,
                         "note": "ns_oracle.py (einsum element loop): the parity specification, not a fair timing baseline"},
        "_y_last": np.asarray(y_c).tolist(),
        "_dE_last": float(dE_c),
    }


def spmv_probe(fs, include_large: bool) -> dict:
    from flowcontrol_amd._lib import SLOT_BDF2, SLOT_SCRATCH

    dev = fs.th.device()
    out = {}
    ms = dev.bench_spmv(SLOT_BDF2, 1000)
    byt = dev.nnz * 12 + dev.N * 16 + (dev.N + 1) * 4
    out["O1_bdf2"] = {"nnz": dev.nnz, "N": dev.N, "bytes": byt, "us": ms * 1e3, "GB/s": byt / ms / 1e6,
                      "pct_hbm_peak": 100 * byt / ms / 1e6 / HBM_PEAK_GBS, "note": "21 MB: L2/Infinity-Cache resident"}
    if include_large:
        # the assembled BDF2 matrices of the other shipped meshes (synthetic uniform base flow: the
        # pattern and the value count are what matters for an SpMV), x = default_rng(0).standard_normal(N)
        from flowcontrol_amd.device import DeviceSolver
        from flowcontrol_amd.examples.cylinder.cylinderflowsolver import refined_cylinder_mesh
        from flowcontrol_amd.fem.mesh import read_xdmf_mesh
        from flowcontrol_amd.fem.spaces import TaylorHood

        cases = [("O1_refined1_bdf2", refined_cylinder_mesh(1), 300.0, 0.01), ("cavity_coarse_bdf2", mesh_file("cavity_coarse"), 3750.0, 1.0 / 7500.0),
                 ("pinball_bdf2", mesh_file("mesh_middle_gmsh"), 300.0, 0.01), ("cavity_fine_bdf2", mesh_file("cavity_fine"), 3750.0, 1.0 / 7500.0)]
        for key, path, mass, nu in cases:
            th = TaylorHood(read_xdmf_mesh(path))
            big = DeviceSolver(th, dev_index(fs))
            U = np.r_[np.ones(th.nn), np.zeros(th.nn)]
            big.assemble_matrix(SLOT_SCRATCH, mass=mass, nu=nu, adv=U, lin=U)
            big.spmv(SLOT_SCRATCH, np.random.default_rng(0).standard_normal(big.N))
            ms = big.bench_spmv(SLOT_SCRATCH, 1000 if big.nnz < 10_000_000 else 200)
            byt = big.nnz * 12 + big.N * 16 + (big.N + 1) * 4
            out[key] = {"nnz": big.nnz, "N": big.N, "bytes": byt, "us": ms * 1e3, "GB/s": byt / ms / 1e6,
                        "pct_hbm_peak": 100 * byt / ms / 1e6 / HBM_PEAK_GBS,
                        "note": ("> 256 MiB Infinity Cache: HBM streaming" if byt > 256 * 2**20 else "Infinity-Cache resident") + "; synthetic uniform base flow"}
            big.close()
    return out


def batched_replicas(fs, steps: int, single_rate: float, ks=(1, 4, 8, 16, 32)) -> dict:
    """Shared-operator batched stepping (fc_step_batch): k lock-step replicas of the headline workload on ONE handle —
    the reference's IC / controller sweeps run k FlowSolver instances instead (batch_run_lidcavity.py:197-215,
    utils/optim.py:95-102).  Every batched step is one public BatchedFlowSolver.step() (host-synchronised, measurements and
    energy of all k runs returned).  The factor sweeps are block products on the fp64 matrix cores; their roofline entry is
    HBM: (factor bytes + operand / result / fold bytes of one batched apply) / apply time by HIP events."""
    from flowcontrol_amd._lib import SLOT_BDF2
    from flowcontrol_amd.batch import BatchedFlowSolver

    dev = fs.th.device()
    out = {"note": "k replicas of the headline workload advanced together on one handle; replicas_steps_per_s = k x batched steps/s",
           "single_steps_per_s": single_rate, "per_k": {}}
    for k in ks:
        bfs = BatchedFlowSolver(fs, k)
        bfs.initialize_time_stepping(ics=[fs.params_ic] * k)
        u = np.zeros((k, 2))
        for _ in range(20):
            bfs.step(u)
        t0 = time.perf_counter()
        for _ in range(steps):
            bfs.step(u)
        dt = (time.perf_counter() - t0) / steps
        ms_apply = dev.bench_batch_apply(SLOT_BDF2, 100)
        info = dev.batch_info()
        apply_bytes = info["factor_bytes"] + info["vector_bytes"]
        # per simulated step, besides the apply: element loop 448 B/cell, gather ~40 B/row, tail: matrix 12 nnz / k + ~100 B/row
        nnz = dev.nnz
        other = 448.0 * fs.th.nc + 140.0 * fs.th.N + 12.0 * nnz / k + 170.0 * fs.th.nc
        out["per_k"][str(k)] = {
            "k": k, "KB": info["KB"], "replicas_steps_per_s": k / dt, "ms_per_batched_step": dt * 1e3, "x_single": k / dt / single_rate,
            "bytes_per_simulated_step": apply_bytes / k + other,
            "y_last_run0": bfs.y_meas[0].tolist(), "residual_max": float(np.max(bfs.solve_info[:, 1])),
            "roofline": {"bound": "hbm", "kernel": f"fc_nd_block_b<{info['KB']}> + fc_nd_fold_b (batched factor sweeps, v_mfma_f64_16x16x4_f64)",
                         "achieved": apply_bytes / ms_apply / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": apply_bytes / ms_apply / 1e6 / HBM_PEAK_GBS,
                         "traffic": None, "apply_us": ms_apply * 1e3, "factor_bytes_per_apply": info["factor_bytes"],
                         "vector_bytes_per_apply": info["vector_bytes"], "launches_per_apply": info["block_launches"] + info["fold_launches"],
                         "mfma_tflops": 2.0 * info["factor_bytes"] / 8.0 * info["KB"] / (ms_apply * 1e-3) / 1e12,
                         "mfma_peak_tflops_f64": 78.6},
        }
        bfs.close()
    return out


def dev_index(fs) -> int:
    return int(fs.th.device().device_index)


# ── the workloads: BASELINE configs[1] (headline), [3] (refined cylinder), [4] (pinball), [2] (open cavity) ─────────────────
class Case:
    """One workload: ``prepare(comm, device)`` → data every rank needs (e.g. a base flow computed once and broadcast),
    ``make(device, shared)`` → a FlowSolver ready to step, ``controller(fs)`` → ``fs -> u_ctrl`` of the loop."""

    def __init__(self, key, workload, make, controller=None, prepare=None, steps_cap=300, warm=10):
        self.key, self.workload, self.make, self.steps_cap, self.warm = key, workload, make, steps_cap, warm
        self._controller, self._prepare = controller, prepare

    def prepare(self, comm, device):
        return self._prepare(comm, device) if self._prepare else None

    def controller(self, fs):
        if self._controller:
            return self._controller(fs)
        u0 = np.zeros(fs.params_control.actuator_number)
        return lambda: u0


def _make_cylinder(refine):
    def make(device, shared):
        return build_solver(device, distributed=True, refine=refine)

    return make


def _make_pinball(device, shared):
    from flowcontrol_amd.actuator import CYLINDER_ACTUATION_MODE
    from flowcontrol_amd.examples.pinball.pinballflowsolver import PinballFlowSolver
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC

    g = np.load(GOLDEN / "pinball_re100_rotation.npz")
    fs = PinballFlowSolver.make_default(Re=100, mode_actuation=CYLINDER_ACTUATION_MODE.ROTATION, path_out=tempfile.mkdtemp(prefix="fc_bench_"), num_steps=0,
                                        save_every=0)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    fs.th.device(device)
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    return fs


def _pinball_controller(fs):
    """Three sensors → 3-in / 3-out LTI controller → three rotating cylinders (flowcontrol_amd/examples/pinball/scenarios.py: the system of the golden fixture; its
    output gain, sized for the first 50 steps, is scaled down so that a long run stays a small-amplitude closed loop)."""
    from flowcontrol_amd.controller import Controller
    from flowcontrol_amd.examples.pinball.scenarios import PINBALL_K

    K = Controller(A=PINBALL_K["A"], B=PINBALL_K["B"], C=PINBALL_K["C"] / 2.0e6, D=PINBALL_K["D"])
    y0, dt = fs.y_meas.copy(), fs.params_time.dt
    return lambda: np.asarray(K.step(y=(fs.y_meas - y0)[:3], dt=dt)).reshape(-1)


def _cavity_fs(device):
    from flowcontrol_amd.examples.cavity.cavityflowsolver import CavityFlowSolver

    fs = CavityFlowSolver.make_default(Re=7500, path_out=tempfile.mkdtemp(prefix="fc_bench_"), num_steps=0, save_every=0,
                                       meshpath=mesh_file("cavity_fine"))
    fs.th.device(device)
    return fs


def _prepare_cavity(comm, device):
    """cavity_fine ships no base flow: rank 0 runs a few Picard sweeps on its own GPU (a throughput run, not a converged
    base flow — every iteration assembled / factorised / solved on the device) and the field is broadcast."""
    up0 = None
    if comm.rank == 0:
        fs = _cavity_fs(device)
        fs.distributed = False
        fs.compute_steady_state(method="picard", max_iter=4, tol=1e-7, u_ctrl=[0.0])
        up0 = fs.fields.UP0.vector().get_local().copy()
        fs.th.release_device()
    return comm.bcast(up0, src=0)


def _make_cavity(device, shared):
    from flowcontrol_amd.fem.spaces import Function

    fs = _cavity_fs(device)
    U0, P0 = Function(fs.W, shared).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    return fs


def _cavity_controller(fs):
    from flowcontrol_amd.controller import Controller

    K = Controller(A=[[-100.0]], B=[[1.0]], C=[[0.5]], D=[[0.0]])  # first-order low-pass on the wall-shear sensor
    y0, dt = fs.y_meas.copy(), fs.params_time.dt
    return lambda: np.asarray(K.step(y=(fs.y_meas - y0)[:1], dt=dt)).reshape(-1)


CASES = {
    "config4": Case("config4", "cylinder Re=100, mesh O1 red-refined x1, open loop, same protocol as the headline", _make_cylinder(1), steps_cap=300),
    "config5": Case("config5", "fluidic pinball Re=100, ROTATION, 3 actuators <- 3-in/3-out LTI Controller <- 3 sensors (closed loop)", _make_pinball,
                    _pinball_controller, steps_cap=300),
    "config3": Case("config3", "open cavity Re=7500 on cavity_fine, Gaussian FORCE actuator <- low-pass Controller <- wall-shear sensor (closed loop); "
                    "base flow = 4 device Picard sweeps (throughput run)", _make_cavity, _cavity_controller, _prepare_cavity, steps_cap=100),
}


def run_case(case, comm, device, steps, with_roofline):
    """One workload on the ranks of ``comm`` (world 1: the single-GPU run): synchronous public loop  y -> controller ->
    FlowSolver.step, timed between barriers, max over ranks; then an instrumented replay for the per-rank phase split
    (HIP-event marks inside fc_step) and, single GPU, the factor sweeps' roofline."""
    from flowcontrol_amd._lib import SLOT_BDF2

    shared = case.prepare(comm, device)
    fs = case.make(device, shared)
    fs.comm = comm if comm.world > 1 else None
    fs.distributed = comm.world > 1
    ctrl = case.controller(fs)
    t0 = time.time()
    fs.step(ctrl())  # assembles and factorises both operators (partitioned: joins the ranks first)
    setup_s = time.time() - t0
    for _ in range(case.warm):
        fs.step(ctrl())
    steps = max(10, min(steps, case.steps_cap))
    comm.barrier()
    t1 = time.perf_counter()
    fs.residual_max = 0.0  # (the solver keeps the running maximum itself: reading solve_info after every step would make the host wait for the
    for _ in range(steps):  #  residual monitor of the step that just ended, which the loop under test never does)
        fs.step(ctrl())
    comm.barrier()
    elapsed = comm.allreduce_max(time.perf_counter() - t1)
    worst = float(np.nanmax([float(fs.solve_info[1]), fs.residual_max]))  # (solve_info collects the last step's: NaN off the monitor's cadence)
    dev = fs.th.device()
    out = {"workload": f"{case.workload}; {fs.th.nc} cells, {fs.th.N} dofs, dt={fs.params_time.dt}", "n_gpus": comm.world, "steps": steps,
           "steps_per_s": steps / elapsed, "ms_per_step": 1e3 * elapsed / steps, "scaling": "strong" if comm.world > 1 else "none",
           "worst_relative_residual": worst,
           # cadence of the residual monitor in the timed loop (the solver's default: every step while the factors stay in the Infinity Cache,
           # every 8th step where they stream from HBM; the reference forms no residual at all) -- finiteness, sensors and energy: every step
           "residual_monitor_every": (getattr(fs, "check_residual_every", None) if getattr(fs, "check_residual_every", None) is not None
                                      else (8 if (comm.world == 1 and dev.factor_storage(1)[1] > 268435456) else 1)),
           "y_last": np.asarray(fs.y_meas).tolist(), "setup_s": setup_s,
           "refactor_ms": {str(k): float(v) for k, v in dev.refactor_ms.items()}}
    # per-rank phase split (instrumented replay: event marks at the phase boundaries of every step)
    n_rep = max(5, min(steps, 50))
    dev.set_phase_timing(True)
    for _ in range(n_rep):
        fs.step(ctrl())
    ph = dev.get_phase_timing()
    dev.set_phase_timing(False)
    names = list(dev.PHASES)
    rows = comm.gather_rows(np.array([ph[k] for k in names]))
    out["phase_us"] = {"phases": names, "per_rank": [[round(float(v), 2) for v in r] for r in rows],
                       "note": "mean microseconds per step and rank from HIP-event marks inside fc_step (instrumented replay of "
                               f"{n_rep} steps; exchange phases include the wait for the slowest rank)"}
    if comm.world > 1:
        part = dev.part
        info = comm.gather_rows(np.array([part.local_cells.size, dev.local_factor_nnz, dev._n_factor_values, dev.partition_info()["matrix_cells"]], dtype=float))
        ci = dev.comm_info()
        out["partition"] = {"root_dofs": int(part.ar_n), "exchanges_per_step": 3, "exchange_transport": ci["transport"],
                            "rccl_ranks": ci["nranks"] if ci["transport"] == "rccl" else None,
                            "local_cells": [int(v) for v in info[:, 0]], "local_factor_nnz": [int(v) for v in info[:, 1]],
                            "stored_factor_nnz": [int(v) for v in info[:, 2]], "matrix_cells": [int(v) for v in info[:, 3]]}
    elif with_roofline:
        dev.set_timing(True)
        for _ in range(n_rep):
            fs.step(ctrl())
        tim = dev.get_timing()
        dev.set_timing(False)
        sweep_bytes, _ = dev.algorithmic_bytes(SLOT_BDF2)
        apply_ms = tim["sweep_ms"] / n_rep
        out["roofline"] = {"bound": "hbm", "kernel": "fc_nd_sweep + fc_nd_down_block + fc_nd_flat_block (factor sweeps)", "achieved": sweep_bytes / apply_ms / 1e6, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": sweep_bytes / apply_ms / 1e6 / HBM_PEAK_GBS, "traffic": None, "bytes_per_apply": sweep_bytes,
                           "apply_us": 1e3 * apply_ms, "launches_per_apply": tim["sweep_launches"] / n_rep, "factor_values": int(dev._n_factor_values),
                           "note": ("factors exceed the 256 MiB Infinity Cache: HBM streaming" if sweep_bytes > 256 * 2**20 else "factors fit the 256 MiB Infinity Cache")}
        out["tail_us"] = out["phase_us"]["per_rank"][0][names.index("tail")]
    fs.th.release_device()
    return out


def krylov_factor_free(device: int, steps: int, factor_bytes: int | None) -> dict:
    """The Krylov mode that factorises NOTHING (fc_setup_krylov: device GMRES right-preconditioned by the SIMPLE / AMG block
    preconditioner) on the headline workload: iterations per step, steps/s of the same public loop, device bytes held."""
    from flowcontrol_amd._lib import SLOT_BDF2
    from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC

    fs = CylinderFlowSolver.make_default(Re=100, path_out=tempfile.mkdtemp(prefix="fc_bench_"), num_steps=0, save_every=0)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    fs.krylov_precond, fs.krylov_method, fs.krylov_max_iter, fs.krylov_rtol = "schur_amg", "gmres", 300, 1e-10
    fs.th.device(device)
    U0, P0 = Function(fs.W, np.load(GOLDEN / "cylinder_O1.npz")["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    u0 = np.zeros(2)
    for _ in range(5):
        fs.step(u0)
    its, res = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        fs.step(u0)
        its.append(int(fs.solve_info[0])), res.append(float(fs.solve_info[1]))
    dt = time.perf_counter() - t0
    info = fs.th.device().krylov_info(SLOT_BDF2)
    out = {"preconditioner": f"SIMPLE: {info['jacobi_sweeps']} damped-Jacobi sweeps on the velocity block + one smoothed-aggregation AMG V(1,1)-cycle on "
                             f"B diag(F)^-1 Bt ({info['amg_levels']} levels, coarsest {info['coarsest_rows']} rows), right-preconditioned GMRES, warm start; nothing is factorised",
           "rtol": fs.krylov_rtol, "steps": steps, "iterations_per_step": float(np.mean(its)), "iterations_max": int(np.max(its)),
           "steps_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps, "rel_residual_max": float(np.max(res)),
           "bytes_held": info["bytes"], "launches_per_precond_apply": info["launches_per_apply"], "setup_ms": info["setup_ms"],
           "factor_bytes_of_the_direct_mode": factor_bytes, "y_last": fs.y_meas.tolist()}
    fs.th.release_device()
    try:
        out["cavity_fine"] = krylov_factor_free_large(device)
    except Exception as e:  # the probe of the big mesh must not take the headline leg down
        out["cavity_fine"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def krylov_factor_free_large(device: int) -> dict:
    """The point of the mode is memory: one factorisation-free solve of a BDF2 operator on BASELINE config 3's mesh (cavity_fine,
    877 k dofs) from a zero guess -- iterations, milliseconds, device bytes held -- next to the factor bytes of the direct mode."""
    from flowcontrol_amd._lib import SLOT_BDF2
    from flowcontrol_amd.device import DeviceSolver
    from flowcontrol_amd.fem.mesh import read_xdmf_mesh
    from flowcontrol_amd.fem.spaces import TaylorHood

    th = TaylorHood(read_xdmf_mesh(mesh_file("cavity_fine")))
    dev = DeviceSolver(th, device)
    try:
        x = th.node_coords
        m = th.mesh
        be = m.boundary_edges()
        be = be[m.edge_midpoints()[be, 0] < m.coords[:, 0].max() - 1e-9]  # (one open side: the operator is regular without a pressure pin)
        nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
        dofs = np.sort(np.r_[nodes, nodes + th.nn])
        dev.set_bc(dofs, np.zeros((dofs.size, 1)))
        U0 = np.r_[0.5 * np.sin(2.0 * x[:, 0]) * np.cos(1.3 * x[:, 1]), 0.3 * np.cos(1.1 * x[:, 0] + 0.2) * np.sin(2.0 * x[:, 1])]
        dev.assemble_matrix(SLOT_BDF2, mass=1.5 / 4e-4, nu=1.0 / 7500.0, adv=U0, lin=U0)
        dev.apply_bc(SLOT_BDF2)
        b = np.random.default_rng(5).standard_normal(dev.N)
        b[dofs] = 0.0
        info = dev.setup_krylov(SLOT_BDF2, sweeps=2, method="gmres", max_iter=300, rtol=1e-10)
        dev.solve(SLOT_BDF2, b)
        t0 = time.perf_counter()
        xs, si = dev.solve(SLOT_BDF2, b)
        ms = 1e3 * (time.perf_counter() - t0)
        dev.setup_solver(SLOT_BDF2)  # the same slot factorised: what the direct mode holds, and its answer
        xd, _ = dev.solve(SLOT_BDF2, b)
        return {"dofs": int(dev.N), "iterations_from_zero_guess": int(si[0]), "rel_residual": float(si[1]), "ms_per_solve": ms,
                "bytes_held": int(info["bytes"]), "factor_bytes_of_the_direct_mode": int(8 * dev.factor_nnz[SLOT_BDF2]),
                "rel_diff_to_the_direct_solve": float(np.linalg.norm(xs - xd) / np.linalg.norm(xd))}
    finally:
        dev.close()


def headline(comm, device, args):
    """The driver's contract: W untimed warm-up steps, EXACTLY K timed steps between barrier + synchronize, max over ranks."""
    import torch

    def barrier():
        comm.barrier()
        torch.cuda.synchronize()

    partitioned = comm.world > 1 and not args.replicas
    t_setup = time.time()
    u0 = np.zeros(2)
    # a failure of the partitioned path is a failure of the run: no silent fall-back to replicas
    fs = build_solver(device, distributed=partitioned)
    fs.comm = comm if partitioned else None
    fs.step(u0)  # BDF1 step: assembles, factorises (partitioned: creates the RCCL communicator, runs its pre-flight); setup, untimed
    log(f"[rank {comm.rank}] setup {time.time() - t_setup:.1f}s; N={fs.th.N}; partitioned={partitioned}")
    for _ in range(max(args.warmup - 1, 0)):
        fs.step(u0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fs.step(u0)
    barrier()
    elapsed = comm.allreduce_max(time.perf_counter() - t0)
    return fs, elapsed, partitioned


def run_rank(comm, args, device):
    """Everything one rank does; rank 0 returns the result dictionary."""
    world, rank = comm.world, comm.rank
    u0 = np.zeros(2)
    fs, elapsed, partitioned = headline(comm, device, args)
    y_last = fs.y_meas.copy()
    mesh_nc, mesh_N, resid_last = fs.th.nc, fs.th.N, float(fs.solve_info[1])

    part_info, single_rate, headline_phases, strong = None, None, None, {}
    if partitioned:
        dev = fs.th.device()
        ci = dev.comm_info()
        part_info = {"local_cells": int(dev.part.local_cells.size), "root_dofs": int(dev.part.ar_n),
                     "local_factor_nnz": int(dev.local_factor_nnz),  # factor values this rank sweeps per solve
                     "stored_factor_nnz": int(dev._n_factor_values),  # ... and stores: its sub-tree + its rows of the root block
                     "matrix_cells": int(dev.partition_info()["matrix_cells"]),  # cells whose element matrices it assembles
                     "exchanges_per_step": 3,
                     # read back from the communicator inside the library (ncclCommCount / ncclCommUserRank), not from the environment
                     "rccl_ranks": ci["nranks"] if ci["transport"] == "rccl" else None, "exchange_transport": ci["transport"],
                     # not None: the in-library RCCL communicator could not be created; the run went on over the host exchange (and is NOT an xGMI figure)
                     "exchange_fallback": getattr(fs, "exchange_fallback", None)}
        n_rep = max(5, min(args.steps, 50))
        dev.set_phase_timing(True)
        for _ in range(n_rep):
            fs.step(u0)
        ph = dev.get_phase_timing()
        dev.set_phase_timing(False)
        rows = comm.gather_rows(np.array([ph[k] for k in dev.PHASES]))
        headline_phases = {"phases": list(dev.PHASES), "per_rank": [[round(float(v), 2) for v in r] for r in rows]}
        fs.th.release_device()
        fs = None
        comm.barrier()
        # the meshes the row partition is meant for, same protocol, fewer steps; the headline stays the O1 figure BASELINE.json names
        for key in ([] if REFINE else [k for k in ("config4", "config5", "config3") if k not in args.skip]):
            try:
                leg = run_case(CASES[key], comm, device, args.steps, with_roofline=False)
            except Exception as err:  # a leg that fails is reported, the run and the other legs go on
                leg = {"error": repr(err)}
                log(f"[rank {rank}] strong-scaling leg {key} failed: {err!r}")
            strong[key] = leg
            comm.barrier()
        if rank == 0 and not args.no_extras:
            # single-GPU rates of the same workloads, measured in this run on rank 0's GPU (the other ranks wait)
            one = SingleCommProxy()
            for key, leg in strong.items():
                if "error" not in leg:
                    leg["single_gpu_steps_per_s"] = run_case(CASES[key], one, device, args.steps, with_roofline=False)["steps_per_s"]
            fs = build_solver(device, distributed=False)
            fs.step(u0)
            for _ in range(20):
                fs.step(u0)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                fs.step(u0)
            single_rate = args.steps / (time.perf_counter() - t1)
        comm.barrier()

    result = None
    roofline = phases = spmv = cpu = replicas = other = kff = None
    t_batched = None
    depth_str = None
    if rank == 0 and not args.no_extras:
        from flowcontrol_amd._lib import SLOT_BDF2

        dev = fs.th.device()
        # batched (no per-step host sync) throughput of the same kernels
        tb = time.perf_counter()
        fs.run(args.steps, u0)
        t_batched = time.perf_counter() - tb
        # instrumented replay: HIP-event pair around every sweep / SpMV launch on the solver stream
        dev.set_timing(True)
        for _ in range(args.steps):
            fs.step(u0)
        tim = dev.get_timing()
        dev.set_timing(False)
        sweep_bytes, spmv_bytes = dev.algorithmic_bytes(SLOT_BDF2)
        applies = args.steps  # one factor apply per step (direct mode, no refinement sweeps)
        n_stage = tim["sweep_launches"] / max(applies, 1)  # launches per apply: 2 depth + 1 in the row form of the up-sweep (+ depth folds in the column form)
        mean_launch_ms = tim["sweep_ms"] / max(tim["sweep_launches"], 1)
        bytes_per_launch = sweep_bytes / n_stage
        achieved = bytes_per_launch / mean_launch_ms / 1e6  # GB/s
        apply_us = 1e3 * tim["sweep_ms"] / max(applies, 1)
        tree = dev.tree_info(min_tree=True)
        traffic, traffic_commit = None, None
        tfile = ROOT / "profiles" / "traffic.json"  # PMC passes of the DEFAULT workload (scripts/profile_gpu.sh)
        if tfile.exists() and REFINE == 0 and not partitioned:
            try:
                tj = json.loads(tfile.read_text())
                traffic, traffic_commit = tj.get("fc_nd_sweep_bytes_per_launch"), tj.get("commit")
            except Exception:
                traffic = None
        roofline = {
            "bound": "hbm", "kernel": "fc_nd_sweep + fc_nd_down_block + fc_nd_flat_block (factor sweeps)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": (f"profiles/traffic.json (builder's rocprofv3 --pmc passes of this workload at commit {traffic_commit}, NOT measured in this run)"
                               if traffic else None),
            "bytes_per_launch": bytes_per_launch, "launches_per_step": tim["sweep_launches"] / args.steps,
            "mean_launch_us": mean_launch_ms * 1e3, "factor_nnz": dev.factor_nnz.get(SLOT_BDF2),
            "applies_per_step": applies / args.steps,
            # bisections fused per level of the elimination tree, root first (fc_get_tree_info; 2 x levels + 1 launches)
            "tree_bits": tree["bits"],
            # fractions that stay comparable across rounds (the design's own byte count grows with the fill of the tree it picks): the
            # apply's wall time per call, the same time priced at 8 B per stored factor value only, and at the bytes of the
            # all-binary-pairs tree [2,2,...] of the same depth (the shape with the least fill; bytes scaled by the ratio of factor values)
            "apply_us": apply_us,
            "frac_values_only": 8.0 * dev.factor_nnz.get(SLOT_BDF2) / (apply_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "frac_min_tree": (sweep_bytes * tree["nnz_min_tree"] / dev.factor_nnz.get(SLOT_BDF2)) / (apply_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "factor_nnz_min_tree": tree["nnz_min_tree"],
            "spmv_in_step": ({"bytes": spmv_bytes, "mean_us": 1e3 * tim["spmv_ms"] / tim["spmv_launches"],
                              "GB/s": spmv_bytes / (tim["spmv_ms"] / tim["spmv_launches"]) / 1e6}
                             if tim["spmv_launches"] and tim["spmv_ms"] > 0 else
                             "fused into fc_tail (residual monitor + energy, one launch)"),
            "note": (f"factors ({sweep_bytes / 1e6:.0f} MB) are re-read every step and largely stay in the 256 MiB Infinity Cache"
                     if sweep_bytes < 256e6 else
                     f"factors ({sweep_bytes / 1e6:.0f} MB) exceed the 256 MiB Infinity Cache: HBM streaming"),
        }
        if not partitioned and dev.world == 1 and not args.no_replicas:
            replicas = batched_replicas(fs, steps=min(args.steps, 400), single_rate=args.steps / elapsed)
        dev.set_phase_timing(True)  # the step's phase split from HIP-event marks inside fc_step (instrumented replay)
        for _ in range(200):
            fs.step(u0)
        phases = dev.get_phase_timing()
        dev.set_phase_timing(False)
        spmv = spmv_probe(fs, include_large=not args.no_large_spmv)
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(fs)
            cpu.pop("_y_last"), cpu.pop("_dE_last")
        depth_str = f"ND selected-inverse depth {dev.depth}, {fs.refine_steps} refinement"
        factor_bytes = 8 * int(dev.factor_nnz.get(SLOT_BDF2) or 0)
        fs.th.release_device()
        if world == 1 and not args.no_krylov:
            try:
                kff = krylov_factor_free(device, min(args.steps, 200), factor_bytes)
            except Exception as err:
                kff = {"error": repr(err)}
                log(f"krylov_factor_free failed: {err!r}")
        if world == 1 and not args.no_other_configs and REFINE == 0:
            # BASELINE configs 3 / 4 / 5 on this GPU, in the driver-timed run: closed-loop steps/s, the factor sweeps' roofline from
            # their own bytes and HIP-event time, the step's phase split (tail = fc_tail + fc_final), fc_refactor milliseconds
            other = {}
            one = SingleCommProxy()
            for key in ("config4", "config5", "config3"):
                if key in args.skip:
                    continue
                try:
                    other[key] = run_case(CASES[key], one, device, args.steps, with_roofline=True)
                except Exception as err:
                    other[key] = {"error": repr(err)}
                    log(f"other_configs {key} failed: {err!r}")
    if rank == 0:
        value = (1 if partitioned else world) * args.steps / elapsed
        c4 = strong.get("config4")
        result = {
            "metric": "timesteps/s (cylinder Re=100, fixed mesh)",
            "value": value,
            "unit": "timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if partitioned else ("none" if world == 1 else "weak"),
            "vs_baseline": None,
            "dtype": "f64",
            "data": "the reference's shipped mesh O1 (converted data file) + base flow computed by the oracle (golden fixture); "
                    "IC and actuation as run_cylinder_example.py",
            "config": {
                "workload": f"cylinder Re=100, mesh O1{' red-refined x' + str(REFINE) if REFINE else ''} ({mesh_nc} cells, {mesh_N} dofs), "
                "dt=0.005, BDF2, open loop, IC div-free vortex (2,0) r=0.5, sensors+energy every step",
                "parallelism": "single GPU" if world == 1 else (
                    f"row-partitioned over {world} GPUs: one elimination sub-tree + its cells per rank, the root's rows split "
                    f"over the ranks, 3 RCCL all-reduces per step" if partitioned
                    else f"{world} independent replicas (no data-path collective)"),
                "partition": part_info,
                "solver": depth_str,
                "ranks_are": args.ranks_are,
            },
            "batched_steps_per_s": (args.steps / t_batched) if t_batched else None,
            "strong_scaling_config4": c4,
            "strong_scaling_config5": strong.get("config5"),
            "strong_scaling_config3": strong.get("config3"),
            "other_configs": other,
            "replicas_steps_per_s": (world * single_rate) if single_rate else (replicas["per_k"]["8"]["replicas_steps_per_s"] if replicas else None),
            "replicas": replicas,
            "roofline": roofline,
            "phase_us": headline_phases if headline_phases is not None else (
                {"phases": [k for k in phases if k != "steps"], "per_rank": [[round(v, 2) for k, v in phases.items() if k != "steps"]]} if phases else None),
            "spmv": spmv,
            "cpu_baseline": cpu,
            "speedup_vs_cpu_baseline": (value / cpu["value"]) if cpu else None,
            "krylov_factor_free": kff,
            "solve_rel_residual_pre_refine": resid_last,
            "y_last": y_last.tolist(),
            # compact copy of other_configs at the END of the line (a log tail that cuts the long line still shows it): [steps/s, sweep roofline fraction]
            "other_configs_summary": ({k: ([round(v["steps_per_s"], 1), round(v["roofline"]["frac"], 3)] if "error" not in v else "error")
                                       for k, v in other.items()} if other else None),
        }
    comm.barrier()
    return result


def SingleCommProxy():
    from flowcontrol_amd.comm import SingleComm

    return SingleComm()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-spmv", action="store_true")
    ap.add_argument("--replicas", action="store_true", help="N > 1: independent replicas instead of the partitioned run")
    ap.add_argument("--no-replicas", action="store_true", help="skip the batched-replicas section (profiling passes of the single-simulation step)")
    ap.add_argument("--no-extras", action="store_true", help="skip rank 0's single-GPU extras (roofline replay, SpMV probe, batched replicas, CPU baseline)")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the BASELINE config 3 / 4 / 5 section")
    ap.add_argument("--no-krylov", action="store_true", help="N = 1: skip the factorisation-free Krylov leg")
    ap.add_argument("--skip", default="", help="comma-separated legs to leave out: config3,config4,config5")
    ap.add_argument("--no-config4", action="store_true", help="N > 1: skip the strong-scaling leg on the BASELINE config-4 mesh (same as --skip config4)")
    ap.add_argument("--refine", type=int, default=0, help="red-refine the O1 mesh K times (BASELINE config 4: K=1); not the headline workload")
    args = ap.parse_args()
    args.skip = {k for k in args.skip.split(",") if k} | ({"config4"} if args.no_config4 else set())
    global REFINE
    REFINE = args.refine

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    thread_ranks = int(os.environ.get("FC_BENCH_THREAD_RANKS", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if thread_ranks > 1:
        # rehearsal of an N-GPU run on ONE GPU: the ranks are threads of this process, every rank its own handle on GPU 0, the
        # exchanges staged through the host (a box admits only a few processes on its card, so eight process ranks cannot share it)
        from flowcontrol_amd.comm import run_threaded

        if world > 1:
            raise SystemExit("FC_BENCH_THREAD_RANKS is a single-process rehearsal: do not combine it with torchrun")
        torch.cuda.set_device(0)
        args.ranks_are = f"{thread_ranks} threads of one process sharing GPU 0, exchanges through the host (rehearsal, NOT a multi-GPU measurement)"
        result = run_threaded(thread_ranks, run_rank, args, 0, timeout=1800.0)[0]
        print(json.dumps(result), flush=True)
        return
    import torch.distributed as dist

    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}")
    same_dev = os.environ.get("FC_BENCH_SAME_DEVICE", "0") == "1"  # rehearsal on a 1-GPU box: all ranks on GPU 0, gloo
    if same_dev:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        from flowcontrol_amd.comm import TorchComm

        if same_dev:
            dist.init_process_group(backend="gloo")
            args.ranks_are = f"{world} processes sharing GPU 0, exchanges through the host over gloo (rehearsal, NOT a multi-GPU measurement)"
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
            args.ranks_are = "one process per GPU, RCCL"
        comm = TorchComm(local if not same_dev else None)
    else:
        comm = SingleCommProxy()
        args.ranks_are = "one process, one GPU"
    result = run_rank(comm, args, local)
    if world > 1:
        dist.destroy_process_group()
    if comm.rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
