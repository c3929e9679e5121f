"""BASELINE configs 3 (full size), 4 and 5 on the device, against the CPU oracle.

config 4  cylinder Re=100, shipped mesh red-refined once (222 962 dofs): base flow on the device (Picard ×3 →
          Newton, every iteration assembled / factorised / solved on the GPU) vs the oracle's base flow, 50
          actuated steps vs the oracle's series, and the same mesh through the 2- and 4-rank partition.
config 5  fluidic pinball Re=100, ROTATION mode (three ActuatorBCRotation): 50 open-loop steps with three
          different Gaussian bumps, and 50 closed-loop steps through a fixed synthetic stable 3-in / 3-out
          Controller, vs the oracle's series.
config 3  open cavity Re=7500 on cavity_fine (876 645 dofs), FORCE actuator + wall-shear sensor in a closed loop:
          base flow Picard x10 -> Newton x10 on the device, 20 closed-loop steps vs the oracle's series
          (tests/golden/make_config3_fixture.py), assembled operators and right-hand sides vs the oracle's on that mesh.

Fixtures: tests/golden/make_config45_fixtures.py (oracle); scenario definitions are imported from it.
"""
import os
import socket
import sys
import tempfile
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests" / "golden"))
from flowcontrol_amd.examples.data import mesh_file  # noqa: E402


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


def _ycols(ts):
    return [c for c in ts.columns if c.startswith("y_meas_")]


# ── config 4 ─────────────────────────────────────────────────────────────────────────────────────────────────
def _config4_solver(path_out, num_steps=50):
    from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver, refined_cylinder_mesh
    from flowcontrol_amd.flowsolverparameters import ParamIC

    fs = CylinderFlowSolver.make_default(Re=100, path_out=path_out, num_steps=num_steps, meshpath=refined_cylinder_mesh(1))
    fs.check_residual_every = 1  # these tests assert every step's residual (the default cadence is 8 where the factors stream from HBM)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    return fs


def test_config4_refined_cylinder_base_flow_and_actuated_steps(tmp_path_factory, golden_dir):
    from flowcontrol_amd.examples.cylinder.scenarios import config4_actuation

    g = np.load(golden_dir / "cylinder_O1_refined1.npz")
    fs = _config4_solver(tmp_path_factory.mktemp("config4"))
    assert fs.th.nc == int(g["ncells"]) == 49136 and fs.th.N == int(g["ndofs"]) == 222962
    fs.compute_steady_state(method="picard", max_iter=3, tol=1e-7, u_ctrl=[0.0, 0.0])
    fs.compute_steady_state(method="newton", max_iter=25, u_ctrl=[0.0, 0.0], initial_guess=fs.fields.UP0)
    nv2 = 2 * fs.th.nn
    assert _rel(fs.fields.UP0.vector().get_local()[:nv2], g["UP0"][:nv2]) < 1e-10
    fs.initialize_time_stepping(ic=None)
    u = config4_actuation(50)
    for k in range(50):
        fs.step(u[k])
        assert fs.solve_info[1] < 1e-12
    ts = fs.timeseries
    assert _rel(ts[_ycols(ts)].to_numpy(), g["y"]) < 1e-8
    assert _rel(ts["dE"].to_numpy(), g["dE"]) < 1e-8
    U = fs.fields.u_.vector().get_local() + fs.fields.U0.vector().get_local()
    assert np.isclose(U.max(), float(g["umax"]), rtol=1e-8) and np.isclose(U.mean(), float(g["umean"]), rtol=1e-8)
    fs.th.release_device()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _config4_worker(rank, world, port, out, nsteps):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flowcontrol_amd.fem.spaces import Function
        from flowcontrol_amd.examples.cylinder.scenarios import config4_actuation

        g = np.load(ROOT / "tests" / "golden" / "cylinder_O1_refined1.npz")
        fs = _config4_solver(tempfile.mkdtemp(), nsteps)
        U0, P0 = Function(fs.W, g["UP0"]).split()
        fs._assign_steady_state(U0, P0)
        fs.initialize_time_stepping(ic=None)
        u = config4_actuation(nsteps)
        for k in range(nsteps):
            fs.step(u[k])
        ts = fs.timeseries
        dev = fs.th.device()
        if rank == 0:
            out["y"] = ts[_ycols(ts)].to_numpy()
            out["dE"] = ts["dE"].to_numpy()
            out["resid"] = float(fs.solve_info[1])
        out[f"cells{rank}"] = int(dev.part.local_cells.size)
        out[f"nnz{rank}"] = int(dev.local_factor_nnz)
        fs.th.release_device()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_config4_refined_cylinder_partitioned(world):
    """The refined mesh row-partitioned over 2 and 4 ranks sharing this GPU (exchange staged through the host, every
    kernel the one an N-GPU run launches): the merged series must be the oracle's, the cells must be split evenly."""
    import torch.multiprocessing as mp

    nsteps = 12
    g = np.load(ROOT / "tests" / "golden" / "cylinder_O1_refined1.npz")
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_config4_worker, args=(world, _free_port(), out, nsteps), nprocs=world, join=True)
        assert _rel(out["y"], g["y"][: nsteps + 1]) < 1e-8
        assert _rel(out["dE"], g["dE"][: nsteps + 1]) < 1e-8
        assert out["resid"] < 1e-9
        cells = [out[f"cells{r}"] for r in range(world)]
        assert sum(cells) == 49136 and max(cells) - min(cells) <= 1


# ── config 5 ─────────────────────────────────────────────────────────────────────────────────────────────────
def _pinball(path_out):
    from flowcontrol_amd.actuator import CYLINDER_ACTUATION_MODE
    from flowcontrol_amd.examples.pinball.pinballflowsolver import PinballFlowSolver
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC

    g = np.load(ROOT / "tests" / "golden" / "pinball_re100_rotation.npz")
    fs = PinballFlowSolver.make_default(Re=100, mode_actuation=CYLINDER_ACTUATION_MODE.ROTATION, path_out=path_out, num_steps=50)
    fs.check_residual_every = 1  # (see _config4_solver)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    assert fs.th.N == int(g["ndofs"]) == 302035
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    return fs, g


def test_config5_pinball_rotation_open_loop_bumps(tmp_path_factory):
    from flowcontrol_amd.examples.pinball.scenarios import pinball_bumps

    fs, g = _pinball(tmp_path_factory.mktemp("config5_ol"))
    us = []
    for _ in range(50):
        u = pinball_bumps(fs.t)
        us.append(u)
        fs.step(u)
        assert fs.solve_info[1] < 1e-12
    us = np.array(us)
    assert np.allclose(us, g["ol_u"], rtol=0, atol=1e-15) and np.all(np.abs(us).max(axis=0) > 1.0)  # three different, time-varying inputs
    ts = fs.timeseries
    assert _rel(ts[_ycols(ts)].to_numpy(), g["ol_y"]) < 1e-8
    assert _rel(ts["dE"].to_numpy(), g["ol_dE"]) < 1e-8
    U = fs.fields.u_.vector().get_local() + fs.fields.U0.vector().get_local()
    assert np.isclose(U.max(), float(g["umax"]), rtol=1e-8) and np.isclose(U.mean(), float(g["umean"]), rtol=1e-8)
    fs.th.release_device()


def test_config5_pinball_rotation_closed_loop(tmp_path_factory):
    """Sensors → Controller (ZOH-discretised 3-in / 3-out LTI system) → three rotating cylinders, 50 steps."""
    from flowcontrol_amd.controller import Controller
    from flowcontrol_amd.examples.pinball.scenarios import PINBALL_K

    fs, g = _pinball(tmp_path_factory.mktemp("config5_cl"))
    K = Controller(A=PINBALL_K["A"], B=PINBALL_K["B"], C=PINBALL_K["C"], D=PINBALL_K["D"])
    us = []
    for _ in range(50):
        u = K.step(y=fs.y_meas, dt=fs.params_time.dt)
        us.append(np.asarray(u).reshape(-1))
        fs.step(u_ctrl=us[-1])
    us = np.array(us)
    assert np.abs(us).max() > 0.5  # the loop is closed in earnest
    assert _rel(us, g["cl_u"]) < 1e-8
    ts = fs.timeseries
    assert _rel(ts[_ycols(ts)].to_numpy(), g["cl_y"]) < 1e-8
    assert _rel(ts["dE"].to_numpy(), g["cl_dE"]) < 1e-8
    fs.th.release_device()


def test_config5_pinball_rotation_base_flow_on_device(tmp_path_factory):
    """run_pinball_rotation_example.py:88-97: Picard ×15 from the antisymmetric_bot guess, then Newton, on the device."""
    from flowcontrol_amd.actuator import CYLINDER_ACTUATION_MODE
    from flowcontrol_amd.examples.pinball.pinballflowsolver import PinballCustomInitialGuess, PinballFlowSolver

    g = np.load(ROOT / "tests" / "golden" / "pinball_re100_rotation.npz")
    fs = PinballFlowSolver.make_default(Re=100, mode_actuation=CYLINDER_ACTUATION_MODE.ROTATION, path_out=tmp_path_factory.mktemp("config5_ss"))
    guess = PinballCustomInitialGuess(mode="antisymmetric_bot").as_dolfin_function(function_space=fs.W)
    fs.compute_steady_state(method="picard", max_iter=15, tol=1e-7, u_ctrl=[0.0, 0.0, 0.0], initial_guess=guess)
    fs.compute_steady_state(method="newton", max_iter=10, u_ctrl=[0.0, 0.0, 0.0], initial_guess=fs.fields.UP0)
    nv2 = 2 * fs.th.nn
    assert _rel(fs.fields.UP0.vector().get_local()[:nv2], g["UP0"][:nv2]) < 1e-9
    fs.th.release_device()


# ── config 3 at full size ────────────────────────────────────────────────────────────────────────────────────
def test_config3_cavity_fine_closed_loop_vs_oracle(tmp_path_factory, golden_dir):
    """cavity_fine (876 645 dofs), the reference's base-flow recipe (tests/integration/test_cavity.py:64-66: Picard x10,
    tol 1e-7 -> Newton x10, every iteration assembled / factorised / solved on the device), then 20 closed-loop steps
    (wall-shear sensor -> first-order low-pass Controller -> Gaussian FORCE actuator) against the CPU oracle's series for
    exactly this scenario (tests/golden/make_config3_fixture.py: 22 SuperLU factorisations of the 877 k-dof matrices)."""
    from flowcontrol_amd._lib import SLOT_BDF2
    from flowcontrol_amd.controller import Controller
    from flowcontrol_amd.examples.cavity.cavityflowsolver import CavityFlowSolver
    from flowcontrol_amd.examples.cavity.scenarios import CAVITY_K
    from make_config3_fixture import N_STEPS, SAMPLE_STRIDE
    from oracle import ns_oracle as O

    g = np.load(golden_dir / "cavity_fine_re7500.npz")
    fs = CavityFlowSolver.make_default(Re=7500, path_out=tmp_path_factory.mktemp("config3"), num_steps=N_STEPS,
                                       meshpath=mesh_file("cavity_fine"))
    assert fs.th.N == int(g["ndofs"]) == 876645 and fs.th.nc == int(g["ncells"])
    fs.check_residual_every = 1  # every step's residual is asserted below (default where the factors stream from HBM: every 8th step)
    fs.compute_steady_state(method="picard", max_iter=10, tol=1e-7, u_ctrl=[0.0])
    fs.compute_steady_state(method="newton", max_iter=10, u_ctrl=[0.0], initial_guess=fs.fields.UP0)
    up0 = fs.fields.UP0.vector().get_local()
    U0 = up0[: 2 * fs.th.nn]
    assert np.isclose(U0.max(), float(g["u0max"]), rtol=1e-9) and np.isclose(U0.mean(), float(g["u0mean"]), rtol=1e-9)
    nv2 = 2 * fs.th.nn
    vel = np.arange(0, fs.th.N, SAMPLE_STRIDE) < nv2
    assert _rel(up0[::SAMPLE_STRIDE][vel], g["UP0_sample"][vel]) < 1e-8  # velocity dofs of the strided sample
    fs.initialize_time_stepping(ic=None)
    K = Controller(A=CAVITY_K["A"], B=CAVITY_K["B"], C=CAVITY_K["C"], D=CAVITY_K["D"])
    y0 = fs.y_meas[0]
    us = []
    for _ in range(N_STEPS):
        u = K.step(y=fs.y_meas[0] - y0, dt=fs.params_time.dt)
        us.append(float(u[0]))
        fs.step(u_ctrl=[u[0]])
        assert fs.solve_info[1] < 1e-12
    ts = fs.timeseries
    assert np.abs(g["u"]).max() > 1.0  # the loop acts in earnest
    assert _rel(np.array(us), g["u"][:, 0]) < 1e-8
    assert _rel(ts[_ycols(ts)].to_numpy(), g["y"]) < 1e-8
    assert _rel(ts["dE"].to_numpy(), g["dE"]) < 1e-8
    U = fs.fields.u_.vector().get_local() + U0
    assert np.isclose(U.max(), float(g["umax"]), rtol=1e-8) and np.isclose(U.mean(), float(g["umean"]), rtol=1e-8)
    # the operators at full size, entry by entry against the oracle's assembly on this mesh (kept from round 2)
    th, dev = fs.th, fs.th.device()
    d = O.Disc.from_taylor_hood(th)
    dt, Re = fs.params_time.dt, fs.params_flow.Re
    dofs, prof = fs._bc_tables()
    A_ref, _ = O.apply_bc_symmetric(O.assemble_matrix(d, mass=1.5 / dt, nu=1.0 / Re, adv=U0, lin=U0), None, dofs, np.zeros(len(dofs)))
    diff = (dev.matrix(SLOT_BDF2) - A_ref).tocoo()
    assert np.abs(diff.data).max() <= 1e-12 * np.abs(A_ref.data).max()
    u_n, u_nn, _ = dev.get_state()
    fprof = fs._force_tables().T
    uc = np.array([0.7])
    b_dev = dev.assemble_rhs(SLOT_BDF2, uc)
    b_ref = O.rhs_transient(d, 2, dt, u_n, u_nn, fprof @ uc, True)
    g_ = np.zeros(th.N)
    g_[dofs] = prof @ uc
    b_ref = b_ref - O.assemble_matrix(d, mass=1.5 / dt, nu=1.0 / Re, adv=U0, lin=U0) @ g_
    b_ref[dofs] = g_[dofs]
    assert _rel(b_dev, b_ref) < 1e-12
    fs.th.release_device()


def test_example_scripts_run_end_to_end(tmp_path_factory):
    """The run scripts a user of the reference starts from (run_lidcavity_example.py, run_pinball_suction_example.py): a few steps of
    each, finite measurements, the time series file, residuals at round-off."""
    from flowcontrol_amd.examples.lidcavity import run_lidcavity_example
    from flowcontrol_amd.examples.pinball import run_pinball_suction_example

    fs = run_lidcavity_example.main(num_steps=5, path_out=tmp_path_factory.mktemp("ex_lid"), Re=1000)
    assert np.all(np.isfinite(fs.y_meas)) and fs.solve_info[1] < 1e-10 and len(fs.timeseries) == 6
    assert any(p.suffix == ".csv" for p in Path(fs.params_save.path_out).rglob("*"))
    fs.th.release_device()
    fs = run_pinball_suction_example.main(num_steps=4, path_out=tmp_path_factory.mktemp("ex_pinball"))
    assert np.all(np.isfinite(fs.y_meas)) and 0.0 < fs.residual_max < 1e-10  # (the monitor ran on the first step at least: default cadence)
    assert np.abs(fs.timeseries[["u_ctrl_1", "u_ctrl_2", "u_ctrl_3"]].to_numpy()[1:]).max() > 0  # the bumps were applied
    fs.th.release_device()
