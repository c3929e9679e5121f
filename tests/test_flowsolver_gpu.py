"""End-to-end parity of the MI355X ``FlowSolver`` (public API, through the C ABI).

Reads like the reference's ``tests/integration/test_cylinder.py``: same scenario, same constants,
same tolerances; in addition every series is compared with the oracle's golden vectors at 1e-8
relative (fp64 everywhere; the device solve is a different but LU-grade algorithm).
"""
import numpy as np
import pytest

from flowcontrol_amd import utils as flu
from flowcontrol_amd.controller import Controller
from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
from flowcontrol_amd.fem.spaces import Function
from flowcontrol_amd.flowsolverparameters import ParamIC
from flowcontrol_amd.examples.data import controller_file  # noqa: E402

pytestmark = pytest.mark.gpu

# reference tests/integration/test_cylinder.py:66-74
_U0_MAX_REF = np.float64(1.1921615450014942)
_U0_MEAN_REF = np.float64(0.336746427968607)
_U_MAX_REF = np.float64(1.325070045534714)
_U_MEAN_REF = np.float64(0.3376859329866094)
_LAST_TIME_REF = np.float64(0.1)
_LAST_Y_MEAS_1_REF = np.float64(0.011615482723602308)
_LAST_Y_MEAS_2_REF = np.float64(0.003860524805395703)
_LAST_Y_MEAS_3_REF = np.float64(0.0038461597025207803)
_LAST_DE_REF = np.float64(0.09462807324653322)


def _rel_l2(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


def _load_baseflow(fs, golden_dir):
    up0 = np.load(golden_dir / "cylinder_O1.npz")["UP0"]
    U0, P0 = Function(fs.W, up0).split()
    fs._assign_steady_state(U0, P0)


def test_cylinder_smoke(tmp_path_factory):
    """Pipeline runs; velocity values are finite after 3 steps (reference test_cylinder_smoke)."""
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp("cylinder_smoke"), num_steps=3)
    fs.compute_steady_state(method="picard", max_iter=3, tol=1e-7, u_ctrl=[0.0, 0.0])
    fs.initialize_time_stepping(ic=None)
    for _ in range(fs.params_time.num_steps):
        fs.step(u_ctrl=[0.0, 0.0])
    u_vals = fs.fields.u_.vector().get_local()
    assert np.all(np.isfinite(u_vals)), "velocity field contains non-finite values"
    fs.th.release_device()


def test_cylinder_regression(tmp_path_factory, golden_dir):
    """10-step closed-loop run + JSON-based restart must reproduce the reference values
    (mirror of the reference's test_cylinder_regression)."""
    path_out = tmp_path_factory.mktemp("cylinder_regression")
    fs = CylinderFlowSolver.make_default(Re=100, path_out=path_out, num_steps=10, save_every=5)
    fs.compute_steady_state(method="picard", max_iter=3, tol=1e-7, u_ctrl=[0.0, 0.0])
    fs.compute_steady_state(method="newton", max_iter=25, u_ctrl=[0.0, 0.0], initial_guess=fs.fields.UP0)

    u0_max = flu.apply_fun(fs.fields.U0, np.max)
    u0_mean = flu.apply_fun(fs.fields.U0, np.mean)
    assert np.isclose(u0_max, _U0_MAX_REF, rtol=1e-6), f"u0_max: {u0_max}"
    assert np.isclose(u0_mean, _U0_MEAN_REF, rtol=1e-6), f"u0_mean: {u0_mean}"

    fs.initialize_time_stepping(ic=None)
    Kss = Controller.from_file(file=controller_file(), x0=None)
    for _ in range(fs.params_time.num_steps):
        y_meas = flu.MpiUtils.mpi_broadcast(fs.y_meas)
        u_ctrl = Kss.step(y=-y_meas[0], dt=fs.params_time.dt)
        fs.step(u_ctrl=[u_ctrl[0], u_ctrl[0]])
    fs.write_timeseries()
    g = np.load(golden_dir / "cylinder_O1.npz")
    ts = fs.timeseries
    assert _rel_l2(ts[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy(), g["cl_y"][:11]) < 1e-8
    assert _rel_l2(ts["dE"].to_numpy(), g["cl_dE"][:11]) < 1e-8
    fs.th.release_device()

    # restart from Tstart=0.05 using the JSON sidecar (no ParamRestart needed)
    fs_restart = CylinderFlowSolver.make_default(Re=100, path_out=path_out, num_steps=10, save_every=5, Tstart=0.05)
    fs_restart.load_steady_state()
    fs_restart.initialize_time_stepping(Tstart=fs_restart.params_time.Tstart)
    for _ in range(fs_restart.params_time.num_steps):
        y_meas = flu.MpiUtils.mpi_broadcast(fs_restart.y_meas)
        u_ctrl = Kss.step(y=-y_meas[0], dt=fs_restart.params_time.dt)
        fs_restart.step(u_ctrl=np.repeat(u_ctrl, repeats=2, axis=0))
    fs_restart.write_timeseries()

    u_max = flu.apply_fun(fs_restart.fields.Usave, np.max)
    u_mean = flu.apply_fun(fs_restart.fields.Usave, np.mean)
    last = fs_restart.timeseries.iloc[-1]
    assert np.isclose(u_max, _U_MAX_REF, rtol=1e-4), f"u_max: {u_max} != {_U_MAX_REF}"
    assert np.isclose(u_mean, _U_MEAN_REF, rtol=1e-6), f"u_mean: {u_mean} != {_U_MEAN_REF}"
    assert np.isclose(last["time"], _LAST_TIME_REF, rtol=1e-6), f"time: {last['time']}"
    assert np.isclose(last["y_meas_1"], _LAST_Y_MEAS_1_REF, rtol=1e-4), f"y_meas_1: {last['y_meas_1']}"
    assert np.isclose(last["y_meas_2"], _LAST_Y_MEAS_2_REF, rtol=1e-4), f"y_meas_2: {last['y_meas_2']}"
    assert np.isclose(last["y_meas_3"], _LAST_Y_MEAS_3_REF, rtol=1e-4), f"y_meas_3: {last['y_meas_3']}"
    assert np.isclose(last["dE"], _LAST_DE_REF, rtol=1e-4), f"dE: {last['dE']}"
    # and far tighter against the oracle's series
    tr = fs_restart.timeseries
    assert _rel_l2(tr[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy(), g["cl_y"][10:21]) < 1e-8
    fs_restart.th.release_device()


def test_cylinder_open_loop_200_steps_vs_oracle(tmp_path_factory, golden_dir):
    """BASELINE config 2: 200 open-loop steps, IC of run_cylinder_example.py:55; sensor and energy
    series within 1e-8 rel-L2 of the oracle (target stated by north_star: 1e-6)."""
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp("cyl_ol"), num_steps=200)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    _load_baseflow(fs, golden_dir)
    fs.initialize_time_stepping(ic=None)
    for _ in range(100):
        fs.step(u_ctrl=[0.0, 0.0])
    # the running maximum of the monitored residual: kept with every step's log row (one step behind: nobody waited for the late record)
    assert 0.0 < fs.residual_max < 1e-12
    y_b, dE_b = fs.run(100, np.zeros(2))  # batched path continues the same trajectory
    g = np.load(golden_dir / "cylinder_O1.npz")
    ts = fs.timeseries
    y = ts[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy()
    assert y.shape == (201, 3)
    assert _rel_l2(y, g["ol_y"]) < 1e-8
    assert _rel_l2(ts["dE"].to_numpy(), g["ol_dE"]) < 1e-8
    assert fs.solve_info[1] < 1e-9
    assert np.isclose(fs.t, 1.0)
    fs.th.release_device()


def test_residual_monitor_cadence_leaves_the_series_bit_identical(tmp_path_factory, golden_dir):
    """check_residual_every = n: the residual monitor (a check the reference never makes, flowsolver.py:728-737) runs on every
    n-th step only; measurements, energy and state are bit for bit those of the every-step run, solve_info[1] is NaN in
    between, the non-finite test still trips on an unmonitored step."""
    runs = {}
    for every in (1, 4, 0):
        fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp(f"cadence{every}"), num_steps=24)
        fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
        fs.check_residual_every = every
        _load_baseflow(fs, golden_dir)
        fs.initialize_time_stepping(ic=None)
        res = []
        for k in range(24):
            fs.step(u_ctrl=[0.1 * np.sin(0.4 * k), -0.05])
            res.append(float(fs.solve_info[1]))
        ts = fs.timeseries
        runs[every] = (ts[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy().copy(), ts["dE"].to_numpy().copy(), fs.fields.u_.vector().get_local().copy(),
                       np.array(res))
        if every == 4:  # an unmonitored step (the 26th of the handle is not a multiple of 4) still reports a non-finite velocity
            dev = fs.th.device()
            fs.step(u_ctrl=[0.0, 0.0])
            u_n, u_nn, p_n = dev.get_state()
            u_n[5] = np.nan
            dev.set_state(u_n, u_nn, p_n)
            with pytest.raises(RuntimeError, match="Failed solving"):
                fs.step(u_ctrl=[0.0, 0.0])
        fs.th.release_device()
    y1, dE1, u1, r1 = runs[1]
    assert np.all(r1 < 1e-12)
    for every in (4, 0):
        y, dE, u, r = runs[every]
        assert np.array_equal(y, y1) and np.array_equal(dE, dE1) and np.array_equal(u, u1)
    r4 = runs[4][3]
    assert np.count_nonzero(np.isfinite(r4)) == 6 and np.all(r4[np.isfinite(r4)] < 1e-12)  # 24 steps, every 4th monitored
    assert not np.any(np.isfinite(runs[0][3]))


def test_overlapped_tail_gives_the_blocking_steps_bits(tmp_path_factory, golden_dir, monkeypatch):
    """fc_step_end(early) + fc_step_collect: the measurements come back right behind the last sweep launch, residual monitor and
    energy run on a second stream and are collected a step later -- the log (y, dE), the solve info and the state are bit for
    bit those of FC_OVERLAP_TAIL=0 (one stream, one record) and of the blocking C-ABI fc_step; a residual breach is reported with
    the following step at the latest."""
    from flowcontrol_amd._lib import SLOT_BDF1, SLOT_BDF2

    def run(overlap):
        monkeypatch.setenv("FC_OVERLAP_TAIL", "1" if overlap else "0")
        fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp(f"overlap{int(overlap)}"), num_steps=30)
        fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
        fs.params_save.energy_every = 3
        _load_baseflow(fs, golden_dir)
        fs.initialize_time_stepping(ic=None)
        infos = []
        for k in range(30):
            y = fs.step(u_ctrl=[0.08 * np.sin(0.5 * k), 0.03])
            if k % 7 == 0:
                infos.append(np.array(fs.solve_info).copy())  # forces the collection now and then; the log collects the rest
        ts = fs.timeseries
        out = (ts[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy().copy(), ts["dE"].to_numpy().copy(), fs.fields.u_.vector().get_local().copy(), np.array(infos), y)
        return fs, out

    fs0, a = run(False)
    fs0.th.release_device()
    fs1, b = run(True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[3], b[3]) and np.all(b[3][:, 1] < 1e-12)
    assert np.count_nonzero(np.isfinite(b[1])) == 11  # IC + every third of 30 steps
    # the blocking C-ABI step continues both trajectories identically
    dev = fs1.th.device()
    y_c, dE_c, info_c = dev.step(SLOT_BDF2, np.array([0.01, -0.02]))
    assert np.all(np.isfinite(y_c)) and np.isfinite(dE_c) and info_c[1] < 1e-12
    # a residual breach (here: an absurd tolerance) surfaces with the next step at the latest
    fs1.params_solver.throw_error = False
    fs1.fields._mark_stale()
    fs1.residual_tol = 1e-30
    r1 = fs1.step(u_ctrl=[0.0, 0.0])
    r2 = fs1.step(u_ctrl=[0.0, 0.0]) if r1 is not None else None
    assert r1 is None or r2 is None
    fs1.th.release_device()
    assert SLOT_BDF1 == 0


def test_missing_baseflow_and_bad_inputs(tmp_path_factory):
    fs = CylinderFlowSolver.make_default(path_out=tmp_path_factory.mktemp("cyl_err"))
    with pytest.raises(ValueError):
        fs.set_actuators_u_ctrl([0.0])
    fs.th.release_device()


# ── open cavity (BASELINE config 3 ingredients) and pinball (config 5) ─────────────────────────────
def _run_case_vs_fixture(fs, g, n_act, nsteps=10):
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    for _ in range(nsteps):
        fs.step(u_ctrl=[0.0] * n_act)
    ts = fs.timeseries
    ycols = [c for c in ts.columns if c.startswith("y_meas_")]
    assert _rel_l2(ts[ycols].to_numpy(), g["y"]) < 1e-8
    assert _rel_l2(ts["dE"].to_numpy(), g["dE"]) < 1e-8
    return ts


def test_cavity_regression_open_loop(tmp_path_factory, golden_dir):
    """Mirror of the reference's test_cavity_regression time-stepping part (Re=7500, dt=4e-4, FORCE
    actuator at u_ctrl=0, wall-shear + point sensors) from the golden base flow."""
    from flowcontrol_amd.examples.cavity.cavityflowsolver import CavityFlowSolver

    fs = CavityFlowSolver.make_default(Re=7500, path_out=tmp_path_factory.mktemp("cavity"), num_steps=10, save_every=5)
    g = np.load(golden_dir / "cavity_coarse.npz")
    ts = _run_case_vs_fixture(fs, g, 1)
    last = ts.iloc[-1]
    assert np.isclose(last["time"], 0.004, rtol=1e-6)
    assert np.isclose(last["y_meas_1"], 6.0488687475121505, rtol=1e-4)
    assert np.isclose(last["y_meas_2"], 0.024799707355708498, rtol=1e-4)
    assert np.isclose(last["dE"], 0.005000924582291293, rtol=1e-4)
    assert np.isclose(flu.apply_fun(fs.fields.Usave, np.mean), 0.3565670457803184, rtol=1e-6)
    assert np.isclose(flu.apply_fun(fs.fields.Usave, np.max), 1.1897880864595587, rtol=1e-4)
    # the FORCE actuator is unit-L2-norm (reference tests/test_actuator.py:155-161) and acts on the flow
    act = fs.params_control.actuator_list[0]
    v = act.expression.profile(fs.th.node_coords)
    assert np.isclose(fs._velocity_l2_norm(np.r_[v[:, 0], v[:, 1]]), 1.0, rtol=1e-12)
    y_before = fs.y_meas.copy()
    fs.step(u_ctrl=[5.0])
    assert np.all(np.isfinite(fs.y_meas)) and not np.allclose(fs.y_meas, y_before)
    fs.th.release_device()


def test_pinball_regression_open_loop(tmp_path_factory, golden_dir):
    from flowcontrol_amd.actuator import CYLINDER_ACTUATION_MODE
    from flowcontrol_amd.examples.pinball.pinballflowsolver import PinballFlowSolver

    fs = PinballFlowSolver.make_default(Re=30, mode_actuation=CYLINDER_ACTUATION_MODE.SUCTION, path_out=tmp_path_factory.mktemp("pinball"),
                                        num_steps=10, save_every=5)
    g = np.load(golden_dir / "pinball_middle.npz")
    ts = _run_case_vs_fixture(fs, g, 3)
    last = ts.iloc[-1]
    assert np.isclose(last["time"], 0.05, rtol=1e-6)
    assert np.isclose(last["y_meas_1"], -0.0007241196930108308, rtol=1e-4)
    assert np.isclose(last["dE"], 0.05722263472621765, rtol=1e-4)
    assert np.isclose(flu.apply_fun(fs.fields.Usave, np.mean), 0.14938204178441114, rtol=1e-6)
    fs.th.release_device()


def test_cylinder_crank_nicolson_vs_oracle(tmp_path_factory, golden_dir):
    """time_scheme="cn" through the public API (ParamSolver.time_scheme, flowsolverparameters.py) against
    the oracle's CN stepper: 10 actuated steps; and CN ≈ BDF2 to O(dt²) on the same trajectory."""
    from tests.support import ndsolver
    from oracle import ns_oracle as O

    g = np.load(golden_dir / "cylinder_O1.npz")
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp("cyl_cn"), num_steps=10)
    fs.params_solver.time_scheme = "cn"
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    _load_baseflow(fs, golden_dir)
    fs.initialize_time_stepping(ic=None)
    assert fs.order == "cn"
    th = fs.th
    d = O.Disc.from_taylor_hood(th)
    dofs, prof = fs._bc_tables()
    skip = np.zeros(th.N, bool)
    skip[dofs] = True
    perm = ndsolver.build_tree(th.cell_dofs, th.mesh.cell_centroids(), th.N, 10, skip).perm
    ts = O.TimeStepperCN(d, 100.0, 0.005, g["UP0"][: 2 * th.nn], dofs, prof, perm=perm)
    rows = [s.row(fs) for s in fs.params_control.sensor_list]
    u_n = fs.fields.ic.u.vector().get_local()
    ys = []
    ucs = np.array([[0.05 * np.sin(0.4 * k), -0.03] for k in range(10)])
    for k in range(10):
        if k < 6:
            fs.step(ucs[k])
        elif k == 6:  # FlowSolver.run with the CN scheme: same steps, same log
            yb, dEb = fs.run(4, ucs[6:])
            assert yb.shape == (4, 3) and dEb.shape == (4,) and np.all(np.isfinite(dEb))
        up = ts.step(u_n, ucs[k])
        u_n = up[: 2 * th.nn]
        ys.append([w @ up[i] for i, w in rows])
    y_dev = fs.timeseries[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy()[1:]
    assert np.array_equal(yb, y_dev[6:])
    assert _rel_l2(y_dev, np.array(ys)) < 1e-8
    assert _rel_l2(fs.fields.u_.vector().get_local(), u_n) < 1e-9
    assert fs.order == "cn" and np.isclose(fs.t, 0.05)
    fs.th.release_device()


def test_cavity_closed_loop_force_actuation_vs_oracle(tmp_path_factory, golden_dir):
    """BASELINE config 3 ingredients in one loop: open cavity Re=7500, Gaussian FORCE actuator driven by
    an LTI Controller fed with the wall-shear sensor (no cavity controller ships with the reference: a
    documented synthetic stable first-order low-pass, K(s) = 0.5 / (0.01 s + 1)), 6 steps; the device
    trajectory must follow the oracle's with the same control sequence."""
    from tests.support import ndsolver
    from flowcontrol_amd.examples.cavity.cavityflowsolver import CavityFlowSolver
    from oracle import ns_oracle as O

    g = np.load(golden_dir / "cavity_coarse.npz")
    fs = CavityFlowSolver.make_default(Re=7500, path_out=tmp_path_factory.mktemp("cavity_cl"), num_steps=6)
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    K = Controller(A=[[-100.0]], B=[[1.0]], C=[[50.0]], D=[[0.0]])
    us = []
    for _ in range(6):
        u = K.step(y=fs.y_meas[0] - g["y"][0][0], dt=fs.params_time.dt)  # feedback on the shear fluctuation
        us.append(float(u[0]))
        fs.step(u_ctrl=[u[0]])
    assert np.max(np.abs(us)) > 1e-6  # the loop is really closed
    # oracle with the recorded control sequence
    th = fs.th
    d = O.Disc.from_taylor_hood(th)
    dofs, prof = fs._bc_tables()
    skip = np.zeros(th.N, bool)
    skip[dofs] = True
    perm = ndsolver.build_tree(th.cell_dofs, th.mesh.cell_centroids(), th.N, 13, skip).perm
    ts = O.TimeStepper(d, 7500.0, fs.params_time.dt, g["UP0"][: 2 * th.nn], dofs, prof, force_profiles=fs._force_tables().T, perm=perm)
    rows = [s.row(fs) for s in fs.params_control.sensor_list]
    u_n = fs.fields.ic.u.vector().get_local()
    u_nn = u_n.copy()
    ys, order = [], 1
    for k in range(6):
        up = ts.step(order, u_n, u_nn, [us[k]])
        order = 2
        u_nn, u_n = u_n, up[: 2 * th.nn]
        ys.append([w @ up[i] for i, w in rows])
    y_dev = fs.timeseries[["y_meas_1", "y_meas_2"]].to_numpy()[1:]
    assert _rel_l2(y_dev, np.array(ys)) < 1e-8
    assert _rel_l2(fs.fields.u_.vector().get_local(), u_n) < 1e-8
    fs.th.release_device()


def test_eager_and_batched_runs_agree_bitwise(tmp_path_factory, golden_dir):
    """step() hands (y, dE) back through a host-mapped record that the host polls; run() synchronises
    once at the end.  Same kernels, same order: the two series must be IDENTICAL.  A record read before
    all of its words were visible (seen at a rate of 5e-4 per step before the record was checksummed)
    shows up as a repeated row."""
    n = 6000
    u = np.stack([0.05 * np.sin(0.01 * np.arange(n)), -0.02 * np.cos(0.013 * np.arange(n))], axis=1)
    series = []
    for mode in ("eager", "batched"):
        fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp(f"poll_{mode}"), num_steps=n)
        fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
        _load_baseflow(fs, golden_dir)
        fs.initialize_time_stepping(ic=None)
        fs.step(u[0])
        if mode == "eager":
            y = np.array([fs.step(u[k]).copy() for k in range(1, n)])
            dE = fs.timeseries["dE"].to_numpy()[2:]
        else:
            y, dE = fs.run(n - 1, u[1:])
        series.append((y, np.asarray(dE)))
        fs.th.release_device()
    (ya, ea), (yb, eb) = series
    assert np.array_equal(ya, yb)
    assert np.array_equal(ea, eb)
    assert not np.any(np.all(ya[1:] == ya[:-1], axis=1))


def test_lidcavity_regression(tmp_path_factory, golden_dir):
    """Mirror of the reference's test_lidcavity_regression (Re = 1000, mesh64, Picard base flow on the
    device, 10 unactuated steps): enclosed flow, pressure pinned (fem.boundary.pressure_pin).  Against the
    oracle's golden series to 1e-8 and the reference's constants with the reference's tolerances."""
    from flowcontrol_amd.examples.lidcavity.lidcavityflowsolver import LidCavityFlowSolver

    fs = LidCavityFlowSolver.make_default(Re=1000, path_out=tmp_path_factory.mktemp("lidcavity"), num_steps=10, save_every=5)
    fs.compute_steady_state(method="picard", max_iter=40, tol=1e-7, u_ctrl=[0.0])
    g = np.load(golden_dir / "lidcavity_mesh64.npz")
    u0_max, u0_mean = flu.apply_fun(fs.fields.U0, np.max), flu.apply_fun(fs.fields.U0, np.mean)
    assert np.isclose(u0_max, 1.000000000000008, rtol=1e-6)
    assert np.isclose(u0_mean, 0.0020234251738529907, rtol=1e-6)
    assert _rel_l2(fs.fields.U0.vector().get_local(), g["UP0"][: 2 * fs.th.nn]) < 1e-8
    fs.initialize_time_stepping(ic=None)
    for _ in range(fs.params_time.num_steps):
        fs.step(u_ctrl=[0.0])
    ts = fs.timeseries
    assert _rel_l2(ts[["y_meas_1", "y_meas_2"]].to_numpy(), g["y"]) < 1e-8
    assert _rel_l2(ts["dE"].to_numpy(), g["dE"]) < 1e-8
    last = ts.iloc[-1]
    assert np.isclose(last["time"], 0.05, rtol=1e-6)
    assert np.isclose(last["y_meas_1"], -0.09584848445257539, rtol=1e-4)
    assert np.isclose(last["y_meas_2"], -0.06060429836866045, rtol=1e-4)
    assert np.isclose(last["dE"], 0.0012665481942387678, rtol=1e-4)
    assert np.isclose(flu.apply_fun(fs.fields.Usave, np.max), 1.000000000000008, rtol=1e-6)
    assert np.isclose(flu.apply_fun(fs.fields.Usave, np.mean), 0.0020222416653700877, rtol=1e-6)
    fs.th.release_device()


def test_cylinder_1000_actuated_steps_vs_oracle(tmp_path_factory, golden_dir):
    """north_star: "sensor timeseries within 1e-6 rel-L2 of reference over 1000 steps".  1000 synchronous
    steps with a time-varying actuation (Dirichlet lifting at every step) against the oracle's series
    (tests/golden/make_cylinder_1000_steps.py); the bar here is 1e-8."""
    g = np.load(golden_dir / "cylinder_O1_ol1000.npz")
    n = 1000
    k = np.arange(n)
    u = np.stack([0.05 * np.sin(0.01 * k), -0.02 * np.cos(0.013 * k)], axis=1)
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp("cyl_1000"), num_steps=n)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    _load_baseflow(fs, golden_dir)
    fs.initialize_time_stepping(ic=None)
    for i in range(n):
        fs.step(u[i])
    ts = fs.timeseries
    y = ts[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy()
    assert y.shape == g["y"].shape == (n + 1, 3)
    assert _rel_l2(y, g["y"]) < 1e-8
    assert _rel_l2(ts["dE"].to_numpy(), g["dE"]) < 1e-8
    assert np.isclose(fs.t, 5.0)
    fs.th.release_device()


def test_steady_state_with_lagged_factors_matches_refactorising_every_iteration(tmp_path_factory):
    """Base flow of the cylinder (Picard ×3 → Newton): the default keeps the factors of an earlier iterate as
    BiCGStab preconditioner; refactorising at every iteration (what the reference does with MUMPS) must give
    the same base flow, and the Krylov path must actually have been taken."""
    import flowcontrol_amd.steadystate as ss_mod

    flows, stats = [], []
    for lag in (True, False):
        fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp(f"ss_lag{int(lag)}"), num_steps=1)
        orig_init = ss_mod.SteadyStateSolver.__init__
        created = []

        def init(self, *a, _o=orig_init, _lag=lag, **k):
            _o(self, *a, **k)
            self.lag_factors = _lag
            created.append(self)

        ss_mod.SteadyStateSolver.__init__ = init
        try:
            fs.compute_steady_state(method="picard", max_iter=3, tol=1e-7, u_ctrl=[0.0, 0.0])
            fs.compute_steady_state(method="newton", max_iter=25, u_ctrl=[0.0, 0.0], initial_guess=fs.fields.UP0)
        finally:
            ss_mod.SteadyStateSolver.__init__ = orig_init
        flows.append(fs.fields.UP0.vector().get_local().copy())
        stats.append([k for s in created for k in s.krylov_iterations])
        fs.th.release_device()
    assert _rel_l2(flows[0], flows[1]) < 1e-9
    assert any(k > 0 for k in stats[0]) and all(k == 0 for k in stats[1])
    assert np.isclose(flows[0][: flows[0].size * 2 // 3].max(), _U0_MAX_REF, rtol=1e-3)  # sanity: it is the cylinder base flow


@pytest.mark.parametrize("case", ["cavity_coarse", "pinball_middle"])
def test_device_base_flow_matches_golden(case, tmp_path_factory, golden_dir):
    """The reference's base-flow recipes (cavity Re=7500: Picard ×10 → Newton; pinball Re=30: Picard ×15 →
    Newton), every iteration on the device (assembly, factorisation or BiCGStab on lagged factors, sweeps),
    against the oracle's golden base flows — which carry the reference's U0 constants to 1e-13."""
    if case == "cavity_coarse":
        from flowcontrol_amd.examples.cavity.cavityflowsolver import CavityFlowSolver

        fs, n_act, picard = CavityFlowSolver.make_default(Re=7500, path_out=tmp_path_factory.mktemp(case)), 1, 10
    else:
        from flowcontrol_amd.actuator import CYLINDER_ACTUATION_MODE
        from flowcontrol_amd.examples.pinball.pinballflowsolver import PinballFlowSolver

        fs, n_act, picard = PinballFlowSolver.make_default(Re=30, mode_actuation=CYLINDER_ACTUATION_MODE.SUCTION,
                                                           path_out=tmp_path_factory.mktemp(case)), 3, 15
    g = np.load(golden_dir / f"{case}.npz")
    fs.compute_steady_state(method="picard", max_iter=picard, tol=1e-7, u_ctrl=[0.0] * n_act)
    fs.compute_steady_state(method="newton", max_iter=10, u_ctrl=[0.0] * n_act, initial_guess=fs.fields.UP0)
    nv2 = 2 * fs.th.nn
    assert _rel_l2(fs.fields.UP0.vector().get_local()[:nv2], g["UP0"][:nv2]) < 1e-10
    fs.th.release_device()


def test_non_finite_state_is_reported_as_divergence(tmp_path_factory, golden_dir):
    """reference flowsolver.py:727-737,816-819: a non-finite velocity after the solve is FC_ERR_DIVERGED at the C ABI,
    RuntimeError("Failed solving…") from FlowSolver.step — or None when ParamSolver.throw_error is False."""
    from flowcontrol_amd._lib import FC_ERR_DIVERGED, SLOT_BDF2, FcDiverged

    for throw in (True, False):
        fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp(f"diverge{int(throw)}"), num_steps=5)
        fs.params_solver.throw_error = throw
        _load_baseflow(fs, golden_dir)
        fs.initialize_time_stepping(ic=None)
        assert fs.step([0.0, 0.0]) is not None
        dev = fs.th.device()
        u_n, u_nn, p_n = dev.get_state()
        u_bad = u_n.copy()
        u_bad[17] = np.inf
        dev.set_state(u_bad, u_nn, p_n)
        if throw:
            with pytest.raises(FcDiverged) as e:  # the C ABI status itself
                dev.step(SLOT_BDF2, np.zeros(2))
            assert e.value.code == FC_ERR_DIVERGED
            # the C-ABI step shifted the (non-finite) solution into the state; fc_undo_step withdraws it
            dev.undo_step()
            a, b, c = dev.get_state()
            assert np.array_equal(np.isinf(a), np.isinf(u_bad)) and np.array_equal(a[np.isfinite(a)], u_bad[np.isfinite(u_bad)])
            assert np.array_equal(b, u_nn) and np.array_equal(c, p_n)
            with pytest.raises(Exception, match="not a single fc_step"):
                dev.undo_step()  # once only
            with pytest.raises(RuntimeError, match="Failed solving"):
                fs.step([0.0, 0.0])
            # FlowSolver.step leaves the fields as the reference does: untouched by the failed step (flowsolver.py:727-751)
            a, b, c = dev.get_state()
            assert np.array_equal(b, u_nn) and np.array_equal(c, p_n) and np.isinf(a[17])
        else:
            assert fs.step([0.0, 0.0]) is None
            a, b, c = dev.get_state()
            assert np.array_equal(b, u_nn) and np.array_equal(c, p_n) and np.isinf(a[17])
            # a pushed finite state makes the solver usable again
            fs.fields.u_n = Function(fs.V, u_n)
            fs.fields.u_nn = Function(fs.V, u_nn)
            y = fs.step([0.0, 0.0])
            assert y is not None and np.all(np.isfinite(y))
        fs.th.release_device()


def test_host_edits_of_the_state_reach_the_device(tmp_path_factory, golden_dir):
    """Assigning fields.u_n (or fields.push() after an in-place edit) makes the next step start from the edited state,
    as in the reference where these Functions ARE the state (flowsolver.py:746-751)."""
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp("push_a"), num_steps=5)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    _load_baseflow(fs, golden_dir)
    fs.initialize_time_stepping(ic=None)
    fs.step([0.0, 0.0])
    fs.step([0.0, 0.0])
    u_n = fs.fields.u_n.vector().get_local().copy()
    u_nn = fs.fields.u_nn.vector().get_local().copy()
    y_plain = fs.step([0.01, 0.0]).copy()
    # rewind by assignment, halve by an in-place edit + push: both must be seen by the next step
    fs.fields.u_n = Function(fs.V, u_n)
    fs.fields.u_nn = Function(fs.V, u_nn)
    y_again = fs.step([0.01, 0.0]).copy()
    assert np.array_equal(y_again, y_plain)
    fs.fields.u_n = Function(fs.V, u_n)
    fs.fields.u_nn = Function(fs.V, u_nn)
    fs.fields.u_n.vector()[:] = 0.5 * u_n
    fs.fields.push()
    y_half = fs.step([0.01, 0.0]).copy()
    assert not np.allclose(y_half, y_plain, rtol=1e-6)
    fs.th.release_device()


def test_batched_run_logs_like_eager_steps(tmp_path_factory, golden_dir):
    """FlowSolver.run must leave the same log as step(): NaN off the energy_every multiples, checkpoints at save_every."""
    logs = []
    for mode in ("eager", "batched"):
        out = tmp_path_factory.mktemp(f"runlog_{mode}")
        fs = CylinderFlowSolver.make_default(Re=100, path_out=out, num_steps=12, save_every=5)
        fs.params_save.energy_every = 3
        fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
        _load_baseflow(fs, golden_dir)
        fs.initialize_time_stepping(ic=None)
        fs.step([0.0, 0.0])
        if mode == "eager":
            for _ in range(11):
                fs.step([0.0, 0.0])
        else:
            fs.run(11, np.zeros(2))
        ts = fs.timeseries
        logs.append((ts["dE"].to_numpy(), ts[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy(), fs.exporter._checkpoints_written,
                     fs.fields.Usave.vector().get_local().copy()))
        fs.th.release_device()
    (ea, ya, ca, ua), (eb, yb, cb, ub) = logs
    assert np.array_equal(np.isnan(ea), np.isnan(eb)) and np.isnan(ea[1]) and not np.isnan(ea[3])
    assert np.array_equal(ea[~np.isnan(ea)], eb[~np.isnan(eb)]) and np.array_equal(ya, yb)
    assert ca == cb == 2 and np.array_equal(ua, ub)


def test_make_solver_plugin_written_against_the_reference_contract(tmp_path_factory, golden_dir):
    """The reference's documented plug-in point (docs/numerical-details.md:44-48, flowsolver.py:812-814): a subclass
    overrides ``_make_solver`` and returns any object with ``set_operator(A)`` / ``solve(x, b)``.  A SuperLU solver that
    reads the operator the way reference code does (``A.mat().getValuesCSR()``) drops in and reproduces the oracle's series;
    the default device solver accepts the operator as a scipy matrix too (reference ``set_operator(A)``) and then gives the
    same steps as through its slot fast path."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    from flowcontrol_amd._lib import SLOT_BDF2
    from flowcontrol_amd.flowsolver import _DeviceNDSolver

    class SuperLUSolver:
        def set_operator(self, A):
            indptr, indices, data = A.mat().getValuesCSR()
            self.lu = spla.splu(sp.csr_matrix((data, indices, indptr), shape=A.shape).tocsc())

        def solve(self, x, b):
            x[:] = self.lu.solve(b)

    class PluggedCylinder(CylinderFlowSolver):
        def _make_solver(self, order):
            return SuperLUSolver()

    g = np.load(golden_dir / "cylinder_O1.npz")
    fs = PluggedCylinder.make_default(Re=100, path_out=tmp_path_factory.mktemp("plugin"), num_steps=8)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    _load_baseflow(fs, golden_dir)
    fs.initialize_time_stepping(ic=None)
    for _ in range(8):
        fs.step(u_ctrl=[0.0, 0.0])
    ts = fs.timeseries
    assert _rel_l2(ts[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy(), g["ol_y"][:9]) < 1e-8
    assert _rel_l2(ts["dE"].to_numpy(), g["ol_dE"][:9]) < 1e-8
    fs.th.release_device()

    # the default solver fed with the operator as a MATRIX (reference contract) vs its slot fast path
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp("plugin_b"), num_steps=4)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    _load_baseflow(fs, golden_dir)
    fs.initialize_time_stepping(ic=None)
    y_slot = [fs.step(u_ctrl=[0.01, -0.01]).copy() for _ in range(3)]
    dev = fs.th.device()
    A = dev.matrix(SLOT_BDF2)
    s2 = _DeviceNDSolver(fs, slot=SLOT_BDF2)
    s2.set_operator(1.0 * A)  # a scipy matrix: values are uploaded, then factorised
    b = np.random.default_rng(2).standard_normal(dev.N)
    x = np.zeros(dev.N)
    s2.solve(x, b)
    assert np.linalg.norm(A @ x - b) < 1e-10 * np.linalg.norm(b)
    with pytest.raises(ValueError):
        s2.set_operator(A + sp.csr_matrix(([1.0], ([0], [dev.N - 1])), shape=A.shape))  # an entry outside the pattern
    fs.initialize_time_stepping(ic=None)
    y_again = [fs.step(u_ctrl=[0.01, -0.01]).copy() for _ in range(3)]
    assert np.allclose(np.array(y_again), np.array(y_slot), rtol=1e-12, atol=1e-15)
    fs.th.release_device()


def test_base_flow_continuation_in_reynolds_number(tmp_path_factory):
    """The reference's Re-continuation script for the lid-driven cavity (compute_steady_state_increasing_Re.py): each base flow
    starts from the previous one, read back from the files the script writes; the last one is a converged steady state of ITS
    Reynolds number (residual of the steady equations) and is what a solver loads with load_steady_state(path_u_p=...)."""
    from flowcontrol_amd.examples.lidcavity.compute_steady_state_increasing_Re import continuation
    from flowcontrol_amd.examples.lidcavity.lidcavityflowsolver import LidCavityFlowSolver

    out = tmp_path_factory.mktemp("lid_continuation")
    found = continuation(re_list=[400, 1000, 2000], path_out=out, picard_iterations=6, newton_iterations=15)
    assert sorted(found) == [400, 1000, 2000] and (out / "steady" / "U0_Re=2000.xdmf").exists()
    fs = LidCavityFlowSolver.make_default(Re=2000, path_out=out, num_steps=3)
    fs.load_steady_state(path_u_p=[out / "steady" / "U0_Re=2000.xdmf", out / "steady" / "P0_Re=2000.xdmf"])
    assert np.allclose(fs.fields.U0.vector().get_local(), found[2000][0].vector().get_local(), rtol=0, atol=1e-12)
    # a steady state: the perturbation equations started from zero perturbation stay at zero
    fs.params_ic = ParamIC(xloc=0.5, yloc=0.5, radius=0.1, amplitude=0.0)  # (the default adds a vortex to whatever ic is passed)
    fs.initialize_time_stepping(ic=Function(fs.W))
    for _ in range(3):
        fs.step([0.0])
    assert np.abs(fs.fields.u_n.vector().get_local()).max() < 1e-8
    # ... and not the base flow of a lower Reynolds number
    assert np.abs(found[2000][0].vector().get_local() - found[1000][0].vector().get_local()).max() > 1e-3
    fs.th.release_device()
