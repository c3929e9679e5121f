"""Shared-operator batched stepping (``fc_step_batch`` / ``BatchedFlowSolver``): k lock-step simulations on one handle.

The reference runs such sweeps as k independent ``FlowSolver`` instances (IC sweeps
``src/examples/lidcavity/batch_run_lidcavity.py:197-215``, controller optimisation ``src/utils/optim.py:95-102``), each
calling ``FlowSolver.step`` (``src/flowcontrol/flowsolver.py:703-799``).  Parity bar: every trajectory of a batch equals
its own single run (same device, public ``FlowSolver.step``) to 1e-12 and, where a golden series of the CPU oracle exists
for the scenario, that series to 1e-8 — the batched factor apply is a different summation order on the fp64 matrix
cores, nothing else.
"""
import ctypes as C
import tempfile

import numpy as np
import pytest

from flowcontrol_amd.controller import Controller
from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
from flowcontrol_amd.fem.spaces import Function
from flowcontrol_amd.flowsolverparameters import ParamIC
from flowcontrol_amd.examples.data import controller_file  # noqa: E402

pytestmark = pytest.mark.gpu

N_STEPS = 20


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


def _ycols(ts):
    return [c for c in ts.columns if c.startswith("y_meas_")]


def _solver(golden_dir, **kw):
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tempfile.mkdtemp(), num_steps=N_STEPS, **kw)
    U0, P0 = Function(fs.W, np.load(golden_dir / "cylinder_O1.npz")["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    return fs


class _Scenario:
    """One trajectory: an initial condition and a control law u = law(step index, y_meas before the step)."""

    def __init__(self, ic, make_law):
        self.ic, self.make_law = ic, make_law


def _scenarios(golden_dir):
    kfile = controller_file()

    def open_loop(amp, w):
        return lambda: (lambda n, y: np.array([amp * np.sin(w * n), -0.5 * amp * np.cos(0.7 * w * n)]))

    def feedback(gain):
        def make():
            K = Controller.from_file(file=kfile, x0=None)

            def law(n, y):
                u = K.step(y=-gain * y[0], dt=0.005)
                return np.array([u[0], u[0]])

            return law

        return make

    def lowpass(c):
        def make():
            K = Controller(A=[[-40.0]], B=[[1.0]], C=[[c]], D=[[0.0]])
            return lambda n, y: np.array([K.step(y=y[1], dt=0.005)[0], -K.x[0] * c])

        return make

    return [
        _Scenario(ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0), open_loop(0.0, 0.0)),  # the oracle's open-loop fixture (ol_*)
        _Scenario(ParamIC(xloc=0.0, yloc=0.0, radius=1.0, amplitude=1.0), feedback(1.0)),  # the oracle's closed-loop fixture (cl_*)
        _Scenario(ParamIC(xloc=2.5, yloc=0.2, radius=0.6, amplitude=0.7), open_loop(0.05, 0.3)),
        _Scenario(ParamIC(xloc=1.5, yloc=-0.3, radius=0.4, amplitude=1.3), open_loop(0.02, 0.9)),
        _Scenario(ParamIC(xloc=0.0, yloc=0.0, radius=1.0, amplitude=0.5), feedback(0.5)),
        _Scenario(ParamIC(xloc=3.0, yloc=0.0, radius=0.8, amplitude=1.0), feedback(2.0)),
        _Scenario(ParamIC(xloc=2.0, yloc=0.5, radius=0.5, amplitude=-1.0), lowpass(30.0)),
        _Scenario(ParamIC(xloc=4.0, yloc=-0.5, radius=1.0, amplitude=2.0), lowpass(-80.0)),
    ]


@pytest.fixture(scope="module")
def single_runs(golden_dir):
    """Every scenario on its own through the public FlowSolver.step (one device handle, re-initialised per run)."""
    fs = _solver(golden_dir)
    out = []
    for sc in _scenarios(golden_dir):
        fs.params_ic = sc.ic
        fs.initialize_time_stepping(ic=None)
        law = sc.make_law()
        for n in range(N_STEPS):
            fs.step(u_ctrl=law(n, fs.y_meas))
            assert fs.solve_info[1] < 1e-12
        ts = fs.timeseries
        out.append((ts[_ycols(ts)].to_numpy().copy(), ts["dE"].to_numpy().copy(), ts[[c for c in ts.columns if c.startswith("u_ctrl_")]].to_numpy()[1:].copy()))
    yield fs, out
    fs.th.release_device()


@pytest.mark.parametrize("k", [8, 3, 16, 20])
def test_batched_trajectories_equal_their_single_runs_and_the_oracle(k, single_runs, golden_dir):
    """k trajectories with different initial conditions, control inputs and controllers (closed loops through host
    Controllers that see only their own run's measurements), advanced together: each must equal its single run to 1e-12;
    runs 0 and 1 are the oracle's open-loop and closed-loop fixtures and must match those to 1e-8."""
    from flowcontrol_amd.batch import BatchedFlowSolver

    fs, singles = single_runs
    scen = _scenarios(golden_dir)
    pick = [i % len(scen) for i in range(k)]  # k = 16: every scenario twice (two columns of one batch must agree bit for bit)
    bfs = BatchedFlowSolver(fs, k)
    bfs.initialize_time_stepping(ics=[scen[i].ic for i in pick])
    laws = [scen[i].make_law() for i in pick]
    for n in range(N_STEPS):
        u = np.stack([laws[j](n, bfs.y_meas[j]) for j in range(k)])
        assert bfs.step(u) is not None
        assert np.all(bfs.solve_info[:, 1] < 1e-12)
    assert np.isclose(bfs.t, N_STEPS * 0.005)
    for j, i in enumerate(pick):
        ts = bfs.timeseries(j)
        y1, dE1, u1 = singles[i]
        assert _rel(ts[_ycols(ts)].to_numpy(), y1) < 1e-12, f"run {j} (scenario {i})"
        assert _rel(ts["dE"].to_numpy(), dE1) < 1e-12
        if np.abs(u1).max() > 0:
            assert _rel(ts[[c for c in ts.columns if c.startswith("u_ctrl_")]].to_numpy()[1:], u1) < 1e-11
    g = np.load(golden_dir / "cylinder_O1.npz")
    ts0 = bfs.timeseries(0)
    assert _rel(ts0[_ycols(ts0)].to_numpy(), g["ol_y"][: N_STEPS + 1]) < 1e-8
    assert _rel(ts0["dE"].to_numpy(), g["ol_dE"][: N_STEPS + 1]) < 1e-8
    if k > 1:
        ts1 = bfs.timeseries(1)
        assert _rel(ts1[_ycols(ts1)].to_numpy(), g["cl_y"][: N_STEPS + 1]) < 1e-8
        assert _rel(ts1["dE"].to_numpy(), g["cl_dE"][: N_STEPS + 1]) < 1e-8
    if k == 16:
        for j in range(8):  # the same scenario in two columns of the batch: bit-identical
            a, b = bfs.timeseries(j), bfs.timeseries(j + 8)
            assert np.array_equal(a[_ycols(a)].to_numpy(), b[_ycols(b)].to_numpy())
    # fields: the batched state of run 2 is the single run's final state
    u_n, u_nn, p_n = bfs.state()
    fs.params_ic = scen[pick[2]].ic
    fs.initialize_time_stepping(ic=None)
    law = scen[pick[2]].make_law()
    for n in range(N_STEPS):
        fs.step(u_ctrl=law(n, fs.y_meas))
    assert _rel(u_n[2], fs.fields.u_n.vector().get_local()) < 1e-12
    assert _rel(u_nn[2], fs.fields.u_nn.vector().get_local()) < 1e-12
    assert _rel(p_n[2], fs.fields.p_n.vector().get_local()) < 1e-11
    bfs.close()


def test_batched_factor_apply_matches_single_solves(single_runs):
    """fc_solve_batch: k random right-hand sides through the matrix-core block sweeps vs fc_solve one by one."""
    from flowcontrol_amd._lib import SLOT_BDF1, SLOT_BDF2

    fs, _ = single_runs
    dev = fs.th.device()
    rng = np.random.default_rng(3)
    for k in (1, 5, 8, 13, 16, 27, 32):
        dev.set_batch(k)
        for slot in (SLOT_BDF1, SLOT_BDF2):
            B = rng.standard_normal((k, dev.N))
            X = dev.solve_batch(slot, B)
            for s in range(k):
                x1, info = dev.solve(slot, B[s])
                assert _rel(X[s], x1) < 1e-12
        info = dev.batch_info()
        assert info["k"] == k and info["KB"] in (4, 8, 16, 32) and info["KB"] >= k
        assert info["factor_bytes"] == 8.0 * dev.factor_nnz[SLOT_BDF2] or info["factor_bytes"] > 0
    dev.set_batch(0)


def test_one_diverged_run_does_not_touch_the_others(single_runs, golden_dir):
    """A non-finite state in ONE column of the batch: fc_step_batch reports FC_ERR_DIVERGED and marks that run only; the
    other runs' measurements are what a clean batch gives (the columns of the block sweeps are independent)."""
    from flowcontrol_amd._lib import SLOT_BDF1, FcDiverged

    fs, _ = single_runs
    dev = fs.th.device()
    k = 4
    dev.set_batch(k)
    nn2, nv = 2 * fs.th.nn, fs.th.nv
    rng = np.random.default_rng(5)
    u0 = 1e-3 * rng.standard_normal((k, nn2))
    dev.set_state_batch(u0, u0, np.zeros((k, nv)))
    y_clean, _, _ = dev.step_batch(SLOT_BDF1, np.zeros((k, 2)))
    bad = u0.copy()
    bad[2, 17] = np.inf
    dev.set_state_batch(bad, bad, np.zeros((k, nv)))
    with pytest.raises(FcDiverged):
        dev.step_batch(SLOT_BDF1, np.zeros((k, 2)))
    info = dev._batch_bufs[4]
    assert info[2, 3] != 0 and np.all(info[[0, 1, 3], 3] == 0)
    y = dev._batch_bufs[2][:, : dev.n_sens]
    assert np.array_equal(y[[0, 1, 3]], y_clean[[0, 1, 3]])
    dev.set_batch(0)


def test_speculative_element_loop_is_dropped_when_the_state_or_the_scheme_changes(single_runs):
    """A batched step ends with the element loop of the NEXT step (predicted scheme, the state as it stands).  Replacing the state,
    or asking for another scheme than predicted, must make the next step redo that loop: the same three steps from the same state
    give the same bits, whatever ran in between."""
    from flowcontrol_amd._lib import SLOT_BDF1, SLOT_BDF2

    fs, _ = single_runs
    dev = fs.th.device()
    k = 5
    dev.set_batch(k)
    nn2, nv = 2 * fs.th.nn, fs.th.nv
    rng = np.random.default_rng(11)
    u0 = 1e-3 * rng.standard_normal((k, nn2))
    u_ctrl = 1e-2 * rng.standard_normal((3, k, 2))

    def three_steps():
        dev.set_state_batch(u0, u0, np.zeros((k, nv)))
        out = []
        for n, slot in enumerate((SLOT_BDF1, SLOT_BDF2, SLOT_BDF2)):
            y, dE, _ = dev.step_batch(slot, u_ctrl[n])
            out.append((y.copy(), dE.copy()))
        return out

    ref = three_steps()
    # (a) straight again: the state is replaced while the speculative loop of a fourth step is on the device
    again = three_steps()
    # (b) a BDF1 step where BDF2 was predicted, then the three steps
    dev.step_batch(SLOT_BDF1, u_ctrl[0])
    dev.step_batch(SLOT_BDF1, u_ctrl[1])
    third = three_steps()
    for a, b, c in zip(ref, again, third):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    dev.set_batch(0)


def test_batch_api_refuses_what_it_cannot_do(single_runs):
    from flowcontrol_amd import _lib
    from flowcontrol_amd._lib import SLOT_BDF1

    fs, _ = single_runs
    dev = fs.th.device()
    with pytest.raises(_lib.FcError):
        dev.set_batch(33)
    dev.set_batch(0)
    with pytest.raises(_lib.FcError):  # no batch allocated
        _lib.check(dev.lib.fc_step_batch(dev._h, SLOT_BDF1, 4, None, None, None, None, 1, None))
    dev.set_batch(4)
    with pytest.raises(_lib.FcError):  # k differs from fc_set_batch
        _lib.check(dev.lib.fc_step_batch(dev._h, SLOT_BDF1, 3, None, None, None, None, 1, None))
    dev.set_solver_options(refine=1)
    with pytest.raises(_lib.FcError):  # refinement sweeps are a single-simulation feature
        dev.step_batch(SLOT_BDF1, np.zeros((4, 2)))
    dev.set_solver_options(refine=0)
    dev.set_batch(0)


def test_batched_cavity_force_actuation_closed_loop(golden_dir):
    """BASELINE config 3's ingredients in a batch (cavity_coarse): body-force actuator (per-run amplitudes enter the element
    loop), wall-shear sensor, low-pass Controller per run with different gains; run 0 (gain of the committed scenario) against
    the oracle's closed-loop series tests/golden/cavity_coarse_re7500.npz, all runs against their single runs."""
    import sys

    sys.path.insert(0, str(golden_dir))
    from flowcontrol_amd.examples.cavity.scenarios import CAVITY_K

    from flowcontrol_amd.batch import BatchedFlowSolver
    from flowcontrol_amd.examples.cavity.cavityflowsolver import CavityFlowSolver

    g = np.load(golden_dir / "cavity_coarse.npz")
    ref = np.load(golden_dir / "cavity_coarse_re7500.npz")
    fs = CavityFlowSolver.make_default(Re=7500, path_out=tempfile.mkdtemp(), num_steps=N_STEPS)
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    gains = [1.0, 0.0, -2.0, 5.0]
    k = len(gains)

    def controller(gain):
        return Controller(A=CAVITY_K["A"], B=CAVITY_K["B"], C=[[gain * CAVITY_K["C"][0][0]]], D=CAVITY_K["D"])

    singles = []
    for gain in gains:
        fs.initialize_time_stepping(ic=None)
        K, y0 = controller(gain), fs.y_meas[0]
        for _ in range(N_STEPS):
            fs.step(u_ctrl=K.step(y=fs.y_meas[0] - y0, dt=fs.params_time.dt))
        ts = fs.timeseries
        singles.append((ts[_ycols(ts)].to_numpy().copy(), ts["dE"].to_numpy().copy()))
    assert _rel(singles[0][0], ref["y"]) < 1e-8 and _rel(singles[0][1], ref["dE"]) < 1e-8
    bfs = BatchedFlowSolver(fs, k)
    bfs.initialize_time_stepping()
    Ks, y0 = [controller(gn) for gn in gains], bfs.y_meas[:, 0].copy()
    for _ in range(N_STEPS):
        bfs.step(np.stack([Ks[j].step(y=bfs.y_meas[j, 0] - y0[j], dt=fs.params_time.dt) for j in range(k)]))
    for j in range(k):
        ts = bfs.timeseries(j)
        assert _rel(ts[_ycols(ts)].to_numpy(), singles[j][0]) < 1e-12
        assert _rel(ts["dE"].to_numpy(), singles[j][1]) < 1e-12
    ts0 = bfs.timeseries(0)
    assert _rel(ts0[_ycols(ts0)].to_numpy(), ref["y"]) < 1e-8
    assert np.abs(ref["u"]).max() > 0.5  # the loop acts
    bfs.close()
    fs.th.release_device()


def test_lidcavity_ic_sweep_example_matches_single_runs(tmp_path_factory):
    """The reference's IC-sweep script (src/examples/lidcavity/batch_run_lidcavity.py) as ONE batch: an enclosed flow (the pressure
    pin lives in the factorisation), k = 4 vortex positions, the snapshot files the reference writes — each run against a single
    ``FlowSolver`` started from the same initial condition."""
    from flowcontrol_amd.examples.lidcavity.batch_run_lidcavity import run_lidcavity_with_ics
    from flowcontrol_amd.examples.lidcavity.lidcavityflowsolver import LidCavityFlowSolver
    from flowcontrol_amd.flowsolverparameters import ParamIC

    out = tmp_path_factory.mktemp("lid_batch")
    ics = [ParamIC(xloc=x, yloc=y, radius=0.1, amplitude=0.1) for x, y in ((0.2, 0.2), (0.8, 0.2), (0.5, 0.5), (0.2, 0.8))]
    dirs = [out / f"run{i + 1}" for i in range(len(ics))]
    bfs = run_lidcavity_with_ics(1000.0, ics, dirs, num_steps=12, save_every=4, picard_iterations=12)
    fs = bfs.fs
    U0 = fs.fields.U0.vector().get_local()
    for i in (0, 2):
        U = np.load(dirs[i] / "U_field_alldata.npy")
        assert U.shape == (fs.V.dim(), 3, 1) and np.load(dirs[i] / "UP0_field_data.npy").shape == (fs.W.dim(),)
        single = LidCavityFlowSolver.make_default(Re=1000.0, path_out=out / f"single{i}", num_steps=12, save_every=0)
        single._assign_steady_state(fs.fields.U0, fs.fields.P0)
        single.params_ic = ics[i]
        single.initialize_time_stepping(ic=None)
        for n in range(12):
            single.step([0.0])
        u_single = single.fields.u_n.vector().get_local() + U0
        assert np.linalg.norm(U[:, -1, 0] - u_single) <= 1e-10 * np.linalg.norm(u_single)
        ts, tb = single.timeseries, bfs.timeseries(i)
        yc = [c for c in ts.columns if c.startswith("y_meas_")]
        assert np.allclose(tb[yc].to_numpy(), ts[yc].to_numpy(), rtol=1e-9, atol=1e-13)
        single.th.release_device()
    bfs.close()
    fs.th.release_device()


def test_batched_residual_monitor_cadence_and_the_gather_that_runs_ahead(golden_dir):
    """``check_residual_every = n`` on the batched path: the residual is formed on every n-th batched step (NaN in between, the
    non-finite test stays on every step) and the trajectories are those of the every-step run (sensors bit for bit) — as they are with
    the control-independent right-hand side gathered one step ahead (the default) or not (FC_SPECULATE_GATHER=0 is read once
    per process, so the second property is checked against the single runs by the parity test above and here by actuating)."""
    from flowcontrol_amd.batch import BatchedFlowSolver

    k, n = 5, 13
    ics = [ParamIC(xloc=2.0 + 0.2 * i, yloc=0.05 * i, radius=0.5, amplitude=1.0 + 0.1 * i) for i in range(k)]
    us = [np.array([[0.03 * (i + 1) * np.sin(0.4 * m), -0.02 * i * np.cos(0.3 * m)] for i in range(k)]) for m in range(n)]
    out = {}
    for every in (1, 4):
        fs = _solver(golden_dir)
        fs.check_residual_every = every
        bfs = BatchedFlowSolver(fs, k)
        bfs.initialize_time_stepping(ics=ics)
        res = []
        for m in range(n):
            bfs.step(us[m])
            res.append(bfs.solve_info[:, 1].copy())
        out[every] = ([bfs.timeseries(i) for i in range(k)], np.array(res))
        bfs.close()
        fs.th.release_device()
    ts1, r1 = out[1]
    ts4, r4 = out[4]
    assert np.all(r1 < 1e-12)
    checked = np.arange(n) % 4 == 0
    assert np.all(r4[checked] < 1e-12) and np.all(np.isnan(r4[~checked]))
    for a, b in zip(ts1, ts4):
        assert np.array_equal(a[_ycols(a)].to_numpy(), b[_ycols(b)].to_numpy())
        # (the energy's partial sums sit at other positions of the fold when the row blocks are absent: another summation tree)
        assert np.allclose(a["dE"].to_numpy(), b["dE"].to_numpy(), rtol=1e-13, atol=0.0)


_SPLIT_SCRIPT = r"""
import sys, tempfile, numpy as np
from flowcontrol_amd._lib import SLOT_BDF2
from flowcontrol_amd.batch import BatchedFlowSolver
from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
from flowcontrol_amd.fem.spaces import Function
from flowcontrol_amd.flowsolverparameters import ParamIC
fs = CylinderFlowSolver.make_default(Re=100, path_out=tempfile.mkdtemp(), num_steps=8)
U0, P0 = Function(fs.W, np.load(sys.argv[1])["UP0"]).split()
fs._assign_steady_state(U0, P0)
fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
fs.initialize_time_stepping(ic=None)
u = lambda n: np.array([0.3 * np.sin(0.4 * n), -0.2 * np.cos(0.3 * n)])
y1 = np.array([fs.step(u_ctrl=u(n)).copy() for n in range(8)])
dev = fs.th.device()
rng = np.random.default_rng(11)
worst = 0.0
for k in (7, 16, 32):
    dev.set_batch(k)
    B = rng.standard_normal((k, dev.N))
    X = dev.solve_batch(SLOT_BDF2, B)
    for _ in range(3):
        assert np.array_equal(dev.solve_batch(SLOT_BDF2, B), X), "a split tile's sum depends on the arrival order of its parts"
    for s in range(k):
        x1, _ = dev.solve(SLOT_BDF2, B[s])
        worst = max(worst, np.linalg.norm(X[s] - x1) / np.linalg.norm(x1))
dev.set_batch(0)
# the batched STEP (element loop, gather, sweeps, tail) against the single run above
for k in (3, 32):
    bfs = BatchedFlowSolver(fs, k)
    bfs.initialize_time_stepping(ics=[fs.params_ic] * k)
    yb = np.array([bfs.step(np.tile(u(n), (k, 1))).copy() for n in range(8)])
    for j in range(k):
        worst = max(worst, np.linalg.norm(yb[:, j] - y1) / np.linalg.norm(y1))
    bfs.close()
print("WORST", worst)
"""


@pytest.mark.parametrize("knobs", [{"FC_BATCH_SPLIT": "2"}, {"FC_BATCH_SPLIT": "0", "FC_BATCH_ELEM": "lds"}], ids=["split2", "nosplit_ldselem"])
def test_split_tiles_add_their_parts_in_a_fixed_order(knobs, golden_dir):
    """Wide tiles of the batched block sweeps are cut into parts (one workgroup each, ``FcBTask::split``); the tile's last-arriving
    part adds the partial products in part order.  FC_BATCH_SPLIT=2 cuts every tile of three or more 32-column chunks (the default, 16,
    only the levels near the root), 0 none: the batched apply must equal the single solves either way and be bit-reproducible from
    call to call.  The second case also runs the LDS-shared element loop (``fc_rhs_elem_b``; the default is the all-register
    ``fc_rhs_elem_breg``): batched steps must equal the single run with either.  (The knobs are read once per process: a child process.)"""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, PYTHONPATH=str(root), **knobs)
    out = subprocess.run([sys.executable, "-c", _SPLIT_SCRIPT, str(golden_dir / "cylinder_O1.npz")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    worst = float([ln for ln in out.stdout.splitlines() if ln.startswith("WORST")][0].split()[1])
    assert worst < 1e-12
