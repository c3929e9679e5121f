"""One-launch factor apply (``fc_nd_dag``: per-node arrival counters instead of a launch per tree level)
against the level launches it replaces, through the C ABI.

Same factors, same right-hand sides: the two applies differ only in summation order (lanes per row), so the
solutions must agree to round-off; the one-launch apply must be bit-reproducible from run to run (any stale
hand-off between workgroups would change bits), and a give-up must be redone transparently.
"""
import numpy as np
import pytest

from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
from flowcontrol_amd.fem.spaces import Function
from flowcontrol_amd.flowsolverparameters import ParamIC

pytestmark = pytest.mark.gpu


def _enable_or_skip(dev):
    """The one-launch apply is an optional part of the library (``FC_HIPCC_FLAGS=-DFC_WITH_DAG``): a default build
    refuses fc_set_dag(1), and these A/B tests have nothing to compare."""
    from flowcontrol_amd._lib import FcError

    try:
        dev.set_dag(True)
    except FcError as e:
        assert "not part of this build" in str(e)
        pytest.skip("libfc_hip.so built without -DFC_WITH_DAG")


def _solver(tmp, golden_dir, n=10):
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp, num_steps=n)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    up0 = np.load(golden_dir / "cylinder_O1.npz")["UP0"]
    U0, P0 = Function(fs.W, up0).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    return fs


def _u(n):
    k = np.arange(n)
    return np.stack([0.05 * np.sin(0.01 * k), -0.02 * np.cos(0.013 * k)], axis=1)


def test_one_launch_apply_matches_level_launches(tmp_path_factory, golden_dir):
    from flowcontrol_amd._lib import SLOT_BDF2

    fs = _solver(tmp_path_factory.mktemp("dag_ab"), golden_dir)
    fs.step([0.0, 0.0])
    dev = fs.th.device()
    _enable_or_skip(dev)
    info = dev.dag_info(SLOT_BDF2)
    assert info["enabled"] and info["tasks"] > 1000 and info["failures"] == 0
    rng = np.random.default_rng(0)
    for k in range(4):
        b = rng.standard_normal(dev.N)
        dev.set_dag(True)
        x1, i1 = dev.solve(SLOT_BDF2, b)
        x1b, _ = dev.solve(SLOT_BDF2, b)
        dev.set_dag(False)
        x0, i0 = dev.solve(SLOT_BDF2, b)
        assert np.array_equal(x1, x1b)  # bit-reproducible
        assert np.linalg.norm(x1 - x0) <= 1e-12 * np.linalg.norm(x0)
        assert i1[1] < 1e-12 and i0[1] < 1e-12
    dev.set_dag(True)
    assert dev.dag_info(SLOT_BDF2)["failures"] == 0
    fs.th.release_device()


def test_one_launch_apply_trajectory_and_reproducibility(tmp_path_factory, golden_dir):
    """2 000 actuated steps: the one-launch trajectory is identical from run to run and follows the
    level-launch trajectory to 1e-10."""
    n = 2000
    u = _u(n)
    runs = []
    for mode in ("dag", "dag", "levels"):
        fs = _solver(tmp_path_factory.mktemp(f"dag_{mode}"), golden_dir, n)
        fs.step(u[0])
        dev = fs.th.device()
        if mode == "dag":
            _enable_or_skip(dev)
        else:
            dev.set_dag(False)
        y = np.array([fs.step(u[k]).copy() for k in range(1, n)])
        runs.append((y, fs.timeseries["dE"].to_numpy()[2:].copy(), dev.get_solution()))
        if mode == "dag":
            from flowcontrol_amd._lib import SLOT_BDF2

            assert dev.dag_info(SLOT_BDF2) == {"tasks": dev.dag_info(SLOT_BDF2)["tasks"], "enabled": True, "failures": 0}
        fs.th.release_device()
    (ya, ea, xa), (yb, eb, xb), (yc, ec, xc) = runs
    assert np.array_equal(ya, yb) and np.array_equal(ea, eb) and np.array_equal(xa, xb)
    assert np.linalg.norm(ya - yc) <= 1e-10 * np.linalg.norm(yc)
    assert np.linalg.norm(xa - xc) <= 1e-10 * np.linalg.norm(xc)


def test_give_up_is_redone_with_level_launches(tmp_path_factory, golden_dir):
    """A workgroup that gives up waiting raises the error word; the step's tail must then leave the state
    untouched and fc_step / fc_run must redo the step with the level launches (same trajectory)."""
    from flowcontrol_amd._lib import SLOT_BDF2, check

    n = 40
    u = _u(n)
    ref = _solver(tmp_path_factory.mktemp("dag_ref"), golden_dir, n)
    y_ref = np.array([ref.step(u[k]).copy() for k in range(n)])
    ref.th.release_device()
    # eager: failure injected behind the apply of step 11
    fs = _solver(tmp_path_factory.mktemp("dag_inj"), golden_dir, n)
    fs.step(u[0])
    dev = fs.th.device()
    _enable_or_skip(dev)
    ys = [fs.y_meas.copy()] + [fs.step(u[k]).copy() for k in range(1, 10)]
    check(dev.lib.fc_debug_inject_dag_failure(dev._h, 0))
    ys += [fs.step(u[k]).copy() for k in range(10, 20)]
    info = dev.dag_info(SLOT_BDF2)
    assert info["failures"] == 1 and not info["enabled"]
    # batched: switch the one-launch apply back on, fail in the 6th step of the batch
    dev.set_dag(True)
    check(dev.lib.fc_debug_inject_dag_failure(dev._h, 5))
    yb, _ = fs.run(20, u[20:])
    assert dev.dag_info(SLOT_BDF2)["failures"] == 2
    y = np.vstack([np.array(ys), yb])
    assert np.linalg.norm(y - y_ref) <= 1e-10 * np.linalg.norm(y_ref)
    fs.th.release_device()
