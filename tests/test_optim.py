"""Optimisation helpers (flowcontrol_amd/optim.py; reference src/utils/optim.py): scalar helpers on the CPU, the batched
closed-loop cost evaluation against one FlowSolver run per candidate on the GPU."""
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from flowcontrol_amd import optim
from flowcontrol_amd.examples.data import controller_file  # noqa: E402

GOLDEN = Path(__file__).resolve().parent / "golden"


def test_cost_functions_and_bookkeeping(tmp_path):
    s = pd.Series([1.0, 2.0, 3.0])
    assert optim.compute_signal_cost(s, 0.5, "integral") == 3.0
    assert optim.compute_signal_cost(s, 0.5, "terminal") == 3.0
    assert optim.compute_signal_cost(s, 0.5, "integral", scaling=lambda v: v**2) == 7.0
    with pytest.raises(ValueError, match="Unknown criterion"):
        optim.compute_signal_cost(s, 1.0, "mean")
    u = pd.DataFrame({"u_ctrl_1": [np.nan, 1.0, 2.0], "u_ctrl_2": [np.nan, -1.0, 0.5]})  # the IC row carries no command
    assert optim.compute_control_cost(u, 2.0) == pytest.approx(2.0 * (1 + 4 + 1 + 0.25))
    best, idx = optim.cummin(np.array([[3.0], [2.0], [2.5], [1.0], [1.0]]))
    assert best.ravel().tolist() == [3.0, 2.0, 2.0, 1.0, 1.0] and idx.tolist() == [0, 1, 1, 3, 3]
    assert optim.fun_array(np.array([[1.0, 2.0], [3.0, 4.0]]), lambda r, a=0.0: r.sum() + a, a=1.0).tolist() == [[4.0], [8.0]]
    optim.write_results(np.array([[0.0, 1.0], [2.0, 3.0], [4.0, 5.0]]), [3.0, 1.0, 2.0], tmp_path, verbose=False)
    all_ = pd.read_csv(tmp_path / "J_costfun.csv")
    run = pd.read_csv(tmp_path / "J_costfun_cummin.csv")
    assert list(all_.columns) == ["J", "x0", "x1"] and all_["J"].tolist() == [3.0, 1.0, 2.0]
    assert run["J"].tolist() == [3.0, 1.0, 1.0] and run["x0"].tolist() == [0.0, 2.0, 2.0]
    optim.write_optim_csv(all_, tmp_path, diverged=True, iteration=7)
    assert (tmp_path / "timeseries" / "timeseries_iter_0007_DIVERGED.csv").exists()
    # NaN costs (closed_loop_costs marks runs that were stopped early so): the bookkeeping still writes its files
    best, idx = optim.cummin(np.array([[3.0], [1.0], [np.nan], [0.5]]))
    assert best.ravel()[:2].tolist() == [3.0, 1.0] and idx.tolist()[:2] == [0, 1] and idx.shape == (4,)
    optim.write_results(np.array([[0.0], [1.0], [2.0]]), [2.0, np.nan, np.inf], tmp_path, verbose=False)
    assert pd.read_csv(tmp_path / "J_costfun.csv").shape == (3, 2) and pd.read_csv(tmp_path / "J_costfun_cummin.csv").shape == (3, 2)


def test_sobol_sample_is_the_low_discrepancy_sequence_in_its_box():
    a = optim.sobol_sample(3, 8)
    assert a.shape == (8, 3) and np.all((a >= 0) & (a < 1))
    assert np.array_equal(a, optim.sobol_sample(3, 8))  # deterministic without a seed
    assert not np.array_equal(a, optim.sobol_sample(3, 8, seed=1))
    b = optim.sobol_sample(3, 8, xlimits=[[0, 1], [10, 20], [-2, 2]])
    assert np.allclose(b, np.array([0, 10, -2]) + a * np.array([1, 10, 4]))
    assert np.allclose(optim.sobol_sample(3, 8, xlimits=np.array([[0, 10, -2], [1, 20, 2]])), b)  # (2, ndim) bounds
    with pytest.raises(ValueError, match="wrong shape"):
        optim.sobol_sample(3, 2, xlimits=[[0, 1]])


@pytest.mark.gpu
def test_batched_candidates_cost_what_their_single_runs_cost(tmp_path_factory):
    """Four candidate controllers (the shipped LTI controller at four gains, one of them zero) evaluated together: each cost
    equals the cost of a FlowSolver of its own driven by that controller."""
    from flowcontrol_amd.controller import Controller
    from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC

    g = np.load(GOLDEN / "cylinder_O1.npz")
    K0 = Controller.from_file(file=controller_file(), x0=None)
    gains = [0.0, 0.5, 1.0, 2.0]
    make = lambda a: Controller(A=K0.A, B=K0.B, C=a * K0.C, D=a * K0.D)  # noqa: E731
    n, pen = 10, 0.3

    def solver(tag):
        fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp(tag), num_steps=n)
        fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
        U0, P0 = Function(fs.W, g["UP0"]).split()
        fs._assign_steady_state(U0, P0)
        return fs

    fs = solver("batch")
    J, series = optim.closed_loop_costs(fs, [make(a) for a in gains], n, u_penalty=pen)
    again = optim.fun_array_batched(np.array(gains)[:, None], lambda r: make(r[0]), fs, n, batch=3, u_penalty=pen)
    assert np.allclose(again[:, 0], J, rtol=1e-12)
    fs.th.release_device()
    for i, a in enumerate(gains):
        one = solver(f"single{i}")
        one.initialize_time_stepping(ic=None)
        K = make(a)
        for _ in range(n):
            cmd = K.step(y=-one.y_meas[0], dt=one.params_time.dt)
            one.step(u_ctrl=[cmd[0], cmd[0]])
        ts = one.timeseries
        Tnorm = one.params_time.dt / float(ts["time"].iloc[-1])
        ref = optim.compute_signal_cost(ts["dE"], Tnorm, "integral") + pen * optim.compute_control_cost(ts[["u_ctrl_1", "u_ctrl_2"]], Tnorm)
        assert J[i] == pytest.approx(ref, rel=1e-10)
        assert np.allclose(series[i]["y_meas_1"].to_numpy(), ts["y_meas_1"].to_numpy(), rtol=1e-9, atol=1e-13)
        one.th.release_device()
    assert J[0] != J[2]  # the controller does change the cost


@pytest.mark.gpu
def test_a_diverging_candidate_ends_alone(tmp_path_factory):
    """One of four candidates is driven to a non-finite state: it gets ``diverged_cost`` from the step it diverges at, the
    other three run to the end and cost exactly what they cost in a clean batch (one diverging run used to end the
    whole batch and leave the healthy candidates unevaluated)."""
    from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC

    g = np.load(GOLDEN / "cylinder_O1.npz")
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp("diverge_batch"), num_steps=8)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)

    class Gain:
        def __init__(self, a, blow_at=None):
            self.a, self.n, self.blow_at = a, 0, blow_at

        def step(self, y, dt):
            self.n += 1
            return np.array([np.inf if self.n == self.blow_at else self.a * y])

    gains = [0.0, 0.3, -0.2, 0.1]
    J_clean, _ = optim.closed_loop_costs(fs, [Gain(a) for a in gains], 8)
    J, series = optim.closed_loop_costs(fs, [Gain(a, blow_at=4 if i == 2 else None) for i, a in enumerate(gains)], 8, diverged_cost=1e9)
    assert J[2] == 1e9 and np.all(np.isfinite(J[[0, 1, 3]]))
    assert np.allclose(J[[0, 1, 3]], J_clean[[0, 1, 3]], rtol=1e-13)
    assert len(series[0]) == 9 and np.isnan(series[2]["y_meas_1"].to_numpy()[-1]) and np.isfinite(series[2]["y_meas_1"].to_numpy()[3])
    optim.write_results(np.array(gains)[:, None], J, tmp_path_factory.mktemp("diverge_out"), verbose=False)
    fs.th.release_device()
