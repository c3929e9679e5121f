"""Host bookkeeping of ``FlowSolver.step`` around the device's deferred results (no GPU: a stand-in device object).

The device publishes y at once and (dE, residual) of a step later (``fc_step_collect``); the host books a step's log row while
the next step runs.  What must hold whatever the timing: a residual-monitor breach on the LAST step of a run is reported (by
``timeseries`` / ``write_timeseries``), a checkpoint is written only after its own step's verdict, the pending row survives
``release_device``, and a failure of the host's bookkeeping between ``step_begin`` and ``step_end`` does not leave a step in
flight."""
import numpy as np
import pytest

from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
from flowcontrol_amd.flowsolver import _DeviceNDSolver


class FakeDevice:
    world = 1

    def __init__(self, residuals):
        self.residuals = list(residuals)
        self.k = 0
        self.in_flight = False
        self.collected = 0
        self.undone = 0
        self.closed = False

    def step_begin(self, slot, u_ctrl, compute_energy=True, u_force=None):
        assert not self.in_flight, "fc_step_begin refused: the previous step was never ended"
        self.in_flight = True

    def step_end(self, early=False):
        assert self.in_flight
        self.in_flight = False
        self.k += 1
        return np.array([0.1 * self.k, 0.0, 0.0]), None, None

    def step_collect(self):
        assert not self.closed, "collect on a released handle"
        self.collected += 1
        return 0.5 * self.k, np.array([0.0, self.residuals[self.k - 1], 1.0, 0.0])

    def undo_step(self):
        self.undone += 1
        self.k -= 1

    def get_state(self):
        n = 4
        return np.zeros(n), np.zeros(n), np.zeros(2)

    def close(self):
        self.closed = True


def _solver(tmp_path, residuals, save_every=0, throw=True):
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path, num_steps=10, save_every=save_every)
    fs.params_solver.throw_error = throw
    dev = FakeDevice(residuals)
    fs.th._device = dev
    fs.first_step, fs.order, fs.iter, fs.t = False, 2, 0, 0.0
    fs._u_ctrl_prev = None
    fs.solvers = {2: _DeviceNDSolver(fs, 1)}
    fs.fields._dirty = False
    fs.exporter.log_ic(t=0.0, y_meas=np.zeros(3), dE=0.0)
    return fs, dev


def test_a_breach_on_the_last_step_is_reported_by_the_readers(tmp_path):
    fs, dev = _solver(tmp_path, [1e-14, 1e-14, 1e-3])
    for _ in range(3):
        assert fs.step([0.0, 0.0]) is not None  # the verdict on step 3 does not exist yet when step 3 returns
    with pytest.raises(RuntimeError, match="exceeds residual_tol"):
        fs.timeseries
    ts = fs.timeseries  # reported once; the log is complete
    assert len(ts) == 4 and ts["dE"].iloc[-1] == pytest.approx(1.5)


def test_write_timeseries_reports_too_and_none_without_throw_error(tmp_path):
    fs, dev = _solver(tmp_path, [1e-14, 1e-3], throw=False)
    fs.step([0.0, 0.0]), fs.step([0.0, 0.0])
    fs.write_timeseries()  # logged (critical), not raised
    assert fs._residual_breach is None and fs.residual_max == pytest.approx(1e-3)


def test_no_checkpoint_of_a_step_the_monitor_rejected(tmp_path, monkeypatch):
    fs, dev = _solver(tmp_path, [1e-14, 1e-3], save_every=2)
    written = []
    monkeypatch.setattr(fs, "_checkpoint", lambda: written.append(fs.iter))
    fs.step([0.0, 0.0])
    with pytest.raises(RuntimeError, match="iteration 2"):
        fs.step([0.0, 0.0])  # a checkpoint step waits for its own verdict
    assert written == []


def test_release_device_books_the_pending_row_first(tmp_path):
    fs, dev = _solver(tmp_path, [1e-14, 1e-14])
    fs.step([0.0, 0.0]), fs.step([0.0, 0.0])
    fs.th.release_device()
    assert dev.closed and dev.collected == 2
    ts = fs.timeseries  # no device any more: everything was fetched before it went
    assert len(ts) == 3 and ts["dE"].iloc[-1] == pytest.approx(1.0)


def test_a_failing_log_flush_does_not_leave_a_step_in_flight(tmp_path, monkeypatch):
    fs, dev = _solver(tmp_path, [1e-14] * 4)
    fs.step([0.0, 0.0])
    real = fs._exporter.log

    def broken(**kw):
        raise OSError("disk full")

    monkeypatch.setattr(fs._exporter, "log", broken)
    with pytest.raises(OSError):
        fs.step([0.0, 0.0])  # the previous step's row is booked between begin and end
    assert not dev.in_flight and dev.undone == 1 and fs.iter == 1
    monkeypatch.setattr(fs._exporter, "log", real)
    assert fs.step([0.0, 0.0]) is not None and fs.iter == 2
