"""``BatchedFlowSolver`` — k lock-step simulations of ONE ``FlowSolver``'s operators on one MI355X.

The reference's outer workloads are sweeps of independent runs of the same case: initial-condition sweeps
(``src/examples/lidcavity/batch_run_lidcavity.py:197-215``), controller optimisation
(``src/utils/optim.py:95-102``: one closed-loop simulation per candidate controller).  There every run owns a
``FlowSolver`` and calls ``step`` (``src/flowcontrol/flowsolver.py:703-799``) once per time step — and re-reads
the whole LU factorisation for one right-hand side.  Here the k runs share the device handle of one prepared
``FlowSolver`` (same mesh, base flow, Δt, actuators, sensors ⇒ same operators and factors) and are advanced
together by ``fc_step_batch``: one pass over the factors serves all of them (``csrc/fc_batch.hip.h``).

What may differ between the runs: the initial condition, the control input of every step (hence the
controller), and what the host does with the measurements.  Everything a single run logs is logged per run::

    fs = CylinderFlowSolver.make_default(...); fs.load_steady_state()
    bfs = BatchedFlowSolver(fs, k=8)
    bfs.initialize_time_stepping(ics=[ParamIC(xloc=x, ...) for x in xs])
    for _ in range(n):
        u = np.stack([K[i].step(y=-bfs.y_meas[i][0], dt=dt) for i in range(8)])
        bfs.step(u_ctrl=u)
    bfs.timeseries(3)          # DataFrame of run 3, same columns as FlowSolver.timeseries
"""

from __future__ import annotations

import logging
import time
from typing import Sequence

import numpy as np
import pandas as pd

from ._lib import SLOT_BDF1, SLOT_BDF2, FcDiverged
from .exporter import FlowExporter
from .fem.spaces import Function
from .flowsolverparameters import ParamIC

logger = logging.getLogger(__name__)


class BatchedFlowSolver:
    def __init__(self, fs, k: int) -> None:
        if not 1 <= int(k) <= 32:
            raise ValueError(f"k must be in [1, 32], got {k}")
        if fs.fields.U0 is None:
            raise RuntimeError("no base flow: call compute_steady_state() or load_steady_state() on the FlowSolver first")
        if fs.refine_steps or fs.nd_truncate:
            raise ValueError("batched stepping applies the full factors directly: refine_steps and nd_truncate must be 0")
        self.fs = fs
        self.k = int(k)
        self.params_time = fs.params_time
        self.params_save = fs.params_save
        self.residual_tol = fs.residual_tol
        self._ready = False

    # ── setup ────────────────────────────────────────────────────────────────
    def _prepare(self) -> None:
        fs = self.fs
        if fs.fields._store.get("u_n") is None:
            fs.initialize_time_stepping(ic=None)  # the operators are assembled from the solver's own (any) state
        if not fs._systems_ready:
            fs._prepare_systems(fs.fields._store["u_n"], fs.fields._store["u_nn"])
        self.dev = fs.th.device()
        if self.dev.world > 1:
            raise RuntimeError("batched stepping runs on single-GPU handles (replicas scale across GPUs by themselves)")
        self.dev.set_batch(self.k)
        # the residual monitor's cadence of the single run applies to the batch (fs.check_residual_every: every n-th batched step)
        self.dev.set_solver_options(refine=0, check_residual=-1 if fs.check_residual_every is None else fs.check_residual_every)
        self._ready = True

    def initialize_time_stepping(self, ics: Sequence[ParamIC | Function | None] | None = None, Tstart: float = 0.0) -> None:
        """Initial condition of every run: a ``ParamIC`` (the solver's default div-free Gaussian vortex with that
        centre / radius / amplitude, as ``FlowSolver._initialize_with_ic``), a mixed ``Function`` (taken as it is) or
        ``None`` (the solver's own ``params_ic``)."""
        if Tstart != 0.0:
            raise NotImplementedError("batched runs start from initial conditions (restart one run with FlowSolver)")
        if not self._ready:
            self._prepare()
        fs, k = self.fs, self.k
        ics = list(ics) if ics is not None else [None] * k
        if len(ics) != k:
            raise ValueError(f"expected {k} initial conditions, got {len(ics)}")
        nn2, nv = 2 * fs.th.nn, fs.th.nv
        u_n = np.zeros((k, nn2))
        p_n = np.zeros((k, nv))
        self.y_meas = np.zeros((k, len(fs.params_control.sensor_list)))
        dE0 = np.zeros(k)
        for i, ic in enumerate(ics):
            if isinstance(ic, Function):
                up = ic.vector().array().copy()
            else:
                pic = fs.params_ic if ic is None else ic
                up = np.zeros(fs.th.N)
                if pic.amplitude:
                    up += pic.amplitude * fs._default_initial_perturbation(xloc=pic.xloc, yloc=pic.yloc, radius=pic.radius).vector().array()
            u_n[i], p_n[i] = up[:nn2], up[nn2:]
            self.y_meas[i] = fs.make_measurement(up=Function(fs.W, up))
            dE0[i] = 0.5 * fs._velocity_l2_norm(u_n[i]) ** 2
        # the log of all runs as arrays (one append per step, not one dict per run and step); timeseries(i) shapes run i's
        # rows exactly as FlowExporter does (exporter.py:169-232: IC row without u_ctrl, then one row per step)
        self._log_ic = (self.params_time.Tstart, self.y_meas.copy(), dE0)
        self._log: list[tuple] = []
        self.dev.set_state_batch(u_n, u_n, p_n)
        self.order: int | str = "cn" if fs.params_solver.time_scheme == "cn" else 1
        self._u_ctrl_prev = None
        self.iter = 0
        self.t = self.params_time.Tstart
        self.diverged = np.zeros(k, dtype=bool)
        self._pending, self._breach, self._solve_info = None, None, None

    # ── stepping ─────────────────────────────────────────────────────────────
    def step(self, u_ctrl) -> np.ndarray | None:
        """Advance all k runs by one Δt; ``u_ctrl``: (k, n_act).  Returns y_meas (k, n_sens).

        A run whose velocity becomes non-finite ENDS there, as a single ``FlowSolver`` run does (flowsolver.py:727-737): it is
        marked in ``self.diverged``, taken out of the batch's dynamics (``fc_reset_sim_batch``) and from then on its rows of
        ``y_meas`` / its log entries are NaN.  The other runs are not affected — the columns are independent — and keep stepping:
        with ``params_solver.throw_error = False`` this call returns ``y_meas`` as usual (NaN rows for the diverged runs); with
        ``throw_error = True`` it raises ``RuntimeError`` once, after the step has been booked, and may be called again for the
        remaining runs.  Only a linear-solve residual above ``residual_tol`` — broken factors, which all runs share — stops
        the batch (``None`` / ``RuntimeError``)."""
        fs, k = self.fs, self.k
        t0 = time.time()
        n_act = fs.params_control.actuator_number
        u_ctrl = np.asarray(u_ctrl, dtype=np.float64).reshape(k, n_act)
        next_iter = self.iter + 1
        want_energy = fs._niter_multiple_of(next_iter, self.params_save.energy_every)
        u_force = None
        if self.order == "cn":
            prev = np.zeros_like(u_ctrl) if self._u_ctrl_prev is None else self._u_ctrl_prev
            u_force = 0.5 * (u_ctrl + prev)
        dev = self.dev
        dev.step_batch_begin(SLOT_BDF2 if self.order == 2 else SLOT_BDF1, u_ctrl, compute_energy=want_energy, u_force=u_force)
        try:
            self._flush()  # the previous step's energies / residuals arrive behind the host's back: book its log row while this step runs
        except BaseException:
            # the host's bookkeeping failed with a batched step in flight: end it, so that the handle does not refuse every later
            # fc_step_batch_begin (its result is dropped: the batch's state has advanced, the caller sees the error)
            try:
                dev.step_batch_end_early()
            except Exception:  # noqa: BLE001 -- the original error is the one to report
                pass
            raise
        newly = np.zeros(k, dtype=bool)
        try:
            y, flags = dev.step_batch_end_early()
        except FcDiverged:
            # the records of all runs were filled before the status came back: the healthy runs' step stands
            y, flags = dev._batch_bufs[2][:, : dev.n_sens].copy(), dev._batch_flags
            newly = (flags != 0) & ~self.diverged
            for s in np.flatnonzero(flags != 0):
                dev.reset_sim_batch(int(s))  # zero state: the column stops producing non-finite values
            logger.critical("Solver diverged (Inf detected) in runs %s", np.flatnonzero(newly).tolist())
            self.diverged |= flags != 0
        if np.any(self.diverged):
            y = np.where(self.diverged[:, None], np.nan, y)
        self.iter = next_iter
        self.t = self.params_time.Tstart + self.iter * self.params_time.dt
        self._u_ctrl_prev = u_ctrl.copy()
        if fs.params_solver.time_scheme != "cn":
            self.order = 2
        self.y_meas = y
        self._pending = (self.t, self._u_ctrl_prev, y, want_energy, (time.time() - t0) / k, self.iter)
        if np.any(newly) and fs.params_solver.throw_error:
            raise RuntimeError(f"Failed solving: Inf found in solution (runs {np.flatnonzero(newly).tolist()}; the other runs go on)")
        # the residual monitor's verdict on a step arrives with the next step at the latest (broken factors: all runs share them)
        if self._report_breach():
            return None
        return self.y_meas

    def _report_breach(self) -> bool:
        """A pending verdict of the residual monitor: logged, raised when ``throw_error`` is set; ``self.breached`` remembers it (a breach
        on the LAST step of a run surfaces through ``timeseries`` / ``close``, which call this)."""
        breach = self._breach
        if breach is None:
            return False
        self._breach = None
        self.breached = True
        msg = f"linear solve residual {breach[0]:.2e} exceeds residual_tol = {self.residual_tol:.1e} at iteration {breach[1]}"
        logger.critical(msg)
        if self.fs.params_solver.throw_error:
            raise RuntimeError(msg)
        return True

    breached = False  # the residual monitor has rejected a step of this batch (broken factors: no run's results are to be trusted)

    _pending = None
    _breach = None
    _solve_info = None

    def _flush(self) -> None:
        """Energy and solve info of the last step (computed on a second stream while the host went on) into the log."""
        row = self._pending
        if row is None:
            return
        self._pending = None
        t, u, y, want_energy, runtime, it = row
        dE, info = self.dev.step_batch_collect()
        self._solve_info = info
        if np.any(self.diverged):
            dE = np.where(self.diverged, np.nan, dE)
        res = info[~self.diverged, 1]
        if res.size and not np.all(np.isnan(res)) and np.nanmax(res) > self.residual_tol:  # (NaN: a step off the monitor's cadence)
            self._breach = (float(np.nanmax(res)), it)
        self._log.append((t, u, y, dE if want_energy else np.full(self.k, np.nan), runtime))

    @property
    def solve_info(self):
        """(0, relative residual, |b|, flag) per run of the last step (collected on access)."""
        self._flush()
        return self._solve_info

    # ── results ──────────────────────────────────────────────────────────────
    def timeseries(self, i: int) -> pd.DataFrame:
        """Log of run ``i``: the columns and rows ``FlowSolver.timeseries`` gives for a single run."""
        self._flush()
        self._report_breach()
        t0, y0, dE0 = self._log_ic
        ex = FlowExporter(paths=self.fs.paths, fields=self.fs.fields, V=self.fs.V, P=self.fs.P, Tstart=t0, dt=self.params_time.dt, save_every=0)
        ex.log_ic(t=t0, y_meas=y0[i], dE=dE0[i])
        for t, u, y, dE, runtime in self._log:
            ex.log(u_ctrl=u[i], y_meas=y[i], dE=dE[i], t=t, runtime=runtime)
        return ex.to_dataframe()

    def state(self):
        """(u_n, u_nn, p_n) of all runs, arrays (k, ·) — one download."""
        return self.dev.get_state_batch()

    def field(self, i: int) -> Function:
        """Current perturbation (u, p) of run ``i`` as a mixed ``Function`` (downloads the batched state)."""
        u_n, _, p_n = self.dev.get_state_batch()
        return Function(self.fs.W, np.r_[u_n[i], p_n[i]])

    def close(self) -> None:
        if self._ready:
            try:
                self._flush()
                self._report_breach()
            finally:
                self.dev.set_batch(0)
                self._ready = False


__all__ = ["BatchedFlowSolver"]
