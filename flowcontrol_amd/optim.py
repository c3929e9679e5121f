"""Controller-optimisation helpers around the time-stepping path (the reference's ``src/utils/optim.py``), plus what that file
cannot have: the candidates of one optimiser iteration evaluated TOGETHER.

The reference evaluates a cost function point by point (``fun_array``: one closed-loop ``FlowSolver`` simulation per candidate
controller, ``optim.py:48-68``).  All candidates of an iteration share the flow solver's operators — only the controller differs
— so here up to 32 of them run as one ``BatchedFlowSolver`` (``closed_loop_costs``): the factors are read once per time step for
all candidates.  The scalar helpers keep the reference's names, arguments and file formats (``J_costfun.csv``,
``J_costfun_cummin.csv``, ``timeseries/timeseries_iter_XXXX[_DIVERGED].csv``).
"""

from __future__ import annotations

import logging
from collections.abc import Callable, Sequence
from pathlib import Path

import numpy as np
import pandas as pd

logger = logging.getLogger(__name__)


# ── scalar helpers (reference optim.py) ──────────────────────────────────────────────────────────────────────────────
def fun_array(x: np.ndarray, fun: Callable[..., float], **kwargs) -> np.ndarray:
    """``fun`` on every row of ``x`` (n_points, dim) → costs (n_points, 1), one evaluation after the other (optim.py:48-68)."""
    x = np.atleast_2d(np.asarray(x, dtype=float))
    return np.array([[float(fun(row, **kwargs))] for row in x])


def cummin(y: np.ndarray, return_index: bool = True):
    """Running minimum of a column vector (n, 1) and, optionally, the index where each running minimum was first reached
    (optim.py:112-137)."""
    y = np.asarray(y, dtype=float).reshape(-1, 1)
    best = np.minimum.accumulate(y, axis=0)
    if not return_index:
        return best
    # first position (up to step i) whose value equals the running minimum of step i; a NaN cost (a run that was stopped early)
    # propagates through the running minimum and matches nothing: index 0 then, as the reference's argmax over an all-False row gives
    hit = np.isclose(best, y.T) & np.tri(y.shape[0], dtype=bool)
    return best, hit.argmax(1)


def write_results(x_data, y_data, optim_path, verbose: bool = True) -> None:
    """All evaluations (``J_costfun.csv``) and the best-so-far sequence (``J_costfun_cummin.csv``), columns J, x0, x1, …
    (optim.py:140-174)."""
    optim_path = Path(optim_path)
    X = np.atleast_2d(np.asarray(x_data, dtype=float))
    J = np.asarray(y_data, dtype=float).reshape(-1, 1)
    cols = ["J"] + [f"x{i}" for i in range(X.shape[1])]
    if verbose:
        logger.info("Logging results to: %s", optim_path)
    pd.DataFrame(np.hstack([J, X]), columns=cols).to_csv(optim_path / "J_costfun.csv", index=False)
    best, idx = cummin(J)
    pd.DataFrame(np.hstack([best, X[idx]]), columns=cols).to_csv(optim_path / "J_costfun_cummin.csv", index=False)


def sobol_sample(ndim: int, npt: int, xlimits=None, skip: int = 1000, seed: int | None = None) -> np.ndarray:
    """``npt`` points of the (unscrambled) Sobol sequence in ``ndim`` dimensions after skipping ``skip`` points (+ a random offset
    when ``seed`` is given), scaled into ``xlimits`` (ndim, 2) or (2, ndim) (optim.py:177-224)."""
    from scipy.stats.qmc import Sobol

    engine = Sobol(d=ndim, scramble=False)
    n_skip = int(skip) + (int(np.random.default_rng(seed).integers(10000)) if seed is not None else 0)
    if n_skip > 0:
        engine.fast_forward(n_skip)
    pts = engine.random(npt)
    if xlimits is not None:
        lim = np.asarray(xlimits, dtype=float)
        if lim.shape == (2, ndim):  # rows = lower / upper bounds (also how the reference reads a square array)
            lim = lim.T
        if lim.shape != (ndim, 2):
            raise ValueError(f"xlimits has wrong shape {lim.shape}, expected ({ndim}, 2)")
        pts = lim[:, 0] + pts * (lim[:, 1] - lim[:, 0])
    return pts


def compute_signal_cost(signal, Tnorm: float, criterion: str, scaling: Callable | None = None) -> float:
    """Time-averaged (``"integral"``: Σ scaling(signal) · Tnorm) or final (``"terminal"``) cost of a time series; Tnorm =
    dt / (t − Tc) (optim.py:230-272)."""
    if criterion not in ("integral", "terminal"):
        raise ValueError(f"Unknown criterion {criterion!r}: expected 'integral' or 'terminal'.")
    f = scaling if scaling is not None else (lambda v: v)
    s = pd.Series(signal) if not isinstance(signal, pd.Series) else signal
    return float(np.sum(f(s)) * Tnorm) if criterion == "integral" else float(f(s.iloc[-1]))


def compute_control_cost(u_ctrl, Tnorm: float) -> float:
    """Time-normalised control effort Σ_t Σ_channels u² · Tnorm (optim.py:275-291)."""
    return float(np.nansum(np.asarray(u_ctrl, dtype=float) ** 2) * Tnorm)


def write_optim_csv(timeseries: pd.DataFrame, savedir, diverged: bool, iteration: int) -> None:
    """``<savedir>/timeseries/timeseries_iter_XXXX[_DIVERGED].csv`` (optim.py:294-318)."""
    d = Path(savedir) / "timeseries"
    d.mkdir(parents=True, exist_ok=True)
    timeseries.to_csv(d / f"timeseries_iter_{int(iteration):04d}{'_DIVERGED' if diverged else ''}.csv", index=False)


# ── the candidates of one iteration, together ────────────────────────────────────────────────────────────────────────
def closed_loop_costs(fs, controllers: Sequence, num_steps: int, u_penalty: float = 0.0, signal: str = "dE", criterion: str = "integral",
                      feedback: Callable | None = None, ics=None, Tc: float = 0.0, diverged_cost: float = np.inf):
    """Cost J = xQx + u_penalty · uRu of every controller in ``controllers`` (≤ 32) for ``num_steps`` closed-loop steps of ``fs``'s case,
    all candidates advanced in lock step on one handle (``BatchedFlowSolver``).

    ``controllers[i].step(y=…, dt=…)`` is called once per time step with ``feedback(y_meas_i)`` (default: minus the first
    measurement, the reference's cylinder loop) and must return the actuator command(s) (a scalar is applied to every actuator).
    ``signal``: a column of the time series (``"dE"``: full-state energy, ``"y_meas_1"`` …).  Returns ``(J, timeseries)``:
    costs (k,) — ``diverged_cost`` for a run that became non-finite — and the k time-series DataFrames."""
    from .batch import BatchedFlowSolver

    k = len(controllers)
    n_act = fs.params_control.actuator_number
    fb = feedback if feedback is not None else (lambda y: -y[0])
    dt = fs.params_time.dt
    bfs = BatchedFlowSolver(fs, k)
    bfs.initialize_time_stepping(ics=ics)
    throw = fs.params_solver.throw_error
    fs.params_solver.throw_error = False  # a diverging candidate is a data point, not an error
    try:
        alive = True
        for _ in range(num_steps):
            u = np.zeros((k, n_act))
            for i, K in enumerate(controllers):
                if bfs.diverged[i]:
                    continue  # that run has ended (its measurements are NaN): no command
                cmd = np.atleast_1d(np.asarray(K.step(y=fb(bfs.y_meas[i]), dt=dt), dtype=float)).ravel()
                u[i] = cmd if cmd.size == n_act else cmd[0]
            if bfs.step(u) is None:  # a residual breach: the factors all candidates share are broken
                alive = False
                break
        series = [bfs.timeseries(i) for i in range(k)]
        if bfs.breached:  # the monitor's verdict on the LAST step arrives with the log: costs of a broken last step are not ranked
            alive = False
        J = np.empty(k)
        for i, ts in enumerate(series):
            if bool(bfs.diverged[i]):  # a diverging candidate ends there; the others run to the end
                J[i] = diverged_cost
                continue
            t_end = float(ts["time"].iloc[-1])
            Tnorm = dt / max(t_end - Tc, dt)
            ucols = [c for c in ts.columns if c.startswith("u_ctrl_")]
            J[i] = compute_signal_cost(ts[signal].dropna(), Tnorm, criterion) + u_penalty * compute_control_cost(ts[ucols], Tnorm)
        if not alive:  # stopped early: mark the finite runs as unevaluated rather than rank them
            J[~np.asarray(bfs.diverged, dtype=bool)] = np.nan
        return J, series
    finally:
        fs.params_solver.throw_error = throw
        bfs.close()


def fun_array_batched(x: np.ndarray, make_controller: Callable, fs, num_steps: int, batch: int = 32, **kwargs) -> np.ndarray:
    """``fun_array`` for closed-loop costs: rows of ``x`` are controller parameters, ``make_controller(row)`` builds the controller;
    the points are evaluated ``batch`` at a time on one handle.  Returns costs (n_points, 1)."""
    x = np.atleast_2d(np.asarray(x, dtype=float))
    out = np.empty((x.shape[0], 1))
    for a in range(0, x.shape[0], batch):
        rows = x[a : a + batch]
        J, _ = closed_loop_costs(fs, [make_controller(r) for r in rows], num_steps, **kwargs)
        out[a : a + rows.shape[0], 0] = J
    return out


__all__ = ["fun_array", "cummin", "write_results", "sobol_sample", "compute_signal_cost", "compute_control_cost", "write_optim_csv",
           "closed_loop_costs", "fun_array_batched"]
