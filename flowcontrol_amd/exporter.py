"""Timeseries logging and checkpoints (mirror of the reference's ``src/flowcontrol/exporter.py``).

Timeseries rows ``{time, dE, runtime, u_ctrl_i, y_meas_i}`` (1-based suffixes, the IC row has no
``u_ctrl``) → pandas DataFrame → CSV, exactly the reference's schema (``exporter.py:169-267``):
parity is judged on this series.  Field checkpoints keep the reference's naming
(``U_restart<T>.xdmf`` + ``.h5`` …, series names ``U`` / ``U_n`` / ``P``, JSON sidecar
``meta_restart<T>.json`` with the same keys, ``exporter.py:85-165,234-262``); the files are written by
``flowcontrol_amd/io.py`` (ParaView-readable XDMF, own HDF5 payload layout).
"""

from __future__ import annotations

import json
import logging
from pathlib import Path

import pandas as pd

from .fem.spaces import Function
from .flowfield import FlowFieldCollection, SimPaths

logger = logging.getLogger(__name__)


def write_frame(path: Path, func: Function, counter: int, name: str = "f", time: float = 0.0) -> None:
    """Frame ``counter`` of the checkpoint series at ``path`` (XDMF + HDF5, ``io.write_xdmf``)."""
    from .io import write_xdmf

    got = write_xdmf(path, func, name, time_step=time, append=counter > 0)
    if got != counter:
        raise RuntimeError(f"{path}: expected to write frame {counter}, series has {got}")


def read_frame(path: Path, func: Function, counter: int, name: str = "f") -> None:
    from .io import read_xdmf

    if not Path(path).with_suffix(".h5").exists():
        raise FileNotFoundError(f"checkpoint {Path(path).with_suffix('.h5')} not found")
    read_xdmf(path, func, name, counter)


class FlowExporter:
    def __init__(self, paths: SimPaths, fields: FlowFieldCollection, V, P, Tstart: float = 0.0, dt: float = 0.0, save_every: int = 0) -> None:
        self.paths = paths
        self.fields = fields
        self.V = V
        self.P = P
        self._Tstart = Tstart
        self._dt = dt
        self._save_every = save_every
        self._records: list[dict] = []
        self._checkpoints_written = 0
        self._frames = 0
        self._u_cols: list[str] | None = None
        self._y_cols: list[str] | None = None

    # ── field export ─────────────────────────────────────────────────────────
    def export_xdmf(self, u_n, u_nn, p_n, time: float, append: bool = True, write_mesh: bool = False, adjust_baseflow: float = 0.0) -> None:
        """Store (U, U_prev, P) = perturbation + adjust_baseflow · base flow (``exporter.py:85-165``)."""
        f = self.fields
        if f.Usave is None:
            f.Usave = Function(self.V)
        if f.Usave_n is None:
            f.Usave_n = Function(self.V)
        if f.Psave is None:
            f.Psave = Function(self.P)
        U0v = f.U0.vector().array()
        P0v = f.P0.vector().array()
        f.Usave.vector()[:] = u_n.vector().array() + adjust_baseflow * U0v
        f.Usave_n.vector()[:] = u_nn.vector().array() + adjust_baseflow * U0v
        f.Psave.vector()[:] = p_n.vector().array() + adjust_baseflow * P0v
        if not append:
            self._frames = 0
        self._checkpoints_written += 1
        for path, func, nm in ((self.paths.U_restart, f.Usave, "U"), (self.paths.Uprev_restart, f.Usave_n, "U_n"), (self.paths.P_restart, f.Psave, "P")):
            write_frame(path, func, self._frames, name=nm, time=time)
        self._frames += 1

    # ── timeseries ───────────────────────────────────────────────────────────
    def log_ic(self, t: float, y_meas, dE: float) -> None:
        row: dict = {"time": t, "dE": dE, "runtime": 0.0}
        for i, v in enumerate(y_meas):
            row[f"y_meas_{i + 1}"] = float(v)
        self._records.append(row)

    def log(self, u_ctrl, y_meas, dE: float, t: float, runtime: float) -> None:
        if self._u_cols is None:
            self._u_cols = [f"u_ctrl_{i + 1}" for i in range(len(u_ctrl))]
            self._y_cols = [f"y_meas_{i + 1}" for i in range(len(y_meas))]
        row: dict = {"time": t, "dE": dE, "runtime": runtime}
        row.update(zip(self._u_cols, (float(v) for v in u_ctrl)))
        row.update(zip(self._y_cols, (float(v) for v in y_meas)))
        self._records.append(row)

    def to_dataframe(self) -> pd.DataFrame:
        return pd.DataFrame(self._records)

    def write_metadata(self, restart_order: int | str = 2) -> None:
        meta = {
            "Tstart": self._Tstart,
            "dt": self._dt,
            "save_every": self._save_every,
            "checkpoints_written": self._checkpoints_written,
            "restart_order": restart_order,
            "files": {
                "U": self.paths.U_restart.name,
                "Uprev": self.paths.Uprev_restart.name,
                "P": self.paths.P_restart.name,
            },
        }
        self.paths.metadata.parent.mkdir(parents=True, exist_ok=True)
        self.paths.metadata.write_text(json.dumps(meta, indent=2))

    def write_timeseries(self) -> None:
        self.paths.timeseries.parent.mkdir(parents=True, exist_ok=True)
        self.to_dataframe().to_csv(self.paths.timeseries, sep=",", index=False)

    def log_progress(self, iter: int, num_steps: int, t: float, t_end: float, runtime: float) -> None:
        logger.info("--- iter: %5d/%5d --- time: %3.3f/%3.3f --- elapsed %5.5f ---", iter, num_steps, t, t_end, runtime)

    def reset(self) -> None:
        """Clear the log.  As in the reference (``exporter.py:287-290``) the checkpoint *counter*
        restarts at 0 while frame 0 written by the initial export stays on disk."""
        self._records.clear()
        self._checkpoints_written = 0
