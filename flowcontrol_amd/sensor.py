"""Sensors (mirror of the reference's ``src/flowcontrol/sensor.py``).

Every sensor the reference ships is a *fixed linear functional* of the mixed field
(cf. ``operatorgetter.py:194-239``): a point probe is one row of P2/P1 basis values
(``utils/mpi.py:22-37`` evaluates exactly that, after a bounding-box-tree search each step); the
wall-shear sensor is the assembled row of ``∫ ∂u_x/∂y ds`` over whole marked facets
(``sensor.py:191-213``).  ``row(flowsolver)`` returns that sparse row once; the device evaluates
all rows inside ``fc_step``.  ``eval(up)`` is the host-side evaluation of the same row.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from enum import IntEnum
from typing import TYPE_CHECKING, Any

import numpy as np
from numpy.typing import NDArray

from .fem import element as el
from .fem.boundary import DOLFIN_EPS, SubDomain, near

if TYPE_CHECKING:
    from .flowsolver import FlowSolver

SENSOR_INDEX_DEFAULT = 10000


class SENSOR_TYPE(IntEnum):
    U = 0
    V = 1
    P = 2
    OTHER = 3


@dataclass(kw_only=True)
class Sensor(ABC):
    sensor_type: SENSOR_TYPE
    require_loading: bool

    @abstractmethod
    def eval(self, up) -> float:
        ...

    @abstractmethod
    def row(self, flowsolver: "FlowSolver") -> tuple[np.ndarray, np.ndarray]:
        """(W dof ids, weights) with ``eval(up) == weights @ up[ids]``."""


@dataclass(kw_only=True)
class SensorPoint(Sensor):
    position: NDArray[np.float64]
    require_loading: bool = False

    def eval(self, up) -> float:
        return float(up(np.asarray(self.position, dtype=np.float64))[int(self.sensor_type)])

    def row(self, flowsolver: "FlowSolver"):
        if self.sensor_type not in (SENSOR_TYPE.U, SENSOR_TYPE.V, SENSOR_TYPE.P):
            raise ValueError("SensorPoint needs sensor_type U, V or P")
        return flowsolver.th.point_eval_row(self.position, int(self.sensor_type))


@dataclass(kw_only=True)
class SensorIntegral(Sensor):
    ds: Any = None
    subdomain: SubDomain | None = None
    sensor_index: int = SENSOR_INDEX_DEFAULT
    require_loading: bool = True
    _row: tuple | None = None

    @abstractmethod
    def load(self, flowsolver: "FlowSolver") -> None:
        ...

    def linear_form(self, v):
        """Assembled row of the sensor functional (the reference returns the UFL form)."""
        if self._row is None:
            raise RuntimeError("SensorIntegral.load(flowsolver) must be called first")
        return self._row

    def row(self, flowsolver: "FlowSolver"):
        if self._row is None:
            self.load(flowsolver)
        return self._row

    def eval(self, up) -> float:
        idx, w = self.linear_form(up)
        return float(w @ up.vector().array()[idx])


@dataclass(kw_only=True)
class SensorHorizontalWallShear(SensorIntegral):
    """∫ ∂u_x/∂y ds over the whole boundary facets inside [x_left, x_right] × {y_sensor}."""

    x_sensor_left: float = 1.0
    x_sensor_right: float = 1.1
    y_sensor: float = 0.0

    def load(self, flowsolver: "FlowSolver") -> None:
        xl, xr, ys = self.x_sensor_left, self.x_sensor_right, self.y_sensor
        self.subdomain = SubDomain(
            lambda x, ob: ob & near(x[:, 1], ys, DOLFIN_EPS) & (x[:, 0] >= xl) & (x[:, 0] <= xr), name="wall_shear_sensor"
        )
        th = flowsolver.th
        mesh = th.mesh
        facets = np.nonzero(self.subdomain.mark_facets(mesh))[0]
        self.ds = facets
        acc: dict[int, float] = {}
        g = 0.5 / np.sqrt(3.0)
        for e in facets:
            c = int(mesh.edge_cells[e, 0])
            k = int(np.nonzero(mesh.cell_edges[c] == e)[0][0])  # local edge opposite vertex k
            i, j = (k + 1) % 3, (k + 2) % 3
            length = float(np.linalg.norm(mesh.coords[mesh.cells[c, i]] - mesh.coords[mesh.cells[c, j]]))
            for s in (0.5 - g, 0.5 + g):  # 2-point Gauss: exact for the P1 gradient trace
                lam = np.zeros(3)
                lam[i], lam[j] = 1.0 - s, s
                dphi = el.p2_grad_ref(lam) @ th.Jinv[c]  # (6, 2) physical gradients
                for a in range(6):
                    dof = int(th.cell_nodes[c, a])  # ux dof
                    acc[dof] = acc.get(dof, 0.0) + 0.5 * length * dphi[a, 1]
        idx = np.array(sorted(acc), dtype=np.int64)
        self._row = (idx, np.array([acc[i] for i in idx]))
