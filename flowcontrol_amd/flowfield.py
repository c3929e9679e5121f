"""State containers of the hot path (mirror of the reference's ``src/flowcontrol/flowfield.py``).

``FlowFieldCollection`` differs from the reference in one MI355X-driven way: while time stepping,
the authoritative copies of ``u_, p_, up_, u_n, u_nn, p_n`` live in HBM.  The attributes are
therefore *lazy*: reading one triggers a single device→host download of the current state (and
only if a step happened since the last read), so a closed loop that only consumes ``y_meas`` never
moves a field across PCIe.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import Callable

from .fem.spaces import Function


@dataclass(frozen=True)
class SimPaths:
    U0: Path
    P0: Path
    U: Path
    P: Path
    Uprev: Path
    U_restart: Path
    Uprev_restart: Path
    P_restart: Path
    timeseries: Path
    metadata: Path
    steady_meta: Path
    mesh: Path


@dataclass
class FlowField:
    """(u, p, up) triple; ``u`` and ``p`` are split copies of the mixed function ``up``."""

    u: Function = field(init=False)
    p: Function = field(init=False)
    up: Function

    def __post_init__(self) -> None:
        self.u, self.p = self.up.split(deepcopy=True)


_LAZY = ("u_", "p_", "up_", "u_n", "u_nn", "p_n")


class FlowFieldCollection:
    """All fields of a simulation: base flow, initial condition, current/previous perturbations,
    save buffers.  Same attribute names as the reference."""

    def __init__(self) -> None:
        self.U0: Function | None = None
        self.P0: Function | None = None
        self.UP0: Function | None = None
        self.ic: FlowField | None = None
        self.Usave: Function | None = None
        self.Psave: Function | None = None
        self.Usave_n: Function | None = None
        self._store: dict[str, Function | None] = {k: None for k in _LAZY}
        self._sync: Callable[[], None] | None = None
        self._stale = False
        self._dirty = False  # a field was replaced on the host after the state went to the device

    def _set_sync(self, fn: Callable[[], None] | None) -> None:
        self._sync = fn

    def _mark_stale(self) -> None:
        self._stale = True

    def push(self) -> None:
        """Declare the host copies of ``u_n, u_nn, p_n`` authoritative: the next ``step()`` / ``run()`` uploads them
        before it advances.  Assigning a field (``fields.u_n = f``) does this by itself; IN-PLACE edits
        (``fields.u_n.vector()[:] = ...``) cannot be seen and need this call.  (In the reference these Functions
        are the state the next right-hand side reads; here that state lives in HBM while stepping.)
        On several ranks reading a field is collective (every rank contributes its dofs): read on all ranks."""
        self._get("u_n")  # make sure the host copies are current before they become the source
        self._dirty = True

    def _get(self, name: str):
        if self._stale and self._sync is not None:
            self._stale = False
            self._sync()
        return self._store[name]


def _lazy_property(name: str):
    def getter(self):
        return self._get(name)

    def setter(self, value):
        if self._sync is not None and name in ("u_n", "u_nn", "p_n"):
            self._get(name)  # bring the other fields up to date first: the upload takes all three from the host
            self._dirty = True
        self._store[name] = value

    return property(getter, setter)


for _n in _LAZY:
    setattr(FlowFieldCollection, _n, _lazy_property(_n))


@dataclass
class BoundaryConditions:
    bcu: list
    bcp: list
