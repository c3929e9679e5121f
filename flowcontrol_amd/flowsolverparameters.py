"""Configuration dataclasses of a FlowSolver run.

API mirror of the reference's ``src/flowcontrol/flowsolverparameters.py`` (same class names,
fields, defaults and derived attributes) so case files and run scripts port unchanged.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import Any


@dataclass(kw_only=True)
class ParamFlowSolver:
    """Common base: a free-form ``user_data`` dict (case geometry etc.)."""

    user_data: dict = field(default_factory=dict)


@dataclass
class ParamFlow(ParamFlowSolver):
    Re: float
    uinf: float = 1.0


@dataclass
class ParamMesh(ParamFlowSolver):
    meshpath: Path


@dataclass
class ParamControl(ParamFlowSolver):
    sensor_list: list[Any]
    sensor_number: int = field(init=False)
    actuator_list: list[Any]
    actuator_number: int = field(init=False)

    def __post_init__(self) -> None:
        self.sensor_number = len(self.sensor_list)
        self.actuator_number = len(self.actuator_list)


@dataclass
class ParamTime(ParamFlowSolver):
    num_steps: int
    dt: float
    Tstart: float
    Tfinal: float = field(init=False)

    def __post_init__(self) -> None:
        self.Tfinal = self.num_steps * self.dt


@dataclass
class ParamRestart(ParamFlowSolver):
    save_every_old: int = 0
    restart_order: int = 2
    dt_old: float = 0.0
    Trestartfrom: float = 0.0


@dataclass
class ParamSave(ParamFlowSolver):
    path_out: Path
    save_every: int
    energy_every: int = 1


@dataclass
class ParamSolver(ParamFlowSolver):
    throw_error: bool = True
    shift: float = 0.0
    is_eq_nonlinear: bool = True
    time_scheme: str = "bdf"  # "bdf": BDF1 start-up then BDF2 ; "cn": Crank–Nicolson


@dataclass
class ParamIC(ParamFlowSolver):
    xloc: float = 0.0
    yloc: float = 0.0
    radius: float = 1.0
    amplitude: float = 1.0
