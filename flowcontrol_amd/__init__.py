"""flowcontrol_amd — the per-timestep hot path of FlowControl on MI355X (gfx950).

Public names mirror the reference's ``src/flowcontrol/__init__.py``.
"""

__version__ = "0.1.0"

from .actuator import (
    ACTUATOR_TYPE,
    CYLINDER_ACTUATION_MODE,
    Actuator,
    ActuatorBC,
    ActuatorBCParabolicV,
    ActuatorBCRotation,
    ActuatorBCUniformU,
    ActuatorForceGaussianV,
)
from .controller import Controller
from .exporter import FlowExporter
from .flowfield import BoundaryConditions, FlowField, FlowFieldCollection, SimPaths
from .flowsolver import FlowSolver
from .flowsolverparameters import (
    ParamControl,
    ParamFlow,
    ParamIC,
    ParamMesh,
    ParamRestart,
    ParamSave,
    ParamSolver,
    ParamTime,
)
from .nsforms import NSForms
from .sensor import SENSOR_TYPE, Sensor, SensorHorizontalWallShear, SensorIntegral, SensorPoint
from .steadystate import SteadyStateSolver

__all__ = [
    "__version__",
    "FlowSolver",
    "Controller",
    "NSForms",
    "FlowExporter",
    "SteadyStateSolver",
    "FlowField",
    "FlowFieldCollection",
    "BoundaryConditions",
    "SimPaths",
    "ParamFlow",
    "ParamTime",
    "ParamSave",
    "ParamSolver",
    "ParamMesh",
    "ParamControl",
    "ParamIC",
    "ParamRestart",
    "Actuator",
    "ActuatorBC",
    "ActuatorBCParabolicV",
    "ActuatorBCRotation",
    "ActuatorBCUniformU",
    "ActuatorForceGaussianV",
    "ACTUATOR_TYPE",
    "CYLINDER_ACTUATION_MODE",
    "Sensor",
    "SensorPoint",
    "SensorIntegral",
    "SensorHorizontalWallShear",
    "SENSOR_TYPE",
]
