"""flowcontrol_amd — the per-timestep hot path of FlowControl on MI355X (gfx950).

The names re-exported here are the public surface of the reference's ``flowcontrol`` package
(``src/flowcontrol/__init__.py:16-85``) so that ``from flowcontrol_amd import FlowSolver, Controller``
replaces ``from flowcontrol import ...`` one to one.
"""

import importlib as _importlib

__version__ = "0.1.0"

#: module → public names it contributes
_EXPORTS = {
    "flowsolver": ("FlowSolver",),
    "controller": ("Controller",),
    "nsforms": ("NSForms",),
    "exporter": ("FlowExporter",),
    "steadystate": ("SteadyStateSolver",),
    "operatorgetter": ("OperatorGetter",),
    "flowfield": ("FlowField", "FlowFieldCollection", "BoundaryConditions", "SimPaths"),
    "flowsolverparameters": ("ParamFlow", "ParamTime", "ParamSave", "ParamSolver", "ParamMesh", "ParamControl", "ParamIC", "ParamRestart"),
    "actuator": ("Actuator", "ActuatorBC", "ActuatorBCParabolicV", "ActuatorBCRotation", "ActuatorBCUniformU",
                 "ActuatorForceGaussianV", "ACTUATOR_TYPE", "CYLINDER_ACTUATION_MODE"),
    "sensor": ("Sensor", "SensorPoint", "SensorIntegral", "SensorHorizontalWallShear", "SENSOR_TYPE"),
}

__all__ = ["__version__"]
for _mod, _names in _EXPORTS.items():
    _m = _importlib.import_module(f"{__name__}.{_mod}")
    for _n in _names:
        globals()[_n] = getattr(_m, _n)
        __all__.append(_n)
del _mod, _names, _m, _n
