"""Parameter bundles shared by the example cases' ``make_default`` constructors."""

from __future__ import annotations

from pathlib import Path

import numpy as np

from .. import flowsolverparameters as fsp
from ..sensor import SENSOR_TYPE, SensorPoint


def probes(spec):
    """[("V", (3.0, 0.0)), ...] → list of point sensors."""
    return [SensorPoint(sensor_type=SENSOR_TYPE[kind], position=np.asarray(xy, dtype=float)) for kind, xy in spec]


def bundle(*, Re, dt, num_steps, Tstart, save_every, path_out, mesh, mesh_extent, sensors, actuators, uinf=1.0, D=1.0,
           solver=None, flow_extra=None):
    """Keyword arguments for ``FlowSolver.__init__`` from a compact case description (the parameter classes
    are the reference's: flowsolverparameters.py)."""
    flow = fsp.ParamFlow(Re=Re, uinf=uinf)
    flow.user_data["D"] = D
    flow.user_data.update(flow_extra or {})
    msh = fsp.ParamMesh(meshpath=Path(mesh))
    msh.user_data.update(mesh_extent)
    return dict(
        params_flow=flow,
        params_time=fsp.ParamTime(num_steps=num_steps, dt=dt, Tstart=Tstart),
        params_save=fsp.ParamSave(save_every=save_every, path_out=Path(path_out)),
        params_solver=fsp.ParamSolver(**(solver or dict(throw_error=True, is_eq_nonlinear=True, shift=0.0))),
        params_mesh=msh,
        params_control=fsp.ParamControl(sensor_list=list(sensors), actuator_list=list(actuators)),
        params_ic=fsp.ParamIC(),
    )
