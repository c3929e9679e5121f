"""Fluidic pinball (three cylinders): BASELINE config 5.

Counterpart of the reference's ``src/examples/pinball/pinballflowsolver.py``: SUCTION mode (no-slip
bodies + three parabolic slots) or ROTATION mode (the whole cylinder surfaces are the actuators),
steady-state BCs with the uniform profile on inlet *and* walls.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pandas as pd

from ... import flowsolver
from ...actuator import CYLINDER_ACTUATION_MODE
from ...fem.boundary import DOLFIN_EPS, Constant, DirichletBC, SubDomain, between, near
from ...flowfield import BoundaryConditions

DEFAULT_MESH = Path(__file__).resolve().parent / "data_input" / "mesh_middle_gmsh.npz"


class PinballCustomInitialGuess:
    """Initial guess of the base-flow iterations (reference ``pinballflowsolver.py:328-358``): a uniform
    (u, v, p) chosen by ``mode`` — symmetric (1, 0, 0), antisymmetric_top (1/√2, +1/√2, 0) or
    antisymmetric_bot (1/√2, −1/√2, 0); the antisymmetric ones steer Picard towards one of the two
    asymmetric steady branches the pinball has at Re ≳ 70."""

    _MODES = {"symmetric": (1.0, 0.0), "antisymmetric_top": (2.0**-0.5, 2.0**-0.5), "antisymmetric_bot": (2.0**-0.5, -(2.0**-0.5))}

    def __init__(self, mode: str = "symmetric"):
        if mode not in self._MODES:
            raise ValueError(f"Unknown mode '{mode}'")
        self.mode = mode

    def __call__(self, x):
        out = np.zeros((np.shape(x)[0], 3))
        out[:, 0], out[:, 1] = self._MODES[self.mode]
        return out

    def value_shape(self):
        return (3,)

    def as_dolfin_function(self, function_space, interp: bool = True):
        """The guess as a field on ``function_space`` (nodal interpolation; for a constant the L2
        projection of the reference's ``interp=False`` branch is the same field)."""
        from ...fem.spaces import Function

        f = Function(function_space)
        f.interpolate(self)
        return f


class PinballFlowSolver(flowsolver.FlowSolver):
    """Flow past 3 cylinders (fluidic pinball). Proposed Re=100."""

    def _make_boundaries(self):
        TOL = DOLFIN_EPS
        ud = self.params_mesh.user_data
        xinfa, xinf, yinf = ud["xinfa"], ud["xinf"], ud["yinf"]
        mode = self.params_control.user_data["mode_actuation"]
        radius = self.params_flow.user_data["D"] / 2
        xm = -1.5 * np.cos(np.pi / 6)
        X = lambda x: x[:, 0]  # noqa: E731
        Y = lambda x: x[:, 1]  # noqa: E731
        close_top = lambda x: between(X(x), -radius, radius) & between(Y(x), radius / 2, 5 * radius / 2)  # noqa: E731
        close_bot = lambda x: between(X(x), -radius, radius) & between(Y(x), -5 * radius / 2, -radius / 2)  # noqa: E731
        close_mid = lambda x: between(X(x), -radius + xm, radius + xm) & between(Y(x), -radius, radius)  # noqa: E731
        names = ["inlet", "outlet", "walls"]
        preds = [
            lambda x, ob: ob & near(X(x), xinfa, TOL),
            lambda x, ob: ob & near(X(x), xinf, TOL),
            lambda x, ob: ob & (near(Y(x), -yinf, TOL) | near(Y(x), yinf, TOL)),
        ]
        if mode == CYLINDER_ACTUATION_MODE.SUCTION:
            ld = self.params_control.actuator_list[0].width
            names += ["cylinder_top", "cylinder_bot", "cylinder_mid", "actuator_mid", "actuator_top", "actuator_bot"]
            preds += [
                lambda x, ob: ob & close_top(x),
                lambda x, ob: ob & close_bot(x),
                lambda x, ob: ob & close_mid(x),
                lambda x, ob: ob & close_mid(x) & between(X(x), -ld + xm, xm + ld),
                lambda x, ob: ob & close_top(x) & between(X(x), -ld, ld),
                lambda x, ob: ob & close_bot(x) & between(X(x), -ld, ld),
            ]
        else:
            names += ["actuator_mid", "actuator_top", "actuator_bot"]
            preds += [lambda x, ob: ob & close_mid(x), lambda x, ob: ob & close_top(x), lambda x, ob: ob & close_bot(x)]
        return pd.DataFrame(index=names, data={"subdomain": [SubDomain(f, n) for n, f in zip(names, preds)]})

    def _make_bcs(self):
        W = self.W
        mode = self.params_control.user_data["mode_actuation"]
        g = self.get_subdomain
        acts = self.params_control.actuator_list
        bcu = [DirichletBC(W.sub(0), Constant((0, 0)), g("inlet")), DirichletBC(W.sub(0).sub(1), Constant(0), g("walls"))]
        if mode == CYLINDER_ACTUATION_MODE.SUCTION:
            bcu += [DirichletBC(W.sub(0), Constant((0, 0)), g(n)) for n in ("cylinder_top", "cylinder_bot", "cylinder_mid")]
        for a, name in zip(acts, ("actuator_mid", "actuator_top", "actuator_bot")):
            a.boundary = g(name)  # as the reference's _make_bcs leaves it for OperatorGetter.get_B
        bcu += [
            DirichletBC(W.sub(0), acts[0].expression, g("actuator_mid")),
            DirichletBC(W.sub(0), acts[1].expression, g("actuator_top")),
            DirichletBC(W.sub(0), acts[2].expression, g("actuator_bot")),
        ]
        return BoundaryConditions(bcu=bcu, bcp=[])

    def _make_BCs(self) -> BoundaryConditions:
        """Steady-state BCs: uniform flow at inlet and walls (reference ``pinballflowsolver.py:186-192``)."""
        uniform = Constant((self.params_flow.uinf, 0))
        bcs = self._make_bcs()
        return BoundaryConditions(
            bcu=[DirichletBC(self.W.sub(0), uniform, self.get_subdomain("inlet")), DirichletBC(self.W.sub(0), uniform, self.get_subdomain("walls"))]
            + bcs.bcu[2:],
            bcp=[],
        )

    def compute_force_coefficients(self, u, p) -> dict:
        """{surface name: (cl, cd)} of every cylinder surface (reference ``pinballflowsolver.py:202-232``)."""
        from ...fem.forces import force_coefficients

        if self.params_control.user_data["mode_actuation"] == CYLINDER_ACTUATION_MODE.SUCTION:
            surfaces = ["cylinder_mid", "actuator_mid", "cylinder_top", "actuator_top", "cylinder_bot", "actuator_bot"]
        else:
            surfaces = ["actuator_mid", "actuator_top", "actuator_bot"]
        return force_coefficients(self, u, p, surfaces)

    @classmethod
    def make_default(cls, Re: float = 50, mode_actuation=None, path_out=None, num_steps: int = 10, save_every: int = 0,
                     Tstart: float = 0.0, verbose: int = 0, meshpath: str | Path | None = None) -> "PinballFlowSolver":
        from ... import flowsolverparameters as fsp
        from ...actuator import ActuatorBCParabolicV, ActuatorBCRotation
        from ...sensor import SENSOR_TYPE, SensorPoint

        if path_out is None:
            path_out = Path.cwd() / "data_output"
        if mode_actuation is None:
            mode_actuation = CYLINDER_ACTUATION_MODE.ROTATION
        params_flow = fsp.ParamFlow(Re=Re, uinf=1.0)
        params_flow.user_data["D"] = 1.0
        params_time = fsp.ParamTime(num_steps=num_steps, dt=0.005, Tstart=Tstart)
        params_save = fsp.ParamSave(save_every=save_every, path_out=Path(path_out))
        params_solver = fsp.ParamSolver(throw_error=True, is_eq_nonlinear=True, shift=0.0)
        params_mesh = fsp.ParamMesh(meshpath=Path(meshpath or DEFAULT_MESH))
        params_mesh.user_data.update({"xinf": 20, "xinfa": -6, "yinf": 6})
        D = params_flow.user_data["D"]
        pos_mid = [-1.5 * np.cos(np.pi / 6), 0.0]
        pos_top = [0.0, 0.75]
        if mode_actuation == CYLINDER_ACTUATION_MODE.SUCTION:
            width = ActuatorBCParabolicV.angular_size_deg_to_width(10, D / 2)
            actuator_list = [
                ActuatorBCParabolicV(width=width, position_x=pos_mid[0]),
                ActuatorBCParabolicV(width=width, position_x=pos_top[0]),
                ActuatorBCParabolicV(width=width, position_x=pos_top[0]),
            ]
        else:
            actuator_list = [
                ActuatorBCRotation(position_x=pos_mid[0], position_y=pos_mid[1], diameter=D),
                ActuatorBCRotation(position_x=pos_top[0], position_y=+pos_top[1], diameter=D),
                ActuatorBCRotation(position_x=pos_top[0], position_y=-pos_top[1], diameter=D),
            ]
        params_control = fsp.ParamControl(
            sensor_list=[SensorPoint(sensor_type=SENSOR_TYPE.V, position=np.array([xs, 0.0])) for xs in (8.0, 10.0, 12.0)],
            actuator_list=actuator_list,
            user_data={"mode_actuation": mode_actuation},
        )
        params_ic = fsp.ParamIC()
        return cls(params_flow=params_flow, params_time=params_time, params_save=params_save, params_solver=params_solver,
                   params_mesh=params_mesh, params_control=params_control, params_ic=params_ic, verbose=verbose)
