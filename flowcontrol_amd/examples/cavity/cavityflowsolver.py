"""Flow over an open cavity (Re = 7500): BASELINE config 3.

Counterpart of the reference's ``src/examples/cavity/cavityflowsolver.py``: ten boundaries, nine
Dirichlet conditions (slip = v only on the far walls, no-slip on the cavity and the adjacent
lower-wall segments), a Gaussian FORCE actuator and a wall-shear integral sensor.  Predicates are
the reference's C++ strings (``cavityflowsolver.py:54-120``) written as numpy comparisons.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pandas

from ... import flowsolver
from ...fem.boundary import DOLFIN_EPS, Constant, DirichletBC, SubDomain, between, near
from ...flowfield import BoundaryConditions

DEFAULT_MESH = Path(__file__).resolve().parent / "data_input" / "cavity_coarse.npz"


class CavityFlowSolver(flowsolver.FlowSolver):
    """Flow over an open cavity. Proposed Re=7500."""

    def _make_boundaries(self):
        TOL = DOLFIN_EPS
        L = self.params_flow.user_data["L"]
        D = self.params_flow.user_data["D"]
        ud = self.params_mesh.user_data
        xinfa, xinf, yinf = ud["xinfa"], ud["xinf"], ud["yinf"]
        x0l, x0r = ud["x0ns_left"], ud["x0ns_right"]
        X = lambda x: x[:, 0]  # noqa: E731
        Y = lambda x: x[:, 1]  # noqa: E731
        subs = {
            "inlet": lambda x, ob: ob & near(X(x), xinfa, TOL),
            "outlet": lambda x, ob: ob & near(X(x), xinf, TOL),
            "upper_wall": lambda x, ob: ob & near(Y(x), yinf, TOL),
            "cavity_left": lambda x, ob: ob & near(X(x), 0.0, TOL) & between(Y(x), -D, 0.0),
            "cavity_botm": lambda x, ob: ob & near(Y(x), -D, TOL) & between(X(x), 0.0, L),
            "cavity_right": lambda x, ob: ob & near(X(x), L, TOL) & between(Y(x), -D, 0.0),
            "lower_wall_left_sf": lambda x, ob: ob & (X(x) >= xinfa) & (X(x) <= x0l + 10 * TOL) & near(Y(x), 0.0, TOL),
            "lower_wall_left_ns": lambda x, ob: ob & (X(x) >= x0l - 10 * TOL) & (X(x) <= 0.0) & near(Y(x), 0.0, TOL),
            "lower_wall_right_ns": lambda x, ob: ob & near(Y(x), 0.0, TOL) & between(X(x), L, x0r),
            "lower_wall_right_sf": lambda x, ob: ob & near(Y(x), 0.0, TOL) & between(X(x), x0r, xinf),
        }
        return pandas.DataFrame(index=list(subs), data={"subdomain": [SubDomain(f, n) for n, f in subs.items()]})

    def _make_bcs(self):
        W = self.W
        zero2, zero = Constant((0, 0)), Constant(0)
        g = self.get_subdomain
        bcu = [
            DirichletBC(W.sub(0), zero2, g("inlet")),
            DirichletBC(W.sub(0).sub(1), zero, g("upper_wall")),
            DirichletBC(W.sub(0).sub(1), zero, g("lower_wall_left_sf")),
            DirichletBC(W.sub(0), zero2, g("lower_wall_left_ns")),
            DirichletBC(W.sub(0), zero2, g("lower_wall_right_ns")),
            DirichletBC(W.sub(0).sub(1), zero, g("lower_wall_right_sf")),
            DirichletBC(W.sub(0), zero2, g("cavity_left")),
            DirichletBC(W.sub(0), zero2, g("cavity_botm")),
            DirichletBC(W.sub(0), zero2, g("cavity_right")),
        ]
        return BoundaryConditions(bcu=bcu, bcp=[])

    def _default_steady_state_initial_guess(self):
        """u = 1 in the channel, 0 inside the cavity (reference ``cavityflowsolver.py:195-209``)."""

        def guess(x):
            out = np.zeros((x.shape[0], 3))
            out[:, 0] = np.where(x[:, 1] >= 0, 1.0, 0.0)
            return out

        return guess

    @classmethod
    def make_default(cls, Re: float = 7500, path_out=None, num_steps: int = 10, save_every: int = 0, Tstart: float = 0.0,
                     verbose: int = 0, meshpath: str | Path | None = None) -> "CavityFlowSolver":
        """Standard parameters of the reference (``cavityflowsolver.py:211-279``): dt = 4e-4, one Gaussian
        force actuator at (-0.1, 0.02), wall-shear sensor on [1, 1.1] × {0} and a U probe at (0.1, 0.1)."""
        from ... import flowsolverparameters as fsp
        from ...actuator import ActuatorForceGaussianV
        from ...sensor import SENSOR_TYPE, SensorHorizontalWallShear, SensorPoint

        if path_out is None:
            path_out = Path.cwd() / "data_output"
        params_flow = fsp.ParamFlow(Re=Re, uinf=1.0)
        params_flow.user_data.update({"L": 1.0, "D": 1.0})
        params_time = fsp.ParamTime(num_steps=num_steps, dt=0.0004, Tstart=Tstart)
        params_save = fsp.ParamSave(save_every=save_every, path_out=Path(path_out))
        params_solver = fsp.ParamSolver(throw_error=True, is_eq_nonlinear=True, shift=0.0)
        params_mesh = fsp.ParamMesh(meshpath=Path(meshpath or DEFAULT_MESH))
        params_mesh.user_data.update({"xinf": 2.5, "xinfa": -1.2, "yinf": 0.5, "x0ns_left": -0.4, "x0ns_right": 1.75})
        params_control = fsp.ParamControl(
            sensor_list=[
                SensorHorizontalWallShear(sensor_index=100, x_sensor_left=1.0, x_sensor_right=1.1, y_sensor=0.0, sensor_type=SENSOR_TYPE.OTHER),
                SensorPoint(sensor_type=SENSOR_TYPE.U, position=np.array([0.1, 0.1])),
            ],
            actuator_list=[ActuatorForceGaussianV(sigma=0.0849, position=np.array([-0.1, 0.02]))],
        )
        params_ic = fsp.ParamIC()
        return cls(params_flow=params_flow, params_time=params_time, params_save=params_save, params_solver=params_solver,
                   params_mesh=params_mesh, params_control=params_control, params_ic=params_ic, verbose=verbose)
