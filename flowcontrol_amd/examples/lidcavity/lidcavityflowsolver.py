"""Lid-driven cavity (supercritical Hopf bifurcation near Re ≈ 7700; the reference proposes Re = 8000).

Counterpart of the reference's ``src/examples/lidcavity/lidcavityflowsolver.py``: unit square, the lid
(top wall) moves at ``uinf`` and carries the actuation (uniform tangential velocity), the three other
walls are no-slip.  The velocity is prescribed on the whole boundary, so the pressure level is free:
the reference leaves the singular matrix to MUMPS, here one pressure dof is pinned
(``fem.boundary.pressure_pin``) — velocities, sensors and energies are the same.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pandas

from ... import flowsolver
from ...fem.boundary import DOLFIN_EPS, Constant, DirichletBC, SubDomain, near
from ...flowfield import BoundaryConditions

DEFAULT_MESH = Path(__file__).resolve().parents[3] / "tests" / "golden" / "meshes" / "lidcavity_mesh64.npz"


class LidCavityFlowSolver(flowsolver.FlowSolver):
    """Lid-driven cavity flow. Proposed Re=8000."""

    def _make_boundaries(self):
        """lid (top wall, actuated), leftwall, rightwall, bottomwall of [xle, xri] × [ylo, yup]
        (reference ``lidcavityflowsolver.py:22-55``)."""
        ud = self.params_mesh.user_data
        TOL = DOLFIN_EPS
        subs = {
            "lid": lambda x, ob: ob & near(x[:, 1], ud["yup"], TOL),
            "leftwall": lambda x, ob: ob & near(x[:, 0], ud["xle"], TOL),
            "rightwall": lambda x, ob: ob & near(x[:, 0], ud["xri"], TOL),
            "bottomwall": lambda x, ob: ob & near(x[:, 1], ud["ylo"], TOL),
        }
        return pandas.DataFrame(index=list(subs), data={"subdomain": [SubDomain(f, n) for n, f in subs.items()]})

    def _make_bcs(self):
        """Perturbation BCs: actuator expression on the lid, no-slip on the three other walls (the walls
        come later in the list, so they win at the two upper corners) — reference ``:57-68``."""
        W = self.W
        g = self.get_subdomain
        zero2 = Constant((0, 0))
        bcu = [
            DirichletBC(W.sub(0), self.params_control.actuator_list[0].expression, g("lid")),
            DirichletBC(W.sub(0), zero2, g("leftwall")),
            DirichletBC(W.sub(0), zero2, g("rightwall")),
            DirichletBC(W.sub(0), zero2, g("bottomwall")),
        ]
        return BoundaryConditions(bcu=bcu, bcp=[])

    def _make_BCs(self) -> BoundaryConditions:
        """Steady-state BCs: the lid moves at uinf, walls no-slip (reference ``:70-78``)."""
        bcu_lid_ss = DirichletBC(self.W.sub(0), Constant((self.params_flow.uinf, 0)), self.get_subdomain("lid"))
        bcs = self._make_bcs()
        return BoundaryConditions(bcu=[bcu_lid_ss] + bcs.bcu[1:], bcp=[])

    def _default_steady_state_initial_guess(self):
        """Zero everywhere — the cavity starts from rest (reference ``:80-92``)."""
        return lambda x: np.zeros((x.shape[0], 3))

    @classmethod
    def make_default(cls, Re: float = 8000, path_out=None, num_steps: int = 10, save_every: int = 0, Tstart: float = 0.0,
                     verbose: int = 0, meshpath: str | Path | None = None) -> "LidCavityFlowSolver":
        """Standard parameters of the reference (``:94-148``): dt = 0.005, uniform-U lid actuator, a V probe
        at (0.05, 0.5) and a U probe at (0.5, 0.95), 64 × 64 mesh."""
        from ... import flowsolverparameters as fsp
        from ...actuator import ActuatorBCUniformU
        from ...sensor import SENSOR_TYPE, SensorPoint

        if path_out is None:
            path_out = Path(__file__).parent / "data_output"
        params_flow = fsp.ParamFlow(Re=Re, uinf=1.0)
        params_flow.user_data["D"] = 1.0
        params_time = fsp.ParamTime(num_steps=num_steps, dt=0.005, Tstart=Tstart)
        params_save = fsp.ParamSave(save_every=save_every, path_out=path_out)
        params_solver = fsp.ParamSolver(throw_error=True, is_eq_nonlinear=True, shift=0.0)
        params_mesh = fsp.ParamMesh(meshpath=meshpath or DEFAULT_MESH)
        params_mesh.user_data.update({"yup": 1, "ylo": 0, "xri": 1, "xle": 0})
        params_control = fsp.ParamControl(
            sensor_list=[
                SensorPoint(sensor_type=SENSOR_TYPE.V, position=np.array([0.05, 0.5])),
                SensorPoint(sensor_type=SENSOR_TYPE.U, position=np.array([0.5, 0.95])),
            ],
            actuator_list=[ActuatorBCUniformU(boundary_name="lid")],
        )
        return cls(params_flow=params_flow, params_time=params_time, params_save=params_save, params_solver=params_solver,
                   params_mesh=params_mesh, params_control=params_control, params_ic=fsp.ParamIC(), verbose=verbose)
