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

DEFAULT_MESH = Path(__file__).resolve().parent / "data_input" / "lidcavity_mesh64.npz"


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
        """The reference's standard case (``lidcavityflowsolver.py:94-148``): dt = 0.005, uniform-U actuator on the
        lid, a V probe at (0.05, 0.5) and a U probe at (0.5, 0.95), the 64 × 64 unit-square mesh."""
        from ...actuator import ActuatorBCUniformU
        from .._defaults import bundle, probes

        return cls(verbose=verbose, **bundle(
            Re=Re, dt=0.005, num_steps=num_steps, Tstart=Tstart, save_every=save_every,
            path_out=path_out if path_out is not None else Path(__file__).parent / "data_output",
            mesh=meshpath or DEFAULT_MESH, mesh_extent={"yup": 1, "ylo": 0, "xri": 1, "xle": 0},
            sensors=probes([("V", (0.05, 0.5)), ("U", (0.5, 0.95))]), actuators=[ActuatorBCUniformU(boundary_name="lid")]))
