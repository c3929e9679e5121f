"""Flow past a cylinder (Re = 100): the case file of BASELINE configs 1/2/4.

Counterpart of the reference's ``src/examples/cylinder/cylinderflowsolver.py``: same six
boundaries (inlet, outlet, walls, cylinder body, two actuator slots at the poles), same five
Dirichlet conditions in the same order, same ``make_default`` parameters.  The C++ predicate
strings of the reference (``cylinderflowsolver.py:35-83``) are written as vectorised numpy
predicates with identical comparisons and tolerances.
"""

from __future__ import annotations

from pathlib import Path

import pandas

from ... import flowsolver
from ...fem.boundary import DOLFIN_EPS, Constant, DirichletBC, SubDomain, between, near
from ...flowfield import BoundaryConditions

#: mesh fixture shipped with this repository (converted from the reference's O1.xdmf data file)
DEFAULT_MESH = Path(__file__).resolve().parent / "data_input" / "O1.npz"


def refined_cylinder_mesh(levels: int = 1, path: str | Path | None = None, source: str | Path | None = None) -> Path:
    """BASELINE config 4: the shipped mesh red-refined ``levels`` times (each triangle → 4), new boundary
    midpoints on the cylinder projected back onto r = 0.5.  Deterministic; written as
    an ``.npz`` mesh file (``coords``, ``cells``) and returned as a path for ``make_default(meshpath=...)``."""
    import tempfile

    import numpy as np

    from ...fem.mesh import read_xdmf_mesh

    m = read_xdmf_mesh(source or DEFAULT_MESH, reorder=False)

    def project(mid, is_boundary):
        r = np.hypot(mid[:, 0], mid[:, 1])
        on_cyl = is_boundary & (r < 0.6)
        out = mid.copy()
        out[on_cyl] *= (0.5 / r[on_cyl])[:, None]
        return out

    for _ in range(levels):
        m = m.refine(project)
    path = Path(path) if path else Path(tempfile.mkdtemp(prefix="fc_mesh_")) / f"O1_refined{levels}.npz"
    path.parent.mkdir(parents=True, exist_ok=True)
    np.savez(path, coords=m.coords, cells=m.cells)
    return path


class CylinderFlowSolver(flowsolver.FlowSolver):
    """Flow past a cylinder. Proposed Re=100."""

    def _make_boundaries(self):
        TOL = DOLFIN_EPS
        xinfa = self.params_mesh.user_data["xinfa"]
        xinf = self.params_mesh.user_data["xinf"]
        yinf = self.params_mesh.user_data["yinf"]
        radius = self.params_flow.user_data["D"] / 2
        ldelta = self.params_control.actuator_list[0].width

        def close_to_cylinder(x):
            return between(x[:, 0], -radius, radius) & between(x[:, 1], -radius, radius)

        inlet = SubDomain(lambda x, ob: ob & near(x[:, 0], xinfa, TOL), "inlet")
        outlet = SubDomain(lambda x, ob: ob & near(x[:, 0], xinf, TOL), "outlet")
        walls = SubDomain(lambda x, ob: ob & (near(x[:, 1], -yinf, TOL) | near(x[:, 1], yinf, TOL)), "walls")
        cylinder = SubDomain(
            lambda x, ob: ob & close_to_cylinder(x) & (between(x[:, 0], -radius, -ldelta) | between(x[:, 0], ldelta, radius)),
            "cylinder",
        )
        actuator_up = SubDomain(
            lambda x, ob: ob & close_to_cylinder(x) & between(x[:, 0], -ldelta, ldelta, tol=0.01) & between(x[:, 1], 0.0, radius),
            "actuator_up",
        )
        actuator_lo = SubDomain(
            lambda x, ob: ob & close_to_cylinder(x) & between(x[:, 0], -ldelta, ldelta, tol=0.01) & between(x[:, 1], -radius, 0.0),
            "actuator_lo",
        )
        return pandas.DataFrame(
            index=["inlet", "outlet", "walls", "cylinder", "actuator_up", "actuator_lo"],
            data={"subdomain": [inlet, outlet, walls, cylinder, actuator_up, actuator_lo]},
        )

    def _make_bcs(self):
        """Zero on inlet / walls (v only) / cylinder body; actuator expressions on the two slots."""
        W = self.W
        acts = self.params_control.actuator_list
        bcu_inlet = DirichletBC(W.sub(0), Constant((0, 0)), self.get_subdomain("inlet"))
        bcu_walls = DirichletBC(W.sub(0).sub(1), Constant(0), self.get_subdomain("walls"))
        bcu_cylinder = DirichletBC(W.sub(0), Constant((0, 0)), self.get_subdomain("cylinder"))
        bcu_actuation_up = DirichletBC(W.sub(0), acts[0].expression, self.get_subdomain("actuator_up"))
        bcu_actuation_lo = DirichletBC(W.sub(0), acts[1].expression, self.get_subdomain("actuator_lo"))
        return BoundaryConditions(bcu=[bcu_inlet, bcu_walls, bcu_cylinder, bcu_actuation_up, bcu_actuation_lo], bcp=[])

    def compute_steady_state(self, u_ctrl, method="newton", **kwargs):
        """Base flow, then the lift / drag coefficients of it as ``self.cl0, self.cd0`` (reference ``:110-113``)."""
        super().compute_steady_state(method=method, u_ctrl=u_ctrl, **kwargs)
        self.cl0, self.cd0 = self.compute_force_coefficients(self.fields.U0, self.fields.P0)

    def compute_force_coefficients(self, u, p):
        """(cl, cd) of the whole cylinder surface: body + both actuator slots (reference ``:115-126``)."""
        from ...fem.forces import force_coefficients

        parts = force_coefficients(self, u, p, ["cylinder", "actuator_up", "actuator_lo"])
        return sum(v[0] for v in parts.values()), sum(v[1] for v in parts.values())

    @classmethod
    def make_default(cls, Re: float = 100, path_out=None, num_steps: int = 10, save_every: int = 0, Tstart: float = 0.0,
                     verbose: int = 0, meshpath: str | Path | None = None) -> "CylinderFlowSolver":
        """The reference's standard case (``cylinderflowsolver.py:128-186``): dt = 0.005, two parabolic BC
        actuators of 10° at the poles of the cylinder, V probes at (3, 0) and (3.1, ±1), mesh O1."""
        from ...actuator import ActuatorBCParabolicV
        from .._defaults import bundle, probes

        width = ActuatorBCParabolicV.angular_size_deg_to_width(10, 0.5)  # 10 degrees on the radius-0.5 cylinder
        poles = [ActuatorBCParabolicV(width=width, position_x=0.0, boundary_name=name) for name in ("actuator_up", "actuator_lo")]
        return cls(verbose=verbose, **bundle(
            Re=Re, dt=0.005, num_steps=num_steps, Tstart=Tstart, save_every=save_every,
            path_out=path_out if path_out is not None else Path.cwd() / "data_output",
            mesh=meshpath or DEFAULT_MESH, mesh_extent={"xinf": 20, "xinfa": -10, "yinf": 10},
            sensors=probes([("V", (3.0, 0.0)), ("V", (3.1, 1.0)), ("V", (3.1, -1.0))]), actuators=poles))
