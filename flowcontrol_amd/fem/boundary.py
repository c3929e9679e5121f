"""Boundary sub-domains, facet marking and Dirichlet conditions with dolfin's semantics.

Restates the third-party behaviour the reference's case files depend on (SURVEY §8a row 13,
Appendix A):

* ``SubDomain.mark`` on facets: a facet belongs to the sub-domain iff **all its vertices and
  its midpoint** satisfy ``inside(x, on_boundary)``, with ``on_boundary`` = facet is exterior.
* ``near(a, b, eps)`` ⇔ ``|a - b| <= eps``; ``DOLFIN_EPS = 3e-16``.
* ``DirichletBC`` (topological): constrains every DoF attached to the marked facets; a
  sub-space BC such as ``W.sub(0).sub(1)`` constrains only that component; in a list of
  BCs later entries overwrite earlier ones on shared DoFs.

Used by the case files (``flowcontrol_amd/examples/*``) in place of
``dolfin.CompiledSubDomain`` / ``dolfin.DirichletBC`` (reference
``src/examples/cylinder/cylinderflowsolver.py:20-108``).
"""

from __future__ import annotations

from typing import Callable, Sequence

import numpy as np

from .spaces import FunctionSpace, TaylorHood

DOLFIN_EPS = 3.0e-16


def near(x, x0, eps: float = DOLFIN_EPS):
    return (x >= x0 - eps) & (x <= x0 + eps)


def between(x, lo, hi, tol: float = 0.0):
    """Inclusive interval test with additive tolerance (reference ``utils/fem.py:57-58``)."""
    return (x >= lo - tol) & (x <= hi + tol)


class SubDomain:
    """Vectorised predicate ``inside(x (n,2), on_boundary (n,) bool) -> (n,) bool``."""

    def __init__(self, inside: Callable[[np.ndarray, np.ndarray], np.ndarray], name: str = ""):
        self._inside = inside
        self.name = name

    def inside(self, x: np.ndarray, on_boundary) -> np.ndarray:
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        ob = np.broadcast_to(np.asarray(on_boundary, dtype=bool), (x.shape[0],))
        return np.asarray(self._inside(x, ob), dtype=bool)

    def mark_facets(self, th_or_mesh) -> np.ndarray:
        """Boolean mask over mesh edges (facets) that belong to this sub-domain."""
        mesh = th_or_mesh.mesh if isinstance(th_or_mesh, TaylorHood) else th_or_mesh
        on_b = mesh.edge_cells[:, 1] < 0
        v0 = mesh.coords[mesh.edges[:, 0]]
        v1 = mesh.coords[mesh.edges[:, 1]]
        mid = 0.5 * (v0 + v1)
        return self.inside(v0, on_b) & self.inside(v1, on_b) & self.inside(mid, on_b)

    def mark(self, markers: np.ndarray, value: int, mesh) -> None:
        markers[self.mark_facets(mesh)] = value


class Constant:
    def __init__(self, value):
        self.value = np.atleast_1d(np.asarray(value, dtype=np.float64))

    def __call__(self, x: np.ndarray) -> np.ndarray:
        x = np.atleast_2d(x)
        return np.broadcast_to(self.value, (x.shape[0], self.value.size)).copy()

    def values(self) -> np.ndarray:
        return self.value


class DirichletBC:
    """``DirichletBC(W.sub(0)[.sub(i)], value, subdomain)``.

    ``value`` is a :class:`Constant` or any callable ``x (n,2) -> (n, value_size)``
    (actuator expressions).  The value is re-evaluated on every
    :meth:`get_boundary_values` call, as ``SystemAssembler`` re-fetches BC values on each
    ``assemble``.
    """

    def __init__(self, space: FunctionSpace, value, subdomain: SubDomain):
        if space.kind != "W" or not space.component or space.component[0] != 0:
            raise ValueError("velocity BCs must be given on W.sub(0) or W.sub(0).sub(i)")
        self.space = space
        self.value = value
        self.subdomain = subdomain
        th = space.th
        facets = subdomain.mark_facets(th)
        self.facets = np.nonzero(facets)[0]
        edges = th.mesh.edges[self.facets]
        nodes = np.unique(np.r_[edges.reshape(-1), th.nv + self.facets]).astype(np.int64)
        self.nodes = nodes
        comps = (0, 1) if len(space.component) == 1 else (space.component[1],)
        self.comps = comps
        self.dofs = np.concatenate([nodes + c * th.nn for c in comps]) if nodes.size else np.zeros(0, np.int64)

    def function_space(self) -> FunctionSpace:
        return self.space

    def _values(self) -> np.ndarray:
        th = self.space.th
        if self.nodes.size == 0:
            return np.zeros(0)
        x = th.node_coords[self.nodes]
        val = np.asarray(self.value(x), dtype=np.float64)
        if val.ndim == 1:
            val = val[:, None]
        if len(self.comps) == 2:
            return np.concatenate([val[:, 0], val[:, 1]])
        # scalar BC on a component: a scalar value or the matching component of a vector value
        col = 0 if val.shape[1] == 1 else self.comps[0]
        return val[:, col].copy()

    def get_boundary_values(self) -> dict[int, float]:
        return dict(zip(self.dofs.tolist(), self._values().tolist()))

    def dof_values(self) -> tuple[np.ndarray, np.ndarray]:
        return self.dofs, self._values()


def combine_bcs(bcs: Sequence[DirichletBC], N: int) -> tuple[np.ndarray, np.ndarray]:
    """Merge a BC list into (sorted unique dofs, values); later BCs win on shared DoFs."""
    val = np.zeros(N)
    mask = np.zeros(N, dtype=bool)
    for bc in bcs:
        d, v = bc.dof_values()
        val[d] = v
        mask[d] = True
    dofs = np.nonzero(mask)[0]
    return dofs, val[dofs]


def pressure_pin(th, dofs: np.ndarray) -> int | None:
    """Pressure dof to pin (to 0) when the velocity is prescribed on the WHOLE boundary, else None.

    An enclosed flow (lid-driven cavity) leaves the pressure defined up to a constant and the monolithic
    matrix singular.  The reference hands that matrix to MUMPS as it is (``bcp=[]``,
    examples/lidcavity/lidcavityflowsolver.py:57-72) and lives with whatever level the null pivot
    produces; a block factorisation with explicit pivot inverses cannot, so the pressure is fixed at the
    vertex nearest the lower-left corner of the bounding box.  Velocities (all the reference tests look at)
    are unaffected: the dropped continuity row is the negative sum of the others."""
    m = th.mesh
    be = m.boundary_edges()
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    isbc = np.zeros(th.N, dtype=bool)
    isbc[np.asarray(dofs, dtype=np.int64)] = True
    if not (np.all(isbc[nodes]) and np.all(isbc[nodes + th.nn])):
        return None
    x = th.node_coords[: th.nv]
    corner = x.min(axis=0)
    return int(2 * th.nn + np.argmin(((x - corner) ** 2).sum(axis=1)))


def with_pressure_pin(th, dofs: np.ndarray, vals: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """(dofs, vals) extended by the pinned pressure dof when :func:`pressure_pin` asks for one
    (``vals`` may be 1-D values or a 2-D table of per-actuator profiles: the pin is 0 in all)."""
    pin = pressure_pin(th, dofs)
    if pin is None or pin in set(np.asarray(dofs).tolist()):
        return dofs, vals
    d = np.append(np.asarray(dofs, dtype=np.int64), pin)
    v = np.concatenate([vals, np.zeros((1,) + np.shape(vals)[1:])])
    o = np.argsort(d, kind="stable")
    return d[o], v[o]


__all__ = ["DOLFIN_EPS", "near", "between", "SubDomain", "Constant", "DirichletBC", "combine_bcs", "pressure_pin",
           "with_pressure_pin"]
