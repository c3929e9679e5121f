"""Function spaces, DoF maps and the dolfin-``Function`` subset the FlowSolver API exposes.

Replaces ``FlowSolver._make_function_spaces`` (reference ``src/flowcontrol/flowsolver.py:242-250``)
and the field-object protocol user scripts rely on (SURVEY §8b "Field-object protocol"):
``.vector().get_local()``, ``.vector()[:]``, ``.set_local`` + ``.apply("insert")``,
``.vector().norm``, ``.copy(deepcopy=True)``, ``.assign``, ``.split(deepcopy=True)``,
``V.dim()``, point evaluation ``f(x)``.

DoF numbering (ours, locality ordered through ``Mesh.from_arrays``):
  P2 scalar nodes  : vertex v → v ; edge e → nv + e            (nn = nv + ne)
  V = P2 × P2      : [ux(0..nn), uy(0..nn)]                   (2 nn)
  P = P1           : vertex v → v                             (nv)
  W = V × P        : [ux, uy, p]                              (2 nn + nv)
All quantities the reference's tests pin are numbering independent.
"""

from __future__ import annotations

import numpy as np

from . import element as el
from .mesh import Mesh


class Vector:
    """Thin numpy-backed stand-in for ``dolfin.GenericVector``."""

    __slots__ = ("_a",)

    def __init__(self, a: np.ndarray):
        self._a = a

    def get_local(self) -> np.ndarray:
        return self._a.copy()

    def set_local(self, a) -> None:
        self._a[:] = np.asarray(a, dtype=np.float64)

    def apply(self, mode: str = "insert") -> None:  # no ghost sync on a single host array
        return None

    def __getitem__(self, k):
        return self._a[k].copy() if isinstance(k, slice) else self._a[k]

    def __setitem__(self, k, v) -> None:
        self._a[k] = v

    def __len__(self) -> int:
        return self._a.size

    def size(self) -> int:
        return self._a.size

    def norm(self, kind: str = "l2") -> float:
        if kind == "l2":
            return float(np.sqrt(np.dot(self._a, self._a)))
        if kind == "linf":
            return float(np.abs(self._a).max()) if self._a.size else 0.0
        if kind == "l1":
            return float(np.abs(self._a).sum())
        raise ValueError(f"unknown norm {kind!r}")

    def max(self) -> float:
        return float(self._a.max())

    def min(self) -> float:
        return float(self._a.min())

    def sum(self) -> float:
        return float(self._a.sum())

    def array(self) -> np.ndarray:
        return self._a

    def __sub__(self, other: "Vector") -> "Vector":
        return Vector(self._a - other._a)

    def __add__(self, other: "Vector") -> "Vector":
        return Vector(self._a + other._a)


class FunctionSpace:
    """One of V (kind 'V'), P ('P') or W ('W') on a Taylor–Hood discretisation."""

    def __init__(self, th: "TaylorHood", kind: str, component: tuple[int, ...] = ()):
        self.th = th
        self.kind = kind
        self.component = component

    def mesh(self) -> Mesh:
        return self.th.mesh

    def dim(self) -> int:
        th = self.th
        if self.component:
            return len(self.dofs())
        return {"V": 2 * th.nn, "P": th.nv, "W": th.N}[self.kind]

    def sub(self, i: int) -> "FunctionSpace":
        return FunctionSpace(self.th, self.kind, self.component + (int(i),))

    def num_sub_spaces(self) -> int:
        if self.kind == "W" and not self.component:
            return 2
        if (self.kind == "V" and not self.component) or (self.kind == "W" and self.component == (0,)):
            return 2
        return 0

    def dofs(self) -> np.ndarray:
        """Global DoF ids (in the numbering of this space's root) covered by this (sub)space."""
        th = self.th
        nn, nv = th.nn, th.nv
        if self.kind == "P":
            return np.arange(nv)
        if self.kind == "V":
            if not self.component:
                return np.arange(2 * nn)
            return np.arange(nn) + self.component[0] * nn
        c = self.component
        if not c:
            return np.arange(th.N)
        if c == (1,):
            return 2 * nn + np.arange(nv)
        if c == (0,):
            return np.arange(2 * nn)
        if len(c) == 2 and c[0] == 0:
            return np.arange(nn) + c[1] * nn
        raise ValueError(f"no subspace {c} of W")

    def collapse(self) -> "FunctionSpace":
        if self.kind == "W" and self.component == (0,):
            return self.th.V
        if self.kind == "W" and self.component == (1,):
            return self.th.P
        return self

    def tabulate_dof_coordinates(self) -> np.ndarray:
        th = self.th
        if self.kind == "P":
            return th.mesh.coords.copy()
        if self.kind == "V":
            return np.vstack([th.node_coords, th.node_coords])
        return np.vstack([th.node_coords, th.node_coords, th.mesh.coords])

    def ufl_element(self):
        return (self.kind, self.component)


class Function:
    """Nodal finite-element function on V, P or W."""

    def __init__(self, space: FunctionSpace, values: np.ndarray | None = None, name: str = "f"):
        if space.component:
            raise ValueError("Functions live on root spaces")
        self.space = space
        n = space.dim()
        self._a = np.zeros(n) if values is None else np.array(values, dtype=np.float64).reshape(n)
        self._name = name

    def function_space(self) -> FunctionSpace:
        return self.space

    def vector(self) -> Vector:
        return Vector(self._a)

    def copy(self, deepcopy: bool = True) -> "Function":
        return Function(self.space, self._a.copy() if deepcopy else self._a, self._name)

    def assign(self, other: "Function") -> None:
        if other.space.kind != self.space.kind or other._a.size != self._a.size:
            raise ValueError("assign: function spaces differ")
        self._a[:] = other._a

    def rename(self, name: str, label: str = "") -> None:
        self._name = name

    def name(self) -> str:
        return self._name

    def value_shape(self) -> tuple[int, ...]:
        return {"V": (2,), "P": (), "W": (3,)}[self.space.kind]

    def split(self, deepcopy: bool = True) -> tuple["Function", "Function"]:
        if self.space.kind != "W":
            raise ValueError("split() is defined for mixed (W) functions")
        th = self.space.th
        u = Function(th.V, self._a[: 2 * th.nn].copy())
        p = Function(th.P, self._a[2 * th.nn :].copy())
        return u, p

    def sub(self, i: int, deepcopy: bool = True) -> "Function":
        th = self.space.th
        if self.space.kind == "W":
            return self.split()[i]
        if self.space.kind == "V":
            raise NotImplementedError("scalar P2 sub-functions are not exposed")
        raise ValueError("scalar function has no sub-functions")

    def interpolate(self, expr) -> None:
        """Nodal interpolation of ``expr(x) -> (n, value_size)``."""
        th = self.space.th
        if self.space.kind == "P":
            self._a[:] = np.asarray(expr(th.mesh.coords)).reshape(-1)
        elif self.space.kind == "V":
            v = np.asarray(expr(th.node_coords))
            self._a[: th.nn], self._a[th.nn :] = v[:, 0], v[:, 1]
        else:
            v = np.asarray(expr(th.node_coords))
            self._a[: th.nn], self._a[th.nn : 2 * th.nn] = v[:, 0], v[:, 1]
            self._a[2 * th.nn :] = np.asarray(expr(th.mesh.coords))[:, 2]

    def __call__(self, *x):
        pt = np.asarray(x[0] if len(x) == 1 else x, dtype=np.float64).reshape(-1)[:2]
        th = self.space.th
        cell, lam = th.locate(pt)
        if cell < 0:
            raise RuntimeError(f"point {pt} is outside the mesh")
        nn = th.nn
        nodes = th.cell_nodes[cell]
        verts = th.mesh.cells[cell]
        phi2 = el.p2_basis(lam)
        phi1 = el.p1_basis(lam)
        k = self.space.kind
        if k == "P":
            return float(phi1 @ self._a[verts])
        ux = float(phi2 @ self._a[nodes])
        uy = float(phi2 @ self._a[nn + nodes])
        if k == "V":
            return np.array([ux, uy])
        return np.array([ux, uy, float(phi1 @ self._a[2 * nn + verts])])


class TaylorHood:
    """P2/P1 discretisation on a triangle mesh: DoF maps, geometry factors, point location."""

    def __init__(self, mesh: Mesh):
        self.mesh = mesh
        nv, ne, nc = mesh.num_vertices, mesh.num_edges, mesh.num_cells
        self.nv, self.ne, self.nc = nv, ne, nc
        self.nn = nv + ne
        self.N = 2 * self.nn + nv
        self.cell_nodes = np.hstack([mesh.cells, mesh.cell_edges + nv]).astype(np.int32)  # (nc, 6)
        self.node_coords = np.vstack([mesh.coords, mesh.edge_midpoints()])
        # (nc, 15) mixed dofs: 6 ux, 6 uy, 3 p
        self.cell_dofs = np.hstack(
            [self.cell_nodes, self.cell_nodes + self.nn, mesh.cells + 2 * self.nn]
        ).astype(np.int32)
        # affine geometry: x = x0 + J [ξ, η]
        p = mesh.coords[mesh.cells]
        J = np.empty((nc, 2, 2))
        J[:, :, 0] = p[:, 1] - p[:, 0]
        J[:, :, 1] = p[:, 2] - p[:, 0]
        det = J[:, 0, 0] * J[:, 1, 1] - J[:, 0, 1] * J[:, 1, 0]
        self.detJ = det  # > 0 (cells are CCW)
        Jinv = np.empty_like(J)
        Jinv[:, 0, 0], Jinv[:, 0, 1] = J[:, 1, 1] / det, -J[:, 0, 1] / det
        Jinv[:, 1, 0], Jinv[:, 1, 1] = -J[:, 1, 0] / det, J[:, 0, 0] / det
        self.Jinv = Jinv  # d(ξ,η)/d(x,y): grad_x φ = Jinvᵀ grad_ref φ
        self.V = FunctionSpace(self, "V")
        self.P = FunctionSpace(self, "P")
        self.W = FunctionSpace(self, "W")
        self._grid = None

    # ── geometry array handed to the device: (nc, 5) = Jinv (row-major 4) + detJ
    def geometry_table(self) -> np.ndarray:
        g = np.empty((self.nc, 5))
        g[:, :4] = self.Jinv.reshape(self.nc, 4)
        g[:, 4] = self.detJ
        return g

    # ── point location (uniform bucket grid; replaces dolfin's bounding-box tree) ──
    def _build_grid(self) -> None:
        m = self.mesh
        p = m.coords[m.cells]
        lo, hi = p.min(axis=1), p.max(axis=1)
        glo, ghi = m.coords.min(axis=0), m.coords.max(axis=0)
        n = int(max(8, min(1024, np.sqrt(self.nc))))
        h = np.maximum((ghi - glo) / n, 1e-300)
        i0 = np.clip(((lo - glo) / h).astype(int), 0, n - 1)
        i1 = np.clip(((hi - glo) / h).astype(int), 0, n - 1)
        buckets: dict[int, list[int]] = {}
        for c in range(self.nc):
            for ix in range(i0[c, 0], i1[c, 0] + 1):
                for iy in range(i0[c, 1], i1[c, 1] + 1):
                    buckets.setdefault(ix * n + iy, []).append(c)
        self._grid = (glo, h, n, buckets)

    def locate(self, pt: np.ndarray, tol: float = 1e-12) -> tuple[int, np.ndarray]:
        """Return (cell, barycentric coords) of the first cell containing ``pt`` or (-1, None)."""
        if self._grid is None:
            self._build_grid()
        glo, h, n, buckets = self._grid
        ij = ((pt - glo) / h).astype(int)
        if np.any(ij < -1) or np.any(ij > n):
            return -1, None
        ij = np.clip(ij, 0, n - 1)
        best, best_lam, best_min = -1, None, -np.inf
        for c in buckets.get(int(ij[0]) * n + int(ij[1]), []):
            x0 = self.mesh.coords[self.mesh.cells[c, 0]]
            xi = self.Jinv[c] @ (pt - x0)
            lam = np.array([1.0 - xi[0] - xi[1], xi[0], xi[1]])
            mn = lam.min()
            if mn >= -tol and mn > best_min:
                best, best_lam, best_min = c, lam, mn
                if mn >= 0.0:
                    break
        return best, best_lam

    def point_eval_row(self, pt, component: int) -> tuple[np.ndarray, np.ndarray]:
        """Sparse row (dof ids, weights) of the functional up ↦ up(pt)[component] on W."""
        cell, lam = self.locate(np.asarray(pt, dtype=np.float64)[:2])
        if cell < 0:
            raise RuntimeError(f"point {pt} is outside the mesh")
        if component in (0, 1):
            return self.cell_nodes[cell].astype(np.int64) + component * self.nn, el.p2_basis(lam)
        return self.mesh.cells[cell].astype(np.int64) + 2 * self.nn, el.p1_basis(lam)


__all__ = ["Vector", "FunctionSpace", "Function", "TaylorHood"]


def default_device_index() -> int:
    """GPU of this process: ``FC_DEVICE`` if set; else, in a one-process-per-GPU launch (``LOCAL_RANK`` in the
    environment, as ``torchrun`` sets it), the local rank folded onto the visible devices; else torch's current
    device when torch has already initialised the GPU runtime in this process; else 0."""
    import os
    import sys

    from .. import _lib

    if os.environ.get("FC_DEVICE"):
        return int(os.environ["FC_DEVICE"])
    if os.environ.get("LOCAL_RANK"):
        return int(os.environ["LOCAL_RANK"]) % max(_lib.device_count(), 1)
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        return int(torch.cuda.current_device())
    return 0


def _th_device(self, device_index: int | None = None):
    """The MI355X handle bound to this discretisation (created on first use; no CPU fallback).
    ``device_index=None``: :func:`default_device_index` — one process per GPU binds its own GPU."""
    if getattr(self, "_device", None) is None:
        from ..device import DeviceSolver

        self._device = DeviceSolver(self, default_device_index() if device_index is None else int(device_index))
    elif device_index is not None and int(device_index) != self._device.device_index:
        raise RuntimeError(f"this discretisation is already bound to GPU {self._device.device_index}, not {device_index}")
    return self._device


def _th_release(self) -> None:
    dev = getattr(self, "_device", None)
    if dev is not None:
        # whoever still waits for results of the handle's last step (a FlowSolver's deferred log row: energy, residual) fetches them now
        for hook in list(getattr(self, "_release_hooks", ())):
            hook()
        dev.close()
    self._device = None


TaylorHood.device = _th_device
TaylorHood.release_device = _th_release
