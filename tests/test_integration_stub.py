"""INTEGRATION.md §B made runnable: ``integration/hip_solver.py`` (the file a maintainer of the reference would add) against a
dolfin-shaped stand-in — sorted-vertex cells of both orientations, dolfin-style edge numbering, an interleaved mixed dof
numbering, ``as_backend_type(A).mat().getValuesCSR()``, vectors with ``get_local / set_local / apply``.  The host logic
(CCW flip, "edge opposite vertex k", dof matching, values onto the fc pattern) is checked without a GPU; with one, the
plug-in's set_operator / solve is checked against SuperLU."""
import sys
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "integration"))
import hip_solver  # noqa: E402

from oracle import ns_oracle as O  # noqa: E402


class _Conn:
    def __init__(self, rows):
        self.rows = rows

    def __call__(self, i):
        return self.rows[i]


class _Topology:
    def __init__(self, c21, e10):
        self.tab = {(2, 1): _Conn(c21), (1, 0): _Conn(e10)}

    def __call__(self, a, b):
        return self.tab[(a, b)]


class FakeDolfinMesh:
    """What dolfin gives for a triangulated square: vertex-sorted cells (orientation mixed), edges numbered dolfin-fashion
    (sorted by vertex pair), connectivities available after init(1) only."""

    def __init__(self, n):
        xs = np.linspace(0.0, 1.0, n + 1)
        X, Y = np.meshgrid(xs, xs, indexing="ij")
        self._x = np.stack([X.ravel(), Y.ravel()], axis=1)
        vid = lambda i, j: i * (n + 1) + j  # noqa: E731
        cells = []
        for i in range(n):
            for j in range(n):
                a, b, c, d = vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)
                cells += [sorted((a, b, c)), sorted((a, c, d))]
        self._cells = np.array(cells, dtype=np.uintp)
        self._ready = False

    def init(self, dim):
        pairs = sorted({tuple(sorted((int(t[(k + 1) % 3]), int(t[(k + 2) % 3])))) for t in self._cells for k in range(3)})
        eid = {p: i for i, p in enumerate(pairs)}
        self._e10 = [np.array(p, dtype=np.uintp) for p in pairs]
        self._c21 = [np.array([eid[tuple(sorted((int(t[(k + 1) % 3]), int(t[(k + 2) % 3]))))] for k in (2, 0, 1)], dtype=np.uintp) for t in self._cells]  # NOT in "opposite vertex" order
        self._ready = True

    def coordinates(self):
        return self._x

    def cells(self):
        return self._cells

    def num_edges(self):
        assert self._ready
        return len(self._e10)

    def topology(self):
        assert self._ready, "mesh.init(1) first"
        return _Topology(self._c21, self._e10)


class _DofMap:
    def __init__(self, dofs):
        self._d = list(dofs)

    def dofs(self):
        return self._d


class _Sub:
    def __init__(self, dofs=None, subs=None):
        self._dofs, self._subs = dofs, subs

    def sub(self, k):
        return self._subs[k]

    def dofmap(self):
        return _DofMap(self._dofs)


class FakeMixedSpace:
    """Taylor-Hood dofs numbered entity by entity, components interleaved: vertex v -> (ux, uy, p), then edge e -> (ux, uy)."""

    def __init__(self, mesh):
        nv, ne = len(mesh.coordinates()), mesh.num_edges()
        x = mesh.coordinates()
        mid = np.array([0.5 * (x[e[0]] + x[e[1]]) for e in [mesh.topology()(1, 0)(i) for i in range(ne)]])
        self.N = 3 * nv + 2 * ne
        self._xy = np.zeros((self.N, 2))
        ux, uy, pp = [], [], []
        for v in range(nv):
            ux.append(3 * v), uy.append(3 * v + 1), pp.append(3 * v + 2)
            self._xy[3 * v : 3 * v + 3] = x[v]
        for e in range(ne):
            ux.append(3 * nv + 2 * e), uy.append(3 * nv + 2 * e + 1)
            self._xy[3 * nv + 2 * e : 3 * nv + 2 * e + 2] = mid[e]
        self._s = _Sub(subs=[_Sub(subs=[_Sub(ux), _Sub(uy)]), _Sub(pp)])

    def tabulate_dof_coordinates(self):
        return self._xy

    def sub(self, k):
        return self._s.sub(k)


class FakeMatrix:
    def __init__(self, A):
        self.A = sp.csr_matrix(A)

    def mat(self):
        return self

    def getValuesCSR(self):
        return self.A.indptr, self.A.indices, self.A.data


class FakeVector:
    def __init__(self, a):
        self.a = np.array(a, dtype=float)
        self.applied = False

    def get_local(self):
        return self.a.copy()

    def set_local(self, v):
        self.a[:] = v

    def apply(self, mode):
        assert mode == "insert"
        self.applied = True


def _system(mesh, W):
    """BC-eliminated BDF2-like operator in the FAKE (dolfin) numbering, and the pieces to rebuild it in fc numbering."""
    coords, cells, cell_edges, edges = hip_solver.mesh_arrays(mesh)
    nv, ne = len(coords), len(edges)
    nn = nv + ne
    d = O.Disc(coords, cells, np.hstack([cells, cell_edges + nv]), nn)
    node_xy = np.vstack([coords, 0.5 * (coords[edges[:, 0]] + coords[edges[:, 1]])])
    U = np.r_[1 + 0.3 * np.sin(node_xy[:, 0]), 0.2 * np.cos(node_xy[:, 1])]
    wall = (node_xy[:, 1] < 1e-12) | (node_xy[:, 1] > 1 - 1e-12) | (node_xy[:, 0] < 1e-12)
    nodes = np.flatnonzero(wall)
    bc_fc = np.r_[nodes, nodes + nn]
    A_fc, _ = O.apply_bc_symmetric(O.assemble_matrix(d, mass=300.0, nu=0.01, adv=U, lin=U), None, bc_fc, np.zeros(bc_fc.size))
    to_fc = hip_solver.fc_dof_map(W, coords, edges)
    inv = np.empty_like(to_fc)
    inv[to_fc] = np.arange(to_fc.size)
    A_dolfin = sp.csr_matrix(A_fc)[inv][:, inv].tocsr()  # A_dolfin[to_fc[i], to_fc[j]] = A_fc[i, j]
    return A_fc, A_dolfin, to_fc[bc_fc], to_fc, (coords, cells, cell_edges, edges)


def test_host_side_of_the_stub_without_a_gpu():
    mesh = FakeDolfinMesh(6)
    coords, cells, cell_edges, edges = hip_solver.mesh_arrays(mesh)
    p = coords[cells]
    det = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 2, 0] - p[:, 0, 0]) * (p[:, 1, 1] - p[:, 0, 1])
    assert np.all(det > 0)  # every cell counter-clockwise although dolfin's come vertex-sorted
    assert not np.array_equal(cells, np.sort(cells, axis=1))  # ... so some really were flipped
    for c in range(len(cells)):
        for k in range(3):
            assert set(edges[cell_edges[c, k]]) == set(cells[c]) - {cells[c, k]}  # edge k is opposite local vertex k
    W = FakeMixedSpace(mesh)
    to_fc = hip_solver.fc_dof_map(W, coords, edges)
    nv, nn = len(coords), len(coords) + len(edges)
    assert np.array_equal(np.sort(to_fc), np.arange(W.N))
    node_xy = np.vstack([coords, 0.5 * (coords[edges[:, 0]] + coords[edges[:, 1]])])
    xy = W.tabulate_dof_coordinates()
    assert np.allclose(xy[to_fc[:nn]], node_xy) and np.allclose(xy[to_fc[nn : 2 * nn]], node_xy) and np.allclose(xy[to_fc[2 * nn :]], node_xy[:nv])
    assert np.all(to_fc[:nv] % 3 == 0) and np.all(to_fc[2 * nn :] % 3 == 2)  # ux of vertex v, p of vertex v in the fake's numbering
    # values onto the fc pattern: the oracle's matrix in dolfin numbering comes back as the fc-numbered one
    A_fc, A_dolfin, bc_dolfin, to_fc2, _ = _system(mesh, W)
    assert np.array_equal(to_fc, to_fc2)
    from flowcontrol_amd.fem.mesh import Mesh
    from flowcontrol_amd.fem.spaces import TaylorHood

    solver = hip_solver.HipNDSolver.__new__(hip_solver.HipNDSolver)
    solver.N = W.N
    full = sp.csr_matrix(O.assemble_matrix(O.Disc(coords, cells, np.hstack([cells, cell_edges + nv]), nn), mass=1.0, nu=1.0, adv=np.ones(2 * nn), lin=np.ones(2 * nn)))
    full.sort_indices()
    solver.pattern = (full.indptr.astype(np.int32), full.indices.astype(np.int32))  # the Taylor-Hood pattern (superset of A's)
    indptr, indices, data = FakeMatrix(A_dolfin).getValuesCSR()
    Afc = sp.csr_matrix((data, indices, indptr), shape=(W.N, W.N))[to_fc][:, to_fc].tocsr()
    vals = solver._values_on_pattern(Afc)
    back = sp.csr_matrix((vals, full.indices, full.indptr), shape=full.shape)
    assert abs(back - sp.csr_matrix(A_fc)).max() == 0.0
    with pytest.raises(ValueError):
        solver._values_on_pattern((Afc + sp.csr_matrix(([1.0], ([0], [W.N - 1])), shape=Afc.shape)).tocsr())  # two dofs that share no cell
    assert Mesh is not None and TaylorHood is not None


@pytest.mark.gpu
def test_plugin_solver_of_the_stub_on_the_gpu():
    from flowcontrol_amd import _lib

    _lib.load()  # builds the library if needed; the stub itself only gets the path
    lib = hip_solver.load_library(str(_lib.LIB_PATH))
    mesh = FakeDolfinMesh(10)
    mesh.init(1)
    W = FakeMixedSpace(mesh)
    A_fc, A_dolfin, bc_dolfin, to_fc, _ = _system(mesh, W)
    h, to_fc2 = hip_solver.hip_handle(lib, mesh, W, bc_dolfin)
    try:
        assert np.array_equal(to_fc, to_fc2)
        solver = hip_solver.HipNDSolver(lib, h, hip_solver.FC_SLOT_BDF2, to_fc, as_backend_type=lambda A: A)
        solver.set_operator(FakeMatrix(A_dolfin))
        b = FakeVector(np.random.default_rng(0).standard_normal(W.N))
        x = FakeVector(np.zeros(W.N))
        solver.solve(x, b)
        assert x.applied
        ref = spla.splu(A_dolfin.tocsc()).solve(b.a)
        assert np.linalg.norm(x.a - ref) < 1e-10 * np.linalg.norm(ref)
        free = np.ones(W.N)
        free[bc_dolfin] = 0.0  # (Dirichlet rows stay identity rows, as SystemAssembler leaves them)
        free[to_fc[W.N - len(mesh.coordinates()) :]] = 0.0  # ... and there is no pressure-pressure block to add to
        A2 = (A_dolfin + sp.diags(25.0 * free)).tocsr()
        solver.set_operator(FakeMatrix(A2))  # a new operator: numeric phase only
        solver.solve(x, b)
        ref2 = spla.splu(A2.tocsc()).solve(b.a)
        assert np.linalg.norm(ref2 - ref) > 1e-3 * np.linalg.norm(ref)
        assert np.linalg.norm(x.a - ref2) < 1e-10 * np.linalg.norm(ref2)
    finally:
        lib.fc_destroy(h)
