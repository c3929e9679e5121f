"""Triangle mesh, XDMF reading, topology and locality ordering (host side, setup only).

Replaces ``FlowSolver._make_mesh`` (reference ``src/flowcontrol/flowsolver.py:233-240``)
and the parts of dolfin's ``Mesh`` the hot path needs: vertex coordinates, cell→vertex,
cell→edge, exterior facets.  Cells are re-ordered along a Morton curve and vertices are
renumbered by first touch so that the per-cell gathers issued by the HIP element-loop
kernels hit neighbouring addresses (coalescing / L2 locality on large meshes).
"""

from __future__ import annotations

import re
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from .hdf5_min import MinimalHDF5


def _morton_key(xy: np.ndarray, bits: int = 16) -> np.ndarray:
    lo = xy.min(axis=0)
    span = np.maximum(xy.max(axis=0) - lo, 1e-300)
    q = np.minimum(((xy - lo) / span * (2**bits - 1)).astype(np.uint64), 2**bits - 1)

    def spread(v: np.ndarray) -> np.ndarray:
        v = v & np.uint64(0xFFFFFFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
        return v

    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))


@dataclass
class Mesh:
    """2-D affine triangle mesh.

    Attributes
    ----------
    coords : (nv, 2) float64 vertex coordinates (renumbered).
    cells : (nc, 3) int32 vertex ids, counter-clockwise.
    edges : (ne, 2) int32 sorted vertex pairs.
    cell_edges : (nc, 3) int32; local edge k is opposite local vertex k.
    edge_cells : (ne, 2) int32 incident cells (-1 if none).
    orig_vertex : (nv,) original (file) vertex id of each renumbered vertex.
    orig_cell : (nc,) original (file) cell id of each re-ordered cell.
    """

    coords: np.ndarray
    cells: np.ndarray
    edges: np.ndarray = field(init=False)
    cell_edges: np.ndarray = field(init=False)
    edge_cells: np.ndarray = field(init=False)
    orig_vertex: np.ndarray | None = None
    orig_cell: np.ndarray | None = None

    def __post_init__(self) -> None:
        self.coords = np.ascontiguousarray(self.coords, dtype=np.float64)
        cells = np.ascontiguousarray(self.cells, dtype=np.int64)
        # counter-clockwise orientation
        p = self.coords[cells]
        det = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 2, 0] - p[:, 0, 0]) * (
            p[:, 1, 1] - p[:, 0, 1]
        )
        if np.any(det == 0.0):
            raise ValueError("degenerate (zero-area) cell in mesh")
        flip = det < 0
        cells[flip] = cells[flip][:, [0, 2, 1]]
        self.cells = cells.astype(np.int32)
        self._build_edges()

    def _build_edges(self) -> None:
        c = self.cells.astype(np.int64)
        nc = c.shape[0]
        # local edge k opposite vertex k: (k+1, k+2)
        pairs = np.stack([c[:, [1, 2]], c[:, [2, 0]], c[:, [0, 1]]], axis=1).reshape(-1, 2)
        pairs.sort(axis=1)
        key = pairs[:, 0] * (self.coords.shape[0] + 1) + pairs[:, 1]
        uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
        # number edges by first appearance in the (locality-ordered) cell list
        order = np.argsort(first, kind="stable")
        rank = np.empty_like(order)
        rank[order] = np.arange(order.size)
        eid = rank[inv]
        self.edges = pairs[first[order]].astype(np.int32)
        self.cell_edges = eid.reshape(nc, 3).astype(np.int32)
        ne = self.edges.shape[0]
        ec = -np.ones((ne, 2), dtype=np.int32)
        cell_of = np.repeat(np.arange(nc, dtype=np.int32), 3)
        # first / second incidence
        srt = np.argsort(eid, kind="stable")
        e_s, c_s = eid[srt], cell_of[srt]
        start = np.r_[True, e_s[1:] != e_s[:-1]]
        ec[e_s[start], 0] = c_s[start]
        second = ~start
        ec[e_s[second], 1] = c_s[second]
        self.edge_cells = ec

    # ── sizes ────────────────────────────────────────────────────────────────
    @property
    def num_vertices(self) -> int:
        return self.coords.shape[0]

    @property
    def num_cells(self) -> int:
        return self.cells.shape[0]

    @property
    def num_edges(self) -> int:
        return self.edges.shape[0]

    def coordinates(self) -> np.ndarray:
        return self.coords

    # ── geometry ─────────────────────────────────────────────────────────────
    def boundary_edges(self) -> np.ndarray:
        """Ids of exterior facets (edges with a single incident cell)."""
        return np.nonzero(self.edge_cells[:, 1] < 0)[0].astype(np.int32)

    def edge_midpoints(self) -> np.ndarray:
        return 0.5 * (self.coords[self.edges[:, 0]] + self.coords[self.edges[:, 1]])

    def cell_centroids(self) -> np.ndarray:
        return self.coords[self.cells].mean(axis=1)

    def hmin(self) -> float:
        e = self.coords[self.edges[:, 0]] - self.coords[self.edges[:, 1]]
        return float(np.sqrt((e * e).sum(axis=1)).min())

    # ── construction helpers ─────────────────────────────────────────────────
    @classmethod
    def from_arrays(cls, coords: np.ndarray, cells: np.ndarray, reorder: bool = True) -> "Mesh":
        coords = np.asarray(coords, dtype=np.float64)[:, :2]
        cells = np.asarray(cells, dtype=np.int64)
        nv = coords.shape[0]
        if cells.min() < 0 or cells.max() >= nv:
            raise ValueError("cell vertex index out of range")
        # orient counter-clockwise *before* renumbering so the numbering does not depend on
        # the orientation the file happened to use
        p = coords[cells]
        det = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 2, 0] - p[:, 0, 0]) * (
            p[:, 1, 1] - p[:, 0, 1]
        )
        cells = cells.copy()
        cells[det < 0] = cells[det < 0][:, [0, 2, 1]]
        if reorder:
            ckey = _morton_key(coords[cells].mean(axis=1))
            corder = np.argsort(ckey, kind="stable")
            cells = cells[corder]
            flat = cells.reshape(-1)
            _, first = np.unique(flat, return_index=True)
            touched = flat[np.sort(first)]
            if touched.size != nv:  # isolated vertices go last
                rest = np.setdiff1d(np.arange(nv), touched)
                touched = np.r_[touched, rest]
            new_of_old = np.empty(nv, dtype=np.int64)
            new_of_old[touched] = np.arange(nv)
            coords = coords[touched]
            cells = new_of_old[cells]
            return cls(coords, cells, orig_vertex=touched.astype(np.int64), orig_cell=corder.astype(np.int64))
        return cls(coords, cells, orig_vertex=np.arange(nv), orig_cell=np.arange(cells.shape[0]))

    @classmethod
    def unit_square(cls, nx: int, ny: int, reorder: bool = True) -> "Mesh":
        """Right-diagonal structured mesh of [0,1]², the layout of ``dolfin.UnitSquareMesh``."""
        x = np.linspace(0.0, 1.0, nx + 1)
        y = np.linspace(0.0, 1.0, ny + 1)
        X, Y = np.meshgrid(x, y, indexing="xy")
        coords = np.stack([X.ravel(), Y.ravel()], axis=1)
        idx = np.arange((nx + 1) * (ny + 1)).reshape(ny + 1, nx + 1)
        v0, v1 = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel()
        v2, v3 = idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
        cells = np.concatenate([np.stack([v0, v1, v3], 1), np.stack([v0, v2, v3], 1)])
        return cls.from_arrays(coords, cells, reorder=reorder)

    def refine(self, project=None) -> "Mesh":
        """Uniform red refinement (each triangle → 4).  ``project(xy, on_boundary_mask)`` may
        move new boundary midpoints (e.g. onto the cylinder)."""
        nv = self.num_vertices
        mid = self.edge_midpoints()
        if project is not None:
            is_b = self.edge_cells[:, 1] < 0
            mid = project(mid, is_b)
        coords = np.vstack([self.coords, mid])
        c = self.cells.astype(np.int64)
        e = self.cell_edges.astype(np.int64) + nv
        cells = np.concatenate(
            [
                np.stack([c[:, 0], e[:, 2], e[:, 1]], 1),
                np.stack([c[:, 1], e[:, 0], e[:, 2]], 1),
                np.stack([c[:, 2], e[:, 1], e[:, 0]], 1),
                np.stack([e[:, 0], e[:, 1], e[:, 2]], 1),
            ]
        )
        return Mesh.from_arrays(coords, cells)


_HDF_ITEM = re.compile(r"^\s*([^:\s]+):(\S+)\s*$")


def read_xdmf_mesh(path: str | Path, reorder: bool = True) -> Mesh:
    """Read a triangle mesh from an XDMF file with an HDF5 heavy-data file next to it.

    Handles both the meshio layout (``/data0``, ``/data1``) and the dolfin layout
    (``/Mesh/mesh/{geometry,topology}``) — the item paths are taken from the XML.
    ``.npz`` files with ``coords`` / ``cells`` arrays (the committed fixtures under
    ``flowcontrol_amd/examples/*/data_input``) are accepted as well.
    """
    path = Path(path)
    if path.suffix == ".npz":
        z = np.load(path)
        return Mesh.from_arrays(z["coords"], z["cells"], reorder=reorder)
    root = ET.parse(path).getroot()
    geo = topo = None
    for grid in root.iter("Grid"):
        g = grid.find("Geometry")
        t = grid.find("Topology")
        if g is not None and t is not None:
            geo, topo = g.find("DataItem"), t.find("DataItem")
            ttype = t.get("TopologyType", "Triangle")
            if ttype.lower() != "triangle":
                raise ValueError(f"only Triangle meshes are supported, got {ttype}")
            break
    if geo is None or topo is None:
        raise ValueError(f"{path}: no Geometry/Topology pair found")

    def load(item) -> np.ndarray:
        m = _HDF_ITEM.match(item.text or "")
        if item.get("Format", "HDF").upper() != "HDF" or not m:
            vals = np.array((item.text or "").split(), dtype=float)
            dims = tuple(int(s) for s in item.get("Dimensions").split())
            return vals.reshape(dims)
        return MinimalHDF5(path.parent / m.group(1)).read(m.group(2))

    coords = np.asarray(load(geo), dtype=np.float64)
    cells = np.asarray(load(topo)).astype(np.int64)
    return Mesh.from_arrays(coords, cells, reorder=reorder)


__all__ = ["Mesh", "read_xdmf_mesh"]
