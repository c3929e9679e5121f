"""Field checkpoints as XDMF + HDF5 in the layout of ``dolfin.XDMFFile.write_checkpoint`` (counterpart of the
reference's ``utils/io.py:21-50`` ``write_xdmf`` / ``read_xdmf``).

What dolfin 2019.1 writes for ``write_checkpoint(func, name, time_step, append)`` and what its ``read_checkpoint``
consumes, per frame ``<name>_<counter>`` of the temporal collection ``<name>``:

    /<name>/<name>_<c>/vector        (N, 1)    f8   dof values, in ANY global numbering
    /<name>/<name>_<c>/cell_dofs     (Σ nd, 1) i8   per cell the global dofs in the ELEMENT-LOCAL order of the reader's cell
    /<name>/<name>_<c>/x_cell_dofs   (nc+1, 1) i8   offsets into cell_dofs
    /<name>/<name>_<c>/cells         (nc, 1)   i8   global index (in the reader's mesh) of the cell each row describes
    /<name>/<name>_<c>/mesh/{topology (nc, 3), geometry (nv, 2)}

``read_checkpoint`` assigns, for every cell of ITS mesh, ``x[dofmap.cell_dofs(cell)[j]] = vector[cell_dofs[x[row] + j]]``
— so the file's dof numbering is free, but rows must name the reader's cells and list dofs in dolfin's local order.
dolfin's meshes are ordered: the local vertices of a cell ascend in global vertex index, P2 local dofs are
(v0, v1, v2, e0, e1, e2) with edge i opposite vertex i, a vector element lists component 0 then component 1, the
Taylor–Hood mixed element (ux, uy, p).  The mapping from our numbering (Morton-ordered CCW cells, vertices numbered by
first touch, ``Mesh.orig_vertex`` / ``Mesh.orig_cell`` remember the mesh file's ids) is therefore: rows in the order of
the ORIGINAL cell ids, local vertices sorted by ORIGINAL vertex id, topology / geometry in the original numbering — the
mesh the reference loads from the same ``.xdmf`` mesh file.  ``vector`` is our own dof vector, unchanged.

Files: ``<stem>.xdmf`` (index), ``<stem>.h5`` (frame 0 + mesh + the three cell tables, dolfin's own file name) and
``<stem>.<counter>.h5`` for later frames (vector only; their XDMF items point at the tables of ``<stem>.h5``), so that
appending a checkpoint costs one frame, not the series so far (the pure-Python HDF5 writer cannot extend a file).
Every frame also carries plain node-centred vertex values for viewers without FiniteElementFunction support.

:func:`read_xdmf` reads this layout through the tables (not through our numbering), so it accepts any file that follows
it — a different dof numbering or cell order in the file is fine (``tests/test_host_logic.py`` feeds it one).  Not
verified against dolfin itself: neither dolfin nor h5py exist in this image.
"""

from __future__ import annotations

import logging
import re
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np

from .fem.hdf5_min import MinimalHDF5, write_hdf5
from .fem.spaces import Function

logger = logging.getLogger(__name__)

_ELEMENT = {"V": ("CG", 2, "Vector"), "P": ("CG", 1, "Scalar"), "W": ("Mixed", 2, "Vector")}


def _h5(path: Path) -> Path:
    """File of frame 0, the mesh and the cell tables."""
    return Path(path).with_suffix(".h5")


def _h5_frame(path: Path, counter: int) -> Path:
    return _h5(path) if counter == 0 else Path(path).with_suffix(f".{counter}.h5")


def _frame_times(path: Path) -> list[float]:
    """Times of the frames already listed in the .xdmf index."""
    if not Path(path).exists():
        return []
    return [float(m) for m in re.findall(r'<Time Value="([^"]+)"\s*/>', Path(path).read_text())]


def dolfin_tables(th) -> dict:
    """Per-cell tables of the checkpoint layout for discretisation ``th``, rows in original cell order:
    ``p2`` (nc, 6) / ``p1`` (nc, 3) our scalar P2 node / vertex ids in dolfin's local order, ``topology``, ``geometry``
    in the mesh file's numbering, ``our_cell`` our cell index of every row."""
    cached = getattr(th, "_dolfin_tables", None)
    if cached is not None:
        return cached
    m = th.mesh
    ov = np.asarray(m.orig_vertex if m.orig_vertex is not None else np.arange(m.num_vertices), dtype=np.int64)
    oc = np.asarray(m.orig_cell if m.orig_cell is not None else np.arange(m.num_cells), dtype=np.int64)
    order = np.argsort(ov[m.cells], axis=1, kind="stable")  # local vertices by ascending original id
    v = np.take_along_axis(m.cells.astype(np.int64), order, axis=1)
    e = np.take_along_axis(m.cell_edges.astype(np.int64), order, axis=1)  # edge i opposite (sorted) vertex i
    rows = np.argsort(oc, kind="stable")  # file row r describes original cell oc[rows[r]] = r-th smallest
    geometry = np.empty_like(np.asarray(m.coords, dtype=np.float64))
    geometry[ov] = m.coords
    tab = dict(p2=np.hstack([v, th.nv + e])[rows], p1=v[rows], topology=ov[v][rows], geometry=geometry, our_cell=rows, cells=oc[rows])
    th._dolfin_tables = tab
    return tab


def _cell_dofs(th, kind: str) -> np.ndarray:
    """(nc, nd) our dof indices of the function space ``kind`` per file row, dolfin's element-local order."""
    t = dolfin_tables(th)
    if kind == "P":
        return t["p1"]
    if kind == "V":
        return np.hstack([t["p2"], th.nn + t["p2"]])
    return np.hstack([t["p2"], th.nn + t["p2"], 2 * th.nn + t["p1"]])


def _vertex_values(func: Function) -> np.ndarray:
    """(nv, ncomp) values at the vertices, original vertex numbering."""
    th = func.function_space().th
    a = func.vector().array()
    k = func.function_space().kind
    if k == "P":
        comps = [a[: th.nv]]
    else:
        comps = [a[: th.nv], a[th.nn : th.nn + th.nv]]
        if k == "W":
            comps.append(a[2 * th.nn :])
    ours = np.stack(comps, axis=1)
    m = th.mesh
    ov = np.asarray(m.orig_vertex if m.orig_vertex is not None else np.arange(m.num_vertices), dtype=np.int64)
    out = np.empty_like(ours)
    out[ov] = ours
    return out


def _xml(path: Path, name: str, th, kind: str, n_dofs: int, times: list[float], ncomp: int) -> str:
    nv, nc = th.nv, th.nc
    nd = {"V": 12, "P": 3, "W": 15}[kind]
    fam, deg, atype = _ELEMENT[kind]
    base = _h5(path).name
    g0 = f"{base}:/{name}/{name}_0"
    grids = []
    for c, t in enumerate(times):
        g = f"{_h5_frame(path, c).name}:/{name}/{name}_{c}"
        fe = (f'<Attribute ItemType="FiniteElementFunction" ElementFamily="{fam}" ElementDegree="{deg}" ElementCell="triangle" '
              f'Name="{name}" Center="Other" AttributeType="{atype}">'
              f'<DataItem Dimensions="{nc * nd} 1" NumberType="UInt" Format="HDF">{g0}/cell_dofs</DataItem>'
              f'<DataItem Dimensions="{n_dofs} 1" NumberType="Float" Format="HDF">{g}/vector</DataItem>'
              f'<DataItem Dimensions="{nc + 1} 1" NumberType="UInt" Format="HDF">{g0}/x_cell_dofs</DataItem>'
              f'<DataItem Dimensions="{nc} 1" NumberType="UInt" Format="HDF">{g0}/cells</DataItem></Attribute>')
        nodal = ""
        for j in range(ncomp):  # plain node scalars next to it: understood by every XDMF reader
            label = name if ncomp == 1 else f"{name}_{'xyp'[j]}"
            nodal += (f'<Attribute Name="{label}_vertex" AttributeType="Scalar" Center="Node"><DataItem ItemType="HyperSlab" Dimensions="{nv} 1">'
                      f'<DataItem Dimensions="3 2" Format="XML">0 {j} 1 1 {nv} 1</DataItem>'
                      f'<DataItem Dimensions="{nv} {ncomp}" NumberType="Float" Precision="8" Format="HDF">{g}/vertex_values</DataItem></DataItem></Attribute>')
        grids.append(
            f'<Grid Name="{name}_{c}" GridType="Uniform">'
            f'<Topology NumberOfElements="{nc}" TopologyType="Triangle" NodesPerElement="3"><DataItem Dimensions="{nc} 3" NumberType="UInt" Format="HDF">{g0}/mesh/topology</DataItem></Topology>'
            f'<Geometry GeometryType="XY"><DataItem Dimensions="{nv} 2" NumberType="Float" Precision="8" Format="HDF">{g0}/mesh/geometry</DataItem></Geometry>'
            f'<Time Value="{t:.16g}" />{fe}{nodal}</Grid>')
    return (f'<?xml version="1.0"?><Xdmf Version="3.0"><Domain><Grid Name="{name}" GridType="Collection" CollectionType="Temporal">'
            + "".join(grids) + "</Grid></Domain></Xdmf>")


def write_xdmf(filename, func: Function, name: str, time_step: float = 0.0, append: bool = False, write_mesh: bool = True) -> int:
    """Append (or start) a checkpoint series; returns the frame counter written (``utils/io.py:21-41``).

    ``write_mesh`` is accepted for signature parity: the mesh and the cell tables are written once, with frame 0."""
    path = Path(filename)
    path.parent.mkdir(parents=True, exist_ok=True)
    space = func.function_space()
    th, kind = space.th, space.kind
    times = _frame_times(path) if append else []
    if times and not _h5(path).exists():
        times = []
    counter = len(times)
    vec = np.ascontiguousarray(func.vector().get_local(), dtype=np.float64).reshape(-1, 1)
    vv = _vertex_values(func)
    frame = {"vector": vec, "vertex_values": vv, "time": np.array([float(time_step)])}
    if counter == 0:
        t = dolfin_tables(th)
        cd = _cell_dofs(th, kind)
        frame.update({
            "cell_dofs": cd.reshape(-1, 1).astype(np.int64),
            "x_cell_dofs": (np.arange(th.nc + 1, dtype=np.int64) * cd.shape[1]).reshape(-1, 1),
            "cells": t["cells"].reshape(-1, 1).astype(np.int64),
            "mesh": {"topology": t["topology"].astype(np.int64), "geometry": t["geometry"]},
        })
    write_hdf5(_h5_frame(path, counter), {name: {f"{name}_{counter}": frame}})
    times.append(float(time_step))
    path.write_text(_xml(path, name, th, kind, vec.shape[0], times, vv.shape[1]))
    return counter


def _items(path: Path, name: str, counter: int):
    """(time, {dataset role: (h5 file, h5 path)}) of frame ``counter`` (−1: last) of series ``name`` from the XDMF index."""
    root = ET.parse(path).getroot()
    series = [g for g in root.iter("Grid") if g.get("CollectionType") == "Temporal"]
    pick = [g for g in series if g.get("Name") == name] or series
    if not pick:
        raise KeyError(f"{path}: no temporal collection")
    frames = [g for g in pick[0] if g.tag == "Grid"]
    c = len(frames) - 1 if counter < 0 else counter
    if c < 0 or c >= len(frames):
        raise FileNotFoundError(f"{path}: series has no frame {c} ({len(frames)} frames)")
    grid = frames[c]
    fe = [a for a in grid.findall("Attribute") if a.get("ItemType") == "FiniteElementFunction"]
    if not fe:
        raise KeyError(f"{path}: frame {c} holds no FiniteElementFunction attribute")
    if pick[0].get("Name") != name and fe[0].get("Name") != name:
        raise KeyError(f"{path}: no series {name!r}; have {sorted(g.get('Name') or '' for g in series)}")
    items = [d.text.strip() for d in fe[0].findall("DataItem")]
    if len(items) != 4:
        raise ValueError(f"{path}: expected cell_dofs, vector, x_cell_dofs, cells in frame {c}")
    roles = {}
    for role, text in zip(("cell_dofs", "vector", "x_cell_dofs", "cells"), items):
        f, _, p = text.partition(":")
        roles[role] = (path.parent / f, p)
    t = grid.find("Time")
    return (float(t.get("Value")) if t is not None else 0.0), roles


def read_xdmf(filename, func: Function, name: str, counter: int = -1) -> float:
    """Load frame ``counter`` (−1: last) of series ``name`` into ``func``; returns its time (``utils/io.py:44-50``).

    The assignment goes through the file's cell tables, exactly as ``dolfin``'s ``read_checkpoint`` does."""
    path = Path(filename)
    if not path.exists():
        raise FileNotFoundError(f"{filename}: no such checkpoint series")
    time, roles = _items(path, name, counter)
    opened: dict[Path, MinimalHDF5] = {}

    def load(role):
        f, p = roles[role]
        if not f.exists():
            raise FileNotFoundError(f"{filename}: frame data {f} not found")
        if f not in opened:
            opened[f] = MinimalHDF5(f)
        return np.asarray(opened[f].read(p)).reshape(-1)

    space = func.function_space()
    th, kind = space.th, space.kind
    ours = _cell_dofs(th, kind)  # rows: original cell order
    x = load("x_cell_dofs").astype(np.int64)
    cells = load("cells").astype(np.int64)
    cd = load("cell_dofs").astype(np.int64)
    vec = load("vector").astype(np.float64)
    nd = ours.shape[1]
    if cells.size != th.nc or np.any(np.diff(x) != nd) or vec.size != func.vector().size():
        raise ValueError(f"{filename}: frame has {vec.size} dofs / {cells.size} cells / {int(np.diff(x).max()) if x.size > 1 else 0} dofs per cell, "
                         f"the function space has {func.vector().size()} / {th.nc} / {nd} (different mesh or element?)")
    if np.any(cells < 0) or np.any(cells >= th.nc):
        raise ValueError(f"{filename}: cell index outside the mesh")
    out = np.empty(func.vector().size())
    # row r of the file describes the reader's cell cells[r]; our table is indexed by original cell id
    out[ours[cells].reshape(-1)] = vec[cd[x[0] : x[-1]]]
    func.vector().set_local(out)
    return time


def export_sparse_matrix(A, figname=None) -> bool:
    """Spy plot of a sparse (or dense) matrix as PNG (reference ``utils/io.py:254-272``).  Needs matplotlib, which this image does
    not ship: returns False (and writes nothing) when it cannot be imported."""
    try:
        import matplotlib

        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        return False
    import scipy.sparse as sp

    M = A if sp.issparse(A) else sp.csr_matrix(np.asarray(A))
    fig, ax = plt.subplots()
    ax.spy(M, markersize=1)
    ax.set_title("Sparse matrix plot")
    fig.savefig(figname if figname is not None else "spy.png")
    plt.close(fig)
    return True


def export_square_operators(path, operators, operators_names) -> None:
    """``<name>.npz`` (CSR) and ``<name>_coo.npz`` (COO) per operator — the files the reference's control-design scripts load
    (``utils/io.py:237-251``) — plus ``<name>.png`` when matplotlib is available."""
    import scipy.sparse as sp

    d = Path(path)
    d.mkdir(parents=True, exist_ok=True)
    for M, name in zip(operators, operators_names):
        csr = sp.csr_matrix(M)
        export_sparse_matrix(csr, d / f"{name}.png")
        sp.save_npz(d / f"{name}.npz", csr)
        sp.save_npz(d / f"{name}_coo.npz", csr.tocoo())


def export_subdomains(mesh, subdomains_list, filename="subdomains.xdmf") -> np.ndarray:
    """Facet markers of the boundary sub-domains for visualisation (reference ``utils/io.py:171-185``: a facet
    ``MeshFunction`` with 0 everywhere and i + 1 on the facets of ``subdomains_list[i]``, later entries overwriting earlier
    ones), written as an XDMF file over the mesh EDGES (``Polyline`` cells with one integer attribute ``f``; heavy data in
    ``<stem>.h5``) that ParaView opens.  Usage as in the reference: ``export_subdomains(fs.mesh, fs.boundaries.subdomain, path)``.
    Returns the marker array (one entry per mesh edge)."""
    path = Path(filename)
    path.parent.mkdir(parents=True, exist_ok=True)
    markers = np.zeros(mesh.edges.shape[0], dtype=np.int64)
    for i, sub in enumerate(subdomains_list):
        sub.mark(markers, i + 1, mesh)
        logger.info("Marking subdomain nr: %d", i + 1)
    h5 = path.with_suffix(".h5")
    write_hdf5(h5, {"mesh": {"topology": mesh.edges.astype(np.int64), "geometry": np.asarray(mesh.coords, dtype=np.float64)},
                    "f": markers.reshape(-1, 1)})
    ne, nv = mesh.edges.shape[0], mesh.coords.shape[0]
    path.write_text(
        '<?xml version="1.0"?>\n<Xdmf Version="3.0"><Domain><Grid Name="subdomains" GridType="Uniform">\n'
        f'<Topology TopologyType="Polyline" NodesPerElement="2" NumberOfElements="{ne}">'
        f'<DataItem Dimensions="{ne} 2" NumberType="Int" Precision="8" Format="HDF">{h5.name}:/mesh/topology</DataItem></Topology>\n'
        f'<Geometry GeometryType="XY"><DataItem Dimensions="{nv} 2" NumberType="Float" Precision="8" Format="HDF">{h5.name}:/mesh/geometry</DataItem></Geometry>\n'
        f'<Attribute Name="f" AttributeType="Scalar" Center="Cell"><DataItem Dimensions="{ne} 1" NumberType="Int" Precision="8" Format="HDF">{h5.name}:/f</DataItem></Attribute>\n'
        "</Grid></Domain></Xdmf>\n")
    logger.info("Writing subdomains file: %s", path)
    return markers


__all__ = ["write_xdmf", "read_xdmf", "dolfin_tables", "export_sparse_matrix", "export_square_operators", "export_subdomains"]
